cd $GRAFT_REPO_ROOT
echo "== before (checked sweep)"; MH_LIB=tools/jobs/libmultih_hip_before_symlean.so timeout 300 python3 tools/symmetric_probe.py 2>&1 | tail -1
echo "== lean symmetric sweep"; timeout 300 python3 tools/symmetric_probe.py 2>&1 | tail -1
echo "== before"; MH_LIB=tools/jobs/libmultih_hip_before_symlean.so timeout 300 python3 tools/symmetric_probe.py 2>&1 | tail -1
echo "== lean"; timeout 300 python3 tools/symmetric_probe.py 2>&1 | tail -1
timeout 1500 python3 -m pytest tests/test_symmetric_exact.py tests/test_gpu_parity.py -x -q -m gpu -k "symmetric or sym" 2>&1 | grep "passed\|failed\|rror" | tail -3
