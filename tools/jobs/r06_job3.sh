cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_postfilter.py -x -q -m gpu 2>&1 | grep -v "^\[Multi-H\]" | tail -12
for i in 1 2; do N=50000 K=10 REPEAT=1 MULTIH_TIMING=1 timeout 600 python3 tools/loop_bench.py 2>&1 | grep -E "done|^N=|Compat|total_s_second" | cut -c1-300 | tail -14; done
