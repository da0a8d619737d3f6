cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_concurrent_moves.py -x -q -m gpu 2>&1 | grep "passed\|failed\|rror" | tail -3
SECONDS=150 SEED=611 timeout 400 python3 tools/stress_parity.py 2>&1 | tail -1
rm -rf /tmp/bks; rocprofv3 --kernel-trace --output-format csv -d /tmp/bks -- python3 tools/batch_kernel_split.py > /dev/null 2>&1
F=$(find /tmp/bks -name "*kernel_trace.csv" | head -1)
python3 tools/batch_kernel_split.py --summarize $F | head -9
for i in 1 2; do N=20000 K=6 REPEAT=1 INIT=stable python3 tools/loop_bench.py 2>&1 | grep -o '"loop_s": [0-9.]*\|"total_s_second_call": [0-9.]*\|"digest": "[0-9a-f]*"' | tr '\n' ' '; echo; done
REPEAT=1 INIT=stable python3 tools/loop_bench.py 2>&1 | grep -o '"loop_s": [0-9.]*\|"total_s_second_call": [0-9.]*\|"digest": "[0-9a-f]*"' | tr '\n' ' '; echo
