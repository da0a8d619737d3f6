cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_gpu_concurrent_moves.py -x -q -m gpu 2>&1 | grep "passed\|failed\|rror" | tail -3
for i in 1 2; do
  N=20000 K=6 REPEAT=1 INIT=stable python3 tools/loop_bench.py 2>&1 | grep -o '"loop_s": [0-9.]*\|"total_s_second_call": [0-9.]*\|"digest": "[0-9a-f]*"' | tr '\n' ' '; echo
  REPEAT=1 INIT=stable python3 tools/loop_bench.py 2>&1 | grep -o '"loop_s": [0-9.]*\|"total_s_second_call": [0-9.]*\|"digest": "[0-9a-f]*"' | tr '\n' ' '; echo
  REPEAT=1 python3 tools/loop_bench.py 2>&1 | grep -o '"loop_s": [0-9.]*\|"total_s_second_call": [0-9.]*\|"digest": "[0-9a-f]*"' | tr '\n' ' '; echo
done
SECONDS=240 SEED=712 timeout 500 python3 tools/stress_parity.py 2>&1 | tail -1
SECONDS=120 SEED=713 timeout 400 python3 tools/stress_alternation.py > /tmp/a.txt 2>&1; tail -1 gpurun_out/stress_alternation.txt | cut -c1-200
