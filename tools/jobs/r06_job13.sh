cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_postfilter.py tests/test_gpu_alternation.py -x -q -m gpu 2>&1 | grep "passed\|failed\|rror" | tail -3
for i in 1 2 3; do MULTIH_TIMING=1 REPEAT=1 python3 tools/loop_bench.py 2>&1 | grep -a "Compatibility check time\|total_s_second_call" | tail -2 | grep -o 'Compatibility check time = [0-9.]*\|"total_s_second_call": [0-9.]*\|"digest": "[0-9a-f]*"' | tr '\n' ' '; echo; done
python3 tools/small_scene_bench.py 2>&1 | grep "== N=" | grep "call 2" | cut -c1-160
SECONDS=100 SEED=77 timeout 300 python3 tools/stress_alternation.py > /tmp/a.txt 2>&1; tail -c 300 /tmp/a.txt | grep -a -o "Process() stress.*"
