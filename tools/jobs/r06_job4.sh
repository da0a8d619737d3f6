cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 3000 python3 -m pytest tests -x -q -m gpu 2>&1 | grep "passed\|failed\|rror" | tail -5 > gpurun_out/r06/gputests_b.txt
cat gpurun_out/r06/gputests_b.txt
timeout 1500 python3 bench.py > gpurun_out/r06/bench_b.json 2> gpurun_out/r06/bench_b.err
tail -c 300 gpurun_out/r06/bench_b.err
SECONDS=200 SEED=61 timeout 500 python3 tools/stress_parity.py 2>&1 | tail -2
SECONDS=120 timeout 400 python3 tools/stress_alternation.py 2>&1 | tail -2
