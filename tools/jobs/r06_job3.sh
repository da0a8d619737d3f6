cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_concurrent_moves.py tests/test_gpu_parity.py -x -q -m gpu -k "concurrent or batch or expand or label or solver or recycle or dense or every or hundreds or loop" 2>&1 | grep "passed\|failed\|rror" | tail -3
SECONDS=90 SEED=7 timeout 400 python3 tools/stress_parity.py 2>&1 | tail -1
for n in 20000 50000; do
N=$n K=$( [ $n = 20000 ] && echo 6 || echo 10 ) INIT=stable REPEAT=1 timeout 600 python3 tools/loop_bench.py 2>&1 | grep -E "^\{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print($n, {k:d[k] for k in ('loop_s','total_s_second_call','digest')})"
done
