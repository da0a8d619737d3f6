cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_alternation.py tests/test_gpu_barrsmith.py -x -q -m gpu 2>&1 | grep "passed\|failed\|rror" | tail -3
for n in 20000 50000; do
N=$n K=$( [ $n = 20000 ] && echo 6 || echo 10 ) INIT=stable REPEAT=1 MULTIH_TIMING=1 timeout 600 python3 tools/loop_bench.py 2>&1 | grep -E "^\{|stable sets:" | cut -c1-200 | tail -2
done
