cd $GRAFT_REPO_ROOT
for n in 20000 50000; do
for sp in 0 1 0 1; do
N=$n K=$( [ $n = 20000 ] && echo 6 || echo 10 ) INIT=stable REPEAT=1 TUNE=39=$sp timeout 600 python3 tools/loop_bench.py 2>&1 | grep -E "^\{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print($n, 'speculate', $sp, {k:d[k] for k in ('loop_s','total_s_second_call')})"
done; done
