# rocprofv3 kernel stats of Process() by the reference's own route (tools/loop_bench.py INIT=stable) at 20 000 points; runs on the GPU box.
export TMPDIR=/tmp
cd /tmp && N=20000 K=6 HYP=50000 INIT=stable rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stable -- python3 $GRAFT_REPO_ROOT/tools/loop_bench.py > /tmp/prof_stable.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_stable -name "*kernel_stats.csv" | head -1)
head -14 $f | cut -c1-60,300-400 | awk -F, '{print}' 
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/prof_stable/**/*kernel_stats.csv',recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    n=r['Name'].split('(')[0][-40:]
    print(f"{n:42s} calls {int(r['Calls']):7d} total {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Percentage']}%")
    tot+=float(r['TotalDurationNs'])
print('sum', tot/1e6,'ms')
PY
grep "^N=\|Alternating\|Stable" /tmp/prof_stable.log | cut -c1-200
