cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf /tmp/lp; MULTIH_TIMING=1 REPEAT=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/lp -- python3 tools/loop_bench.py > /tmp/lp.txt 2>&1
grep -a "after Process\|time =" /tmp/lp.txt | tail -12 | cut -c1-200
python3 tools/process_trace_split.py $(find /tmp/lp -name "*kernel_trace.csv" | head -1) 40
