set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
python3 tools/batch_kernel_split.py > gpurun_out/r06/batch_split_plain.txt 2>&1
rm -rf /tmp/bks; rocprofv3 --kernel-trace --output-format csv -d /tmp/bks -- python3 tools/batch_kernel_split.py > gpurun_out/r06/batch_split_traced.txt 2>&1
F=$(find /tmp/bks -name "*kernel_trace.csv" | head -1)
python3 tools/batch_kernel_split.py --summarize $F > gpurun_out/r06/batch_split_summary.txt 2>&1
cat gpurun_out/r06/batch_split_plain.txt | grep -v "^\[Multi" | tail -3
cat gpurun_out/r06/batch_split_summary.txt; head -3 $F | cut -c1-400
