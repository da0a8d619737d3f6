# The short form of tools/evidence_r04.sh: tests, bench, rocprof passes, proxies, timelines, loop timing.
#   tools/evidence_r04_refresh.sh <head-sha>
set -x
SHA=${1:-unknown}
export TMPDIR=/tmp
mkdir -p gpurun_out/ev4
python -m pytest tests -m gpu -q 2>&1 | grep -v '^\[Multi-H\]\|^Median\|^Iteration\|^$' | tail -8 > gpurun_out/ev4/tests.log
python bench.py > gpurun_out/ev4/bench.json 2> gpurun_out/ev4/bench.err
bash tools/profile_bench.sh r04 $SHA > gpurun_out/ev4/profile_bench.log 2>&1
python tools/shard_proxy.py > gpurun_out/ev4/shard_proxy.txt 2>&1
HEADROOM=-1,0,64 SIZES=100000,12500 python tools/shard_proxy.py > gpurun_out/ev4/shard_proxy_ab.txt 2>&1
DEPTH=1 python tools/shard_proxy.py > gpurun_out/ev4/shard_proxy_depth1.txt 2>&1
python tools/enqueue_probe.py > gpurun_out/ev4/enqueue_probe.txt 2>&1
DLTFORM=1 SIZES=100000,25000,12500 timeout 300 python tools/shard_proxy.py 2>&1 | grep -a 'DLT form\|^== the prefetched' > gpurun_out/ev4/shard_proxy_dltform.txt
timeout 120 python tools/dlt_probe.py > gpurun_out/ev4/dlt_probe.txt 2>&1
for M in 12500 100000; do
  SIZES=$M STEPS=20 WARM=3 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ev4/tl_$M -- python3 tools/shard_proxy.py > gpurun_out/ev4/tl_$M.log 2>&1
  python3 tools/timeline.py $(find gpurun_out/ev4/tl_$M -name "*kernel_trace.csv" | head -1) 8 3 > gpurun_out/ev4/timeline_$M.txt
done
find gpurun_out/ev4 -name "*.csv" -size +8M -delete
MULTIH_TIMING=1 python tools/loop_bench.py > gpurun_out/ev4/loop_timing.txt 2>&1
tail -3 gpurun_out/ev4/tests.log
