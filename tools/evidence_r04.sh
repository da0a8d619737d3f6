# Runs on the GPU box: tests, bench, rocprof passes, the strong-scaling proxy and its timeline, labeling / loop timings,
# the r04 measurements (core components, barrsmith agreement, at-size oracle runs), stress runs.
#   tools/evidence_r04.sh <head-sha>     (the sha is passed in: .git does not travel to the box)
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
timeout 200 python tools/meanshift_probe.py > gpurun_out/ev4/meanshift_probe.txt 2>&1
for M in 12500 100000; do
  SIZES=$M STEPS=20 WARM=3 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ev4/tl_$M -- python3 tools/shard_proxy.py > gpurun_out/ev4/tl_$M.log 2>&1
  python3 tools/timeline.py $(find gpurun_out/ev4/tl_$M -name "*kernel_trace.csv" | head -1) 8 3 > gpurun_out/ev4/timeline_$M.txt
done
find gpurun_out/ev4 -name "*.csv" -size +8M -delete
python tools/score_bench.py > gpurun_out/ev4/score_bench.txt 2>&1
CPU=1 python tools/label_bench.py > gpurun_out/ev4/label_bench.txt 2>&1
bash tools/profile_label.sh r04_label $SHA > gpurun_out/ev4/profile_label.log 2>&1
python tools/core_components.py > gpurun_out/ev4/core_components.txt 2>&1
SWEEP=1 python tools/barrsmith_agreement.py > gpurun_out/ev4/barrsmith.txt 2>&1
python tools/at_size_alternation.py > gpurun_out/ev4/at_size_init.txt 2>&1
ROUTE=dlt python tools/at_size_alternation.py > gpurun_out/ev4/at_size_dlt.txt 2>&1
MULTIH_TIMING=1 python tools/small_scene_bench.py > gpurun_out/ev4/small_scenes.txt 2>&1
MULTIH_TIMING=1 python tools/loop_bench.py > gpurun_out/ev4/loop_timing.txt 2>&1
MULTIH_TIMING=1 INIT=stable python tools/loop_bench.py > gpurun_out/ev4/loop_timing_reference_init.txt 2>&1
ITER_HYP=100000 python tools/loop_bench.py > gpurun_out/ev4/loop_reproposal.txt 2>&1
SECONDS=150 SEED=4 python tools/stress_parity.py > gpurun_out/ev4/stress_parity.txt 2>&1
SECONDS=200 SEED=4 python tools/stress_alternation.py > gpurun_out/ev4/stress_process.txt 2>&1
SECONDS=90 SEED=4 python tools/stress_residual_edges.py > gpurun_out/ev4/stress_residual_edges.txt 2>&1
tail -n 3 gpurun_out/ev4/tests.log gpurun_out/ev4/stress_parity.txt gpurun_out/ev4/stress_process.txt gpurun_out/ev4/stress_residual_edges.txt
