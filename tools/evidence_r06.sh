# Round-6 evidence session on the GPU box: tests, bench, rocprof passes of the bench command (kernel trace + the PMC passes for the
# residual kernel's HBM traffic), labeling on the three scenes, the loop by both routes with the per-iteration split, the concurrent
# alpha-moves (batch probe, per-move solver times alone / beside others), small scenes, barrsmith from the reference's rows and from
# the raw file (twelve seeds), the at-size alternation against the oracle.
#   tools/evidence_r06.sh <head-sha>
set -x
SHA=${1:-unknown}
export TMPDIR=/tmp
E=gpurun_out/ev6
rm -rf $E; mkdir -p $E
python -m pytest tests -m gpu -q 2>&1 | grep -v '^\[Multi-H\]\|^Median\|^Iteration\|^$' | tail -8 > $E/tests.log
python bench.py > $E/bench.json 2> $E/bench.err
bash tools/profile_bench.sh r06 $SHA > $E/profile_bench.log 2>&1
python tools/label_bench.py > $E/label_bench.txt 2>&1
SEPARATION=2 python tools/label_bench.py > $E/label_bench_intermediate_scene.txt 2>&1
LEGACY=1 python tools/label_bench.py > $E/label_bench_r04_scene.txt 2>&1
MULTIH_TIMING=1 REPEAT=1 python tools/loop_bench.py > $E/loop_timing.txt 2>&1
MULTIH_TIMING=1 REPEAT=1 INIT=stable python tools/loop_bench.py > $E/loop_timing_reference_route.txt 2>&1
MULTIH_TIMING=1 REPEAT=1 INIT=stable TUNE=37=1 python tools/loop_bench.py > $E/loop_timing_reference_route_sequential_moves.txt 2>&1
N=20000 K=6 MULTIH_TIMING=1 REPEAT=1 INIT=stable python tools/loop_bench.py > $E/loop_timing_reference_route_20000.txt 2>&1
N=20000 K=6 MULTIH_TIMING=1 REPEAT=1 INIT=stable TUNE=37=1 python tools/loop_bench.py > $E/loop_timing_reference_route_20000_sequential_moves.txt 2>&1
REPS=5 CTX=16 python tools/batch_probe.py > $E/batch_probe.txt 2>&1
python tools/batch_trace_probe.py > $E/batch_trace_probe.txt 2>&1
python tools/small_scene_bench.py > $E/small_scenes.txt 2>&1
SEEDS=1234,7,99,1,2,3,4,5,6,8,9,10 python tools/barrsmith_agreement.py > $E/barrsmith.txt 2>&1
python tools/at_size_alternation.py > $E/at_size_init.txt 2>&1
ROUTE=dlt python tools/at_size_alternation.py > $E/at_size_dlt.txt 2>&1
SECONDS=120 SEED=66 python tools/stress_parity.py > $E/stress_parity.txt 2>&1
find $E -name "*.csv" -size +8M -delete
tail -3 $E/tests.log
