# Runs on the GPU box: tests, bench, rocprof passes, energy per variant, labeling and loop timings, stress runs.
#   tools/evidence_r03.sh <head-sha>     (the sha is passed in: .git does not travel to the box)
set -x
SHA=${1:-unknown}
export TMPDIR=/tmp
mkdir -p gpurun_out/ev
python -m pytest tests -m gpu -q 2>&1 | grep -v '^\[Multi-H\]\|^Median\|^$' | tail -8 > gpurun_out/ev/tests.log
python bench.py > gpurun_out/ev/bench.json 2> gpurun_out/ev/bench.err
bash tools/profile_bench.sh r03 $SHA > gpurun_out/ev/profile_bench.log 2>&1
MH_LIB=multi-h_amd/libmultih_hip_tuning.so RV=32,20,22,34,0,3,10,7 OUT=ev/energy.json python tools/energy_probe.py > gpurun_out/ev/energy.log 2>&1
python tools/score_bench.py > gpurun_out/ev/score_bench.txt 2>&1
CPU=1 python tools/label_bench.py > gpurun_out/ev/label_bench.txt 2>&1
bash tools/profile_label.sh r03_label $SHA > gpurun_out/ev/profile_label.log 2>&1
python tools/cascade_sweep.py > gpurun_out/ev/cascade_sweep.txt 2>&1
MULTIH_TIMING=1 python tools/small_scene_bench.py > gpurun_out/ev/small_scenes.txt 2>&1
MULTIH_TIMING=1 python tools/loop_bench.py > gpurun_out/ev/loop_timing.txt 2>&1
MULTIH_TIMING=1 INIT=stable python tools/loop_bench.py > gpurun_out/ev/loop_timing_reference_init.txt 2>&1
ITER_HYP=100000 python tools/loop_bench.py > gpurun_out/ev/loop_reproposal.txt 2>&1
SECONDS=150 SEED=3 python tools/stress_parity.py > gpurun_out/ev/stress_parity.txt 2>&1
SECONDS=200 SEED=3 python tools/stress_alternation.py > gpurun_out/ev/stress_process.txt 2>&1
SECONDS=60 python tools/stress_residual_edges.py > gpurun_out/ev/stress_residual_edges.txt 2>&1
tail -n 3 gpurun_out/ev/tests.log gpurun_out/ev/stress_parity.txt gpurun_out/ev/stress_process.txt gpurun_out/ev/stress_residual_edges.txt
