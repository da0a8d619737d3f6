# Round-5 evidence session on the GPU box: tests, bench, rocprof passes of the bench command, the PMC passes over k_cost32,
# the timeline of the exchange beside the sweep (one-rank RCCL communicator), labeling / loop / small-scene / mean-shift timings.
#   tools/evidence_r05.sh <head-sha>
set -x
SHA=${1:-unknown}
export TMPDIR=/tmp
E=gpurun_out/ev5
rm -rf $E; mkdir -p $E
python -m pytest tests -m gpu -q 2>&1 | grep -v '^\[Multi-H\]\|^Median\|^Iteration\|^$' | tail -8 > $E/tests.log
python bench.py > $E/bench.json 2> $E/bench.err
bash tools/profile_bench.sh r05 $SHA > $E/profile_bench.log 2>&1
# --- k_cost32: where the 0.9 ms beyond its store stream go ---
C=$E/cost32_pmc; mkdir -p $C
rocprofv3 --kernel-trace --stats --output-format csv -d $C/trace -- python3 tools/cost32_pmc_driver.py > /dev/null 2> $C/trace.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $C/sq1 -- python3 tools/cost32_pmc_driver.py > /dev/null 2> $C/sq1.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --output-format csv -d $C/sq2 -- python3 tools/cost32_pmc_driver.py > /dev/null 2> $C/sq2.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $C/write -- python3 tools/cost32_pmc_driver.py > /dev/null 2> $C/write.err
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $C/wrreq -- python3 tools/cost32_pmc_driver.py > /dev/null 2> $C/wrreq.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $C/fetch -- python3 tools/cost32_pmc_driver.py > /dev/null 2> $C/fetch.err
python3 tools/cost32_pmc_summary.py $C > $E/cost32_pmc.txt 2>&1
grep -h "k_cost32" $(find $C/trace -name "*kernel_stats.csv" | head -1) >> $E/cost32_pmc.txt
# --- the exchange beside the sweep: one-rank native communicator, 12 500-hypothesis steps ---
SIZES=12500 STEPS=20 WARM=3 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $E/tl_rccl -- python3 tools/shard_proxy.py > $E/tl_rccl.log 2>&1
python3 tools/timeline.py $(find $E/tl_rccl -name "*kernel_trace.csv" | head -1) -6 3 $(find $E/tl_rccl -name "*memory_copy_trace.csv" | head -1) > $E/rccl_beside_sweep.txt 2>&1
# --- labeling, loop, scenes, mean shift ---
python tools/label_bench.py > $E/label_bench.txt 2>&1
LEGACY=1 python tools/label_bench.py > $E/label_bench_r04_scene.txt 2>&1
MULTIH_TIMING=1 python tools/loop_bench.py > $E/loop_timing.txt 2>&1
MULTIH_TIMING=1 INIT=stable python tools/loop_bench.py > $E/loop_timing_reference_init.txt 2>&1
ITER_HYP=100000 python tools/loop_bench.py > $E/loop_reproposal.txt 2>&1
python tools/small_scene_bench.py > $E/small_scenes.txt 2>&1
MS_INDEXED=0 MS_PERSIST=0 python tools/meanshift_probe.py > $E/meanshift_probe_r04_schedule.txt 2>&1
MS_INDEXED=0 python tools/meanshift_probe.py > $E/meanshift_probe_persistent_schedule.txt 2>&1
MULTIH_MS_STATS=1 python tools/meanshift_probe.py > $E/meanshift_probe.txt 2>&1
python tools/barrsmith_agreement.py > $E/barrsmith.txt 2>&1
python tools/at_size_alternation.py > $E/at_size_init.txt 2>&1
find $E -name "*.csv" -size +8M -delete
tail -3 $E/tests.log
