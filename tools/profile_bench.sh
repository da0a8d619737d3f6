#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of the bench command and separate PMC passes for HBM
# traffic and instruction mix.  Outputs under gpurun_out/prof_<tag>/; the summary names the source revision it was taken
# from (HEAD sha passed in by the caller — .git does not travel to the box — and the hash of bench.py as it ran).
#   tools/profile_bench.sh <tag> <head-sha>
TAG=${1:-r04}
HEAD_SHA=${2:-unknown}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 1 --no-cpu-baseline --no-rehearsal"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-rehearsal > $OUT/bench_under_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_sq2.err
HEAD_SHA=$HEAD_SHA WARMUP=3 STEPS=20 python3 tools/summarize_prof.py $OUT | tee $OUT/summary.txt
