#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of three LabelingSteps at 50k sites / 11 labels
# (tools/label_bench.py) and one PMC pass over the alpha-expansion kernels.  Outputs under gpurun_out/prof_<tag>/.
TAG=${1:-r02_label}
HEAD_SHA=${2:-unknown}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
export CPU=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/label_bench.py > $OUT/label_bench_under_trace.txt 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_label -- python3 tools/label_bench.py > /dev/null 2> $OUT/pmc_label.err
python3 tools/label_bench.py > $OUT/label_bench.txt 2>&1
HEAD_SHA=$HEAD_SHA KERNELS=k_solve,k_move_setup,k_reduce,k_delta,k_energy,k_apply_pending STAT_ROWS=24 python3 tools/summarize_prof.py $OUT | tee $OUT/summary.txt
