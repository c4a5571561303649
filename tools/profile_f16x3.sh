#!/bin/bash
# rocprofv3 passes of the f16x3 mode on BASELINE configs[4] (R101, COCO shapes, 6 views): kernel stats + MFMA-busy counters per kernel.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full-pool --no-f16x3 --no-train --model frcnn101 --shape coco --augs FCDRG --precision f16x3"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- $CMD > $OUT/pmc_mfma.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_lds -- $CMD > $OUT/pmc_lds.log 2>&1
find $OUT -name "*kernel_trace.csv" -path "*pmc_*" -delete
du -sh $OUT
