#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of the bench command.
# Writes everything under gpurun_out/ (scratch); tools/summarize_profile.py turns it into profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the driver's command shape (20 steps = 1 280 images -> 14 batches of 92 images) without the informational legs
CMD="python $ROOT/bench.py --steps ${PROF_STEPS:-20} --warmup ${PROF_WARMUP:-5} --no-cpu-baseline --no-full-pool --no-f16x3 --no-train --no-cfg4"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1
# counters in their own runs (no other trace domains): HBM traffic of every dispatch
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
# (the three-counter pass serialises the dispatches: at 20 steps it runs past ten minutes -- 6 steps give the same per-kernel ratios)
CMD6="python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-pool --no-f16x3 --no-train --no-cfg4"
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -- $CMD6 > $OUT/pmc_mfma.log 2>&1
find $OUT -name "*.csv" | head -20
# keep the merge small: drop the per-dispatch kernel traces of the pmc runs, keep counter_collection
find $OUT -name "*kernel_trace.csv" -path "*pmc_*" -delete
du -sh $OUT
