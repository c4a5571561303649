#!/bin/bash
# Runs on the GPU box (via gpurun): same-run HIP-event vs kernel-trace comparison of the GEMM launches (tools/event_vs_trace.py),
# Usage: tools/profile_event_vs_trace.sh <tag>   (tools/profile_gpu.sh <tag> takes the stats + PMC passes)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full-pool --no-f16x3 --no-train --no-cfg4"
CALD_PROFILE_DUMP=$OUT/launches.csv timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
python $ROOT/tools/event_vs_trace.py $OUT/launches.csv $OUT/trace $OUT/event_vs_trace.json > $OUT/event_vs_trace.txt 2>&1
find $OUT/trace -name "*kernel_trace.csv" -delete
CALD_PROFILE_DUMP=$OUT/launches_plain.csv $CMD > $OUT/bench_plain.json 2> $OUT/bench_plain.err
cat $OUT/event_vs_trace.txt | head -40
