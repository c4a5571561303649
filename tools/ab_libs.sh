#!/bin/bash
# Same-box A/B of two builds of libcaldhip.so: cald_amd/lib/libcaldhip_base.so (kept aside before an edit) against the current one.
# Usage (on the GPU box): bash tools/ab_libs.sh <tag> [reps]   -> gpurun_out/<tag>_{conv,bench}_{base,new}_<rep>.txt
tag=${1:-ab}; reps=${2:-2}
mkdir -p gpurun_out
cp cald_amd/lib/libcaldhip.so /tmp/new.so
FL="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-full-pool --no-f16x3 --no-cfg4 --no-train"
for rep in $(seq 1 $reps); do
  for w in base new; do
    if [ $w = base ]; then cp cald_amd/lib/libcaldhip_base.so cald_amd/lib/libcaldhip.so; else cp /tmp/new.so cald_amd/lib/libcaldhip.so; fi
    python tools/bench_conv.py $w > gpurun_out/${tag}_conv_${w}_$rep.txt 2>&1
    python bench.py $FL > gpurun_out/${tag}_bench_${w}_$rep.txt 2>&1
  done
done
cp /tmp/new.so cald_amd/lib/libcaldhip.so
python - <<'P'
import glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("TAG", "ab")
P
for f in gpurun_out/${tag}_bench_*; do echo $f; tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'])"; done
paste <(grep TF gpurun_out/${tag}_conv_base_1.txt) <(grep TF gpurun_out/${tag}_conv_new_1.txt | awk '{print $(NF-3), $(NF-2), $(NF-1), $NF}')
