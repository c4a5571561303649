#!/bin/bash
# bench_conv (+ optionally bench.py) under a list of "VAR=VAL" settings on one box: bash tools/env_sweep.sh "A=1 B=2" <with_bench 0|1> [layer filter]
mkdir -p gpurun_out
FL="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-full-pool --no-f16x3 --no-cfg4 --no-train"
i=0
for KV in $1; do
  i=$((i+1))
  env $KV python tools/bench_conv.py $KV $3 > gpurun_out/env_conv_$i.txt 2>&1
  echo "== $KV"; grep TF gpurun_out/env_conv_$i.txt | awk '{printf "%s ", $(NF-1)} END{print ""}'
  if [ "$2" = "1" ]; then env $KV python bench.py $FL > gpurun_out/env_bench_$i.txt 2>&1; tail -1 gpurun_out/env_bench_$i.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'])"; fi
done
