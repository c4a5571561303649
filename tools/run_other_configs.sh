#!/bin/bash
# Informational runs of the other BASELINE.json configs on one MI355X (per GPU): images/s and GEMM TFLOP/s, both precisions.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/other_configs_$1.jsonl
: > $OUT
run() { python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-full-pool --no-f16x3 "$@" 2>/dev/null | tail -1 >> $OUT; }
for P in fp32 f16x3; do
  run --model retinanet --shape voc --augs FCD --precision $P          # configs[2]
  run --model frcnn --shape coco --augs FCD --precision $P             # configs[3] (per GPU)
  run --model frcnn101 --shape coco --augs FCDRG --precision $P        # configs[4] (per GPU)
  run --model frcnn --shape voc --augs FCDR --precision $P             # reference default --augs
done
python3 - <<PY
import json
for l in open("$OUT"):
    d = json.loads(l)
    print("%-95s %-6s %7.1f img/s  GEMM %6.1f TF-eq" % (d["config"]["workload"][15:110], d["dtype"][:5], d["value"], d["roofline"]["achieved"]))
PY
