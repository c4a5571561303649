#!/bin/bash
# Same-box A/B of the training step: alternates environment settings, N rounds; prints ms per step of every run.
#   tools/ab_train.sh "A_ENV=1" "B_ENV=1 C_ENV=0" [rounds]
R=${3:-3}
for r in $(seq 1 $R); do
  for cfg in "$1" "$2"; do
    ms=$(env $cfg python tools/bench_train.py --steps 30 --warmup 6 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $r  [$cfg]  $ms ms"
  done
done
