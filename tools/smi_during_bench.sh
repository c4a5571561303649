#!/bin/bash
# Samples clocks / power / temperature while the headline sweep runs (is the fp32-MFMA rate bounded by the sustained clock?).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/smi_$1
mkdir -p $OUT
python $ROOT/bench.py --steps 40 --warmup 2 --no-full-pool --no-cpu-baseline --no-f16x3 > $OUT/bench.json 2> $OUT/bench.err &
BP=$!
sleep 6
for i in $(seq 1 30); do
  rocm-smi --showclocks --showpower --showtemp --showperflevel 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature|Performance" | tr '\n' ';' >> $OUT/smi.txt; echo >> $OUT/smi.txt
  sleep 1
done
wait $BP
sed -e 's/=*//g' -e 's/GPU\[0\]\s*: //g' -e 's/Temperature (Sensor \(junction\|memory\)) (C)/T_\1/g' $OUT/smi.txt | cut -c1-260
rocm-smi --showmaxpower 2>/dev/null | grep -i "max\|power" | head -5
