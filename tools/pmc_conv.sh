#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of one microbench layer under a kernel variant:  tools/pmc_conv.sh <tag> <layer substring> [ENV=VAL ...]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; LAYER=$2; shift 2
OUT=$ROOT/gpurun_out/pmcconv_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -- python $ROOT/tools/bench_conv.py $TAG "$LAYER" > $OUT/$C.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                a = agg[r["Kernel_Name"].split("(")[0][:70]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (s, n) in agg.items():
        if "conv" in k:
            print("$TAG %-12s %-60s dispatches %3d  avg %.3f GB (raw counter, KB units)" % (c, k, n, s / n / 1e6))
PY
find $OUT -name "*.csv" -delete
