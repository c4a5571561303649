"""Per-layer-class summary of a CALD_PROFILE_DUMP launch log (bench.py): time, share and TFLOP/s per (Cin, Cout, filter, stride, group)."""
import collections
import csv
import re
import sys


def main(path, top=40):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        k = re.sub(r"mt=\d+,", "", r["desc"])
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1; a[1] += float(r["ms"]); a[2] += float(r["gflop"])
    tot = sum(a[1] for a in agg.values()); fl = sum(a[2] for a in agg.values())
    print("%s: %d launches, %.1f ms, %.1f TFLOP/s overall" % (path, sum(a[0] for a in agg.values()), tot, fl / tot))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("  %-46s n=%5d %9.2f ms %5.1f%%  %7.1f TF" % (k[:46], a[0], a[1], 100 * a[1] / tot, a[2] / a[1]))


if __name__ == "__main__":
    for p in sys.argv[1:]:
        main(p)
