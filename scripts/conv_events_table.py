#!/usr/bin/env python3
"""Per-shape table of the conv launches INSIDE the step, from `bench.py --dump-conv-events f.json`:
(family, calling op, algorithmic GFLOP) -> launches, average ms, algorithmic / executed TFLOP/s, share of the step's conv time."""
import collections
import json
import sys

rows = json.load(open(sys.argv[1]))
agg = collections.OrderedDict()
for fam, who, fl, ex, ms in rows:
    a = agg.setdefault((fam, who, round(fl / 1e9, 2)), [0, 0.0, 0.0, 0.0])
    a[0] += 1
    a[1] += ms
    a[2] += fl
    a[3] += ex
tot = sum(a[1] for a in agg.values())
print(f"{'family':11s} {'op':48s} {'GFLOP':>8s} {'n':>4s} {'avg ms':>8s} {'sum ms':>8s} {'alg TF':>7s} {'exe TF':>7s} {'share':>6s}")
for (fam, who, gf), (n, ms, fl, ex) in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[1][1])):
    print(f"{fam:11s} {who:48s} {gf:8.2f} {n:4d} {ms / n:8.3f} {ms:8.2f} {fl / ms / 1e9:7.1f} {ex / ms / 1e9:7.1f} {100 * ms / tot:5.1f}%")
print(f"total conv time in the step: {tot:.2f} ms")
