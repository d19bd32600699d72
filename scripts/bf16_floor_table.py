#!/usr/bin/env python3
"""Per-shape floors of the bf16-storage conv launches from `bench.py --storage bf16 --dump-conv-events f.json`:
operand + result bytes of the convolution itself (bf16 activations, fused-epilogue streams of the residual tails counted), the time they
take at the 5.5 TB/s the elementwise passes of the same step reach, the MFMA time at 1.3 PFLOP/s (what the bf16 pipe sustains on random
data, MI355X_MICROARCH.md), and the measured time: where is the step's conv time above max(floors)?"""
import collections
import json
import sys

rows = json.load(open(sys.argv[1]))
BW, PF = 5.5e12, 1.3e15
agg = collections.OrderedDict()
for fam, who, fl, ex, ms in rows:
    a = agg.setdefault((fam, who, round(fl / 1e9, 2)), [0, 0.0])
    a[0] += 1
    a[1] += ms
out = []
for (fam, who, gf), (n, ms) in agg.items():
    parts = who.split()
    if len(parts) < 2 or "x" not in parts[1]:
        out.append((fam, who, gf, n, ms / n, 0.0, gf * 1e9 / PF * 1e3, ""))
        continue
    N, H, W, C = (int(v) for v in parts[1].split("x"))
    M = N * H * W
    ck2 = gf * 1e9 / (2.0 * M * C)              # other side's channels x taps
    k = 3 if abs(ck2 - 9 * C) < 1 else 1
    co = ck2 / (k * k)
    stride = 1
    tail = "tail" in who
    if fam == "conv_fwd":
        b = 2 * M * C + 2 * M * co
    elif fam == "conv_dgrad":
        b = 2 * M * C + 2 * M * co
        if tail:                                   # + identity gradient + previous z3 (+ bits): two more output-wide streams
            b += 2 * 2 * M * co
    else:
        b = 2 * M * C + 2 * M * co
    t_b = b / BW * 1e3
    t_f = gf * 1e9 / PF * 1e3
    out.append((fam, who, gf, n, ms / n, t_b, t_f, f"{C}->{int(co + .5)} k{k}"))
tot = sum(r[3] * r[4] for r in out)
tfl = sum(r[3] * max(r[5], r[6]) for r in out)
print(f"{'family':11s} {'op':40s} {'shape':14s} {'n':>3s} {'ms':>7s} {'t_hbm':>7s} {'t_mfma':>7s} {'x floor':>7s} {'excess ms':>9s}")
for r in sorted(out, key=lambda r: -(r[4] - max(r[5], r[6])) * r[3]):
    fl = max(r[5], r[6])
    print(f"{r[0]:11s} {r[1]:40s} {r[7]:14s} {r[3]:3d} {r[4]:7.3f} {r[5]:7.3f} {r[6]:7.3f} {r[4] / fl if fl else 0:7.2f} {(r[4] - fl) * r[3]:9.2f}")
print(f"total {tot:.2f} ms, sum of floors {tfl:.2f} ms")
