#!/usr/bin/env python3
"""Launches per STEADY-STATE step by kernel name: two `kernel_stats.csv` files of the same bench.py command that differ only in
--steps (a and b), counts subtracted and divided by the step difference -- the set-up launches (model.cuda() is ~330 copyBuffer
launches, the zero-fills of the arenas, the one-time weight transposes) cancel.
usage: launches_per_step.py short.csv long.csv <steps_long - steps_short>"""
import csv
import sys


def load(p):
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationUs"])) for r in csv.DictReader(open(p))}


a, b, d = load(sys.argv[1]), load(sys.argv[2]), int(sys.argv[3])
rows = []
for name, (cb, ub) in b.items():
    ca, ua = a.get(name, (0, 0.0))
    if cb != ca:
        rows.append(((cb - ca) / d, (ub - ua) / d, name))
plumbing = ("copyBuffer", "fillBuffer", "at::native", "weight_transpose")
tot = sum(r[0] for r in rows)
pl = [r for r in rows if any(k in r[2] for k in plumbing)]
print(f"launches per step: {tot:.1f}; device time per step {sum(r[1] for r in rows) / 1e3:.2f} ms")
print(f"plumbing launches per step (copyBuffer / fillBuffer / torch elementwise / weight_transpose): {sum(r[0] for r in pl):.1f}, "
      f"{sum(r[1] for r in pl) / 1e3:.3f} ms")
for n, us, name in sorted(pl, key=lambda r: -r[0]):
    print(f"  {n:7.1f}  {us:9.1f} us  {name[:150]}")
print("everything else:")
for n, us, name in sorted((r for r in rows if r not in pl), key=lambda r: -r[1]):
    print(f"  {n:7.1f}  {us:9.1f} us  {name[:150]}")
