#!/usr/bin/env python3
"""HBM bytes per conv launch BY SHAPE inside the step (a per-shape PMC table for the round's kernels).
Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs; KiB units; FETCH_SIZE doubled on gfx950, as
MI355X_MICROARCH.md prescribes) of
    UEM_PROF_MARK=1 python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-other-configs --no-hipgraph --dump-conv-events ev.json
where every profiled conv op of the one timed step is preceded by a one-element marker launch (`scale_kernel`): the dispatches between
two markers are that op's kernels (a Winograd op = transforms + GEMM), matched to the op list bench.py dumped in the same order.
    pmc_per_op.py <fetch_dir> <write_dir> <events.json>"""
import collections
import csv
import glob
import json
import os
import re
import sys

CONV = re.compile(r"conv_dma_kernel|conv_fwd_kernel|conv_wgrad_kernel|wgrad_dma_kernel|wino4?_|conv_bf16_kernel|wgrad_bf16_kernel|stem_fwd_kernel|stem_wgrad")


def ops_of(dirname, counter, nops):
    rows = []
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if r[1].startswith("scale_kernel")]
    assert len(marks) >= nops, (len(marks), nops)
    marks = marks[-nops:] + [len(rows)]
    out = []
    for a, b in zip(marks[:-1], marks[1:]):
        seg = [r for r in rows[a + 1:b] if CONV.search(r[1])]
        out.append((sum(r[2] for r in seg) * 1024.0, [r[1] for r in seg]))
    return out


def main():
    fetch_dir, write_dir, evf = sys.argv[1:4]
    ev = json.load(open(evf))
    fe, wr = ops_of(fetch_dir, "FETCH_SIZE", len(ev)), ops_of(write_dir, "WRITE_SIZE", len(ev))
    agg = collections.OrderedDict()
    for (fam, who, fl, ex, ms), (fb, names), (wb, _) in zip(ev, fe, wr):
        kern = max(names, key=lambda n: ("gemm" in n or "conv_dma" in n or "wgrad_dma" in n, len(n)))[:46] if names else "?"
        a = agg.setdefault((fam, who, round(fl / 1e9, 2)), [0, 0.0, 0.0, 0.0, kern, len(names)])
        a[0] += 1; a[1] += ms; a[2] += 2.0 * fb; a[3] += wb
    print(f"{'family':11s} {'op / input -> other side':58s} {'GFLOP':>7s} {'n':>3s} {'ms':>7s} {'rd MB':>8s} {'wr MB':>8s} {'alg MB':>8s} {'ratio':>6s} {'GB/s':>6s}  kernels")
    tot = collections.Counter()
    for (fam, who, gf), (n, ms, f, w, kern, nk) in sorted(agg.items(), key=lambda kv: (kv[0][0], -(kv[1][2] + kv[1][3]))):
        alg = float("nan")
        m = re.search(r" (\d+)x(\d+)x(\d+)x(\d+) ->(\d+) k(\d+)", who)
        if m:
            nb, h, wd, c, co, k = map(int, m.groups())
            px = nb * h * wd
            # the tensor named in the label + the tensor on the other side (its pixel count from the flops: strided layers) + the filter bank
            other_px = gf * 1e9 / (2.0 * co * k * k * c)
            alg = 4.0 * (px * c + other_px * co + co * c * k * k)
        per = (f + w) / n
        tot[fam] += f + w
        print(f"{fam:11s} {who:58s} {gf:7.2f} {n:3d} {ms / n:7.3f} {f / n / 1e6:8.1f} {w / n / 1e6:8.1f} {alg / 1e6:8.1f} {per / alg if alg == alg else float('nan'):6.2f} "
              f"{per / (ms / n * 1e-3) / 1e9:6.0f}  {nk}x {kern}")
    print("HBM bytes per step by family (GB): " + ", ".join(f"{k} {v / 1e9:.1f}" for k, v in tot.items()))


if __name__ == "__main__":
    main()
