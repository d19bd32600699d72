#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of ONE rank of a data-parallel rehearsal (scripts/dp_rehearsal.sh):
per-queue busy time, how much of the traced window this process had a kernel on the device, the longest gaps,
and the average duration of the conv families -- the evidence behind DESIGN.md's side-stream note.

    python scripts/dp_trace_report.py gpurun_out/dp_<tag>/prof [--last-ms 1500]
"""
import argparse
import csv
import glob
import os
import re


def family(name):
    if "conv_wgrad_kernel" in name:
        return "conv_wgrad"
    m = re.search(r"conv_fwd_kernel<\d+, \d+, \d+, (\d)", name)
    if m:
        return "conv_dgrad" if m.group(1) == "1" else "conv_fwd"
    if name.startswith("bn_") or "affine_act" in name:
        return "batchnorm"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--last-ms", type=float, default=1500.0, help="look at the last N ms of the trace (steady state)")
    args = ap.parse_args()
    files = glob.glob(os.path.join(args.dir, "**", "*kernel_trace.csv"), recursive=True)
    dbs = glob.glob(os.path.join(args.dir, "**", "*_results.db"), recursive=True)
    assert files or dbs, "no kernel_trace.csv / results.db under " + args.dir
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
    if not files:                                   # ROCm 7.2 default output: the rocpd sqlite database
        import sqlite3
        for f in dbs:
            for s, e, q, n in sqlite3.connect(f).execute("select start, end, queue_id, name from kernels"):
                rows.append((int(s), int(e), str(q), n))
    rows.sort()
    t_end = max(r[1] for r in rows)
    t0 = t_end - int(args.last_ms * 1e6)
    rows = [r for r in rows if r[0] >= t0]
    span = (t_end - rows[0][0]) / 1e6
    print(f"window: last {span:.1f} ms, {len(rows)} kernel dispatches of this process")
    # per-queue busy time and union busy time
    by_q = {}
    for s, e, q, n in rows:
        by_q.setdefault(q, []).append((s, e))
    for q, iv in sorted(by_q.items()):
        print(f"  queue {q}: {len(iv):6d} kernels, sum of durations {sum(e - s for s, e in iv) / 1e6:9.1f} ms")
    merged, cur_s, cur_e = [], None, None
    for s, e, _, _ in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                merged.append((cur_s, cur_e))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    merged.append((cur_s, cur_e))
    busy = sum(e - s for s, e in merged) / 1e6
    print(f"  a kernel of this process on the device: {busy:.1f} ms = {100 * busy / span:.1f} % of the window")
    gaps = sorted(((merged[i + 1][0] - merged[i][1]) / 1e6, merged[i][1]) for i in range(len(merged) - 1))[::-1][:8]
    print("  longest gaps with no kernel of this process running (ms): " + ", ".join(f"{g:.2f}" for g, _ in gaps))
    print(f"  gaps > 0.2 ms: {sum(1 for i in range(len(merged) - 1) if merged[i + 1][0] - merged[i][1] > 2e5)}; total gap time "
          f"{span - busy:.1f} ms")
    fam = {}
    for s, e, q, n in rows:
        a = fam.setdefault(family(n), [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e6
    print("  family        launches   total ms   avg ms")
    for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k:12s} {c:9d} {t:10.1f} {t / c:8.4f}")
    # concurrency: time with >= 2 kernels of this process in flight (main + side stream)
    ev = sorted([(s, 1) for s, e, _, _ in rows] + [(e, -1) for s, e, _, _ in rows])
    depth, last, conc = 0, ev[0][0], 0
    for t, d in ev:
        if depth >= 2:
            conc += t - last
        depth += d
        last = t
    print(f"  >= 2 kernels of this process in flight: {conc / 1e6:.1f} ms")


if __name__ == "__main__":
    main()
