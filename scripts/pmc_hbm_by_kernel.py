#!/usr/bin/env python3
"""HBM bytes per kernel NAME from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, KiB units, FETCH_SIZE
doubled on gfx950 as MI355X_MICROARCH.md prescribes) joined with the average durations of a --kernel-trace --stats run of the
same command (scripts/kernel_stats.py CSV): MB per launch, achieved GB/s, share of the step's HBM traffic.
    pmc_hbm_by_kernel.py <fetch_dir> <write_dir> <kernel_stats.csv> [top N]"""
import csv
import glob
import os
import sys


def collect(dirname, counter):
    agg = {}
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = agg.setdefault(r["Kernel_Name"], [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


def short(name):
    return name if len(name) <= 70 else name[:67] + "..."


def main():
    fetch_dir, write_dir, stats = sys.argv[1], sys.argv[2], sys.argv[3]
    top = int(sys.argv[4]) if len(sys.argv) > 4 else 25
    fe, wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    dur = {}
    for r in csv.DictReader(open(stats)):
        dur[r["Name"]] = (int(r["Calls"]), float(r["AverageUs"]))
    rows, total = [], 0.0
    for k, (n, v) in fe.items():
        nw, vw = wr.get(k, (0, 0.0))
        fetch = 2.0 * v * 1024
        write = vw * 1024
        rows.append((fetch + write, k, n, fetch / n, write / max(nw, 1)))
        total += fetch + write
    rows.sort(reverse=True)
    print(f"total HBM bytes over the profiled run: {total / 1e9:.2f} GB (fetch x2-corrected + write)")
    print(f"{'kernel':70s} {'launches':>8s} {'rd MB':>9s} {'wr MB':>9s} {'avg us':>8s} {'GB/s':>7s} {'share':>6s}")
    for tot, k, n, f, w in rows[:top]:
        d = dur.get(k, (0, 0.0))[1]
        gbs = (f + w) / (d * 1e-6) / 1e9 if d > 0 else 0.0
        print(f"{short(k):70s} {n:8d} {f / 1e6:9.2f} {w / 1e6:9.2f} {d:8.1f} {gbs:7.0f} {100 * tot / total:5.1f}%")


if __name__ == "__main__":
    main()
