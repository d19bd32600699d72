#!/usr/bin/env python3
"""Per-kernel-family averages of rocprofv3 --pmc counters (CSV output: *counter_collection.csv, or the rocpd db).
    python scripts/pmc_summary.py <dir> [substring of the kernel name to keep]
Counters are summed over the dispatches of a family and divided by the dispatch count; durations come with them."""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for key in ("conv_wgrad_kernel", "conv_fwd_kernel", "wgrad_dma_kernel", "conv_dma_kernel", "stem_fwd_kernel", "stem_wgrad_kernel"):
        i = name.find(key)
        if i >= 0:
            return name[i:i + 60]
    return name[:60]


def main():
    d = sys.argv[1]
    keep = sys.argv[2] if len(sys.argv) > 2 else ""
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if keep and keep not in n:
                continue
            k = short(n)
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k in sorted(agg):
        print(f"{k}   avg duration {sum(dur[k]) / len(dur[k]):.1f} us")
        for c in sorted(agg[k]):
            print(f"    {c:36s} {agg[k][c] / cnt[k][c]:18.1f}   ({cnt[k][c]} dispatches)")


if __name__ == "__main__":
    main()
