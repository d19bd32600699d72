#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes)
into per-launch HBM traffic of the conv kernel families -> profiles/traffic_latest.json (read by bench.py).
gfx950 correction: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streams => doubled;
both counters are in KiB."""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def family(name):
    if "conv_wgrad_kernel" in name or "wgrad_dma_kernel" in name:
        return "conv_wgrad"
    m = re.search(r"conv_fwd_kernel<\d+, \d+, \d+, (\d)", name) or re.search(r"conv_dma_kernel<\d+, \d+, (\d)", name)   # <BN, KB, MODE, ...>
    if m:
        return "conv_dgrad" if m.group(1) == "1" else "conv_fwd"
    return None


def collect(dirname, counter):
    agg = {}
    for f in glob.glob(os.path.join(dirname, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            fam = family(r["Kernel_Name"])
            if fam is None:
                continue
            a = agg.setdefault(fam, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


def main():
    # usage: pmc_traffic.py <fetch_dir> <write_dir> "<config key>"   (key = bench.py's: "<model>-<head> <workload> B=.. size=.. prec=..")
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    config = sys.argv[3] if len(sys.argv) > 3 else "resnet50-aspp ssl B=32 size=512 prec=fp32"
    fe, wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    out = {"config": config}
    for fam in fe:
        nf, vf = fe[fam]
        nw, vw = wr.get(fam, (1, 0.0))
        fetch_b = 2.0 * vf * 1024 / nf          # gfx950: doubled (see docstring)
        write_b = vw * 1024 / max(nw, 1)
        out[fam] = round(fetch_b + write_b)
        print(f"{fam:11s} launches={nf:5d}  fetch/launch={fetch_b / 1e6:9.2f} MB (corrected x2)  write/launch={write_b / 1e6:9.2f} MB")
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
