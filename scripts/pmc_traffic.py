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
    """kernel name -> conv family, as ops.PROF attributes the launches (a Winograd conv is its transforms + its GEMM).  The Winograd
    input transforms carry the pass in their template arguments (<AFFINE, PASS>: 0 forward, 1 data gradient, 2 weight gradient)."""
    if "stem_fwd_kernel" in name:
        return "conv_fwd"
    if "stem_wgrad" in name:                                # the stem's weight-gradient kernel and the two folds of its per-block banks
        return "conv_wgrad"
    if "conv_wgrad_kernel" in name or "wgrad_dma_kernel" in name or re.search(r"wino4?_dy_kernel", name) or re.search(r"wino4?_filter_grad_kernel", name):
        return "conv_wgrad"
    m = re.search(r"conv_fwd_kernel<\d+, \d+, \d+, (\d)", name) or re.search(r"conv_dma_kernel<\d+, \d+, (\d)", name)   # <BN, KB, MODE, ...>
    if m:
        return "conv_dgrad" if m.group(1) == "1" else "conv_fwd"
    m = re.search(r"wino4?_input_kernel<\w+, (\d)>", name)
    if m:
        return ("conv_fwd", "conv_dgrad", "conv_wgrad")[int(m.group(1))]
    m = re.search(r"wino4?_filter_kernel<(\w+)>", name)
    if m:
        return "conv_dgrad" if m.group(1) in ("true", "1") else "conv_fwd"
    m = re.search(r"wino4?_output_kernel<(\d)>", name)      # <0>: plain (eval-mode forward; the PPM head's data gradient, counted as forward here)
    if m:
        return "conv_dgrad" if m.group(1) == "2" else "conv_fwd"
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
    # usage: pmc_traffic.py <fetch_dir> <write_dir> <steps profiled> "<config key>"
    #   (key = bench.py's: "<model>-<head> <workload> B=.. size=.. prec=..").  Output: HBM bytes per STEP and family (bench.py divides
    #   by its launches per step), stamped with the hash of the conv kernel sources they were measured on.
    sys.path.insert(0, ROOT)
    import bench
    fetch_dir, write_dir, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    config = sys.argv[4] if len(sys.argv) > 4 else "resnet50-aspp ssl B=32 size=512 prec=fp32"
    fe, wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    out = {"config": config, "kernel_source_sha256_16": bench.kernel_source_hash(), "steps_profiled": steps, "per_step": {}}
    for fam in fe:
        nf, vf = fe[fam]
        nw, vw = wr.get(fam, (1, 0.0))
        fetch_b = 2.0 * vf * 1024 / steps       # gfx950: doubled (see docstring)
        write_b = vw * 1024 / steps
        out["per_step"][fam] = round(fetch_b + write_b)
        print(f"{fam:11s} kernel launches/step={nf / steps:7.1f}  fetch/step={fetch_b / 1e9:8.3f} GB (corrected x2)  write/step={write_b / 1e9:8.3f} GB")
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
