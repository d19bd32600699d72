#!/usr/bin/env python3
"""Forward / data-gradient LDS-DMA conv kernels over the ResNet-50 shape inventory: one tile per block (round 2) against persistent
blocks with cross-tile operand prefetch (round 3), variants of a shape interleaved in ONE process; outputs compared bit for bit.
TFLOP/s per (shape, variant)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uemda_amd import _lib, ops
from bench_conv_shapes import SHAPES

lib = _lib.load()
lib.uemdbg_conv_config.argtypes = [ctypes.c_int] * 2
lib.uemdbg_conv_config.restype = None
lib.uemdbg_conv_persist.argtypes = [ctypes.c_int]
lib.uemdbg_conv_persist.restype = None


def timeit(f, reps=3):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    B = int(os.environ.get("B", "32"))
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    # (persist, bn + 1000 * channels per k-step)
    variants = [(0, 0), (1, 0), (1, 16000), (0, 32000)]
    names = ["tile/blk", "persist", "persist k16", "tile/blk k32"]
    print("shape".ljust(24) + "".join(f"fwd {n}".rjust(17) for n in names) + "".join(f"dgrad {n}".rjust(19) for n in names[:2]))
    tot = {(k, v): 0.0 for k in ("f", "d") for v in variants}
    for name, cin, cout, k, s, d, hin, cnt in SHAPES:
        if "aspp" in name or (only and only not in name):
            continue
        pad = d * (k - 1) // 2
        x = torch.randn(B, hin, hin, cin, device="cuda")
        w = torch.randn(cout, k, k, cin, device="cuda") * 0.05
        sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1
        y = ops.conv2d(x, w, stride=s, pad=pad, dil=d)
        dy = torch.randn_like(y)
        wt = ops.weight_transpose(w)
        flops = 2.0 * y.numel() * k * k * cin
        bf, bd = {}, {}
        ref_f = ref_d = None
        for rnd in range(3):
            for v in variants:
                lib.uemdbg_conv_persist(v[0])
                lib.uemdbg_conv_config(1, v[1])
                fw = lambda: ops.conv2d(x, w, stride=s, pad=pad, dil=d, in_scale=sc, in_shift=sh, in_relu=True, out=y)
                dg = lambda: ops.conv2d_dgrad(dy, wt, x.shape, stride=s, pad=pad, dil=d)
                if rnd == 0:
                    of, od = fw().clone(), dg().clone()
                    if ref_f is None:
                        ref_f, ref_d = of, od
                    else:
                        assert torch.equal(of, ref_f), (name, v, "forward differs", float((of - ref_f).abs().max()))
                        assert torch.equal(od, ref_d), (name, v, "dgrad differs", float((od - ref_d).abs().max()))
                bf[v] = min(bf.get(v, 1e9), timeit(fw))
                if v[1] == 0:
                    bd[v] = min(bd.get(v, 1e9), timeit(dg))
        for v in variants:
            tot[("f", v)] += cnt * bf[v]
            if v in bd:
                tot[("d", v)] += cnt * bd[v]
        print(name.ljust(24) + "".join(f"{flops / bf[v] / 1e9:17.1f}" for v in variants) + "".join(f"{flops / bd[v] / 1e9:19.1f}" for v in variants[:2]), flush=True)
    print("per-forward ms".ljust(24) + "".join(f"{tot[('f', v)]:17.2f}" for v in variants) + "".join(f"{tot[('d', v)]:19.2f}" for v in variants[:2]))
    lib.uemdbg_conv_config(-1, 0)
    lib.uemdbg_conv_persist(-1)


if __name__ == "__main__":
    main()
