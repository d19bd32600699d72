#!/usr/bin/env python3
"""Forward / data-gradient conv kernels over the ResNet-50 shape inventory: register-staged main loop vs the LDS-DMA
main loop (N tile 128 or 64), all variants of a shape interleaved in ONE process.  TFLOP/s per (shape, variant)."""
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


def timeit(f):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 3


def main():
    B = int(os.environ.get("B", "32"))
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    variants = [(0, 0), (1, 0), (1, 64), (1, 16000)]          # (dma, bn + 1000 * channels per k-step)
    names = ["regs", "dma", "dma bn64", "dma k16"]
    print("shape".ljust(24) + "".join(f"fwd {n}".rjust(14) for n in names) + "".join(f"dgrad {n}".rjust(15) for n in names))
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
        for rnd in range(3):
            for v in variants:
                lib.uemdbg_conv_config(*v)
                bf[v] = min(bf.get(v, 1e9), timeit(lambda: ops.conv2d(x, w, stride=s, pad=pad, dil=d, in_scale=sc, in_shift=sh, in_relu=True, out=y)))
                bd[v] = min(bd.get(v, 1e9), timeit(lambda: ops.conv2d_dgrad(dy, wt, x.shape, stride=s, pad=pad, dil=d)))
        for v in variants:
            tot[("f", v)] += cnt * bf[v]
            tot[("d", v)] += cnt * bd[v]
        print(name.ljust(24) + "".join(f"{flops / bf[v] / 1e9:14.1f}" for v in variants) + "".join(f"{flops / bd[v] / 1e9:15.1f}" for v in variants), flush=True)
    print("per-forward ms".ljust(24) + "".join(f"{tot[('f', v)]:14.2f}" for v in variants) + "".join(f"{tot[('d', v)]:15.2f}" for v in variants))
    lib.uemdbg_conv_config(-1, 0)


if __name__ == "__main__":
    main()
