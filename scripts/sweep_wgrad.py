#!/usr/bin/env python3
"""Tile / split-K sweep of the LDS-DMA weight-gradient kernel over the ResNet-50 shape inventory, all variants of a
shape interleaved in ONE process (cdna guide rule 24).  Prints TFLOP/s per (shape, tile, rounds)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uemda_amd import _lib, ops
from bench_conv_shapes import SHAPES

lib = _lib.load()
lib.uemdbg_wgrad_config.argtypes = [ctypes.c_int] * 4
lib.uemdbg_wgrad_config.restype = None


def main():
    B = int(os.environ.get("B", "32"))
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    variants = [(tm, tn, r, bk) for bk in (32, 16) for tm, tn in ((128, 64), (128, 128), (64, 64)) for r in (0, 1)]
    print("shape".ljust(24) + "".join(f"{tm}x{tn}r{r}k{bk}".rjust(13) for tm, tn, r, bk in variants))
    for name, cin, cout, k, s, d, hin, cnt in SHAPES:
        if "aspp" in name or (only and only not in name):
            continue
        pad = d * (k - 1) // 2
        x = torch.randn(B, hin, hin, cin, device="cuda")
        ho = ops.conv_out_size(hin, k, s, pad, d)
        dy = torch.randn(B, ho, ho, cout, device="cuda")
        dw = torch.zeros(cout, k, k, cin, device="cuda")
        sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1
        flops = 2.0 * B * ho * ho * cout * k * k * cin
        best = {}
        for rnd in range(3):
            for v in variants:
                tm, tn, r, bk = v
                if (tm == 128 and cout % 128) or (tn == 128 and cin % 128):
                    continue
                lib.uemdbg_wgrad_config(tm, tn, r, bk)
                f = lambda: ops.conv2d_wgrad(x, dy, dw, stride=s, pad=pad, dil=d, in_scale=sc, in_shift=sh, in_relu=True)
                f()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(3):
                    f()
                b.record()
                torch.cuda.synchronize()
                t = a.elapsed_time(b) / 3
                best[v] = min(best.get(v, 1e9), t)
        print(name.ljust(24) + "".join((f"{flops / best[v] / 1e9:13.1f}" if v in best else " " * 13) for v in variants), flush=True)
    lib.uemdbg_wgrad_config(0, 0, 0, 0)


if __name__ == "__main__":
    main()
