#!/usr/bin/env python3
"""Diagnostic: LDS-DMA forward conv main loop on one shape with / without its DMA traffic and its operand prologue."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uemda_amd import _lib, ops
lib = _lib.load()
lib.uemdbg_conv_config.argtypes = [ctypes.c_int] * 2
lib.uemdbg_conv_dbg.argtypes = [ctypes.c_int]


def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for name, cin, cout, k, d, hin in (("l4 3x3 512", 512, 512, 3, 1, 32), ("l4 1x1 512->2048", 512, 2048, 1, 1, 32), ("l2 3x3 128", 128, 128, 3, 1, 64)):
    pad = d * (k - 1) // 2
    x = torch.randn(32, hin, hin, cin, device="cuda")
    w = torch.randn(cout, k, k, cin, device="cuda") * 0.05
    sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1
    y = ops.conv2d(x, w, pad=pad, dil=d)
    flops = 2.0 * y.numel() * k * k * cin
    for label, dma, dbg, aff in (("regs affine", 0, 0, True), ("regs plain", 0, 0, False), ("dma affine", 1, 0, True), ("dma plain", 1, 0, False),
                                 ("dma affine, no DMA in loop", 1, 1, True), ("dma plain, no DMA in loop", 1, 1, False)):
        lib.uemdbg_conv_config(dma, 0); lib.uemdbg_conv_dbg(dbg)
        kw = dict(in_scale=sc, in_shift=sh, in_relu=True) if aff else {}
        best = min(timeit(lambda: ops.conv2d(x, w, pad=pad, dil=d, out=y, **kw)) for _ in range(3))
        print(f"{name:18s} {label:30s} {flops / best / 1e9:7.1f} TFLOP/s")
lib.uemdbg_conv_config(-1, 0); lib.uemdbg_conv_dbg(0)
