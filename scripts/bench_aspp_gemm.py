#!/usr/bin/env python3
"""The ASPP heads' GEMM (2 heads x 4 dilations x 9 taps x C columns = 432 -> 448 at C = 6; K = 2048; M = B*32*32) under the conv
kernel variants: forward, data gradient, weight gradient; TFLOP/s on the 432 useful columns."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uemda_amd import _lib, ops

lib = _lib.load()
lib.uemdbg_conv_config.argtypes = [ctypes.c_int] * 2
lib.uemdbg_conv_config.restype = None
lib.uemdbg_conv_persist.argtypes = [ctypes.c_int]
lib.uemdbg_conv_persist.restype = None


def timeit(f, reps=5):
    f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


B = int(os.environ.get("B", "32"))
feat = torch.randn(B, 32, 32, 2048, device="cuda")
for R in (448, 512):
    wall = torch.randn(R, 1, 1, 2048, device="cuda") * 0.01
    G = ops.conv2d(feat, wall)
    dG = torch.randn_like(G)
    dw = torch.zeros_like(wall)
    wt = ops.weight_transpose(wall)
    fl = 2.0 * B * 1024 * 432 * 2048
    for persist, cfg, name in ((0, 0, "tile/blk rule"), (1, 0, "persist"), (0, 32000, "tile/blk k32"), (0, 64, "tile/blk bn64"), (1, 64, "persist bn64")):
        lib.uemdbg_conv_persist(persist)
        lib.uemdbg_conv_config(1, cfg)
        tf = timeit(lambda: ops.conv2d(feat, wall, out=G))
        td = timeit(lambda: ops.conv2d_dgrad(dG, wt, feat.shape))
        print(f"R={R} {name:16s} fwd {tf:.3f} ms {fl / tf / 1e9:6.1f} TF/s | dgrad {td:.3f} ms {fl / td / 1e9:6.1f} TF/s")
    lib.uemdbg_conv_persist(-1)
    lib.uemdbg_conv_config(-1, 0)
    tw = timeit(lambda: ops.conv2d_wgrad(feat, dG, dw))
    print(f"R={R} wgrad {tw:.3f} ms {fl / tw / 1e9:6.1f} TF/s")
