#!/usr/bin/env python3
"""The residual-tail data gradient (uem_conv2d_dgrad_tail) of the four ResNet50 stages at B=32, 512x512: the 1x1 conv1 of a bottleneck
taken backwards (dy: M x C/4 -> dx: M x C) with the identity gradient, its gate bits and the previous block's bn3 reduction in the
epilogue, option by option, against the bytes each variant moves."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uemda_amd import ops

import ctypes

from uemda_amd import _lib

lib = _lib.load()
HOOKS = hasattr(lib, "uemdbg_conv_persist")
if HOOKS:
    lib.uemdbg_conv_config.argtypes = [ctypes.c_int] * 2
    lib.uemdbg_conv_config.restype = None
    lib.uemdbg_conv_persist.argtypes = [ctypes.c_int]
    lib.uemdbg_conv_persist.restype = None
B = int(os.environ.get("B", "32"))
SHAPES = [("layer1", 128, 64, 256), ("layer2", 64, 128, 512), ("layer3", 32, 256, 1024), ("layer4", 32, 512, 2048)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


print(f"{'stage':8s} {'variant':34s} {'ms':>7s} {'TF/s':>7s} {'GB moved':>9s} {'TB/s':>6s}")
for name, hw, cmid, cout in SHAPES:
    M = B * hw * hw
    dy = torch.randn(B, hw, hw, cmid, device="cuda")
    w_t = torch.randn(cout, 1, 1, cmid, device="cuda") * 0.05
    acc = torch.randn(B, hw, hw, cout, device="cuda")
    z = torch.randn(B, hw, hw, cout, device="cuda")
    bits = torch.randint(0, 2 ** 31, (M * cout // 32,), device="cuda", dtype=torch.int32)
    vec = torch.rand(4, cout, device="cuda") + 0.5
    out = torch.empty(B, hw, hw, cout, device="cuda")
    shape = (B, hw, hw, cout)
    unit = 4.0 * M * cout / 1e9
    small = 4.0 * M * cmid / 1e9
    flops = 2.0 * M * cmid * cout
    variants = [
        ("plain dgrad", lambda: ops.conv2d_dgrad(dy, w_t, shape, out=out), unit + small),
        ("tail: + identity*bits", lambda: ops.conv2d_dgrad_tail(dy, w_t, shape, acc_src=acc, acc_bits=bits, out=out), 2 * unit + small),
        ("tail: + identity*bits + bn3 sums", lambda: ops.conv2d_dgrad_tail(dy, w_t, shape, acc_src=acc, acc_bits=bits, out=out, bn_z=z, bn_vec=vec,
                                                                          bn_bits=bits), 3 * unit + small),
        ("tail: bn3 sums only", lambda: ops.conv2d_dgrad_tail(dy, w_t, shape, out=out, bn_z=z, bn_vec=vec, bn_bits=bits), 2 * unit + small),
        ("torch: out = acc + z (3 streams)", lambda: torch.add(acc, z, out=out), 3 * unit),
    ]
    for label, fn, gb in variants:
        ms = timeit(fn)
        print(f"{name:8s} {label:34s} {ms:7.3f} {flops / ms / 1e9:7.1f} {gb:9.3f} {gb / ms:6.2f}")
    if HOOKS:
        # launch geometry of the full tail: persistent blocks on / off, channels per k-step, N tile (DEBUG_HOOKS build only)
        for persist in (0, 1):
            for kb in (16, 32):
                for bn in (64, 128):
                    if persist and kb == 16:
                        continue
                    lib.uemdbg_conv_persist(persist)
                    lib.uemdbg_conv_config(1, bn + 1000 * kb)
                    ms = timeit(variants[2][1])
                    ms2 = timeit(variants[3][1])
                    print(f"{name:8s}   persist={persist} k-step={kb} BN={bn}: full tail {ms:7.3f} ms ({flops / ms / 1e9:6.1f} TF/s)   bn3 sums only {ms2:7.3f} ms")
        lib.uemdbg_conv_persist(-1)
        lib.uemdbg_conv_config(-1, 0)
