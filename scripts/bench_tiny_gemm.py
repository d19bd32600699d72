#!/usr/bin/env python3
"""The PPM branch convs: 1x1, 2048 -> 512 on pooled maps (32 x s x s rows, s = 1, 2, 3, 6) -- GEMMs with a handful of row tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uemda_amd import ops
from bench_conv_shapes import timeit

for s in (1, 2, 3, 6):
    x = torch.randn(32, s, s, 2048, device="cuda")
    w = torch.randn(512, 1, 1, 2048, device="cuda") * 0.02
    y = ops.conv2d(x, w)
    dy = torch.randn_like(y)
    wt = ops.weight_transpose(w)
    dw = torch.zeros_like(w)
    tf = timeit(lambda: ops.conv2d(x, w), 5)
    td = timeit(lambda: ops.conv2d_dgrad(dy, wt, x.shape), 5)
    tw = timeit(lambda: ops.conv2d_wgrad(x, dy, dw), 5)
    print(f"s={s} M={32 * s * s:5d}: fwd {tf * 1e3:8.1f} us  dgrad {td * 1e3:8.1f} us  wgrad {tw * 1e3:8.1f} us")

# the PPM head's big convs
x = torch.randn(32, 32, 32, 4096, device="cuda")
w = torch.randn(512, 3, 3, 4096, device="cuda") * 0.01
tf = timeit(lambda: ops.conv2d(x, w, pad=1), 3)
print(f"conv0 3x3 4096->512 fwd {tf:8.3f} ms = {2 * 32768 * 512 * 9 * 4096 / tf / 1e9:6.1f} TFLOP/s")
a = torch.randn(32, 32, 32, 512, device="cuda")
w4 = torch.randn(32, 1, 1, 512, device="cuda") * 0.01
b4 = torch.zeros(32, device="cuda")
tf = timeit(lambda: ops.conv2d(a, w4, b4, algo_cout=6), 3)
print(f"conv4 1x1 512->32 fwd {tf * 1e3:8.1f} us")
for s in (1, 2, 3, 6):
    feat = torch.randn(32, 32, 32, 2048, device="cuda")
    from uemda_amd.models.ppm import _avgpool
    tp = timeit(lambda: _avgpool(feat, s), 3)
    print(f"adaptive avgpool s={s}: {tp * 1e3:8.1f} us")
