#!/usr/bin/env python3
"""Probe: do an MFMA-bound conv stream and an HBM-bound BatchNorm stream overlap when issued on two HIP streams?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uemda_amd import ops

B = 32
x = torch.randn(B, 32, 32, 512, device="cuda")
w = torch.randn(512, 3, 3, 512, device="cuda") * 0.02
y = torch.empty(B, 32, 32, 512, device="cuda")
z = torch.randn(B, 64, 64, 512, device="cuda")
dz = torch.randn_like(z)
out = torch.empty_like(z)
bn = torch.nn.BatchNorm2d(512).cuda()
st = ops.bn_stats(z, bn.weight.detach(), bn.bias.detach(), None, None, True)
gg, gb = torch.zeros(512, device="cuda"), torch.zeros(512, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
NC, NB = 20, 60


dyc = torch.randn(B, 32, 32, 512, device="cuda")
dw = torch.zeros_like(w)
KIND = sys.argv[1] if len(sys.argv) > 1 else "fwd"


def conv_once():
    if KIND == "wgrad":
        ops.conv2d_wgrad(x, dyc, dw, pad=1)
    else:
        ops.conv2d(x, w, pad=1, out=y)


def conv_loop():
    for _ in range(NC):
        conv_once()


def bn_loop():
    for _ in range(NB):
        ops.bn_backward(z, dz, st, gg, gb, None, True, dx=out)


def timed(fn):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b)


conv_loop(); bn_loop()
tc, tb = timed(conv_loop), timed(bn_loop)


def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        conv_loop()
    with torch.cuda.stream(s2):
        bn_loop()
    cur.wait_stream(s1); cur.wait_stream(s2)


def interleaved():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    for i in range(NC):
        with torch.cuda.stream(s1):
            conv_once()
        with torch.cuda.stream(s2):
            for _ in range(NB // NC):
                ops.bn_backward(z, dz, st, gg, gb, None, True, dx=out)
    cur.wait_stream(s1); cur.wait_stream(s2)


both()
t2 = timed(both)
t3 = timed(interleaved)
print(f"[{KIND}] conv alone {tc:.2f} ms, bn alone {tb:.2f} ms, sum {tc + tb:.2f} ms; two streams {t2:.2f} ms; interleaved issue {t3:.2f} ms")
