#!/usr/bin/env python3
"""bf16 conv kernel, persistent blocks: counted waits around the epilogue (uemdbg_conv_bf16_lazy 1) against the full waits (0), per
ResNet shape at the benchmark batch, interleaved in one process: forward with the BatchNorm tile statistics, plain data gradient.
    B=32 python scripts/sweep_conv_bf16_lazy.py [name filter]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uemda_amd import _lib, ops_bf16
from bench_conv_shapes import SHAPES, timeit


def main():
    B = int(os.environ.get("B", "32"))
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    lib = _lib.load()
    print(f"{'shape':24s} {'M':>8s} | {'fwd full':>8s} {'counted':>8s} {'change':>7s} | {'dgrad full':>10s} {'counted':>8s} {'change':>7s}")
    tot = [0.0, 0.0, 0.0, 0.0]
    for name, cin, cout, k, s, d, hin, cnt in SHAPES:
        if "aspp" in name or "stem" in name or (only and only not in name):
            continue
        pad = d * (k - 1) // 2
        x = torch.randn(B, hin, hin, cin, device="cuda").bfloat16()
        w = (torch.randn(cout, k, k, cin, device="cuda") * 0.05).bfloat16()
        y = ops_bf16.conv2d(x, w, stride=s, pad=pad, dil=d)
        dy = torch.randn_like(y)
        wt = w.permute(3, 1, 2, 0).contiguous()
        M = y.numel() // cout
        res, outs = [], []
        for lazy in (0, 1):
            lib.uemdbg_conv_bf16_lazy(lazy)
            want = M % 128 == 0
            tf = timeit(lambda: ops_bf16.conv2d(x, w, stride=s, pad=pad, dil=d, out=y, want_stats=want), 5)
            td = timeit(lambda: ops_bf16.conv2d_dgrad(dy, wt, x.shape, stride=s, pad=pad, dil=d), 5)
            o = ops_bf16.conv2d(x, w, stride=s, pad=pad, dil=d, want_stats=want)
            g = ops_bf16.conv2d_dgrad(dy, wt, x.shape, stride=s, pad=pad, dil=d)
            outs.append((o, g))
            res.append((tf, td))
        lib.uemdbg_conv_bf16_lazy(-1)
        same = all(torch.equal(a, b) if torch.is_tensor(a) else all(torch.equal(u, v) for u, v in zip(a, b) if torch.is_tensor(u))
                   for a, b in zip(outs[0], outs[1]))
        (f0, d0), (f1, d1) = res
        print(f"{name:24s} {M:8d} | {f0:8.3f} {f1:8.3f} {100 * (f1 / f0 - 1):+6.1f}% | {d0:10.3f} {d1:8.3f} {100 * (d1 / d0 - 1):+6.1f}%  {'bit-equal' if same else 'DIFFERENT'}")
        tot[0] += cnt * f0; tot[1] += cnt * f1; tot[2] += cnt * d0; tot[3] += cnt * d1
    print(f"per-forward totals (ms): fwd full waits {tot[0]:.2f} / counted {tot[1]:.2f};  dgrad {tot[2]:.2f} / {tot[3]:.2f}")


if __name__ == "__main__":
    main()
