#!/usr/bin/env python3
"""bf16 conv kernel: 128-row tiles (rule) against the 256-row tiles of round 5 forced on (uemdbg_conv_bf16_big), per ResNet shape at
the benchmark batch, interleaved in one process: forward with the BatchNorm tile statistics, plain data gradient.
    B=32 python scripts/sweep_conv_bf16_big.py [name filter]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uemda_amd import _lib, ops_bf16
from bench_conv_shapes import SHAPES, timeit


def main():
    B = int(os.environ.get("B", "32"))
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    lib = _lib.load()
    print(f"{'shape':24s} {'M':>8s} {'tiles256':>8s} | {'fwd 128':>8s} {'fwd 256':>8s} {'change':>7s} | {'dgrad 128':>9s} {'dgrad 256':>9s} {'change':>7s}")
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
        res = []
        for big in (0, 1):
            lib.uemdbg_conv_bf16_big(big)
            want = M % 128 == 0
            tf = timeit(lambda: ops_bf16.conv2d(x, w, stride=s, pad=pad, dil=d, out=y, want_stats=want), 5)
            td = timeit(lambda: ops_bf16.conv2d_dgrad(dy, wt, x.shape, stride=s, pad=pad, dil=d), 5)
            res.append((tf, td))
        lib.uemdbg_conv_bf16_big(-1)
        (f0, d0), (f1, d1) = res
        t256 = (M // 256) * max(1, cout // 128) if M % 256 == 0 else 0
        print(f"{name:24s} {M:8d} {t256:8d} | {f0:8.3f} {f1:8.3f} {100 * (f1 / f0 - 1):+6.1f}% | {d0:9.3f} {d1:9.3f} {100 * (d1 / d0 - 1):+6.1f}%")
        tot[0] += cnt * f0; tot[1] += cnt * f1; tot[2] += cnt * d0; tot[3] += cnt * d1
    print(f"per-forward totals (ms): fwd 128-row {tot[0]:.2f} / 256-row {tot[1]:.2f};  dgrad {tot[2]:.2f} / {tot[3]:.2f}")


if __name__ == "__main__":
    main()
