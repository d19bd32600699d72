#!/usr/bin/env python3
"""bf16-storage conv kernels over the ResNet shape inventory at the benchmark batch: TFLOP/s against the dense bf16 MFMA
peak (2500) and the HBM GB/s of each launch (bf16 activations: these layers are HBM-co-bound long before the matrix peak)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uemda_amd import ops_bf16
from bench_conv_shapes import SHAPES, timeit


def main():
    B = int(os.environ.get("B", "32"))
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    print(f"{'shape':24s} {'M':>8s} | {'fwd ms':>8s} {'TF/s':>7s} {'GB/s':>6s} | {'dgrad ms':>8s} {'TF/s':>7s} | {'wgrad ms':>8s} {'TF/s':>7s}")
    tot = [0.0, 0.0, 0.0]
    for name, cin, cout, k, s, d, hin, cnt in SHAPES:
        if "aspp" in name or (only and only not in name):
            continue
        pad = d * (k - 1) // 2
        x = torch.randn(B, hin, hin, cin, device="cuda").bfloat16()
        w = (torch.randn(cout, k, k, cin, device="cuda") * 0.05).bfloat16()
        y = ops_bf16.conv2d(x, w, stride=s, pad=pad, dil=d)
        dy = torch.randn_like(y)
        wt = w.permute(3, 1, 2, 0).contiguous()
        M = y.numel() // cout
        flops = 2.0 * M * cout * k * k * cin
        t_f = timeit(lambda: ops_bf16.conv2d(x, w, stride=s, pad=pad, dil=d, out=y), 5)
        t_d = timeit(lambda: ops_bf16.conv2d_dgrad(dy, wt, x.shape, stride=s, pad=pad, dil=d), 5)
        t_w = float("nan")
        if hasattr(ops_bf16, "conv2d_wgrad"):
            dw = torch.zeros(cout, k, k, cin, device="cuda")
            t_w = timeit(lambda: ops_bf16.conv2d_wgrad(x, dy, dw, stride=s, pad=pad, dil=d), 5)
        gbs = (x.numel() + y.numel() + w.numel()) * 2 / (t_f * 1e-3) / 1e9
        print(f"{name:24s} {M:8d} | {t_f:8.3f} {flops / t_f / 1e9:7.1f} {gbs:6.0f} | {t_d:8.3f} {flops / t_d / 1e9:7.1f} | {t_w:8.3f} {flops / t_w / 1e9:7.1f}")
        tot[0] += cnt * t_f; tot[1] += cnt * t_d; tot[2] += cnt * t_w
    print(f"per-forward totals (ms): fwd={tot[0]:.2f}  dgrad={tot[1]:.2f}  wgrad={tot[2]:.2f}")


if __name__ == "__main__":
    main()
