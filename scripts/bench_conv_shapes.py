#!/usr/bin/env python3
"""Per-shape microbenchmark of the conv kernels over the ResNet-50 (OS16) + ASPP shape inventory
 at the benchmark batch: achieved TFLOP/s (f32 MFMA peak 157.3) and the
algorithmic HBM GB/s of each launch, for forward / data-gradient / weight-gradient."""
import argparse
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uemda_amd import ops

SHAPES = [  # name, Cin, Cout, k, stride, dil, Hin (at 512^2 input), count per forward
    ("l1 1x1 64->64", 64, 64, 1, 1, 1, 128, 1), ("l1 3x3 64", 64, 64, 3, 1, 1, 128, 3),
    ("l1 1x1 64->256", 64, 256, 1, 1, 1, 128, 4), ("l1 1x1 256->64", 256, 64, 1, 1, 1, 128, 2),
    ("l2 1x1 256->128", 256, 128, 1, 1, 1, 128, 1), ("l2 3x3 128 s2", 128, 128, 3, 2, 1, 128, 1),
    ("l2 1x1 128->512", 128, 512, 1, 1, 1, 64, 4), ("l2 ds 256->512 s2", 256, 512, 1, 2, 1, 128, 1),
    ("l2 1x1 512->128", 512, 128, 1, 1, 1, 64, 3), ("l2 3x3 128", 128, 128, 3, 1, 1, 64, 3),
    ("l3 1x1 512->256", 512, 256, 1, 1, 1, 64, 1), ("l3 3x3 256 s2", 256, 256, 3, 2, 1, 64, 1),
    ("l3 1x1 256->1024", 256, 1024, 1, 1, 1, 32, 6), ("l3 ds 512->1024 s2", 512, 1024, 1, 2, 1, 64, 1),
    ("l3 1x1 1024->256", 1024, 256, 1, 1, 1, 32, 5), ("l3 3x3 256", 256, 256, 3, 1, 1, 32, 5),
    ("l4 1x1 1024->512", 1024, 512, 1, 1, 1, 32, 1), ("l4 3x3 512 d1", 512, 512, 3, 1, 1, 32, 1),
    ("l4 1x1 512->2048", 512, 2048, 1, 1, 1, 32, 3), ("l4 ds 1024->2048", 1024, 2048, 1, 1, 1, 32, 1),
    ("l4 1x1 2048->512", 2048, 512, 1, 1, 1, 32, 2), ("l4 3x3 512 d2", 512, 512, 3, 1, 2, 32, 2),
    ("aspp 3x3 2048->32 d6", 2048, 32, 3, 1, 6, 32, 1), ("aspp 3x3 2048->32 d24", 2048, 32, 3, 1, 24, 32, 1),
]


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--scale", type=int, default=1, help="divide spatial size (1 = 512^2 input)")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--only", default="", help="substring filter on the shape name")
    args = ap.parse_args()
    B = args.batch
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    print(f"{'shape':24s} {'M':>8s} | {'fwd ms':>8s} {'TF/s':>6s} {'GB/s':>6s} | {'dgrad ms':>8s} {'TF/s':>6s} | {'wgrad ms':>8s} {'TF/s':>6s}")
    for name, cin, cout, k, s, d, hin, cnt in SHAPES:
        if args.only and args.only not in name:
            continue
        h = hin // args.scale
        pad = d * (k - 1) // 2
        x = torch.randn(B, h, h, cin, device="cuda")
        w = torch.randn(cout, k, k, cin, device="cuda") * 0.05
        sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1
        y = ops.conv2d(x, w, stride=s, pad=pad, dil=d)
        dy = torch.randn_like(y)
        wt = ops.weight_transpose(w)
        dw = torch.zeros_like(w)
        M = y.numel() // cout
        flops = 2.0 * M * cout * k * k * cin
        t_f = timeit(lambda: ops.conv2d(x, w, stride=s, pad=pad, dil=d, in_scale=sc, in_shift=sh, in_relu=True, out=y), args.reps)
        t_d = timeit(lambda: ops.conv2d_dgrad(dy, wt, x.shape, stride=s, pad=pad, dil=d), args.reps)
        t_w = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, stride=s, pad=pad, dil=d, in_scale=sc, in_shift=sh, in_relu=True), args.reps)
        gbs = (x.numel() + y.numel() + w.numel()) * 4 / (t_f * 1e-3) / 1e9
        print(f"{name:24s} {M:8d} | {t_f:8.3f} {flops / t_f / 1e9:6.1f} {gbs:6.0f} | {t_d:8.3f} {flops / t_d / 1e9:6.1f} | "
              f"{t_w:8.3f} {flops / t_w / 1e9:6.1f}")
        tot["fwd"] += cnt * t_f
        tot["dgrad"] += cnt * t_d
        tot["wgrad"] += cnt * t_w
        del x, w, y, dy, wt, dw
    print("per-forward totals (ms): " + "  ".join(f"{k}={v:.2f}" for k, v in tot.items()))


if __name__ == "__main__":
    main()
