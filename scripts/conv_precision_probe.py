#!/usr/bin/env python3
"""Error of the opt-in conv precisions (bf16x3 split, plain bf16) against the exact-fp32 kernel and an fp64
torch reference, on a few ResNet shapes: max-abs / relative-L2 of forward and data-gradient outputs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uemda_amd import ops


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def main():
    torch.manual_seed(0)
    for cin, cout, k, s, d, h in [(64, 64, 3, 1, 1, 32), (256, 128, 1, 1, 1, 32), (512, 512, 3, 1, 2, 16), (2048, 512, 1, 1, 1, 16),
                                  (128, 128, 3, 2, 1, 32), (2048, 32, 3, 1, 6, 16), (64, 256, 1, 1, 1, 32), (256, 64, 1, 1, 1, 32)]:
        pad = d * (k - 1) // 2
        x = torch.randn(4, h, h, cin, device="cuda")
        w = torch.randn(cout, k, k, cin, device="cuda") * (2.0 / (cin * k * k)) ** 0.5
        sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1
        xa = torch.relu(x * sc + sh)
        ref = torch.nn.functional.conv2d(xa.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), None, s, pad, d)
        ref = ref.permute(0, 2, 3, 1)
        dy = torch.randn_like(ref, dtype=torch.float32).contiguous()
        wt = ops.weight_transpose(w)
        refd = torch.nn.grad.conv2d_input((4, cin, h, h), w.permute(0, 3, 1, 2).double(), dy.permute(0, 3, 1, 2).double(), s, pad, d)
        refd = refd.permute(0, 2, 3, 1)
        refw = torch.nn.grad.conv2d_weight(xa.permute(0, 3, 1, 2).double(), (cout, cin, k, k), dy.permute(0, 3, 1, 2).double(), s, pad, d)
        refw = refw.permute(0, 2, 3, 1)
        line = f"Cin={cin:4d} Cout={cout:4d} k={k} s={s} d={d}:"
        for prec in ("fp32", "bf16x3", "bf16"):
            ops.set_conv_precision(prec)
            y = ops.conv2d(x, w, stride=s, pad=pad, dil=d, in_scale=sc, in_shift=sh, in_relu=True)
            dx = ops.conv2d_dgrad(dy, wt, x.shape, stride=s, pad=pad, dil=d)
            dw = torch.zeros_like(w)
            ops.conv2d_wgrad(x, dy, dw, stride=s, pad=pad, dil=d, in_scale=sc, in_shift=sh, in_relu=True)
            line += f"  {prec}: fwd {rel(y, ref):.2e} dgrad {rel(dx, refd):.2e} wgrad {rel(dw, refw):.2e}"
        ops.set_conv_precision("fp32")
        print(line)


if __name__ == "__main__":
    main()
