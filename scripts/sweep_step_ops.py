#!/usr/bin/env python3
"""Launch geometry of the LDS-DMA conv kernel WITH the epilogues the step runs them with (DEBUG_HOOKS build): forward + BatchNorm
tile statistics (affine prologue) and data gradient + BatchNorm-backward reduction, over the ResNet-50 shape inventory; variants
(persistent blocks, channels per k-step, N tile) interleaved in one process against the dispatch rule.  ms per launch."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uemda_amd import _lib, ops
from bench_conv_shapes import SHAPES

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


def main():
    B = int(os.environ.get("B", "32"))
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    variants = [(-1, 0)] + [(p, bn + 1000 * kb) for p in (0, 1) for kb in (16, 32) for bn in (64, 128)]
    names = ["rule"] + [f"p{p} k{kb} n{bn}" for p in (0, 1) for kb in (16, 32) for bn in (64, 128)]
    print("shape".ljust(22) + "op".ljust(7) + "".join(n.rjust(12) for n in names))
    tot = {}
    for name, cin, cout, k, s, d, hin, cnt in SHAPES:
        if "aspp" in name or (only and only not in name):
            continue
        pad = d * (k - 1) // 2
        x = torch.randn(B, hin, hin, cin, device="cuda")
        w = torch.randn(cout, k, k, cin, device="cuda") * 0.05
        bn = torch.nn.BatchNorm2d(cout).cuda()
        bn_in = torch.nn.BatchNorm2d(cin).cuda()
        sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1
        y, st = ops.conv2d_bn(x, w, bn, stride=s, pad=pad, dil=d, in_scale=sc, in_shift=sh, in_relu=True)
        _, st_in = ops.conv2d_bn(x, torch.randn(cin, 1, 1, cin, device="cuda") * 0.05, bn_in)
        dy = torch.randn_like(y)
        wt = ops.weight_transpose(w)
        gg, gb = torch.zeros(cin, device="cuda"), torch.zeros(cin, device="cuda")
        fw = lambda: ops.conv2d_bn(x, w, bn, stride=s, pad=pad, dil=d, in_scale=sc, in_shift=sh, in_relu=True)
        dg = lambda: ops.conv2d_dgrad_bn_backward(dy, wt, x, st_in, gg, gb, stride=s, pad=pad, dil=d)
        for op, fn in (("fwd", fw), ("dgrad", dg)):
            best = {}
            for rnd in range(2):
                for v in variants:
                    lib.uemdbg_conv_persist(v[0])
                    lib.uemdbg_conv_config(-1 if v[0] < 0 else 1, v[1])
                    best[v] = min(best.get(v, 1e9), timeit(fn))
            lo = min(best.values())
            print(name.ljust(22) + op.ljust(7) + "".join((f"{best[v]:.3f}" + ("*" if best[v] == lo else " ")).rjust(12) for v in variants), flush=True)
            for v in variants:
                tot[(op, v)] = tot.get((op, v), 0.0) + cnt * best[v]
            tot[(op, "best")] = tot.get((op, "best"), 0.0) + cnt * lo
    for op in ("fwd", "dgrad"):
        print(f"per-forward ms {op}".ljust(29) + "".join(f"{tot[(op, v)]:12.2f}" for v in variants) + f"   best-of {tot[(op, 'best')]:.2f}")
    lib.uemdbg_conv_config(-1, 0)
    lib.uemdbg_conv_persist(-1)


if __name__ == "__main__":
    main()
