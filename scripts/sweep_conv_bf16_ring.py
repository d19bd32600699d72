#!/usr/bin/env python3
"""bf16 conv kernel: the round-5 dispatch (128-row tiles / 256-row two-stage tiles) against the three-stage ring on persistent 256-row
blocks (round 6, `uemdbg_conv_bf16_ring`), per ResNet shape, interleaved in one process, outputs compared bit for bit:
  fwd   forward with the BatchNorm tile statistics
  dgrad plain data gradient
  bnbwd data gradient with the BatchNorm-backward partial sums of the layer it feeds (conv3 / conv2 taken backwards)
  tail  the residual tail (conv1 taken backwards: identity gradient through packed bits + the previous block's bn3 sums)
    B=32 SCALE=1 python scripts/sweep_conv_bf16_ring.py [name filter]        (SCALE=2: the 1024x1024 tiles of BASELINE config 5)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uemda_amd import _lib, ops_bf16
from bench_conv_shapes import SHAPES


def timeit(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def same(a, b):
    """bf16 tensors bit for bit; fp32 tensors (the per-128-row partial sums, whose summation order depends on the tile shape) to 1e-5 of
    the column's magnitude"""
    if isinstance(a, (tuple, list)):
        return all(same(x, y) for x, y in zip(a, b))
    if a is None:
        return b is None
    if a.dtype == torch.float32:
        return bool(((a - b).abs() <= 1e-5 * (a.abs() + b.abs()) + 1e-3).all())
    return bool(torch.equal(a, b))


def main():
    B = int(os.environ.get("B", "32"))
    scale = int(os.environ.get("SCALE", "1"))
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    lib = _lib.load()
    what = os.environ.get("SWEEP", "split")                  # ring: the three-stage ring with the classic epilogue; split: + loader / storer waves
    for f in ("uemdbg_conv_bf16_ring", "uemdbg_conv_bf16_pw", "uemdbg_conv_bf16_big", "uemdbg_conv_bf16_areg"):
        getattr(lib, f).argtypes = [ctypes.c_int]
        getattr(lib, f).restype = None

    def switch(on):
        lib.uemdbg_conv_bf16_ring({"ring": 1, "split": 2}.get(what, 0) if on else 0)
        lib.uemdbg_conv_bf16_pw(1 if (on and what == "pw") else 0)
        lib.uemdbg_conv_bf16_areg((1 if on else 0) if what == "areg" else 0)      # the A-stationary pointwise block forced on / off
        if what == "big":                                    # 256-row two-stage tiles (256 x 256 where the columns allow) forced on / rule
            lib.uemdbg_conv_bf16_pw(0)
            lib.uemdbg_conv_bf16_big(1 if on else -1)
    print(f"B={B} scale={scale}   times in ms: round-5 dispatch / {what}, change; '!' = outputs differ beyond the statistics' summation order")
    print(f"{'shape':22s} {'M':>8s} {'tiles':>6s} | {'fwd':>21s} | {'dgrad':>21s} | {'bnbwd':>21s} | {'tail':>21s}")
    tot = {}
    for name, cin, cout, k, s, d, hin, cnt in SHAPES:
        if "aspp" in name or "stem" in name or (only and only not in name):
            continue
        hin *= scale
        pad = d * (k - 1) // 2
        x = torch.randn(B, hin, hin, cin, device="cuda").bfloat16()
        w = (torch.randn(cout, k, k, cin, device="cuda") * 0.05).bfloat16()
        y = ops_bf16.conv2d(x, w, stride=s, pad=pad, dil=d)
        dy = torch.randn_like(y)
        wt = w.permute(3, 1, 2, 0).contiguous()
        M = y.numel() // cout
        Mi = x.numel() // cin
        cases = {"fwd": lambda: ops_bf16.conv2d(x, w, stride=s, pad=pad, dil=d, want_stats=M % 128 == 0),
                 "dgrad": lambda: ops_bf16.conv2d_dgrad(dy, wt, x.shape, stride=s, pad=pad, dil=d)}
        if s == 1 and Mi % 128 == 0:
            z = torch.randn(B, hin, hin, cin, device="cuda").bfloat16()
            vec = (torch.rand(4, cin, device="cuda") + 0.5).contiguous()
            cases["bnbwd"] = lambda: ops_bf16.conv2d_dgrad_tail(dy, wt, x.shape, bn_z=z, bn_vec=vec, pad=pad, dil=d)
            if k == 1 and cin >= 2 * cout:
                acc = torch.randn(B, hin, hin, cin, device="cuda").bfloat16()
                bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (Mi * cin // 32,), device="cuda", dtype=torch.int32)
                cases["tail"] = lambda: ops_bf16.conv2d_dgrad_tail(dy, wt, x.shape, acc_src=acc, acc_bits=bits, bn_z=z, bn_vec=vec, bn_bits=bits)
        cells = []
        for key in ("fwd", "dgrad", "bnbwd", "tail"):
            fn = cases.get(key)
            if fn is None:
                cells.append(f"{'-':>21s}")
                continue
            switch(False)
            ref = fn()
            t0 = timeit(fn)
            switch(True)
            out = fn()
            t1 = timeit(fn)
            switch(False)
            t0 = min(t0, timeit(fn))
            switch(True)
            t1 = min(t1, timeit(fn))
            ok = same(ref, out)
            cells.append(f"{t0:6.3f} {t1:6.3f} {100 * (t1 / t0 - 1):+6.1f}%{' ' if ok else '!'}")
            a = tot.setdefault(key, [0.0, 0.0])
            a[0] += cnt * t0
            a[1] += cnt * t1
        lib.uemdbg_conv_bf16_ring(-1)
        lib.uemdbg_conv_bf16_pw(-1)
        lib.uemdbg_conv_bf16_big(-1)
        lib.uemdbg_conv_bf16_areg(-1)
        mo = y.numel() // cout
        t256 = (mo // 256) * max(1, cout // 128) if mo % 256 == 0 else 0
        print(f"{name:22s} {M:8d} {t256:6d} | " + " | ".join(cells), flush=True)
        del x, w, y, dy, wt
    print("weighted by launches per forward (ms): " + "  ".join(f"{k}: {a[0]:.2f} -> {a[1]:.2f}" for k, a in tot.items()))


if __name__ == "__main__":
    main()
