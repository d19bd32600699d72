#!/usr/bin/env python3
"""Winograd F(2x2, 3x3) and F(4x4, 3x3) against the direct f32-MFMA kernels on the stride-1 3x3 shapes of the network (B = 32, 512^2
tiles): forward (with the BatchNorm prologue and statistics), data gradient (with the fused BatchNorm-backward reduction), weight
gradient; ms per call, stage by stage for the Winograd paths.  WINO_ERR=1 also prints each path's relative L2 error against float64
on a small batch of the same layer."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uemda_amd import ops

SHAPES = [("l3 3x3 256", 256, 256, 1, 32), ("l4 3x3 512 d1", 512, 512, 1, 32), ("l4 3x3 512 d2", 512, 512, 2, 32),
          ("l2 3x3 128", 128, 128, 1, 64), ("l1 3x3 64", 64, 64, 1, 128), ("ppm 3x3 4096->512", 4096, 512, 1, 32)]


def timeit(fn, reps=5):
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
    B = int(os.environ.get("B", "32"))
    ops.WINOGRAD_MIN_CH = 64
    ops.WINOGRAD_MIN_CH_F4 = 64
    for name, cin, cout, d, h in SHAPES:
        x = torch.randn(B, h, h, cin, device="cuda")
        w = (torch.randn(cout, cin, 3, 3, device="cuda") * 0.02).contiguous(memory_format=torch.channels_last)
        sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1
        bn = torch.nn.BatchNorm2d(cout).cuda()
        bn_in = torch.nn.BatchNorm2d(cin).cuda()
        st_in = ops.bn_stats(x, bn_in.weight.detach(), bn_in.bias.detach(), None, None, True)
        dy = torch.randn(B, h, h, cout, device="cuda")
        gg, gb = torch.zeros(cin, device="cuda"), torch.zeros(cin, device="cuda")
        dw = torch.zeros(cout, 3, 3, cin, device="cuda")
        wo, wt = ops.weight_ohwi(w), ops.weight_transpose(ops.weight_ohwi(w))
        flops = 2.0 * B * h * h * cout * 9 * cin
        t_df = timeit(lambda: ops.conv2d_bn(x, wo, bn, pad=d, dil=d, in_scale=sc, in_shift=sh, in_relu=True))
        t_dd = timeit(lambda: ops.conv2d_dgrad_bn_backward(dy, wt, x, st_in, gg, gb, pad=d, dil=d))
        t_dw = timeit(lambda: ops.conv2d_wgrad(x, dy, dw, pad=d, dil=d, in_scale=sc, in_shift=sh, in_relu=True))
        print(f"{name:20s} direct fwd/dgrad/wgrad {t_df:.3f} {t_dd:.3f} {t_dw:.3f} ms", flush=True)
        for m in (2, 4):
            if m == 4 and ((B * h * h // 16) % 128 or h % (4 * d)):
                continue
            npos = (m + 2) ** 2
            y, st, v = ops.conv3x3_wino_bn(x, w, bn, d, in_scale=sc, in_shift=sh, in_relu=True, m=m)
            t_wf = timeit(lambda: ops.conv3x3_wino_bn(x, w, bn, d, in_scale=sc, in_shift=sh, in_relu=True, m=m))
            t_wd = timeit(lambda: ops.conv3x3_wino_dgrad_bn_backward(dy, w, x, st_in, gg, gb, d, m=m))
            t_ww = timeit(lambda: ops.conv3x3_wino_wgrad(v, dy, dw, d))
            t_wr = timeit(lambda: ops.conv3x3_wino_wgrad(None, dy, dw, d, x=x, in_scale=sc, in_shift=sh, in_relu=True, m=m))
            # stages
            u = ops.wino_filter_cached(w, False, m)
            t_in = timeit(lambda: ops.wino_input(x, d, sc, sh, True, m))
            t_g = timeit(lambda: ops.wino_gemm(v, u))
            ut = ops.wino_filter_cached(w, True, m)
            vdy = ops.wino_input(dy, d, m=m)
            t_gd = timeit(lambda: ops.wino_gemm(vdy, ut, data_gradient=True))
            mt = ops.wino_gemm(v, u)
            t_out = timeit(lambda: ops.call("uem_wino_output", ops.ptr(mt), ops.ptr(y), B, h, h, cout, d, m, None, None, None, None, ops.stream()))
            dm = torch.empty((npos, v.shape[1], cout), device="cuda")
            t_dy = timeit(lambda: ops.call("uem_wino_dy", ops.ptr(dy), ops.ptr(dm), B, h, h, cout, d, m, None, 0, ops.stream()))
            du = torch.zeros((npos, cout, cin), device="cuda")
            t_wg = timeit(lambda: ops.call("uem_wino_wgrad_gemm", ops.ptr(v), ops.ptr(dm), ops.ptr(du), v.shape[1], cin, cout, npos, ops.stream()))
            t_fg = timeit(lambda: ops.call("uem_wino_filter_grad", ops.ptr(du), ops.ptr(dw), cout, cin, m, ops.stream()))
            t_fl = timeit(lambda: ops.call("uem_wino_filter", ops.ptr(wo), ops.ptr(u), cout, cin, 0, m, ops.stream()))
            gf = flops / 9.0 * npos / (m * m)
            print(f"{'':20s} F({m}x{m},3x3) fwd/dgrad/wgrad/wgrad-recomputing-V {t_wf:.3f} {t_wd:.3f} {t_ww:.3f} {t_wr:.3f} ms | stages: input {t_in:.3f} "
                  f"gemm {t_g:.3f} ({gf / t_g / 1e9:.0f} TF/s) dgrad-gemm {t_gd:.3f} ({gf / t_gd / 1e9:.0f} TF/s) output {t_out:.3f} dy {t_dy:.3f} "
                  f"wgrad-gemm {t_wg:.3f} ({gf / t_wg / 1e9:.0f} TF/s) filter-grad {t_fg:.3f} filter {t_fl:.3f}", flush=True)
            del v, mt, dm, du, y, vdy
        if os.environ.get("WINO_ERR"):
            import torch.nn.functional as F
            g = torch.Generator().manual_seed(1)
            nb = max(2, 2048 // (h * h) * 2)
            xs = torch.randn(nb, cin, h, h, generator=g)
            ws = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
            ref = F.conv2d(F.relu(xs.double()), ws.double(), padding=d, dilation=d)
            xg = xs.permute(0, 2, 3, 1).contiguous().cuda()
            wg = ws.cuda().contiguous(memory_format=torch.channels_last)
            one, zero = torch.ones(cin, device="cuda"), torch.zeros(cin, device="cuda")

            def err(y):
                yy = y.permute(0, 3, 1, 2).double().cpu()
                return float((yy - ref).norm() / ref.norm())
            e = [err(ops.conv2d(xg, ops.weight_ohwi(wg), None, pad=d, dil=d, in_scale=one, in_shift=zero, in_relu=True))]
            for m in (2, 4):
                ok = (nb * h * h // (m * m)) % 128 == 0 and h % (m * d) == 0
                e.append(err(ops.conv3x3_wino(xg, wg, d, in_scale=one, in_shift=zero, in_relu=True, m=m)) if ok else float("nan"))
            print(f"{'':20s} relative L2 error against float64: direct {e[0]:.2e}  F(2x2) {e[1]:.2e}  F(4x4) {e[2]:.2e}", flush=True)
        del x, w, dy


if __name__ == "__main__":
    main()
