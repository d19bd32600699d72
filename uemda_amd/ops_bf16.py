"""Tensor-level wrappers of the bf16-STORAGE entry points (BASELINE config 5): activations and weights are
torch.bfloat16 NHWC / OHWI tensors, accumulation is fp32 on the bf16 matrix cores.  Same conventions as ops.py."""
import ctypes

import torch

from . import ops
from ._lib import CONV_ACCUMULATE, CONV_TRANSPOSED, ConvShape, UemError, call
from .ops import conv_out_size, need_gpu, ptr, stream


def _bf16c(t, what):
    if t.dtype != torch.bfloat16 or not t.is_contiguous():
        raise UemError(f"{what}: expected a contiguous bfloat16 tensor, got {t.dtype} strides={t.stride()}")
    return t


def conv2d(x, w_ohwi, stride=1, pad=0, dil=1, out=None, accumulate=False, want_stats=False):
    """y = conv(x, w): x (N,H,W,Cin) bf16, w (Cout,KH,KW,Cin) bf16 -> (N,Ho,Wo,Cout) bf16 [, per-tile statistics (tiles,2,Cout)]."""
    need_gpu(x, w_ohwi)
    _bf16c(x, "conv2d_bf16 x"), _bf16c(w_ohwi, "conv2d_bf16 w")
    n, h, w, cin = x.shape
    cout, kh, kw, cin2 = w_ohwi.shape
    if cin != cin2:
        raise UemError(f"conv2d_bf16: Cin mismatch {cin} vs {cin2}")
    s = ConvShape()
    s.N, s.H, s.W, s.Cin = n, h, w, cin
    s.Ho, s.Wo, s.Cout = conv_out_size(h, kh, stride, pad, dil), conv_out_size(w, kw, stride, pad, dil), cout
    s.KH, s.KW, s.stride, s.pad, s.dil = kh, kw, stride, pad, dil
    s.x_ld, s.y_ld = cin, cout
    if out is None:
        out = torch.empty((n, s.Ho, s.Wo, cout), device=x.device, dtype=torch.bfloat16)
    M = n * s.Ho * s.Wo
    ts = torch.empty((M // 128, 2, cout), device=x.device, dtype=torch.float32) if want_stats else None
    flops = 2.0 * M * cout * kh * kw * cin
    ops.PROF.run("conv_fwd", flops, lambda: call("uem_conv2d_bf16", ptr(x), ptr(w_ohwi), ptr(out), ctypes.byref(s),
                                                 CONV_ACCUMULATE if accumulate else 0, ptr(ts), stream()))
    return (out, ts) if want_stats else out


def conv2d_dgrad(dy, w_t, x_shape, stride=1, pad=0, dil=1, out=None, accumulate=False):
    """dx (N,H,W,Cin) bf16 from dy (N,Ho,Wo,Cout) bf16; w_t (Cin,KH,KW,Cout) bf16 = transposed forward weights."""
    need_gpu(dy, w_t)
    _bf16c(dy, "dgrad_bf16 dy"), _bf16c(w_t, "dgrad_bf16 w_t")
    cin, kh, kw, cout = w_t.shape
    n, h, w, _ = x_shape
    s = ConvShape()
    s.N, s.H, s.W, s.Cin = n, h, w, cin
    s.Ho, s.Wo, s.Cout = dy.shape[1], dy.shape[2], cout
    s.KH, s.KW, s.stride, s.pad, s.dil = kh, kw, stride, pad, dil
    s.x_ld, s.y_ld = cin, cout
    if out is None:
        out = torch.empty((n, h, w, cin), device=dy.device, dtype=torch.bfloat16)
    flops = 2.0 * n * dy.shape[1] * dy.shape[2] * cout * kh * kw * cin
    ops.PROF.run("conv_dgrad", flops, lambda: call("uem_conv2d_bf16", ptr(dy), ptr(w_t), ptr(out), ctypes.byref(s),
                                                   CONV_TRANSPOSED | (CONV_ACCUMULATE if accumulate else 0), None, stream()))
    return out


def conv2d_wgrad(x, dy, dw_ohwi, stride=1, pad=0, dil=1):
    """dw (Cout,KH,KW,Cin) fp32 += dy^T * im2col(x) with bf16 x (N,H,W,Cin) and dy (N,Ho,Wo,Cout)."""
    need_gpu(x, dy, dw_ohwi)
    _bf16c(x, "wgrad_bf16 x"), _bf16c(dy, "wgrad_bf16 dy")
    if dw_ohwi.dtype != torch.float32 or not dw_ohwi.is_contiguous():
        raise UemError("wgrad_bf16: dw must be a contiguous float32 tensor")
    cout, kh, kw, cin = dw_ohwi.shape
    n, h, w, _ = x.shape
    s = ConvShape()
    s.N, s.H, s.W, s.Cin = n, h, w, cin
    s.Ho, s.Wo, s.Cout = dy.shape[1], dy.shape[2], cout
    s.KH, s.KW, s.stride, s.pad, s.dil = kh, kw, stride, pad, dil
    s.x_ld, s.y_ld = cin, cout
    flops = 2.0 * n * s.Ho * s.Wo * cout * kh * kw * cin
    ops.PROF.run("conv_wgrad", flops, lambda: call("uem_conv2d_wgrad_bf16", ptr(x), ptr(dy), ptr(dw_ohwi), ctypes.byref(s), stream()))
