"""Tensor-level wrappers of the bf16-STORAGE entry points (BASELINE config 5): activations and weights are
torch.bfloat16 NHWC / OHWI tensors, accumulation is fp32 on the bf16 matrix cores.  Same conventions as ops.py."""
import ctypes

import torch

from . import _lib, ops
from ._lib import CONV_ACCUMULATE, CONV_TRANSPOSED, ConvShape, UemError, call
from .ops import conv_out_size, need_gpu, ptr, stream


import os

# Round 5: the bf16 weight gradients run on a SIDE stream beside the data-gradient / BatchNorm chain of the backward pass.  They
# are leaves of the backward graph (nothing downstream reads them before the optimizer or the gradient all-reduce), and under bf16
# storage every conv launch is short (30-90 us, one to four rounds of blocks) and bound by operand delivery and latency, not by HBM or
# the matrix pipe: two such launches side by side fill each other's ramps and tails.  (The fp32 path dropped the same mechanism in
# round 2 -- its kernels are 4x longer and matrix-bound: 136.1 against 136.4 ms.)  Ordering: the side stream waits for everything the
# main stream has queued at the call (the gradient it reads), the tensors it reads are recorded on it (the caching allocator will not
# hand their memory out again before the launch has run), and the main stream waits for the side stream once per backward pass --
# an autograd end-of-backward callback -- and before a data-parallel bucket goes out.  UEM_BF16_SIDE_WGRAD=0 switches it off.
SIDE_WGRAD = os.environ.get("UEM_BF16_SIDE_WGRAD", "1") != "0"


_Side, side_join, _on_side = ops._Side, ops.side_join, ops.on_side


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
    else:
        ops.guard_write(out, "conv2d_dgrad_bf16(out=)")
    flops = 2.0 * n * dy.shape[1] * dy.shape[2] * cout * kh * kw * cin
    ops.PROF.run("conv_dgrad", flops, lambda: call("uem_conv2d_bf16", ptr(dy), ptr(w_t), ptr(out), ctypes.byref(s),
                                                   CONV_TRANSPOSED | (CONV_ACCUMULATE if accumulate else 0), None, stream()))
    return out


def dgrad_tail_ok(x_shape, cin):
    n, h, w, _ = x_shape
    return (n * h * w) % 128 == 0 and cin % 64 == 0 and ops.FUSE_BN_BACKWARD


def conv2d_dgrad_tail(dy, w_t, x_shape, acc_src=None, acc_bits=None, out=None, accumulate=False, bn_z=None, bn_vec=None, bn_bits=None,
                      pad=None, dil=1):
    """ops.conv2d_dgrad_tail on bf16 tensors (stride-1 data gradient with the residual bookkeeping and / or the first pass of
    a BatchNorm backward in its epilogue) -> (dx, per-tile partial sums or None)."""
    need_gpu(dy, w_t)
    _bf16c(dy, "dgrad_bf16 dy"), _bf16c(w_t, "dgrad_bf16 w_t")
    cin, kh, kw, cout = w_t.shape
    n, h, w, _ = x_shape
    s = ConvShape()
    s.N, s.H, s.W, s.Cin = n, h, w, cin
    s.Ho, s.Wo, s.Cout = dy.shape[1], dy.shape[2], cout
    s.KH, s.KW, s.stride, s.pad, s.dil = kh, kw, 1, (dil * (kh - 1) // 2 if pad is None else pad), dil
    s.x_ld, s.y_ld = cin, cout
    if out is None:
        out = torch.empty((n, h, w, cin), device=dy.device, dtype=torch.bfloat16)
    else:
        ops.guard_write(out, "conv2d_dgrad_tail_bf16(out=)")
    M = n * h * w
    tp = torch.empty((M // 128, 2, cin), device=dy.device, dtype=torch.float32) if bn_z is not None else None
    flops = 2.0 * n * dy.shape[1] * dy.shape[2] * cout * kh * kw * cin
    ops.PROF.run("conv_dgrad", flops, lambda: call("uem_conv2d_dgrad_tail_bf16", ptr(dy), ptr(w_t), ptr(out), ctypes.byref(s), ptr(acc_src),
                                                   ptr(acc_bits), ptr(bn_z), ptr(bn_vec), ptr(bn_bits), ptr(tp),
                                                   CONV_ACCUMULATE if accumulate else 0, stream()))
    return out, tp


def conv2d_dgrad_bn_backward(dy, w_t, z, st, gamma_grad, beta_grad, stride=1, pad=0, dil=1):
    """dA = dgrad(dy) for the conv that consumed relu(bn(z)), then that BatchNorm+ReLU's backward -> dz (in dA's buffer); the
    reduction pass rides in the data-gradient epilogue when the conv has stride 1 and full tiles."""
    cin, kh, kw, cout = w_t.shape
    if stride != 1 or not dgrad_tail_ok(z.shape, cin) or not st.training:
        da = conv2d_dgrad(dy, w_t, z.shape, stride=stride, pad=pad, dil=dil)
        return bn_backward(z, da, st, gamma_grad, beta_grad, relu=1, dx=da)
    vec = st.scale._base
    if vec is None or vec.shape != (4, cin):
        raise UemError("conv2d_dgrad_bn_backward: BNState vectors must live in one (4, C) buffer")
    da, tp = conv2d_dgrad_tail(dy, w_t, z.shape, bn_z=z, bn_vec=vec, pad=pad, dil=dil)
    return bn_backward_from_partials(z, da, st, tp, gamma_grad, beta_grad, relu=1, dx=da)


def conv2d_wgrad(x, dy, dw_ohwi, stride=1, pad=0, dil=1, side=False):
    """dw (Cout,KH,KW,Cin) fp32 += dy^T * im2col(x) with bf16 x (N,H,W,Cin) and dy (N,Ho,Wo,Cout)."""
    if dw_ohwi is None:                      # frozen weight (blocks.grad_ohwi)
        return
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
    try:
        def launch():
            ops.PROF.run("conv_wgrad", flops, lambda: call("uem_conv2d_wgrad_bf16", ptr(x), ptr(dy), ptr(dw_ohwi), ctypes.byref(s), stream()),
                         who="conv2d_wgrad")
        if side and SIDE_WGRAD and ops.in_backward():            # side: only for gradient-arena views (ops.conv2d_wgrad)
            _on_side(launch, (x, dy), "bf16 weight gradient")
        else:
            launch()
    except UemError as e:
        if "code -2" not in str(e):
            raise
        # shapes the bf16 weight-gradient kernel does not take (rows that are not a multiple of 32 pixels): the same bf16
        # values through the fp32 kernel
        ops.conv2d_wgrad(to_f32(x), to_f32(dy), dw_ohwi, stride=stride, pad=pad, dil=dil)


# ---- the two ends of the bf16 region in bf16 (round 5): stem and InstanceNorm ----------------------------------------
def stem_ok(x_shape, bn):
    """Can the stem run with a bf16 conv output?  Training-mode statistics out of the conv epilogue (full 128-pixel tiles) and the
    pooled form of the BatchNorm backward (even map size)."""
    n, _, h, w = x_shape
    ho, wo = conv_out_size(h, 7, 2, 3, 1), conv_out_size(w, 7, 2, 3, 1)
    return (bn.training or bn.running_mean is None) and (n * ho * wo) % 128 == 0 and ho % 2 == 0 and wo % 2 == 0 and ops.FUSE_BN_STATS


def stem_conv_bn(x4, w8, bn):
    """7x7/s2 stem conv (bf16 operands) -> bf16 z + training-mode BatchNorm statistics of the rounded values -> (z, BNState)."""
    n, h, w, _ = x4.shape
    ho, wo = conv_out_size(h, 7, 2, 3, 1), conv_out_size(w, 7, 2, 3, 1)
    M = n * ho * wo
    z = torch.empty((n, ho, wo, 64), device=x4.device, dtype=torch.bfloat16)
    ts = torch.empty((M // 128, 2, 64), device=x4.device, dtype=torch.float32)
    ops.PROF.run("conv_fwd", 2.0 * z.numel() * 147, lambda: call("uem_conv2d_stem_fwd_stats_bf16", ptr(x4), ptr(w8), ptr(z), n, h, w, ptr(ts), stream()))
    st = ops.BNState()
    st.training = True
    buf = torch.empty((4, 64), device=x4.device, dtype=torch.float32)
    st.scale, st.shift, st.mean, st.invstd = buf[0], buf[1], buf[2], buf[3]
    call("uem_bn_stats_from_tiles", ptr(ts), M // 128, M, 64, ptr(bn.weight.detach()), ptr(bn.bias.detach()), float(bn.eps),
         float(bn.momentum if bn.momentum is not None else 0.1), ptr(bn.running_mean), ptr(bn.running_var),
         ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift), stream())
    return z, st


def maxpool_affine_fwd(z, st, want_idx):
    """maxpool3x3s2(relu(z*scale + shift)) on the bf16 z -> bf16 pooled map (+ argmax taps)."""
    n, h, w, c = z.shape
    ho, wo = conv_out_size(h, 3, 2, 1, 1), conv_out_size(w, 3, 2, 1, 1)
    y = torch.empty((n, ho, wo, c), device=z.device, dtype=torch.bfloat16)
    idx = torch.empty((n, ho, wo, c), device=z.device, dtype=torch.uint8) if want_idx else None
    call("uem_maxpool3x3s2_affine_fwd_bf16", ptr(z), ptr(st.scale), ptr(st.shift), ptr(y), ptr(idx), n, h, w, c, stream())
    return y, idx


def bn_backward_pooled(z, dy_pool, idx, st, gamma_grad, beta_grad):
    """ops.bn_backward_pooled on bf16 tensors: z bf16, pooled gradient bf16 -> dz bf16."""
    n, h, w, c = z.shape
    tmp = torch.empty((2, c), device=z.device, dtype=torch.float32)
    ws = torch.empty(_lib.load().uem_bn_workspace_floats(n * h * w, c), device=z.device, dtype=torch.float32)
    call("uem_bn_bwd_reduce_pool_bf16", ptr(z), ptr(dy_pool), ptr(idx), ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd),
         n, h, w, c, 1, ptr(tmp[0]), ptr(tmp[1]), ptr(gamma_grad), ptr(beta_grad), ptr(ws), stream())
    dz = torch.empty_like(z)
    call("uem_bn_bwd_apply_pool_bf16", ptr(z), ptr(dy_pool), ptr(idx), ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd),
         ptr(tmp[0]), ptr(tmp[1]), n, h, w, c, 1, ptr(dz), stream())
    return dz


def stem_wgrad(x4, dz, dw_ohwi):
    if dw_ohwi is None:                      # frozen stem
        return
    n, h, w, _ = x4.shape
    if ops.stem_tiles_ok(h, w) and dw_ohwi.is_contiguous():
        # the stem's own weight-gradient kernel (csrc/stem.hip): bf16 dz widened at the load, fp32 operands, deterministic
        ws = torch.empty(_lib.load().uem_stem_conv_wgrad_workspace_floats(), device=x4.device, dtype=torch.float32)
        ops.PROF.run("conv_wgrad", 2.0 * dz.numel() * 147,
                     lambda: call("uem_stem_conv_wgrad_bf16", ptr(x4), ptr(dz), ptr(dw_ohwi), ptr(ws), n, h, w, stream()),
                     executed=2.0 * dz.numel() * 160)
        return
    dw8 = torch.zeros((64, 7, 8, 4), device=x4.device, dtype=torch.float32)
    ops.PROF.run("conv_wgrad", 2.0 * dz.numel() * 147, lambda: call("uem_conv2d_stem_wgrad_bf16", ptr(x4), ptr(dz), ptr(dw8), n, h, w, stream()))
    call("uem_stem_unpack_grad", ptr(dw8), ptr(dw_ohwi), stream())


def instnorm_fwd(x, eps=1e-5):
    """InstanceNorm of the bf16 layer4 output -> fp32 features (+ invstd for the backward)."""
    n, h, w, c = x.shape
    _bf16c(x, "instnorm_fwd_bf16 x")
    y = torch.empty((n, h, w, c), device=x.device, dtype=torch.float32)
    stats = torch.empty((2, n, c), device=x.device, dtype=torch.float32)
    call("uem_instnorm_fwd_bf16", ptr(x), ptr(y), ptr(stats[0]), ptr(stats[1]), n, h * w, c, eps, stream())
    return y, stats[1]


def instnorm_bwd(y, dy, invstd):
    """-> the bf16 gradient of the InstanceNorm's bf16 input"""
    n, h, w, c = y.shape
    dx = torch.empty((n, h, w, c), device=y.device, dtype=torch.bfloat16)
    call("uem_instnorm_bwd_bf16", ptr(y), ptr(dy), ptr(invstd), ptr(dx), n, h * w, c, stream())
    return dx


# ---- fp32 <-> bf16 ---------------------------------------------------------------------------------------------
def to_bf16(x, out=None):
    need_gpu(x)
    if x.dtype != torch.float32 or not x.is_contiguous():
        raise UemError("to_bf16: expected a contiguous float32 tensor")
    out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16) if out is None else out
    call("uem_cast_f32_bf16", ptr(x), ptr(out), x.numel(), stream())
    return out


def to_f32(x):
    need_gpu(x)
    _bf16c(x, "to_f32")
    out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    call("uem_cast_bf16_f32", ptr(x), ptr(out), x.numel(), stream())
    return out


def weight(param):
    """bf16 OHWI copy of a conv weight: a view into the model's bf16 copy of the whole fp32 parameter arena, which is
    refreshed by ONE cast after every optimizer step (fp32 masters stay in the arena the optimizer and the all-reduce use)."""
    owner = getattr(param, "_uem_owner", None)
    off = getattr(param, "_uem_off", None)
    if owner is not None and off is not None and owner._arena is not None and param.data_ptr() == owner._arena.data_ptr() + 4 * off:
        # The copy is valid while nothing has written the arena since the cast.  Writers: FusedSGD (its kernel writes behind torch's
        # back and bumps ops.WEIGHT_EPOCH), in-place ops on the arena itself (arena._version), and in-place ops THROUGH a parameter
        # -- `p.data` is a view of the arena with its own version counter: torch.optim.SGD, load_state_dict's copy_, p.mul_() under
        # no_grad move p._version and leave arena._version alone (keyed on the arena's version only, the forward kept
        # the weights of the first bf16 step).  So the versions of ALL parameters are recorded at the cast (once per step, a few
        # hundred integers) and the requested parameter's is compared on every call.
        key = (ops.WEIGHT_EPOCH, owner._arena._version)
        hit = getattr(owner, "_uem_arena_bf16", None)
        if hit is None or hit[0] != key or hit[2].get(id(param), -1) != param._version:
            versions = {id(q): q._version for q in owner.parameters()}
            hit = (key, to_bf16(owner._arena), versions)
            owner._uem_arena_bf16 = hit
        o, i, kh, kw = param.shape
        return hit[1][off:off + param.numel()].view(o, kh, kw, i)
    key = (ops.WEIGHT_EPOCH, param._version, param.data_ptr())
    hit = getattr(param, "_uem_wb", None)
    if hit is None or hit[0] != key:
        hit = (key, to_bf16(ops.weight_ohwi(param)))
        param._uem_wb = hit
    return hit[1]


def weight_t(param):
    """bf16 (Cin,KH,KW,Cout) copy for the data gradient: transposed and rounded from the fp32 master (the same bf16 values as weight()),
    refreshed with every other derived filter bank of the model in the ONE uem_weight_prep launch after an optimizer step."""
    cout, cin, kh, kw = param.shape
    return ops.PREP.get(param, _lib.PREP_TRANSPOSE_BF16, (cin, kh, kw, cout), dtype=torch.bfloat16)


# ---- BatchNorm / residual passes on bf16 tensors ------------------------------------------------------------------
def conv2d_bn(x, w_b, bn, stride=1, pad=0, dil=1):
    """bf16 conv + training-mode BatchNorm statistics out of the conv epilogue -> (z bf16, BNState fp32)."""
    cout = w_b.shape[0]
    n, h, w, _ = x.shape
    ho, wo = conv_out_size(h, w_b.shape[1], stride, pad, dil), conv_out_size(w, w_b.shape[2], stride, pad, dil)
    M = n * ho * wo
    if not (bn.training or bn.running_mean is None):
        # eval mode: scale / shift from the running statistics, no statistics pass, any number of output pixels -- the offline
        # pseudo-label pass and evaluation under torch.no_grad(), and the frozen-statistics training graph of
        # ResNetEncoder(batchnorm_trainable=False) (reference resnet.py:112-117,183-190), whose backward is bn_backward below
        z = conv2d(x, w_b, stride=stride, pad=pad, dil=dil)
        return z, ops.bn_stats(z, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, False, bn.eps)
    if M % 128 != 0:
        raise UemError(f"bf16 storage: conv outputs need a multiple of 128 pixels per batch (got {M}); use fp32 storage")
    z, ts = conv2d(x, w_b, stride=stride, pad=pad, dil=dil, want_stats=True)
    st = ops.BNState()
    st.training = True
    buf = torch.empty((4, cout), device=x.device, dtype=torch.float32)
    st.scale, st.shift, st.mean, st.invstd = buf[0], buf[1], buf[2], buf[3]
    call("uem_bn_stats_from_tiles", ptr(ts), M // 128, M, cout, ptr(bn.weight.detach()), ptr(bn.bias.detach()), float(bn.eps),
         float(bn.momentum if bn.momentum is not None else 0.1), ptr(bn.running_mean), ptr(bn.running_var),
         ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift), stream())
    return z, st


def affine_act(x, st, res=None, res_st=None, relu=True, want_bits=False):
    C = x.shape[-1]
    out = torch.empty_like(x)
    bits = torch.empty(x.numel() // 32, device=x.device, dtype=torch.int32) if want_bits else None
    call("uem_affine_act_bf16", ptr(x), ptr(st.scale), ptr(st.shift), ptr(res),
         ptr(res_st.scale) if res_st is not None else None, ptr(res_st.shift) if res_st is not None else None,
         ptr(out), x.numel() // C, C, 1 if relu else 0, ptr(bits), stream())
    return (out, bits) if want_bits else out


def bn_backward(x, dy, st, gamma_grad, beta_grad, relu, bits=None, want_dres=False, dx=None):
    """BatchNorm(+ReLU) backward on bf16 tensors.  relu: 0 none, 1 mask recomputed from x, 2 packed bits.
    Returns dx (a new bf16 tensor unless given; may alias dy) [, dres = dy*mask]."""
    C = x.shape[-1]
    M = x.numel() // C
    tmp = torch.empty((2, C), device=x.device, dtype=torch.float32)
    if st.training or gamma_grad is not None or beta_grad is not None:
        ws = torch.empty(ops._lib.load().uem_bn_workspace_floats(M, C), device=x.device, dtype=torch.float32)
        call("uem_bn_bwd_reduce_bf16", ptr(x), ptr(dy), ptr(bits), ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd), M, C,
             int(relu), ptr(tmp[0]), ptr(tmp[1]), ptr(gamma_grad), ptr(beta_grad), ptr(ws), stream())
    if not st.training:
        # frozen statistics (eval-mode BatchNorm inside a training graph): y = x*scale + shift with constant scale / shift, so
        # dx = dp*scale -- the apply pass with both batch sums zero; gamma / beta, when still trainable, took sum dp*xhat / sum dp
        # (xhat from the running statistics) in the reduce above
        tmp.zero_()
    ops.guard_write(dx, "bn_backward_bf16(dx=)")
    dx = torch.empty_like(x) if dx is None else dx
    dres = torch.empty_like(x) if want_dres else None
    call("uem_bn_bwd_apply_bf16", ptr(x), ptr(dy), ptr(bits), ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd),
         ptr(tmp[0]), ptr(tmp[1]), M, C, int(relu), ptr(dx), ptr(dres), stream())
    return (dx, dres) if want_dres else dx


def bn_backward_pair(x1, x2, dy, bits, st1, st2, tiles1, gg1, gb1, gg2, gb2, dx2=None):
    """bf16 twin of ops.bn_backward_pair: bn3's and the downsample BatchNorm's backward of a block with a downsample branch, one apply
    pass for both (uem_bn_bwd_apply_pair_bf16).  None when a BatchNorm is frozen or the switch is off (the caller runs the two)."""
    if not (ops.BN_PAIR and st1.training and st2.training) or bits is None or x1.shape != x2.shape:
        return None
    C = x1.shape[-1]
    M = x1.numel() // C
    if C % 32 != 0:
        return None
    tmp = torch.empty((4, C), device=x1.device, dtype=torch.float32)
    ws = torch.empty(_lib.load().uem_bn_workspace_floats(M, C), device=x1.device, dtype=torch.float32)
    if tiles1 is not None:
        call("uem_bn_bwd_from_tiles", ptr(tiles1), tiles1.shape[0], C, ptr(tmp[0]), ptr(tmp[1]), ptr(gg1), ptr(gb1), stream())
    else:
        call("uem_bn_bwd_reduce_bf16", ptr(x1), ptr(dy), ptr(bits), ptr(st1.scale), ptr(st1.shift), ptr(st1.mean), ptr(st1.invstd), M, C, 2,
             ptr(tmp[0]), ptr(tmp[1]), ptr(gg1), ptr(gb1), ptr(ws), stream())
    call("uem_bn_bwd_reduce_bf16", ptr(x2), ptr(dy), ptr(bits), ptr(st2.scale), ptr(st2.shift), ptr(st2.mean), ptr(st2.invstd), M, C, 2,
         ptr(tmp[2]), ptr(tmp[3]), ptr(gg2), ptr(gb2), ptr(ws), stream())
    dx1 = torch.empty_like(x1)
    ops.guard_write(dx2, "bn_backward_pair_bf16(dx2=)")
    dx2 = torch.empty_like(x2) if dx2 is None else dx2
    if not _lib.try_call("uem_bn_bwd_apply_pair_bf16", ptr(x1), ptr(x2), ptr(dy), ptr(bits), ptr(st1.scale), ptr(st1.mean), ptr(st1.invstd),
                         ptr(tmp[0]), ptr(tmp[1]), ptr(st2.scale), ptr(st2.mean), ptr(st2.invstd), ptr(tmp[2]), ptr(tmp[3]), M, C,
                         ptr(dx1), ptr(dx2), stream()):
        call("uem_bn_bwd_apply_bf16", ptr(x1), ptr(dy), ptr(bits), ptr(st1.scale), ptr(st1.shift), ptr(st1.mean), ptr(st1.invstd),
             ptr(tmp[0]), ptr(tmp[1]), M, C, 2, ptr(dx1), None, stream())
        call("uem_bn_bwd_apply_bf16", ptr(x2), ptr(dy), ptr(bits), ptr(st2.scale), ptr(st2.shift), ptr(st2.mean), ptr(st2.invstd),
             ptr(tmp[2]), ptr(tmp[3]), M, C, 2, ptr(dx2), None, stream())
    return dx1, dx2


def bn_backward_from_partials(x, dy, st, tp, gamma_grad, beta_grad, relu, bits=None, dx=None):
    """BatchNorm backward whose reduction pass already ran in a data-gradient epilogue (tp = its per-tile partial sums)."""
    C = x.shape[-1]
    M = x.numel() // C
    tmp = torch.empty((2, C), device=x.device, dtype=torch.float32)
    call("uem_bn_bwd_from_tiles", ptr(tp), tp.shape[0], C, ptr(tmp[0]), ptr(tmp[1]), ptr(gamma_grad), ptr(beta_grad), stream())
    ops.guard_write(dx, "bn_backward_from_partials_bf16(dx=)")
    dx = torch.empty_like(x) if dx is None else dx
    call("uem_bn_bwd_apply_bf16", ptr(x), ptr(dy), ptr(bits), ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd),
         ptr(tmp[0]), ptr(tmp[1]), M, C, int(relu), ptr(dx), None, stream())
    return dx
