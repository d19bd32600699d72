"""PPMBilinear head on the MI355X (reference uemda/models/Encoder.py:8-65): adaptive avg-pool at scales
(1,2,3,6) -> 1x1 conv -> BN -> ReLU -> bilinear up (align_corners=False) -> concat with feat (4096 ch) ->
3x3 conv -> BN -> ReLU -> Dropout2d(0.1) -> 1x1 conv (+bias).  One autograd node per head; every arithmetic
step is a HIP kernel (the 3x3 4096->512 conv is the largest GEMM of the PPM network, 38.7 GFLOP/tile/head)."""
import itertools

import torch
from torch.autograd import Function

from .. import ops
from ..ops import UemError, call, ptr, stream
from .blocks import _BN, _st_from, _st_tensor, grad_buffer, grad_ohwi

_on_bwd = ops.on_backward_stream          # every backward of a step on ONE stream (ops: two forward streams)

_drop_counter = itertools.count(1)


def dropout_seed(call_index, rank=None):
    """Seed of one Dropout2d call: process seed x call counter x data-parallel rank (per-rank RNG streams;
    every rank is seeded alike by seed_torch(2333), so without the rank term all replicas would draw one mask)."""
    if rank is None:
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    return ((torch.initial_seed() * 1000003 + call_index) * 8191 + rank * 2654435761) % (2 ** 63 - 1) + 1


def _avgpool(x, s):
    n, h, w, c = x.shape
    y = torch.empty((n, s, s, c), device=x.device, dtype=torch.float32)
    call("uem_adaptive_avgpool_fwd", ptr(x), ptr(y), n, h, w, c, s, stream())
    return y


# layer5 and layer6 pool the SAME feature map (Encoder.py:148-149): inside a `shared_pools()` scope (Deeplabv2._heads) the pooled
# tensors are computed once per forward.  Outside a scope nothing is shared: a cache keyed on (address, shape) alone would hand the
# PREVIOUS feature map's pools to the next same-shape tensor the allocator places at the same address (kernel outputs are written
# through raw pointers, so their _version never moves) -- ADVICE r2.
_pool_scope = {"depth": 0, "feat": None, "pools": {}}


class shared_pools:
    """`with shared_pools():` -- ppm_head calls inside the block that are given the SAME tensor object share their pooled maps."""

    def __enter__(self):
        _pool_scope["depth"] += 1
        return self

    def __exit__(self, *exc):
        _pool_scope["depth"] -= 1
        if _pool_scope["depth"] == 0:
            _pool_scope["feat"], _pool_scope["pools"] = None, {}
        return False


def _shared_pool(feat, s):
    if _pool_scope["depth"] == 0:
        return _avgpool(feat, s)
    if _pool_scope["feat"] is not feat:                # identity of the tensor object, held alive by the scope
        _pool_scope["feat"], _pool_scope["pools"] = feat, {}
    pools = _pool_scope["pools"]
    if s not in pools:
        pools[s] = _avgpool(feat, s)
    return pools[s]


def clear_pool_cache():
    _pool_scope["feat"], _pool_scope["pools"] = None, {}


class PPMHeadFn(Function):
    # operand precision of the head's convs (None = the global setting); Deeplabv2 sets "bf16" for the bf16-storage model
    prec = None

    @staticmethod
    def forward(ctx, feat, head, *params):
        ctx.prec = PPMHeadFn.prec
        with ops.conv_precision(ctx.prec):
            return PPMHeadFn._forward(ctx, feat, head)

    @staticmethod
    @_on_bwd
    def backward(ctx, dout):
        with ops.conv_precision(ctx.prec):
            return PPMHeadFn._backward(ctx, dout)

    @staticmethod
    def _forward(ctx, feat, head):
        n, h, w, cin = feat.shape
        scales = head.pool_scales
        nb = len(scales)
        ctot = cin + 512 * nb
        training = head.training
        cat = torch.empty((n, h, w, ctot), device=feat.device, dtype=torch.float32)
        cat[..., :cin].copy_(feat)
        saved_branch = []
        for i, s in enumerate(scales):
            conv, bn = head.ppm[i][1], head.ppm[i][2]
            if training and n * s * s < 2:
                raise ValueError("Expected more than 1 value per channel when training (PPM scale-1 branch needs B >= 2)")
            p = _shared_pool(feat, s)
            z = ops.conv2d(p, ops.weight_ohwi(conv.weight))
            st = _BN.stats(z, bn)
            call("uem_bilinear_up_fwd", ptr(z), ptr(cat[..., cin + 512 * i:]), n, s, s, 512, h, w, ctot, 0,
                 ptr(st.scale), ptr(st.shift), 1, stream())
            saved_branch += [p, z, _st_tensor(st)]
            ops.nbt_inc(bn)
        conv0, bn0, drop, conv4 = head.conv_last[0], head.conv_last[1], head.conv_last[3], head.conv_last[4]
        # the 4096 -> 512 3x3 conv (99.7 % of the head's FLOPs) takes the Winograd path when the shape allows (ops.wino_ok)
        vcat = None
        plan = ops.wino_plan(cat.shape, conv0.weight.shape[0], conv0.weight.shape[2], conv0.weight.shape[3], 1, 1, 1)
        if plan is not None and plan.mf and (bn0.training or bn0.running_mean is None) and ops.FUSE_BN_STATS:
            zc, stc, vcat = ops.conv3x3_wino_bn(cat, conv0.weight, bn0, 1, m=plan.mf)
        elif plan is not None and plan.mf:
            zc, vcat = ops.conv3x3_wino(cat, conv0.weight, 1, want_v=True, m=plan.mf)
            stc = _BN.stats(zc, bn0)
        else:
            zc = ops.conv2d(cat, ops.weight_ohwi(conv0.weight), pad=1)
            stc = _BN.stats(zc, bn0)
        ops.nbt_inc(bn0)
        a = ops.affine_act(zc, stc, relu=True)
        mask = None
        if training and drop.p > 0:
            if torch.cuda.is_current_stream_capturing():
                # inside a hipGraph capture (step.GraphedStep) a host-side seed would be frozen into the launch and every replay
                # would draw the same mask: torch's generator is graph-safe (its Philox offset advances per replay), so the
                # per-(image, channel) keep mask comes from it and the kernel only applies it (seed 0 = mask given)
                mask = (torch.rand((n, 512), device=feat.device) >= drop.p).to(torch.float32).mul_(1.0 / (1.0 - drop.p))
                call("uem_dropout2d", ptr(a), ptr(a), ptr(mask), n, h * w, 512, 0.0, 0, stream())
            else:
                mask = torch.empty((n, 512), device=feat.device, dtype=torch.float32)
                seed = dropout_seed(next(_drop_counter))
                call("uem_dropout2d", ptr(a), ptr(a), ptr(mask), n, h * w, 512, float(drop.p), seed, stream())
        C = conv4.weight.shape[0]
        w4 = torch.zeros((32, 1, 1, 512), device=feat.device, dtype=torch.float32)
        w4[:C].copy_(ops.weight_ohwi(conv4.weight))
        b4 = torch.zeros(32, device=feat.device, dtype=torch.float32)
        b4[:C].copy_(conv4.bias.detach())
        out32 = ops.conv2d(a, w4, b4, algo_cout=C)
        out = out32[..., :C].contiguous()
        if any(ctx.needs_input_grad):
            ctx.head, ctx.training, ctx.C = head, stc.training, C
            ctx.has_mask = mask is not None
            ctx.wino = plan
            keep_v = plan is not None and plan.keep_v
            # Winograd: the weight gradient reduces over the transformed input; where the forward keeps it (same tile size both ways,
            # under the byte cap) cat itself is not needed again
            ctx.save_for_backward(feat, vcat if keep_v else cat, zc, _st_tensor(stc), a, w4,
                                  mask if mask is not None else a.new_empty(0), *saved_branch)
        return out

    @staticmethod
    def _backward(ctx, dout):
        head, C = ctx.head, ctx.C
        feat, cat, zc, stcb, a, w4, mask = ctx.saved_tensors[:7]
        branch = ctx.saved_tensors[7:]
        n, h, w, cin = feat.shape
        ctot = cat.shape[-1]
        conv0, bn0, conv4 = head.conv_last[0], head.conv_last[1], head.conv_last[4]
        d32 = torch.zeros((n, h, w, 32), device=feat.device, dtype=torch.float32)
        d32[..., :C].copy_(dout)
        ops.bias_grad(d32, grad_buffer(conv4.bias), C, 32)
        dw4 = torch.zeros((32, 1, 1, 512), device=feat.device, dtype=torch.float32)
        ops.conv2d_wgrad(a, d32, dw4, algo_cout=C)
        ops.add_(grad_ohwi(conv4.weight), dw4[:C])
        da = ops.conv2d_dgrad(d32, ops.weight_transpose(w4), a.shape, algo_cout=C)
        if ctx.has_mask:
            call("uem_dropout2d", ptr(da), ptr(da), ptr(mask), n, h * w, 512, 0.0, 0, stream())   # seed 0: reuse mask
        stc = _st_from(stcb, ctx.training)
        dzc = ops.bn_backward(zc, da, stc, grad_buffer(bn0.weight), grad_buffer(bn0.bias), None, True, dx=da)
        plan = ctx.wino
        if plan is not None and plan.mb:
            if plan.keep_v:
                ops.conv3x3_wino_wgrad(cat, dzc, grad_ohwi(conv0.weight), 1)        # `cat` is the saved transformed input here
            elif plan.wgrad:
                ops.conv3x3_wino_wgrad(None, dzc, grad_ohwi(conv0.weight), 1, x=cat, m=plan.mb)
            else:
                ops.conv2d_wgrad(cat, dzc, grad_ohwi(conv0.weight), pad=1)
            if plan.dgrad:
                dcat, _ = ops.conv3x3_wino_dgrad(dzc, conv0.weight, 1, m=plan.mb)
            else:
                dcat = ops.conv2d_dgrad(dzc, ops.weight_transpose_cached(conv0.weight), (n, h, w, ctot), pad=1)
        else:
            ops.conv2d_wgrad(cat, dzc, grad_ohwi(conv0.weight), pad=1)
            dcat = ops.conv2d_dgrad(dzc, ops.weight_transpose_cached(conv0.weight), cat.shape, pad=1)
        dps = []
        for i, s in enumerate(head.pool_scales):
            conv, bn = head.ppm[i][1], head.ppm[i][2]
            p, z, stb = branch[3 * i:3 * i + 3]
            st = _st_from(stb, ctx.training)
            du = torch.empty((n, s, s, 512), device=feat.device, dtype=torch.float32)
            call("uem_bilinear_up_bwd", ptr(dcat[..., cin + 512 * i:]), ptr(du), n, s, s, 512, h, w, ctot, 0, stream())
            dz = ops.bn_backward(z, du, st, grad_buffer(bn.weight), grad_buffer(bn.bias), None, True, dx=du)
            ops.conv2d_wgrad(p, dz, grad_ohwi(conv.weight))
            dps.append(ops.conv2d_dgrad(dz, ops.weight_transpose_cached(conv.weight), p.shape))
        # concat slice + the four adaptive-average-pool backward passes in one kernel (one read of dcat's slice, one write)
        import ctypes
        nb = len(dps)
        if nb > 4:
            raise ops.UemError("PPMBilinear: at most 4 pooled branches")
        dfeat = torch.empty((n, h, w, cin), device=feat.device, dtype=torch.float32)
        dp_ptrs = (ctypes.c_void_p * 4)(*([ptr(t) for t in dps] + [None] * (4 - nb)))
        sc = (ctypes.c_int * 4)(*(list(head.pool_scales) + [1] * (4 - nb)))
        call("uem_ppm_feat_grad", ptr(dcat), ctot, dp_ptrs, sc, nb, ptr(dfeat), n, h, w, cin, stream())
        return (dfeat, None) + (None,) * (len(ctx.needs_input_grad) - 2)


def ppm_head(feat, head):
    return PPMHeadFn.apply(feat, head, *list(head.parameters()))
