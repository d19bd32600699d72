"""bf16-STORAGE autograd nodes of the residual network (BASELINE config 5: "bf16 weights, CDNA4 bf16 MFMA").

Between the max-pool output and the layer4 output every activation and every activation gradient is a bf16 tensor; conv
weights are fp32 masters in the flat arena with bf16 copies refreshed after each optimizer step; accumulation, BatchNorm
statistics, scale / shift and all parameter gradients are fp32.  With bf16 matrix instructions 16x faster than the f32 ones
there is no operand prologue on this path: relu(bn(z)) is materialised for the next conv (the same HBM bytes per activation
as fp32 storage + prologue).  Same reference semantics as blocks.BottleneckFn (uemda/_resnets.py:92-112)."""
import torch
from torch.autograd import Function

from .. import ops, ops_bf16 as ob
from .blocks import _st_from, _st_tensor, grad_buffer, grad_ohwi


class CastFn(Function):
    """fp32 <-> bf16 at the two ends of the bf16 region (gradient cast back on the way down)."""

    @staticmethod
    def forward(ctx, x, to_bf16):
        ctx.to_bf16 = to_bf16
        return ob.to_bf16(x.contiguous()) if to_bf16 else ob.to_f32(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        return (ob.to_f32(dy) if ctx.to_bf16 else ob.to_bf16(dy)), None


class BottleneckBf16Fn(Function):
    @staticmethod
    def forward(ctx, x, blk, *params):
        s, d = blk.stride, blk.dilation
        z1, st1 = ob.conv2d_bn(x, ob.weight(blk.conv1.weight), blk.bn1)
        a1 = ob.affine_act(z1, st1)
        z2, st2 = ob.conv2d_bn(a1, ob.weight(blk.conv2.weight), blk.bn2, stride=s, pad=d, dil=d)
        a2 = ob.affine_act(z2, st2)
        z3, st3 = ob.conv2d_bn(a2, ob.weight(blk.conv3.weight), blk.bn3)
        has_ds = blk.downsample is not None
        if has_ds:
            zd, std = ob.conv2d_bn(x, ob.weight(blk.downsample[0].weight), blk.downsample[1], stride=s)
            y, bits = ob.affine_act(z3, st3, res=zd, res_st=std, want_bits=True)
        else:
            zd, std = None, None
            y, bits = ob.affine_act(z3, st3, res=x, want_bits=True)
        blk._nbt_add()
        if any(ctx.needs_input_grad):
            ctx.blk, ctx.has_ds = blk, has_ds
            saved = [x, bits, z1, a1, z2, a2, z3, _st_tensor(st1), _st_tensor(st2), _st_tensor(st3)]
            if has_ds:
                saved += [zd, _st_tensor(std)]
            ctx.save_for_backward(*saved)
        return y

    @staticmethod
    def backward(ctx, dy):
        blk = ctx.blk
        sv = ctx.saved_tensors
        x, bits, z1, a1, z2, a2, z3 = sv[:7]
        st1, st2, st3 = (_st_from(b) for b in sv[7:10])
        s, d = blk.stride, blk.dilation
        G, gb = grad_ohwi, grad_buffer
        dy = dy.contiguous()
        # BN3 + residual + ReLU (mask = packed bits of the block output); dp = dy*mask feeds the identity / downsample path
        dz3, dp = ob.bn_backward(z3, dy, st3, gb(blk.bn3.weight), gb(blk.bn3.bias), relu=2, bits=bits, want_dres=True)
        ob.conv2d_wgrad(a2, dz3, G(blk.conv3.weight))
        da2 = ob.conv2d_dgrad(dz3, ob.weight_t(blk.conv3.weight), a2.shape)
        del dz3
        dz2 = ob.bn_backward(z2, da2, st2, gb(blk.bn2.weight), gb(blk.bn2.bias), relu=1)
        del da2
        ob.conv2d_wgrad(a1, dz2, G(blk.conv2.weight), stride=s, pad=d, dil=d)
        da1 = ob.conv2d_dgrad(dz2, ob.weight_t(blk.conv2.weight), a1.shape, stride=s, pad=d, dil=d)
        del dz2
        dz1 = ob.bn_backward(z1, da1, st1, gb(blk.bn1.weight), gb(blk.bn1.bias), relu=1)
        del da1
        ob.conv2d_wgrad(x, dz1, G(blk.conv1.weight))
        wt1 = ob.weight_t(blk.conv1.weight)
        if ctx.has_ds:
            zd, std = sv[10], _st_from(sv[11])
            ds_conv, ds_bn = blk.downsample[0], blk.downsample[1]
            dzd = ob.bn_backward(zd, dp, std, gb(ds_bn.weight), gb(ds_bn.bias), relu=0)
            ob.conv2d_wgrad(x, dzd, G(ds_conv.weight), stride=s)
            dx = ob.conv2d_dgrad(dz1, wt1, x.shape)
            ob.conv2d_dgrad(dzd, ob.weight_t(ds_conv.weight), x.shape, stride=s, out=dx, accumulate=True)
        else:
            dx = ob.conv2d_dgrad(dz1, wt1, x.shape, out=dp, accumulate=True)      # identity gradient + conv1's
        cb = getattr(blk, "_uem_after_backward", None)      # data-parallel bucket trigger (uemda_amd.dp)
        if cb is not None:
            cb()
        return (dx, None) + (None,) * (len(ctx.needs_input_grad) - 2)
