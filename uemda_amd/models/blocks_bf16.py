"""bf16-STORAGE autograd nodes of the residual network (BASELINE config 5: "bf16 weights, CDNA4 bf16 MFMA").

Between the max-pool output and the layer4 output every activation and every activation gradient is a bf16 tensor; conv
weights are fp32 masters in the flat arena with bf16 copies refreshed after each optimizer step; accumulation, BatchNorm
statistics, scale / shift and all parameter gradients are fp32.  With bf16 matrix instructions 16x faster than the f32 ones
there is no operand prologue on this path: relu(bn(z)) is materialised for the next conv (the same HBM bytes per activation
as fp32 storage + prologue).  Same reference semantics as blocks.BottleneckFn (uemda/_resnets.py:92-112)."""
import torch
from torch.autograd import Function

from .. import ops, ops_bf16 as ob
from .blocks import _Link, _st_from, _st_tensor, grad_buffer, grad_ohwi

_on_bwd = ops.on_backward_stream          # every backward of a step on ONE stream (ops: two forward streams)


class CastFn(Function):
    """fp32 <-> bf16 at the two ends of the bf16 region (gradient cast back on the way down)."""

    @staticmethod
    def forward(ctx, x, to_bf16):
        ctx.to_bf16 = to_bf16
        return ob.to_bf16(x.contiguous()) if to_bf16 else ob.to_f32(x.contiguous())

    @staticmethod
    @_on_bwd
    def backward(ctx, dy):
        dy = dy.contiguous()
        return (ob.to_f32(dy) if ctx.to_bf16 else ob.to_bf16(dy)), None


class StemBf16Fn(Function):
    """conv7x7 s2 -> BN -> ReLU -> maxpool3x3 s2 with the conv output z, the pooled map and their gradients stored in bf16 (round 5).
    z -- 64 channels at half resolution -- is the largest tensor of the network: written once and read once forward, read twice and its
    gradient written once backward; in bf16 each of those passes moves half the bytes, and the cast in front of layer1 is gone.
    Same reference lines as blocks.StemFn (uemda/_resnets.py:149-153,205-212)."""

    @staticmethod
    def forward(ctx, x, resnet, *params):
        x4 = ops.nchw3_to_nhwc4(x)
        z, st = ob.stem_conv_bn(x4, ops.stem_weight_packed(resnet.conv1.weight), resnet.bn1)
        need = any(ctx.needs_input_grad)
        y, idx = ob.maxpool_affine_fwd(z, st, need)
        ops.nbt_inc(resnet.bn1)
        if need:
            ctx.resnet = resnet
            ctx.save_for_backward(x4, z, _st_tensor(st), idx)
        return y

    @staticmethod
    @_on_bwd
    def backward(ctx, dy):
        x4, z, stbuf, idx = ctx.saved_tensors
        resnet = ctx.resnet
        st = _st_from(stbuf, True)
        dz = ob.bn_backward_pooled(z, dy.contiguous(), idx, st, grad_buffer(resnet.bn1.weight), grad_buffer(resnet.bn1.bias))
        ob.stem_wgrad(x4, dz, grad_ohwi(resnet.conv1.weight))
        return (None, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class InstNormBf16Fn(Function):
    """nn.InstanceNorm2d at the end of the bf16 region: bf16 layer4 output in, fp32 features out; bf16 gradient back (Encoder.py:123,147)"""

    @staticmethod
    def forward(ctx, x, eps):
        y, invstd = ob.instnorm_fwd(x.contiguous(), eps)
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(y, invstd)
        return y

    @staticmethod
    @_on_bwd
    def backward(ctx, dy):
        y, invstd = ctx.saved_tensors
        return ob.instnorm_bwd(y, dy.contiguous(), invstd), None


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
        if blk.bn1.training:
            blk._nbt_add()
        if any(ctx.needs_input_grad):
            ctx.blk, ctx.has_ds, ctx.training = blk, has_ds, st1.training
            saved = [x, bits, z1, a1, z2, a2, z3, _st_tensor(st1), _st_tensor(st2), _st_tensor(st3)]
            if has_ds:
                saved += [zd, _st_tensor(std)]
            ctx.save_for_backward(*saved)
            # bn3's reduction pass rides in the next block's last data-gradient epilogue, the previous block's in ours
            # (blocks._Link, same protocol as the fp32 node)
            ctx.link_in = getattr(x, "_uem_link", None)
            ctx.link_out = None
            if st3.training:                        # frozen statistics (batchnorm_trainable=False): no reduction pass to hand on
                ctx.link_out = _Link(z3, bits, _st_tensor(st3))
                y._uem_link = ctx.link_out
        return y

    @staticmethod
    @_on_bwd
    def backward(ctx, dy):
        blk = ctx.blk
        sv = ctx.saved_tensors
        x, bits, z1, a1, z2, a2, z3 = sv[:7]
        st1, st2, st3 = (_st_from(b, ctx.training) for b in sv[7:10])
        s, d = blk.stride, blk.dilation
        G, gb = grad_ohwi, grad_buffer
        dy = dy.contiguous()
        # BN3 + residual + ReLU.  dp = dy*[y > 0] is not materialised when the tail epilogue takes the identity gradient: bn3's
        # backward, the downsample BatchNorm's backward and the identity path all read dy through the packed mask.
        tail_ok = ob.dgrad_tail_ok(x.shape, x.shape[-1])
        need_dp = not ctx.has_ds and not tail_ok
        lo = ctx.link_out
        # dy may be overwritten in place only when it is the buffer the next block's backward allocated for us
        own = lo is not None and lo.owns(dy)
        dp = None
        dz3 = dzd = None
        if ctx.has_ds:
            # bn3 and the downsample BatchNorm read the same gated dy: their apply passes run as one (dzd takes dy's place when it is ours)
            tiles = lo.tiles if own and lo.tiles is not None else None
            ds_bn = blk.downsample[1]
            pair = ob.bn_backward_pair(z3, sv[10], dy, bits, st3, _st_from(sv[11], ctx.training), tiles, gb(blk.bn3.weight),
                                       gb(blk.bn3.bias), gb(ds_bn.weight), gb(ds_bn.bias), dx2=dy if own else None)
            if pair is not None:
                dz3, dzd = pair
                if tiles is not None:
                    lo.tiles = None
        if dz3 is not None:
            pass
        elif own and lo.tiles is not None:
            dz3 = ob.bn_backward_from_partials(z3, dy, st3, lo.tiles, gb(blk.bn3.weight), gb(blk.bn3.bias), relu=2, bits=bits)
            lo.tiles = None
        elif need_dp:
            dz3, dp = ob.bn_backward(z3, dy, st3, gb(blk.bn3.weight), gb(blk.bn3.bias), relu=2, bits=bits, want_dres=True)
        else:
            dz3 = ob.bn_backward(z3, dy, st3, gb(blk.bn3.weight), gb(blk.bn3.bias), relu=2, bits=bits)
        ob.conv2d_wgrad(a2, dz3, G(blk.conv3.weight), side=True)
        dz2 = ob.conv2d_dgrad_bn_backward(dz3, ob.weight_t(blk.conv3.weight), z2, st2, gb(blk.bn2.weight), gb(blk.bn2.bias))
        del dz3
        ob.conv2d_wgrad(a1, dz2, G(blk.conv2.weight), stride=s, pad=d, dil=d, side=True)
        dz1 = ob.conv2d_dgrad_bn_backward(dz2, ob.weight_t(blk.conv2.weight), z1, st1, gb(blk.bn1.weight), gb(blk.bn1.bias),
                                          stride=s, pad=d, dil=d)
        del dz2
        ob.conv2d_wgrad(x, dz1, G(blk.conv1.weight), side=True)
        wt1 = ob.weight_t(blk.conv1.weight)
        li = ctx.link_in
        fuse = li is not None and li.z3.shape == x.shape and tail_ok
        bn_args = dict(bn_z=li.z3, bn_vec=li.vec, bn_bits=li.bits) if fuse else {}
        tp = None
        if ctx.has_ds:
            zd, std = sv[10], _st_from(sv[11], ctx.training)
            ds_conv, ds_bn = blk.downsample[0], blk.downsample[1]
            if dzd is None:
                dzd = ob.bn_backward(zd, dy, std, gb(ds_bn.weight), gb(ds_bn.bias), relu=2, bits=bits, dx=dy if own else None)
            ob.conv2d_wgrad(x, dzd, G(ds_conv.weight), stride=s, side=True)
            wtd = ob.weight_t(ds_conv.weight)
            if s == 1 and tail_ok:
                dx = ob.conv2d_dgrad(dzd, wtd, x.shape)
                dx, tp = ob.conv2d_dgrad_tail(dz1, wt1, x.shape, out=dx, accumulate=True, **bn_args)
            else:
                # strided downsample: its data gradient reaches one pixel in four, so it goes second (accumulating)
                dx = ob.conv2d_dgrad(dz1, wt1, x.shape)
                ob.conv2d_dgrad(dzd, wtd, x.shape, stride=s, out=dx, accumulate=True)
        elif tail_ok:
            dx, tp = ob.conv2d_dgrad_tail(dz1, wt1, x.shape, acc_src=dy, acc_bits=bits, out=dy if own else None, **bn_args)
        else:
            dx = ob.conv2d_dgrad(dz1, wt1, x.shape, out=dp, accumulate=True)      # identity gradient + conv1's
        if li is not None:
            li.hand_over(dx, tp)
        cb = getattr(blk, "_uem_after_backward", None)      # data-parallel bucket trigger (uemda_amd.dp)
        if cb is not None:
            ob.side_join()                                  # the bucket may go out now: this block's weight gradients first
            cb()
        return (dx, None) + (None,) * (len(ctx.needs_input_grad) - 2)
