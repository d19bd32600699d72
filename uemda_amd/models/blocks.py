"""autograd nodes of the segmentation network.  One coarse node per residual block / stem / head so
that every arithmetic step inside is a HIP kernel launch (uemda_amd.ops) and the saved state is
exactly what the hand-written backward needs:

  raw conv outputs z (pre-BatchNorm), the per-channel BN operands (scale/shift/mean/invstd) and the
  materialised block outputs.  BN-apply + ReLU between convs is never materialised: it is the
  operand prologue of the consuming conv (forward, wgrad) and is recomputed as a mask in backward.

Parameter gradients are accumulated straight into `param.grad` (views of the model's flat gradient
arena; conv wgrad lands there with fp32 atomics) and the nodes return None for them.
Reference semantics: uemda/_resnets.py:72-112,149-153,205-212; uemda/models/Encoder.py:68-84,123.
"""
import torch
from torch.autograd import Function

from .. import ops
from ..ops import UemError

_on_bwd = ops.on_backward_stream          # ops, "two streams for the step's two graphs"


def grad_buffer(p):
    """The tensor wgrad kernels accumulate into: p.grad, (re)attached to the flat arena when None.  None for a frozen
    parameter (requires_grad False: ResNetEncoder freeze_at / batchnorm_trainable, reference resnet.py:112-130): the kernels
    skip its gradient and its .grad stays None, as under torch autograd."""
    if not p.requires_grad:
        return None
    if ops._FWD2 and ops.on_second_stream():
        # a backward node of the step's second graph, on the second stream: it accumulates into the owner's shadow gradient arena,
        # folded into p.grad at the end of the pass (ops: two streams) -- never the same address as the first graph's node next door
        owner = getattr(p, "_uem_owner", None)
        if owner is None or not hasattr(p, "_uem_grad2_view"):
            raise UemError("a backward node is running on the second stream for a parameter outside a flat gradient arena")
        if p.grad is None:
            p.grad = p._uem_grad_view()
        return owner.shadow_grad(p)
    if p.grad is None:
        maker = getattr(p, "_uem_grad_view", None)
        g = maker() if maker is not None else torch.zeros_like(p)
        if maker is not None:
            g.zero_()
        p.grad = g
    return p.grad


def grad_ohwi(p):
    g = grad_buffer(p)
    if g is None:
        return None
    g = g.permute(0, 2, 3, 1)
    if not g.is_contiguous():
        raise UemError("conv weight .grad is not channels_last; let the model own its gradient arena")
    return g


class _BN:
    """view of an nn.BatchNorm2d used as a parameter/buffer holder"""

    @staticmethod
    def stats(x, bn):
        tr = bn.training or bn.running_mean is None
        return ops.bn_stats(x, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, tr,
                            bn.eps, bn.momentum if bn.momentum is not None else 0.1)


def _st_tensor(st):
    # the four vectors of a BNState live in one (4, C) buffer: recover it from the first view
    return st.scale._base if st.scale._base is not None else st.scale


def _st_from(buf, training=True):
    st = ops.BNState()
    st.scale, st.shift, st.mean, st.invstd = buf[0], buf[1], buf[2], buf[3]
    st.training = training
    return st


class StemFn(Function):
    """conv7x7 s2 -> BN -> ReLU -> maxpool3x3 s2   (_resnets.py:149-153, resnet.py:142-143)"""

    # operand precision of the 7x7 conv and its weight gradient (None = the global setting); "bf16" under bf16 storage
    prec = None

    @staticmethod
    def forward(ctx, x, resnet, *params):
        ctx.prec = StemFn.prec
        x4 = ops.nchw3_to_nhwc4(x)
        # BatchNorm statistics out of the conv epilogue; BatchNorm + ReLU applied in the max-pool's fetch: the 64-channel
        # half-resolution map (the largest activation of the network) is written once (z) and read once
        with ops.conv_precision(ctx.prec):
            z, st = ops.stem_conv_bn(x4, ops.weight_ohwi(resnet.conv1.weight), resnet.bn1, w8=lambda: ops.stem_weight_packed(resnet.conv1.weight))
        need = any(ctx.needs_input_grad)
        y, idx = ops.maxpool_affine_fwd(z, st, need)
        ops.nbt_inc(resnet.bn1)
        if need:
            ctx.resnet = resnet
            ctx.training = st.training
            ctx.save_for_backward(x4, z, _st_tensor(st), idx)
        return y

    @staticmethod
    @_on_bwd
    def backward(ctx, dy):
        x4, z, stbuf, idx = ctx.saved_tensors
        resnet = ctx.resnet
        st = _st_from(stbuf, ctx.training)
        if st.training and z.shape[1] % 2 == 0 and z.shape[2] % 2 == 0:
            # the pool's backward is read in gather form inside both BatchNorm-backward passes (no full-size da)
            dz = ops.bn_backward_pooled(z, dy.contiguous(), idx, st, grad_buffer(resnet.bn1.weight), grad_buffer(resnet.bn1.bias))
        else:
            da = ops.maxpool_bwd(dy.contiguous(), idx, z.shape)
            dz = ops.bn_backward(z, da, st, grad_buffer(resnet.bn1.weight), grad_buffer(resnet.bn1.bias), None, True, dx=da)
        with ops.conv_precision(ctx.prec):
            ops.stem_wgrad(x4, dz, grad_ohwi(resnet.conv1.weight))
        return (None, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class _Link:
    """What the NEXT block's backward needs to take this block's bn3 reduction pass into its last data-gradient epilogue
    (ops.conv2d_dgrad_tail): travels as an attribute of the block's output tensor; the next block's forward picks it up."""
    __slots__ = ("z3", "bits", "vec", "tiles", "dx", "dx_version")
    hits = 0                    # how often a block found its incoming gradient to be the handed-over tensor (tests read this)

    def __init__(self, z3, bits, vec):
        self.z3, self.bits, self.vec, self.tiles, self.dx, self.dx_version = z3, bits, vec, None, None, -1

    def hand_over(self, dx, tiles):
        """the next block's backward produced `dx` (this block's incoming gradient) and, with it, the partial sums `tiles` of this
        block's bn3 backward over exactly that tensor"""
        self.tiles, self.dx, self.dx_version = tiles, dx, dx._version

    def owns(self, dy):
        """Is `dy` the very tensor the next block's backward handed over, untouched since?  Ownership is POSITIVE: the
        same tensor OBJECT at the same version -- not merely the same address.  If autograd summed another consumer's gradient
        into it in place (InputBuffer add_), the object is the same but its version moved: the partial sums no longer describe it
        and it may not be overwritten.  (The kernels write through raw pointers and never move a version.)"""
        ok = self.dx is not None and dy is self.dx and dy._version == self.dx_version
        _Link.hits += int(ok)
        return ok


class BottleneckFn(Function):
    """1x1 -> BN -> ReLU -> 3x3(stride, dilation) -> BN -> ReLU -> 1x1 -> BN (+ downsample) -> add -> ReLU
    (_resnets.py:92-112).  `blk` carries conv1..3, bn1..3, downsample, stride, dilation."""

    @staticmethod
    def forward(ctx, x, blk, *params):
        W = ops.weight_ohwi
        s, d = blk.stride, blk.dilation
        z1, st1 = ops.conv2d_bn(x, W(blk.conv1.weight), blk.bn1)
        # the stride-1 3x3 convs take the Winograd paths (2.25x / 4x fewer multiplies, ops.wino_plan); the transformed input V is
        # what the weight gradient reduces over: kept for backward when both directions share a tile size and V fits the byte cap,
        # else recomputed there from z1
        v2 = None
        plan = ops.wino_plan(z1.shape, blk.conv2.weight.shape[0], blk.conv2.weight.shape[2], blk.conv2.weight.shape[3], s, d, d)
        if plan is not None and plan.mf and (blk.bn2.training or blk.bn2.running_mean is None) and ops.FUSE_BN_STATS:
            z2, st2, v2 = ops.conv3x3_wino_bn(z1, blk.conv2.weight, blk.bn2, d, in_scale=st1.scale, in_shift=st1.shift, in_relu=True, m=plan.mf)
        elif plan is not None and plan.mf:
            z2, v2 = ops.conv3x3_wino(z1, blk.conv2.weight, d, in_scale=st1.scale, in_shift=st1.shift, in_relu=True, want_v=True, m=plan.mf)
            st2 = _BN.stats(z2, blk.bn2)
        else:
            z2, st2 = ops.conv2d_bn(z1, W(blk.conv2.weight), blk.bn2, stride=s, pad=d, dil=d, in_scale=st1.scale,
                                    in_shift=st1.shift, in_relu=True)
        if plan is None or not plan.keep_v:
            v2 = None
        z3, st3 = ops.conv2d_bn(z2, W(blk.conv3.weight), blk.bn3, in_scale=st2.scale, in_shift=st2.shift, in_relu=True)
        has_ds = blk.downsample is not None
        need_bwd = any(ctx.needs_input_grad) and ops.RELU_BITS
        if has_ds:
            zd, std = ops.conv2d_bn(x, W(blk.downsample[0].weight), blk.downsample[1], stride=s)
            y = ops.affine_act(z3, st3, res=zd, res_st=std, relu=True, want_bits=need_bwd)
        else:
            zd, std = None, None
            y = ops.affine_act(z3, st3, res=x, relu=True, want_bits=need_bwd)
        y, ybits = y if need_bwd else (y, None)
        if blk.bn1.training:
            blk._nbt_add()
        if any(ctx.needs_input_grad):
            ctx.blk = blk
            ctx.has_ds = has_ds
            ctx.training = st1.training
            # the ReLU mask of the block output travels as packed bits (1/32 of y): y itself is not read in backward
            saved = [x, ybits if ybits is not None else y, z1, z2, z3, _st_tensor(st1), _st_tensor(st2), _st_tensor(st3)]
            if has_ds:
                saved += [zd, _st_tensor(std)]
            ctx.wino = plan
            if v2 is not None:
                saved.append(v2)
            ctx.save_for_backward(*saved)
            # bn3's reduction pass can ride in the next block's last data-gradient epilogue, and the previous block's in ours
            ctx.link_in = getattr(x, "_uem_link", None)
            ctx.link_out = None
            if ybits is not None and st3.training:
                ctx.link_out = _Link(z3, ybits, _st_tensor(st3))
                y._uem_link = ctx.link_out
        return y

    @staticmethod
    @_on_bwd
    def backward(ctx, dy):
        blk = ctx.blk
        sv = ctx.saved_tensors
        x, ybits, z1, z2, z3 = sv[:5]
        st1, st2, st3 = (_st_from(b, ctx.training) for b in sv[5:8])
        s, d = blk.stride, blk.dilation
        W, G, gb = ops.weight_ohwi, grad_ohwi, grad_buffer
        dy = dy.contiguous()
        packed = ybits.dtype == torch.int32
        # BN3 + residual + ReLU.  dp = dy*[y > 0] (the gradient of the pre-ReLU sum) is never materialised: bn3's backward,
        # the downsample BatchNorm's backward and the identity path all read dy through the packed mask.
        tail_ok = packed and ops.dgrad_tail_ok(x.shape, x.shape[-1])
        dp = torch.empty_like(dy) if not ctx.has_ds and not tail_ok else None      # shapes the tail epilogue does not take
        lo = ctx.link_out
        # dy may be overwritten in place only when it is the buffer the next block's backward allocated for us (a gradient
        # handed in by the caller, or one autograd summed from several consumers, is left alone)
        own = lo is not None and lo.owns(dy)
        dz3 = dzd = None
        if ctx.has_ds and packed:
            # bn3 and the downsample BatchNorm read the same gated dy: their apply passes run as one (dzd takes dy's place when it is ours)
            tiles = lo.tiles if own and lo.tiles is not None else None
            ds_bn = blk.downsample[1]
            pair = ops.bn_backward_pair(z3, sv[8], dy, ybits, st3, _st_from(sv[9], ctx.training), tiles, gb(blk.bn3.weight),
                                        gb(blk.bn3.bias), gb(ds_bn.weight), gb(ds_bn.bias), dx2=dy if own else None)
            if pair is not None:
                dz3, dzd = pair
                if tiles is not None:
                    lo.tiles = None
        if dz3 is not None:
            pass
        elif own and lo.tiles is not None:
            dz3 = ops.bn_backward_from_partials(z3, dy, st3, lo.tiles, gb(blk.bn3.weight), gb(blk.bn3.bias), ybits, dres=dp)
            lo.tiles = None
        else:
            kw = dict(ymask_bits=ybits) if packed else dict(ymask=ybits)
            dz3 = ops.bn_backward(z3, dy, st3, gb(blk.bn3.weight), gb(blk.bn3.bias), relu=True, dres=dp, **kw)
        ops.conv2d_wgrad(z2, dz3, G(blk.conv3.weight), in_scale=st2.scale, in_shift=st2.shift, in_relu=True, side=True)
        dz2 = ops.conv2d_dgrad_bn_backward(dz3, ops.weight_transpose_cached(blk.conv3.weight), z2, st2, gb(blk.bn2.weight),
                                           gb(blk.bn2.bias))
        del dz3
        plan = ctx.wino
        if plan is not None and plan.mb:
            mb = plan.mb
            if plan.keep_v:
                ops.conv3x3_wino_wgrad(sv[-1], dz2, G(blk.conv2.weight), d, side=True)
            elif plan.wgrad:
                ops.conv3x3_wino_wgrad(None, dz2, G(blk.conv2.weight), d, x=z1, in_scale=st1.scale, in_shift=st1.shift, in_relu=True, m=mb, side=True)
            else:
                ops.conv2d_wgrad(z1, dz2, G(blk.conv2.weight), stride=s, pad=d, dil=d, in_scale=st1.scale, in_shift=st1.shift, in_relu=True, side=True)
            if not plan.dgrad:
                dz1 = ops.conv2d_dgrad_bn_backward(dz2, ops.weight_transpose_cached(blk.conv2.weight), z1, st1, gb(blk.bn1.weight),
                                                   gb(blk.bn1.bias), stride=s, pad=d, dil=d)
            elif st1.training and ops.FUSE_BN_BACKWARD:
                dz1 = ops.conv3x3_wino_dgrad_bn_backward(dz2, blk.conv2.weight, z1, st1, gb(blk.bn1.weight), gb(blk.bn1.bias), d, m=mb)
            else:
                da1, _ = ops.conv3x3_wino_dgrad(dz2, blk.conv2.weight, d, m=mb)
                dz1 = ops.bn_backward(z1, da1, st1, gb(blk.bn1.weight), gb(blk.bn1.bias), None, True, dx=da1)
        else:
            ops.conv2d_wgrad(z1, dz2, G(blk.conv2.weight), stride=s, pad=d, dil=d, in_scale=st1.scale, in_shift=st1.shift, in_relu=True, side=True)
            dz1 = ops.conv2d_dgrad_bn_backward(dz2, ops.weight_transpose_cached(blk.conv2.weight), z1, st1, gb(blk.bn1.weight),
                                               gb(blk.bn1.bias), stride=s, pad=d, dil=d)
        del dz2
        ops.conv2d_wgrad(x, dz1, G(blk.conv1.weight), side=True)
        wt1 = ops.weight_transpose_cached(blk.conv1.weight)
        # the previous block's bn3 reduction rides in OUR last data-gradient launch when that launch has full dense tiles
        li = ctx.link_in
        fuse = li is not None and li.z3.shape == x.shape and ops.dgrad_tail_ok(x.shape, x.shape[-1])
        bn_args = dict(bn_z=li.z3, bn_vec=li.vec, bn_bits=li.bits) if fuse else {}
        tp = None
        if ctx.has_ds:
            zd, std = sv[8], _st_from(sv[9], ctx.training)
            ds_conv, ds_bn = blk.downsample[0], blk.downsample[1]
            if dzd is None:
                kw = dict(ymask_bits=ybits) if packed else dict(ymask=ybits)
                dzd = ops.bn_backward(zd, dy, std, gb(ds_bn.weight), gb(ds_bn.bias), relu=True, dx=dy if own else None, **kw)
            ops.conv2d_wgrad(x, dzd, G(ds_conv.weight), stride=s, side=True)
            wtd = ops.weight_transpose_cached(ds_conv.weight)
            if s == 1 and tail_ok:
                dx = ops.conv2d_dgrad(dzd, wtd, x.shape)
                dx, tp = ops.conv2d_dgrad_tail(dz1, wt1, x.shape, out=dx, accumulate=True, **bn_args)
            else:
                # strided downsample: its data gradient reaches one pixel in four, so it goes second (accumulating)
                dx = ops.conv2d_dgrad(dz1, wt1, x.shape)
                ops.conv2d_dgrad(dzd, wtd, x.shape, stride=s, out=dx, accumulate=True)
        elif tail_ok:
            dx, tp = ops.conv2d_dgrad_tail(dz1, wt1, x.shape, acc_src=dy, acc_bits=ybits, out=dy if own else None,
                                           **bn_args)                                      # identity gradient + conv1's
        else:
            dx = ops.conv2d_dgrad(dz1, wt1, x.shape, out=dp, accumulate=True)     # identity grad + conv1 dgrad
        if li is not None:
            li.hand_over(dx, tp)
        cb = getattr(blk, "_uem_after_backward", None)      # data-parallel bucket trigger (uemda_amd.dp)
        if cb is not None:
            ops.side_join()                                 # side-stream weight gradients (if any) land before a bucket goes out
            cb()
        return (dx, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class InstNormFn(Function):
    """nn.InstanceNorm2d(affine=False, track_running_stats=False)  (Encoder.py:123,147)"""

    @staticmethod
    def forward(ctx, x, eps):
        y, invstd = ops.instnorm_fwd(x, eps)
        if ctx.needs_input_grad[0]:
            ctx.save_for_backward(y, invstd)
        return y

    @staticmethod
    @_on_bwd
    def backward(ctx, dy):
        y, invstd = ctx.saved_tensors
        return ops.instnorm_bwd(y, dy.contiguous(), invstd), None


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * 8)(*([t.data_ptr() for t in tensors] + [None] * (8 - len(tensors))))


def _aspp_pack(heads, feat):
    """Wall (R, 1, 1, cin): row (d*9 + tap)*nh*C + head*C + c = W_head,d[c, tap, :]; bias (nd, nh, C) -- one kernel launch.
    heads: one or two Classifier_Modules."""
    nh = len(heads)
    C = heads[0].conv2d_list[0].weight.shape[0]
    cin = feat.shape[3]
    nd = len(heads[0].dilations)
    used = nd * 9 * nh * C
    R = (used + 63) // 64 * 64                       # GEMM N tile (64) and dgrad K block (32)
    if nd > 4 or nd * nh * C > 256:
        raise UemError("ASPP heads: at most 4 dilations and heads * nd * classes <= 256")
    wall = torch.empty((R, cin), device=feat.device, dtype=torch.float32)
    bias = torch.empty((nd, nh, C), device=feat.device, dtype=torch.float32)
    convs = [head.conv2d_list[i] for head in heads for i in range(nd)]
    ops.call("uem_aspp_pack", _ptr_array([ops.weight_ohwi(c.weight) for c in convs]), _ptr_array([c.bias.detach() for c in convs]),
             ops.ptr(wall), ops.ptr(bias), C, cin, nd, R, nh, ops.stream())
    return wall.view(R, 1, 1, cin), bias, C, nd, R, used


import os as _os
BF16_HEADS_DMA = _os.environ.get("UEM_BF16_HEADS_DMA", "1") != "0"      # bf16 storage: heads on the LDS-DMA bf16 kernels (0: rounds 2-5)


class ASPPHeadsFn(Function):
    """Both Classifier_Module heads (Encoder.py:68-84): the 2 heads x 4 dilations x 9 taps are the columns of
    ONE 1x1 GEMM G = feat x Wall on the MFMA kernel (feat is read once), then a 36-term gather rebuilds the
    dilated 3x3 sums; see uem_aspp_gather_* in include/uemda_hip.h.  head6 = None: ONE head (the class's single-head default and
    the cascade branch) -- half the columns, one output (round 5; rounds 1-4 ran the pair with the same module twice)."""

    # operand precision of the heads' GEMMs (None = the global setting); Deeplabv2 sets "bf16" for the bf16-storage model
    prec = None

    @staticmethod
    def forward(ctx, feat, head5, head6, *params):
        import ctypes
        heads = (head5,) if head6 is None else (head5, head6)
        nh = len(heads)
        wall, bias, C, nd, R, used = _aspp_pack(heads, feat)
        n, h, w, cin = feat.shape
        dil = (ctypes.c_int * nd)(*head5.dilations)
        ctx.prec = ASPPHeadsFn.prec
        # bf16 storage (round 6): the heads' GEMM and its two gradients on the LDS-DMA bf16 kernels the encoder runs
        # (operands rounded to bf16 either way; rounds 2-5 ran them on the register-staged fp32-tensor kernel with bf16 operands: 1.4 ms
        # of the 38 ms R50 step, 5.6 of R101-1024's 204).  feat / dG are cast once (the mining still reads the fp32 features), G and
        # dfeat come back through one cast each
        ctx.dma = ctx.prec == "bf16" and BF16_HEADS_DMA and cin % 64 == 0 and R % 64 == 0
        if ctx.dma:
            from .. import ops_bf16 as ob
            feat_b, wall_b = ob.to_bf16(feat), ob.to_bf16(wall)
            G = ob.to_f32(ob.conv2d(feat_b, wall_b))
        else:
            with ops.conv_precision(ctx.prec):
                G = ops.conv2d(feat, wall, algo_cout=used)
        x1 = torch.empty((n, h, w, C), device=feat.device, dtype=torch.float32)
        x2 = torch.empty((n, h, w, C), device=feat.device, dtype=torch.float32) if nh == 2 else None
        ops.call("uem_aspp_gather_fwd", ops.ptr(G), ops.ptr(bias), ops.ptr(x1), ops.ptr(x2), n, h, w, nh * C, R, nd, dil, ops.stream())
        if any(ctx.needs_input_grad):
            ctx.heads = heads
            ctx.feat_shape = feat.shape
            ctx.save_for_backward(feat_b if ctx.dma else feat, wall)
        return (x1, x2) if nh == 2 else x1

    @staticmethod
    @_on_bwd
    def backward(ctx, d1, d2=None):
        import ctypes
        feat, wall = ctx.saved_tensors
        heads = ctx.heads
        nh = len(heads)
        head5 = heads[0]
        C = head5.conv2d_list[0].weight.shape[0]
        nd = len(head5.dilations)
        n, h, w, cin = ctx.feat_shape
        R = wall.shape[0]
        used = nd * 9 * nh * C
        d1 = d1.contiguous()
        d2 = d2.contiguous() if nh == 2 else None
        db = torch.zeros(nh * C, device=feat.device, dtype=torch.float32)
        ops.bias_grad(d1, db[:C], C, C)
        if nh == 2:
            ops.bias_grad(d2, db[C:], C, C)
        dG = torch.empty((n, h, w, R), device=feat.device, dtype=torch.float32)
        dil = (ctypes.c_int * nd)(*head5.dilations)
        ops.call("uem_aspp_gather_bwd", ops.ptr(d1), ops.ptr(d2), ops.ptr(dG), n, h, w, nh * C, R, nd, dil, ops.stream())
        dwall = torch.zeros((R, 1, 1, cin), device=feat.device, dtype=torch.float32)
        if ctx.dma:
            from .. import ops_bf16 as ob
            dG_b = ob.to_bf16(dG)
            ob.conv2d_wgrad(feat, dG_b, dwall)                              # `feat` is the bf16 copy here
            wall_t = torch.empty((cin, 1, 1, R), device=feat.device, dtype=torch.bfloat16)
            ops.call("uem_weight_transpose_bf16", ops.ptr(wall), ops.ptr(wall_t), R, 1, 1, cin, ops.stream())
            dfeat = ob.to_f32(ob.conv2d_dgrad(dG_b, wall_t, ctx.feat_shape))
        else:
            with ops.conv_precision(ctx.prec):
                ops.conv2d_wgrad(feat, dG, dwall, algo_cout=used)
                dfeat = ops.conv2d_dgrad(dG, ops.weight_transpose(wall), feat.shape, algo_cout=used)
        convs = [head.conv2d_list[i] for head in heads for i in range(nd)]
        gws, gbs = [grad_ohwi(c.weight) for c in convs], [grad_buffer(c.bias) for c in convs]
        if all(g is not None for g in gws + gbs):
            ops.call("uem_aspp_unpack_grad", ops.ptr(dwall), ops.ptr(db), _ptr_array(gws), _ptr_array(gbs), C, cin, nd, nh, ops.stream())
        else:                                              # a frozen head parameter: per-tensor adds for the trainable ones
            dwv = dwall.view(R, cin)[:used].view(nd, 9, nh, C, cin)
            for k, conv in enumerate(convs):
                hd, i = divmod(k, nd)
                if gws[k] is not None:
                    ops.add_(gws[k], dwv[i, :, hd].permute(1, 0, 2).contiguous())
                if gbs[k] is not None:
                    ops.add_(gbs[k], db[hd * C:(hd + 1) * C])
        return (dfeat, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)
