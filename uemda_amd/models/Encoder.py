"""Deeplabv2 = ResNet encoder + InstanceNorm + two ASPP / PPM heads, MI355X-native.

Drop-in for reference `uemda/models/Encoder.py:87-186` (same constructor dict, same forward contract,
same `state_dict()` keys and OIHW layouts).  nn.Conv2d / nn.BatchNorm2d objects are
used purely as parameter and buffer HOLDERS (so the key names match the reference); their torch
`forward` is never called -- the forward/backward run through uemda_amd.models.blocks on hand-written
HIP kernels.  There is no CPU path: calling the model with CPU tensors raises.
"""
import math

import torch
import torch.nn as nn

from .. import ops
from ..ops import UemError
from ..resnet import ResNetEncoder
from . import blocks
from .config import AttrDict


class Classifier_Module(nn.Module):
    """ASPP head: parameter holder with the reference's layout (Encoder.py:68-84)."""

    def __init__(self, inplanes, dilation_series, padding_series, num_classes):
        super().__init__()
        self.dilations = tuple(dilation_series)
        self.conv2d_list = nn.ModuleList()
        for dilation, padding in zip(dilation_series, padding_series):
            if dilation != padding:
                raise UemError("Classifier_Module: padding must equal dilation")
            self.conv2d_list.append(nn.Conv2d(inplanes, num_classes, kernel_size=3, stride=1, padding=padding,
                                              dilation=dilation, bias=True))
        for m in self.conv2d_list:
            m.weight.data.normal_(0, 0.01)                     # Encoder.py:77-78


class PPMBilinear(nn.Module):
    """PPM head: parameter holder with the reference's layout (Encoder.py:8-41)."""

    def __init__(self, num_classes=7, fc_dim=2048, use_aux=False, pool_scales=(1, 2, 3, 6),
                 norm_layer=nn.BatchNorm2d):
        super().__init__()
        if use_aux:
            raise UemError("PPMBilinear(use_aux=True) is not on the UemDA path (never set by the scripts)")
        self.pool_scales = tuple(pool_scales)
        self.ppm = nn.ModuleList([nn.Sequential(nn.AdaptiveAvgPool2d(s), nn.Conv2d(fc_dim, 512, 1, bias=False),
                                                norm_layer(512), nn.ReLU(inplace=True)) for s in pool_scales])
        self.conv_last = nn.Sequential(
            nn.Conv2d(fc_dim + len(pool_scales) * 512, 512, kernel_size=3, padding=1, bias=False),
            norm_layer(512), nn.ReLU(inplace=True), nn.Dropout2d(0.1), nn.Conv2d(512, num_classes, kernel_size=1))


class Deeplabv2(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = AttrDict()
        self.set_default_config()
        self.config.update(config)
        cfg = self.config
        self.encoder = ResNetEncoder(cfg.backbone)

        def head(inchannels, ppm_cfg):
            if cfg.use_ppm:
                if ppm_cfg is None:
                    raise UemError("Deeplabv2(cascade=True, use_ppm=True) reads config['ppm1'] / config['ppm2'] (Encoder.py:94-96)")
                return PPMBilinear(**{k: v for k, v in ppm_cfg.items() if k != "norm_layer"})
            return Classifier_Module(inchannels, [6, 12, 18, 24], [6, 12, 18, 24], cfg.num_classes)
        if cfg.multi_layer and cfg.cascade:                                       # Encoder.py:93-102: layer5 on the layer3 output
            self.layer5 = head(cfg.inchannels // 2, cfg.get("ppm1", None))
            self.layer6 = head(cfg.inchannels, cfg.get("ppm2", None))
        elif cfg.multi_layer:                                                     # Encoder.py:103-110: what every UemDA script builds
            self.layer5 = head(cfg.inchannels, cfg.ppm)
            self.layer6 = head(cfg.inchannels, cfg.ppm)
        else:                                                                     # Encoder.py:111-116: the class's default
            self.cls_pred = head(cfg.inchannels, cfg.ppm)
        if cfg.is_ins_norm:                                                       # holders (no parameters), Encoder.py:118-123
            if cfg.cascade:
                self.instance_norm1 = nn.InstanceNorm2d(cfg.inchannels)
                self.instance_norm2 = nn.InstanceNorm2d(cfg.inchannels)
            else:
                self.instance_norm = nn.InstanceNorm2d(cfg.inchannels)
        self._arena = None
        self._grad_arena = None

    def set_default_config(self):
        self.config.update(dict(
            backbone=dict(resnet_type='resnet50', output_stride=16, pretrained=True),
            multi_layer=False, cascade=False, use_ppm=False,
            ppm=dict(num_classes=7, use_aux=False, norm_layer=nn.BatchNorm2d),
            inchannels=2048, num_classes=7, is_ins_norm=False))

    # ---- flat parameter / gradient arenas ----------------------------------------------------------------
    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._flatten_parameters()
        return out

    def _flatten_parameters(self):
        """Re-home every parameter in ONE flat fp32 buffer (conv weights in channels_last = OHWI order) and
        give it a matching slot in a flat gradient buffer: the fused optimizer and the RCCL gradient
        all-reduce then work on two contiguous arrays."""
        params = [p for p in self.parameters()]
        if not params:
            return
        dev = params[0].device
        # every tensor starts on a 16-byte boundary: the conv kernels fetch weights by 16-byte LDS-DMA and fall back to the
        # register-staged loop for a misaligned filter bank (a 6-element bias used to shift everything behind it by 24 bytes:
        # the second PPM head's 4096 -> 512 conv ran at half speed); the padding floats stay zero under the optimizer
        total = sum((p.numel() + 3) // 4 * 4 for p in params)
        arena = torch.zeros(total, device=dev, dtype=torch.float32)
        garena = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        for p in params:
            n = p.numel()

            def view_of(buf, p=p, off=off, n=n):
                flat = buf[off:off + n]
                if p.dim() == 4:
                    o, i, kh, kw = p.shape
                    return flat.view(o, kh, kw, i).permute(0, 3, 1, 2)
                return flat.view(p.shape)
            v = view_of(arena)
            v.copy_(p.data)
            p.data = v
            had_grad = p.grad is not None
            if had_grad:
                gv = view_of(garena)
                gv.copy_(p.grad)
                p.grad = gv
            p._uem_grad_view = (lambda vo=view_of, ga=garena: vo(ga))
            p._uem_grad2_view = (lambda vo=view_of, me=self: vo(me._shadow_grad_arena()))
            p.__dict__.pop("_uem_g2", None)
            p._uem_owner = self
            p._uem_off = off
            off += (n + 3) // 4 * 4
        self._arena, self._grad_arena, self._n_params = arena, garena, total
        self._grad_arena2, self._g2_dirty, self._g2_hi, self._g2_task = None, False, total, None
        # one int64 arena for every BatchNorm's num_batches_tracked: a training forward bumps all of them with ONE
        # add instead of one tiny launch per layer (53 per forward on ResNet-50)
        bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d) and m.num_batches_tracked is not None]
        nbt = torch.zeros(max(len(bns), 1), device=dev, dtype=torch.int64)
        for i, bn in enumerate(bns):
            nbt[i] = bn.num_batches_tracked
            bn._buffers["num_batches_tracked"] = nbt[i]
            bn._uem_nbt_arena = True
        self._nbt, self._bns = nbt, bns
        # ... and one fp32 arena for every BatchNorm's running mean / variance, with a zeroed SHADOW of the same layout: while the
        # step's second forward runs on its own stream (step.forward_pair) the modules' buffers point into the shadow, whose zeros
        # turn the kernels' `(1 - m) * r + m * v` into the EMA contribution m * v; `apply_shadow_running_stats` then applies it to the
        # real statistics behind the join, in the reference's order (source batch first) and to the last bit of the sequential form
        tracked = [bn for bn in bns if bn.running_mean is not None and bn.running_var is not None]
        sizes = [(bn.running_mean.numel() + 3) // 4 * 4 for bn in tracked]
        rs = torch.zeros(max(2 * sum(sizes), 4), device=dev, dtype=torch.float32)
        self._rs_views = []
        off = 0
        for bn, n_ in zip(tracked, sizes):
            c = bn.running_mean.numel()
            for name in ("running_mean", "running_var"):
                view = rs[off:off + c]
                view.copy_(bn._buffers[name])
                bn._buffers[name] = view
                self._rs_views.append((bn, name, off, c))
                off += n_
        self._rs, self._rs_shadow, self._rs_tracked = rs, torch.zeros_like(rs), tracked
        self._nbt_skip = False

    # ---- the step's second forward on its own stream (step.forward_pair) --------------------------------------------
    def two_stream_ok(self):
        """may the second train-mode forward of a step run beside the first?  Every BatchNorm in training mode, tracking its running
        statistics with one common momentum; no checkpointed layer (its re-run inside backward would queue work on the second stream)"""
        if self._arena is None or not self.training or not self._rs_tracked or len(self._rs_tracked) != len(self._bns):
            return False
        m0 = self._rs_tracked[0].momentum
        if m0 is None or any((not bn.training) or bn.momentum != m0 for bn in self._bns):
            return False
        return not any(self.encoder.config.with_cp)

    def shadow_running_stats(self):
        """context manager: inside it every BatchNorm's running-statistics buffers are views of the zeroed shadow arena and the
        num_batches_tracked bump is skipped (the caller bumps it on the main stream)"""
        import contextlib

        @contextlib.contextmanager
        def cm():
            for bn, name, off, c in self._rs_views:
                bn._buffers[name] = self._rs_shadow[off:off + c]
            self._nbt_skip = True
            try:
                yield
            except BaseException:
                if self._rs_shadow.is_cuda:      # a forward that raised half way: its partial contribution must not reach the statistics
                    torch.cuda.synchronize()     # (error path: simply let whatever stream it ran on finish first)
                self._rs_shadow.zero_()
                raise
            finally:
                self._nbt_skip = False
                for bn, name, off, c in self._rs_views:
                    bn._buffers[name] = self._rs[off:off + c]
        return cm()

    def apply_shadow_running_stats(self):
        """running = (1 - momentum) * running + shadow; shadow = 0 -- on the current stream, behind the join of the two forwards: what
        the second forward would have done to the statistics the first one left, multiplication and addition rounded as in the
        kernels' own update (bn_stats_finish)"""
        m = float(self._rs_tracked[0].momentum)
        ops.call("uem_scale", ops.ptr(self._rs), self._rs.numel(), 1.0 - m, ops.stream())
        ops.add_(self._rs, self._rs_shadow)
        self._rs_shadow.zero_()
        self._nbt_step()

    # ---- the second graph's backward on its own stream: the shadow gradient arena (ops: two streams) ------------------
    def _shadow_grad_arena(self):
        if self._grad_arena2 is None:
            self._grad_arena2 = torch.zeros_like(self._grad_arena)
        return self._grad_arena2

    def shadow_grad(self, p):
        """the slot of parameter `p` in the shadow gradient arena (same layout as p.grad), marked as holding gradients to fold"""
        g = p.__dict__.get("_uem_g2")
        if g is None:
            g = p._uem_g2 = p._uem_grad2_view()
        if not self._g2_dirty:
            self._g2_dirty = True
        ops.shadow_grads_touched(self)
        return g

    def fold_shadow_grads(self, lo=0, synced=False):
        """gradient arena[lo:hi] += shadow[lo:hi]; shadow[lo:hi] = 0 on the current stream, once it has waited for the side stream and
        the second stream (`synced`: the caller has ordered the current stream behind the writers of [lo:hi] itself); hi is the start
        of what an earlier partial fold (the data-parallel early bucket) already covered.  Runs as the end-of-backward callback of a
        pass that touched the shadow, and from ops.grad_join."""
        if not self._g2_dirty or self._grad_arena2 is None:
            return
        hi = self._g2_hi
        ops.side_join()
        second = ops._FWD2.get(self._grad_arena.device.index)
        cur = torch.cuda.current_stream()
        if second is not None and cur != second and not synced:
            cur.wait_stream(second)
        if hi > lo:
            a, b = self._grad_arena[lo:hi], self._grad_arena2[lo:hi]
            ops.call("uem_add_clear", ops.ptr(a), ops.ptr(b), a.numel(), ops.stream())
        if lo > 0:
            self._g2_hi = lo                 # the rest is folded at the end of the pass
        else:
            self._g2_hi, self._g2_dirty, self._g2_task = self._n_params, False, None

    def _nbt_step(self):
        """BatchNorm2d.forward: `num_batches_tracked += 1` for every layer in training mode."""
        if self._nbt_skip:
            return
        if all(bn.training for bn in self._bns):
            self._nbt.add_(1)
        else:
            for bn in self._bns:
                if bn.training:
                    bn.num_batches_tracked.add_(1)

    def flat_parameters(self):
        return self._arena, self._grad_arena, self._n_params

    def set_storage(self, storage):
        """"fp32" (default: the reference's arithmetic, every parity statement) or "bf16" (BASELINE config 5: bf16
        activations + bf16 copies of the fp32 master weights inside the encoder, bf16 matrix cores; training mode)."""
        if storage not in ("fp32", "bf16"):
            raise UemError(f"storage must be 'fp32' or 'bf16', got {storage!r}")
        self.encoder.storage = storage
        return self

    def zero_grad(self, set_to_none=False):
        """Zero the flat gradient arena (one memset) and keep every .grad attached to it."""
        if self._grad_arena is None:
            return super().zero_grad(set_to_none)
        ops.grad_join()             # weight gradients a failed backward left on the side / second stream must not land after the memset
        self._grad_arena.zero_()
        for p in self.parameters():
            if p.grad is None and p.requires_grad:           # a frozen parameter keeps .grad = None, as under torch autograd
                p.grad = p._uem_grad_view()

    # ---- forward ----------------------------------------------------------------------------------------
    def _head(self, feat, head):
        """one Classifier_Module / PPMBilinear on its own feature map (the single-head default, the cascade branch)"""
        bf16 = self.encoder.storage == "bf16" and self.training
        if self.config.use_ppm:
            from . import ppm
            ppm.PPMHeadFn.prec = "bf16" if bf16 else None
            try:
                return ppm.ppm_head(feat, head)
            finally:
                ppm.PPMHeadFn.prec = None
        # one Classifier_Module: the heads' GEMM with half the columns and one output (rounds 1-4 ran the two-head GEMM with
        # the same module on both sides and discarded the second output)
        blocks.ASPPHeadsFn.prec = "bf16" if bf16 else None
        try:
            return blocks.ASPPHeadsFn.apply(feat, head, None, *list(head.parameters()))
        finally:
            blocks.ASPPHeadsFn.prec = None

    def _heads(self, feat):
        bf16 = self.encoder.storage == "bf16" and self.training
        if self.config.use_ppm:
            from . import ppm
            # bf16 storage (training): the heads' convs take bf16 operands (fp32 tensors in memory, fp32 accumulate)
            ppm.PPMHeadFn.prec = "bf16" if bf16 else None
            try:
                with ppm.shared_pools():         # the pooled maps are shared by the two heads of THIS forward only
                    return ppm.ppm_head(feat, self.layer5), ppm.ppm_head(feat, self.layer6)
            finally:
                ppm.PPMHeadFn.prec = None
        params = list(self.layer5.parameters()) + list(self.layer6.parameters())
        # bf16 storage (training): the two heads' GEMM takes bf16 operands (fp32 feat / logits in memory, fp32 accumulate)
        blocks.ASPPHeadsFn.prec = "bf16" if bf16 else None
        try:
            return blocks.ASPPHeadsFn.apply(feat, self.layer5, self.layer6, *params)
        finally:
            blocks.ASPPHeadsFn.prec = None

    def _prob(self, x1, x2, size):
        """(softmax(up(x1)) + softmax(up(x2))) / 2, bilinear align_corners=True (Encoder.py:139-141,153-155); x2 = x1 gives the
        single head's softmax(up(x1)) (:164-165): (a + a) / 2 is a, exactly"""
        n, h, w, c = x1.shape
        H, W = size
        prob = torch.empty((n, c, H, W), device=x1.device, dtype=torch.float32)
        ops.call("uem_upsample_softmax_avg", ops.ptr(x1.contiguous()), ops.ptr(x2.contiguous()), ops.ptr(prob),
                 n, c, h, w, H, W, ops.stream())
        return prob

    def forward(self, x):
        ops.need_gpu(x)
        if self._arena is None:
            raise UemError("Deeplabv2: move the model to the MI355X device first (model.cuda())")
        cfg = self.config
        # bf16 storage: the InstanceNorm reads the bf16 layer4 output itself (no cast pass in front of it)
        fuse_in = bool(cfg.is_ins_norm) and not (cfg.multi_layer and cfg.cascade)
        stages = self.encoder.forward_nhwc(x, last_bf16=fuse_in)
        if cfg.multi_layer and cfg.cascade:                                       # Encoder.py:129-143
            if self.encoder.storage != "fp32":
                raise UemError("Deeplabv2(cascade=True) reads the layer3 output as an fp32 map: use fp32 storage")
            feat1, feat2 = stages[-2:]
            if cfg.is_ins_norm:
                feat1 = blocks.InstNormFn.apply(feat1, self.instance_norm1.eps)
                feat2 = blocks.InstNormFn.apply(feat2, self.instance_norm2.eps)
            x1, x2 = self._head(feat1, self.layer5), self._head(feat2, self.layer6)
            if self.training:
                self._nbt_step()
                return ops.as_nchw_view(x1), ops.as_nchw_view(feat1), ops.as_nchw_view(x2), ops.as_nchw_view(feat2)
            return self._prob(x1, x2, x.shape[-2:])
        feat = stages[-1]                                                         # Encoder.py:145
        if cfg.is_ins_norm:                                                       # Encoder.py:146-147
            if feat.dtype == torch.bfloat16:
                from .blocks_bf16 import InstNormBf16Fn
                feat = InstNormBf16Fn.apply(feat, self.instance_norm.eps)
            else:
                feat = blocks.InstNormFn.apply(feat, self.instance_norm.eps)
        if not cfg.multi_layer:                                                   # Encoder.py:156-165
            x1 = self._head(feat, self.cls_pred)
            if self.training:
                self._nbt_step()
                return ops.as_nchw_view(x1), ops.as_nchw_view(feat)
            return self._prob(x1, x1, x.shape[-2:])
        x1, x2 = self._heads(feat)
        if self.training:
            self._nbt_step()
            return ops.as_nchw_view(x1), ops.as_nchw_view(x2), ops.as_nchw_view(feat)   # Encoder.py:150-151
        return self._prob(x1, x2, x.shape[-2:])                                   # Encoder.py:153-155
