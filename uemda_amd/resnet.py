"""ResNet-50/101 encoder (OS16: layer4 de-strided and dilated), MI355X-native.

Mirrors reference `uemda/resnet.py:43-208` (ResNetEncoder) and `uemda/_resnets.py:72-227` (Bottleneck,
ResNet): same module tree => same state_dict keys.  The modules hold parameters; the arithmetic runs in
uemda_amd.models.blocks (HIP kernels)."""
from functools import partial

import torch
import torch.nn as nn

from . import ops
from .models import blocks
from .models.config import AttrDict
from .ops import UemError


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    @property
    def stride(self):
        return self.conv2.stride[0]

    @property
    def dilation(self):
        return self.conv2.dilation[0]

    def _bns(self):
        bns = [self.bn1, self.bn2, self.bn3]
        if self.downsample is not None:
            bns.append(self.downsample[1])
        return bns

    def _nbt_add(self):
        for bn in self._bns():
            ops.nbt_inc(bn)

    def forward(self, x):                      # x: (N,H,W,C) dense NHWC
        if x.dtype == torch.bfloat16:           # bf16-storage region (ResNetEncoder.storage = "bf16", BASELINE config 5)
            from .models.blocks_bf16 import BottleneckBf16Fn
            return BottleneckBf16Fn.apply(x, self, *list(self.parameters()))
        return blocks.BottleneckFn.apply(x, self, *list(self.parameters()))


class ResNet(nn.Module):
    def __init__(self, layers):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], stride=2)
        self.layer3 = self._make_layer(256, layers[2], stride=2)
        self.layer4 = self._make_layer(512, layers[3], stride=2)
        for m in self.modules():                                   # _resnets.py:164-169
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, nblocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        for _ in range(1, nblocks):
            layers.append(Bottleneck(self.inplanes, planes))
        return nn.Sequential(*layers)


_LAYERS = {"resnet50": [3, 4, 6, 3], "resnet101": [3, 4, 23, 3]}


class ResNetEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = AttrDict()
        self.config.update(dict(resnet_type='resnet50', include_conv5=True, batchnorm_trainable=True,
                                pretrained=False, freeze_at=0, output_stride=32,
                                with_cp=(False, False, False, False)))          # resnet.py:170-181
        self.config.update(config)
        cfg = self.config
        if cfg.output_stride not in (8, 16, 32):
            raise ValueError('output_stride must be 8, 16 or 32.')               # resnet.py:48-51
        if cfg.resnet_type not in _LAYERS:
            raise UemError(f"resnet_type {cfg.resnet_type!r} not supported (resnet50 / resnet101)")
        if len(tuple(cfg.with_cp)) != 4:
            raise UemError("with_cp takes four flags, one per residual layer")
        self.resnet = ResNet(_LAYERS[cfg.resnet_type])
        if isinstance(cfg.pretrained, str):
            # no network on the box: a local torchvision-format checkpoint path may be given instead of True
            sd = torch.load(cfg.pretrained, map_location="cpu")
            sd = sd.get("state_dict", sd)
            self.resnet.load_state_dict({k: v for k, v in sd.items() if not k.startswith("fc.")}, strict=False)
        if not cfg.batchnorm_trainable:                                          # resnet.py:57-58
            self._frozen_res_bn()
        self._freeze_at(cfg.freeze_at)                                           # resnet.py:60
        if cfg.output_stride == 16:                                              # resnet.py:62-63
            self.resnet.layer4.apply(partial(self._nostride_dilate, dilate=2))
        elif cfg.output_stride == 8:
            self.resnet.layer3.apply(partial(self._nostride_dilate, dilate=2))
            self.resnet.layer4.apply(partial(self._nostride_dilate, dilate=4))

    @staticmethod
    def _nostride_dilate(m, dilate):                                            # resnet.py:192-207
        if isinstance(m, nn.Conv2d):
            if m.stride == (2, 2):
                m.stride = (1, 1)
                if m.kernel_size == (3, 3):
                    m.dilation = (dilate // 2, dilate // 2)
                    m.padding = (dilate // 2, dilate // 2)
            elif m.kernel_size == (3, 3):
                m.dilation = (dilate, dilate)
                m.padding = (dilate, dilate)

    def _frozen_res_bn(self):                                                   # resnet.py:112-117
        """Every BatchNorm of the encoder: parameters frozen, running statistics used (eval mode) also while training."""
        for m in self.resnet.modules():
            if isinstance(m, nn.modules.batchnorm._BatchNorm):
                for p in m.parameters():
                    p.requires_grad = False
                m.eval()

    def _freeze_at(self, at=2):                                                 # resnet.py:119-130
        """Freeze the parameters of the stem (at >= 1) and of layer1 .. layer4 (at >= 2 .. 5): no gradient is computed for
        them (blocks.grad_buffer), the optimizer leaves them alone (FusedSGD._trainable_ranges), and the backward pass stops at
        the first trainable layer.  Their BatchNorms still normalise with batch statistics and update the running ones,
        as in the reference (freeze_params touches requires_grad only)."""
        r = self.resnet
        frozen = [[r.conv1, r.bn1], [r.layer1], [r.layer2], [r.layer3], [r.layer4]][:max(0, min(int(at), 5))]
        for group in frozen:
            for m in group:
                for p in m.parameters():
                    p.requires_grad = False

    def train(self, mode=True):                                                 # resnet.py:183-190
        super().train(mode)
        self._freeze_at(self.config.freeze_at)
        if mode and not self.config.batchnorm_trainable:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return self

    def _run_layer(self, layer, y, with_cp):
        """One residual layer; with_cp: under torch.utils.checkpoint as the reference does (resnet.py:146-165) -- only the
        layer input is kept, the layer's forward runs again inside backward.  As there, the second run updates the
        BatchNorm running statistics (and num_batches_tracked) a second time."""
        def run(t):
            if torch.is_grad_enabled():                   # the re-run inside backward: torch's BatchNorm would count it
                for m in layer.modules():
                    if isinstance(m, nn.BatchNorm2d) and m.training and getattr(m, "_uem_nbt_arena", False):
                        m.num_batches_tracked.add_(1)
            for blk in layer:
                t = blk(t)
            return t
        if with_cp and y.requires_grad:
            import torch.utils.checkpoint as cp
            return cp.checkpoint(run, y, use_reentrant=True)
        for blk in layer:
            y = blk(y)
        return y

    # "fp32" (default, the parity path) or "bf16": activations between the max-pool and the layer4 output stored in bf16,
    # bf16 matrix cores with fp32 accumulation, fp32 master weights (BASELINE config 5; training mode only)
    storage = "fp32"

    def forward_nhwc(self, x, last_bf16=False):
        """-> the four stage outputs (NHWC).  last_bf16 (bf16 storage only): leave the layer4 output in bf16 for a consumer that
        reads bf16 itself (Deeplabv2's InstanceNorm: blocks_bf16.InstNormBf16Fn); otherwise it is cast to fp32 here."""
        r = self.resnet
        params = [r.conv1.weight, r.bn1.weight, r.bn1.bias]
        # bf16 storage: the training step, and (round 3) inference under torch.no_grad() -- the offline pseudo-label pass and
        # evaluation; an eval-mode forward that records a graph stays fp32
        bf16 = self.storage == "bf16" and (self.training or not torch.is_grad_enabled())
        if bf16:
            from .models.blocks_bf16 import CastFn, StemBf16Fn
            from . import ops_bf16 as ob
        if bf16 and self.training and ob.stem_ok(x.shape, r.bn1):
            y = StemBf16Fn.apply(x, r, *params)            # round 5: the stem's z, pooled map and gradients in bf16 (no cast)
        else:
            blocks.StemFn.prec = "bf16" if (bf16 and self.training) else None   # bf16 operands for the stem conv in training (fp32 in memory)
            try:
                y = blocks.StemFn.apply(x, r, *params)
            finally:
                blocks.StemFn.prec = None
            if bf16:
                y = CastFn.apply(y, True)
        outs = []
        for layer, with_cp in zip((r.layer1, r.layer2, r.layer3, r.layer4), self.config.with_cp):
            y = self._run_layer(layer, y, with_cp)
            outs.append(y)
        if bf16 and not last_bf16:
            outs[-1] = CastFn.apply(outs[-1], False)       # the heads stay fp32
        return outs

    def forward(self, inputs):                                                  # resnet.py:140-166
        return [ops.as_nchw_view(t) for t in self.forward_nhwc(inputs)]
