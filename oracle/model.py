"""Oracle (CPU, fp32) restatement of the segmentation network.  TEST INFRASTRUCTURE ONLY.

Functional re-statement of reference
  uemda/models/Encoder.py:87-165   Deeplabv2: the multi_layer branch every script instantiates (:103-110,144-155), the single-head
                                   default (multi_layer=False, :111-116,156-165) and the cascade branch (:93-102,129-143)
  uemda/models/Encoder.py:68-84    Classifier_Module (ASPP head)
  uemda/models/Encoder.py:8-65     PPMBilinear head
  uemda/resnet.py:43-208           ResNetEncoder (OS16 => layer4 de-strided + dilated)
  uemda/_resnets.py:72-227         Bottleneck / ResNet

It is written over a flat {state_dict key: tensor} mapping (same keys and OIHW layouts as the
reference's `state_dict()`) with plain torch.nn.functional calls, so that the
same weights can be pushed through the reference (golden generator), this oracle and the HIP
path.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

RESNET_BLOCKS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3)}
ASPP_DILATIONS = (6, 12, 18, 24)      # Encoder.py:107-110
PPM_SCALES = (1, 2, 3, 6)             # Encoder.py:10
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def layer_plan(resnet_type="resnet50", output_stride=16):
    """[(prefix, inplanes, planes, stride, dilation, has_downsample)] for every Bottleneck.

    Follows _resnets.py:181-203 (_make_layer) then resnet.py:62-63,192-207 (_nostride_dilate
    applied to layer4 with dilate=2 when output_stride == 16): a conv whose stride was 2 gets
    stride 1 and (3x3 only) dilation dilate//2; every other 3x3 conv gets dilation `dilate`.
    """
    assert output_stride in (16, 32)
    plan = []
    inplanes = 64
    for li, (planes, nblk) in enumerate(zip((64, 128, 256, 512), RESNET_BLOCKS[resnet_type])):
        layer_stride = 1 if li == 0 else 2
        for bi in range(nblk):
            stride = layer_stride if bi == 0 else 1
            dilation = 1
            has_ds = bi == 0 and (stride != 1 or inplanes != planes * 4)
            if li == 3 and output_stride == 16:
                if stride == 2:
                    stride, dilation = 1, 1          # dilate // 2 == 1
                else:
                    dilation = 2
            plan.append((f"encoder.resnet.layer{li + 1}.{bi}", inplanes, planes, stride, dilation, has_ds))
            inplanes = planes * 4
    return plan


def param_shapes(resnet_type="resnet50", num_classes=6, use_ppm=False, fc_dim=2048, multi_layer=True, cascade=False):
    """OrderedDict key -> shape, in the reference's state_dict order.  multi_layer=False: one head `cls_pred`
    (Encoder.py:111-116); cascade: layer5 on the layer3 output (fc_dim // 2 channels), layer6 on the layer4 output (:93-102)."""
    sd = OrderedDict()

    def bn(prefix, c):
        sd[prefix + ".weight"] = (c,)
        sd[prefix + ".bias"] = (c,)
        sd[prefix + ".running_mean"] = (c,)
        sd[prefix + ".running_var"] = (c,)
        sd[prefix + ".num_batches_tracked"] = ()

    sd["encoder.resnet.conv1.weight"] = (64, 3, 7, 7)
    bn("encoder.resnet.bn1", 64)
    for prefix, inpl, planes, _s, _d, has_ds in layer_plan(resnet_type):
        sd[prefix + ".conv1.weight"] = (planes, inpl, 1, 1)
        bn(prefix + ".bn1", planes)
        sd[prefix + ".conv2.weight"] = (planes, planes, 3, 3)
        bn(prefix + ".bn2", planes)
        sd[prefix + ".conv3.weight"] = (planes * 4, planes, 1, 1)
        bn(prefix + ".bn3", planes * 4)
        if has_ds:
            sd[prefix + ".downsample.0.weight"] = (planes * 4, inpl, 1, 1)
            bn(prefix + ".downsample.1", planes * 4)
    heads = [("layer5", fc_dim // 2 if cascade else fc_dim), ("layer6", fc_dim)] if multi_layer else [("cls_pred", fc_dim)]
    for head, fc_dim in heads:
        if use_ppm:
            for i in range(4):
                sd[f"{head}.ppm.{i}.1.weight"] = (512, fc_dim, 1, 1)
                bn(f"{head}.ppm.{i}.2", 512)
            sd[f"{head}.conv_last.0.weight"] = (512, fc_dim + 4 * 512, 3, 3)
            bn(f"{head}.conv_last.1", 512)
            sd[f"{head}.conv_last.4.weight"] = (num_classes, 512, 1, 1)
            sd[f"{head}.conv_last.4.bias"] = (num_classes,)
        else:
            for i in range(4):
                sd[f"{head}.conv2d_list.{i}.weight"] = (num_classes, fc_dim, 3, 3)
                sd[f"{head}.conv2d_list.{i}.bias"] = (num_classes,)
    return sd


class OracleDeeplabv2:
    """Functional Deeplabv2 over a state_dict-keyed tensor mapping (CPU, fp32)."""

    def __init__(self, state, resnet_type="resnet50", num_classes=6, use_ppm=False,
                 is_ins_norm=True, requires_grad=True, freeze_at=0, batchnorm_trainable=True,
                 with_cp=(False, False, False, False), multi_layer=True, cascade=False):
        """freeze_at / batchnorm_trainable / with_cp: the ResNetEncoder options of reference uemda/resnet.py:57-60 (ctor),
        :112-130 (_frozen_res_bn, _freeze_at), :146-165 (torch.utils.checkpoint per layer), :183-190 (train())."""
        self.freeze_at, self.batchnorm_trainable, self.with_cp = int(freeze_at), bool(batchnorm_trainable), tuple(with_cp)
        self.multi_layer, self.cascade = bool(multi_layer), bool(cascade)
        self.resnet_type = resnet_type
        self.num_classes = num_classes
        self.use_ppm = use_ppm
        self.is_ins_norm = is_ins_norm
        self.training = True
        self.p = OrderedDict()
        for k, v in state.items():
            t = torch.as_tensor(v).detach().clone()
            if t.is_floating_point():
                t = t.float()
                if requires_grad and not ("running_" in k):
                    t.requires_grad_(True)
            self.p[k] = t
        self.plan = layer_plan(resnet_type)
        # resnet.py:119-130: freeze_params(conv1, bn1) at >= 1, layer1..layer4 at >= 2..5; :112-117: every encoder BatchNorm
        frozen = ["encoder.resnet.conv1.", "encoder.resnet.bn1.", "encoder.resnet.layer1.", "encoder.resnet.layer2.",
                  "encoder.resnet.layer3.", "encoder.resnet.layer4."][:max(0, min(self.freeze_at, 5)) + (1 if self.freeze_at >= 1 else 0)]
        for k, t in self.p.items():
            if not t.requires_grad:
                continue
            if any(k.startswith(f) for f in frozen):
                t.requires_grad_(False)
            if not self.batchnorm_trainable and k.startswith("encoder.resnet.") and (".bn" in k or ".downsample.1." in k):
                t.requires_grad_(False)

    # -- nn.Module-like helpers -------------------------------------------------------------
    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def parameters(self):
        return [t for t in self.p.values() if t.requires_grad]

    def named_parameters(self):
        return [(k, t) for k, t in self.p.items() if t.requires_grad]

    def state_dict(self):
        return OrderedDict((k, v.detach().clone()) for k, v in self.p.items())

    # -- building blocks -----------------------------------------------------------------------
    def _bn(self, x, prefix):
        p = self.p
        frozen_stats = not self.batchnorm_trainable and prefix.startswith("encoder.resnet.")   # resnet.py:186-190: eval() while training
        if self.training and not frozen_stats:
            # torch BatchNorm2d training semantics (_resnets.py:151 etc.): normalise with biased
            # batch variance, update running stats with momentum 0.1 and the unbiased variance.
            y = F.batch_norm(x, p[prefix + ".running_mean"], p[prefix + ".running_var"],
                             p[prefix + ".weight"], p[prefix + ".bias"], True, BN_MOMENTUM, BN_EPS)
            p[prefix + ".num_batches_tracked"] += 1
            return y
        return F.batch_norm(x, p[prefix + ".running_mean"], p[prefix + ".running_var"],
                            p[prefix + ".weight"], p[prefix + ".bias"], False, BN_MOMENTUM, BN_EPS)

    def _bottleneck(self, x, prefix, stride, dilation, has_ds):
        p = self.p                                                     # _resnets.py:92-112
        out = F.conv2d(x, p[prefix + ".conv1.weight"])
        out = F.relu(self._bn(out, prefix + ".bn1"))
        out = F.conv2d(out, p[prefix + ".conv2.weight"], stride=stride, padding=dilation, dilation=dilation)
        out = F.relu(self._bn(out, prefix + ".bn2"))
        out = F.conv2d(out, p[prefix + ".conv3.weight"])
        out = self._bn(out, prefix + ".bn3")
        if has_ds:
            # downsample conv keeps the block's (possibly de-strided) stride: resnet.py:198-199
            idn = F.conv2d(x, p[prefix + ".downsample.0.weight"], stride=stride)
            idn = self._bn(idn, prefix + ".downsample.1")
        else:
            idn = x
        return F.relu(out + idn)

    def encoder(self, x):
        p = self.p                                                     # resnet.py:140-166
        self.stages = []                                               # the four stage outputs (c2..c5)
        x = F.conv2d(x, p["encoder.resnet.conv1.weight"], stride=2, padding=3)
        x = F.relu(self._bn(x, "encoder.resnet.bn1"))
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        for li in range(4):
            blocks = [b for b in self.plan if b[0].startswith(f"encoder.resnet.layer{li + 1}.")]

            def run(t, blocks=blocks):
                for prefix, _inpl, _planes, stride, dilation, has_ds in blocks:
                    t = self._bottleneck(t, prefix, stride, dilation, has_ds)
                return t
            if self.with_cp[li] and x.requires_grad:                   # resnet.py:146-165
                import torch.utils.checkpoint as cp
                x = cp.checkpoint(run, x, use_reentrant=True)
            else:
                x = run(x)
            self.stages.append(x)
        return x

    def aspp_head(self, feat, head):
        p = self.p                                                     # Encoder.py:80-84
        out = None
        for i, d in enumerate(ASPP_DILATIONS):
            y = F.conv2d(feat, p[f"{head}.conv2d_list.{i}.weight"], p[f"{head}.conv2d_list.{i}.bias"],
                         padding=d, dilation=d)
            out = y if out is None else out + y
        return out

    def ppm_head(self, feat, head, dropout=False):
        p = self.p                                                     # Encoder.py:43-55
        h, w = feat.shape[-2:]
        outs = [feat]
        for i, s in enumerate(PPM_SCALES):
            y = F.adaptive_avg_pool2d(feat, s)
            y = F.conv2d(y, p[f"{head}.ppm.{i}.1.weight"])
            y = F.relu(self._bn(y, f"{head}.ppm.{i}.2"))
            outs.append(F.interpolate(y, (h, w), mode="bilinear", align_corners=False))
        y = torch.cat(outs, 1)
        y = F.conv2d(y, p[f"{head}.conv_last.0.weight"], padding=1)
        y = F.relu(self._bn(y, f"{head}.conv_last.1"))
        if dropout and self.training:
            y = F.dropout2d(y, 0.1, True)                              # Encoder.py:39 (stochastic)
        return F.conv2d(y, p[f"{head}.conv_last.4.weight"], p[f"{head}.conv_last.4.bias"])

    def __call__(self, x, dropout=False):
        feat = self.encoder(x)                                         # Encoder.py:145
        head = (lambda f, h: self.ppm_head(f, h, dropout)) if self.use_ppm else self.aspp_head

        def up(t):
            return F.interpolate(t, x.shape[-2:], mode="bilinear", align_corners=True)
        if self.multi_layer and self.cascade:                          # Encoder.py:129-143
            feat1, feat2 = self.stages[-2:]
            if self.is_ins_norm:
                feat1, feat2 = F.instance_norm(feat1, eps=1e-5), F.instance_norm(feat2, eps=1e-5)
            x1, x2 = head(feat1, "layer5"), head(feat2, "layer6")
            if self.training:
                return x1, feat1, x2, feat2
            return (up(x1).softmax(dim=1) + up(x2).softmax(dim=1)) / 2
        if self.is_ins_norm:
            feat = F.instance_norm(feat, eps=1e-5)                     # Encoder.py:123,147
        if not self.multi_layer:                                       # Encoder.py:156-165
            x1 = head(feat, "cls_pred")
            if self.training:
                return x1, feat
            return up(x1).softmax(dim=1)
        x1, x2 = head(feat, "layer5"), head(feat, "layer6")
        if self.training:
            return x1, x2, feat                                        # Encoder.py:150-151
        return (up(x1).softmax(dim=1) + up(x2).softmax(dim=1)) / 2     # Encoder.py:153-155
