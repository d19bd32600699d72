"""Losses of the self-training step on the MI355X (reference uemda/gast/balance.py):
ClassBalance (:15-78), CrossEntropy (:81-101), UVEMLoss (:345-434), loss_calc_uvem (:437-457).

Both losses take LOW-RESOLUTION logits (b, c, h, w) with full-resolution targets: the bilinear
(align_corners=True) upsample that `loss_calc` / `loss_calc_uvem` perform in the reference
(tools.py:249-250, balance.py:446-447) is fused into the loss kernel, forward and backward, so the
(b, c, H, W) logits never exist.  Full-resolution logits are accepted too (h == H).
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from .. import _lib, ops
from ..ops import UemError, call, ptr, stream


def _loss_ws(B, C, h, w, dev):
    return torch.empty(_lib.load().uem_loss_workspace_floats(B, C, h, w), device=dev, dtype=torch.float32)


class _FusedLossFn(Function):
    """mode 'ce' | 'uvem'; one or two heads; returns the scalar loss, saves d(loss)/d(logits)."""

    @staticmethod
    def forward(ctx, p1, p2, label, soft, pixel_weight, mode, hyper, ignore_label):
        l1 = ops.as_nhwc(p1.detach()).contiguous()
        l2 = ops.as_nhwc(p2.detach()).contiguous() if p2 is not None else None
        B, h, w, C = l1.shape
        label = label.contiguous()
        if label.dim() == 4:
            label = label.squeeze(1)
        H, W = label.shape[-2:]
        dev = l1.device
        loss = torch.empty((1,), device=dev, dtype=torch.float32)
        d1 = torch.empty_like(l1)
        d2 = torch.empty_like(l2) if l2 is not None else None
        ws = _loss_ws(B, C, h, w, dev)
        pw = pixel_weight.contiguous() if pixel_weight is not None else None
        if mode == "ce":
            call("uem_ce_upsampled", ptr(l1), ptr(l2), ptr(label), ptr(pw), ptr(loss), ptr(d1), ptr(d2), ptr(ws),
                 B, C, h, w, H, W, int(ignore_label), 1.0, stream())
        else:
            m, t, g = hyper
            soft = soft.detach().contiguous()
            call("uem_uvem_upsampled", ptr(l1), ptr(l2), ptr(label), ptr(soft), ptr(pw), ptr(loss), ptr(d1), ptr(d2),
                 ptr(ws), B, C, h, w, H, W, float(m), float(t), float(g), int(ignore_label), 1.0, stream())
        ctx.save_for_backward(d1, d2 if d2 is not None else d1.new_empty(0))
        ctx.two = d2 is not None
        return loss[0]

    @staticmethod
    def backward(ctx, go):
        d1, d2 = ctx.saved_tensors
        go = go.detach().reshape(1).float().contiguous()
        call("uem_scale_by_scalar", ptr(d1), ptr(d2) if ctx.two else None, d1.numel(), ptr(go), stream())
        g1 = d1.permute(0, 3, 1, 2)
        g2 = d2.permute(0, 3, 1, 2) if ctx.two else None
        return g1, g2, None, None, None, None, None, None


class ClassBalance(nn.Module):
    """EMA of class frequency -> per-pixel loss weight (balance.py:15-78).  The per-class vector math
    (class_num values) is host-side bookkeeping; the per-pixel passes are HIP kernels."""

    def __init__(self, class_num=7, ignore_label=-1, decay=0.99, temperature=0.5, device="cuda", process_group=None):
        super().__init__()
        assert temperature > 0
        self.class_num, self.ignore_label, self.decay = class_num, ignore_label, decay
        self.temperature, self.eps = temperature, 1e-7
        self.freq = torch.ones([class_num], device=device).float() / class_num
        self.process_group = process_group          # data parallel (collective 3): None = the default group

    def _class_counts(self, label):
        """(class_num + 1,) pixel counts of this rank's labels: classes, then ignored pixels (HIP kernel)."""
        lab = label.contiguous().view(-1)
        counts = torch.zeros(self.class_num + 1, device=lab.device, dtype=torch.float32)
        call("uem_class_count", ptr(lab), lab.numel(), self.class_num, int(self.ignore_label), ptr(counts), stream())
        return counts

    def _freq_from_counts(self, counts):
        """class frequency over the GLOBAL batch (balance.py:45-53: class count / valid-pixel count).  Under data parallel the
        counts are all-reduced first -- class counts and, with them, the valid-pixel count -- so every replica's `freq` EMA and
        per-pixel weights stay the single-process ones; rank-local frequencies would let the replicas drift apart."""
        if torch.distributed.is_available() and torch.distributed.is_initialized() and \
                torch.distributed.get_world_size(self.process_group) > 1:
            counts = counts.clone()
            torch.distributed.all_reduce(counts, group=self.process_group)
        cls = counts[: self.class_num]
        return cls / (cls.sum() + self.eps)

    def _local_freq(self, label):
        return self._freq_from_counts(self._class_counts(label))

    def ema_update(self, label):
        # in place (balance.py:55-61 rebinds; same values): a step replayed from a hipGraph reads self.freq at its capture-time address
        self.freq.mul_(self.decay).add_(self._local_freq(label), alpha=1.0 - self.decay)

    def _get_class_wight(self):
        prob = torch.softmax((1.0 - self.freq) / self.temperature, dim=0)
        return prob / (prob.max() + self.eps)

    def get_class_weight_4pixel(self, label):
        self.ema_update(label)
        lab = label.contiguous().view(-1)
        out = torch.empty(lab.numel(), device=lab.device, dtype=torch.float32)
        cw = self._get_class_wight().contiguous()
        call("uem_class_weight_gather", ptr(lab), ptr(cw), ptr(out), lab.numel(), self.class_num,
             int(self.ignore_label), stream())
        return out


class CrossEntropy(nn.Module):
    def __init__(self, ignore_label=-1, class_balancer=None):
        super().__init__()
        self.ignore_label, self.class_balancer = ignore_label, class_balancer

    def forward_multi(self, preds, labels):
        """mean over heads of CrossEntropy.forward; each head's loss is the mean over ALL pixels, ignored
        ones included in the denominator (balance.py:97-101)."""
        pw = self.class_balancer.get_class_weight_4pixel(labels) if self.class_balancer is not None else None
        p2 = preds[1] if len(preds) > 1 else None
        return _FusedLossFn.apply(preds[0], p2, labels.long(), None, pw, "ce", None, self.ignore_label)

    def forward(self, preds, labels):
        return self.forward_multi([preds], labels)


class UVEMLoss(nn.Module):
    def __init__(self, m=0.1, threshold=0.7, gamma=8.0, class_balancer=None, class_num=7, ignore_label=-1):
        super().__init__()
        self.m, self.threshold, self.gamma = m, threshold, gamma
        self.class_balancer, self.class_num, self.ignore_label = class_balancer, class_num, ignore_label

    def forward_multi(self, preds, targets, label_t_soft):
        pw = self.class_balancer.get_class_weight_4pixel(targets) if self.class_balancer is not None else None
        p2 = preds[1] if len(preds) > 1 else None
        return _FusedLossFn.apply(preds[0], p2, targets.long(), label_t_soft, pw, "uvem",
                                  (self.m, self.threshold, self.gamma), self.ignore_label)

    def forward(self, preds, targets, label_t_soft):
        return self.forward_multi([preds], targets, label_t_soft)

    def get_weight(self, uncertainties):
        u = uncertainties.contiguous().float()
        ops.need_gpu(u)
        out = torch.empty_like(u)
        call("uem_uvem_weight", ptr(u), ptr(out), u.numel(), float(self.m), float(self.threshold), float(self.gamma), stream())
        return out


def loss_calc_uvem(pred, label, label_soft, loss_fn, multi=True):
    """balance.py:437-457.  With multi=True both heads go through ONE fused kernel pass."""
    if not isinstance(loss_fn, UVEMLoss):
        raise UemError("loss_calc_uvem: loss_fn must be uemda_amd.gast.balance.UVEMLoss")
    if multi is True:
        if len(pred) > 2:
            raise UemError("loss_calc_uvem: at most two heads")
        return loss_fn.forward_multi(list(pred), label.long(), label_soft)
    return loss_fn(pred, label.long(), label_soft)
