"""PrototypeContrastiveLoss on the MI355X (reference uemda/loss.py:10-47): drop-in signature
`loss_fn(Proto, feat, labels)`; forward and the gradient w.r.t. `feat` come from one fused HIP kernel."""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib, ops
from .ops import UemError, call, ptr, stream


class _PCLFn(Function):
    @staticmethod
    def forward(ctx, feat2d, proto, labels, temperature, ignore_label):
        n, k = feat2d.shape
        C = proto.shape[0]
        loss = torch.empty(1, device=feat2d.device, dtype=torch.float32)
        dfeat = torch.empty_like(feat2d)
        ws = torch.empty(_lib.load().uem_pcl_workspace_floats(k, C), device=feat2d.device, dtype=torch.float32)
        call("uem_pcl_loss", ptr(proto), ptr(feat2d), ptr(labels), ptr(loss), ptr(dfeat), ptr(ws), n, k, C,
             float(temperature), int(ignore_label), stream())
        ctx.save_for_backward(dfeat)
        return loss[0]

    @staticmethod
    def backward(ctx, go):
        (dfeat,) = ctx.saved_tensors
        go = go.detach().reshape(1).float().contiguous()
        call("uem_scale_by_scalar", ptr(dfeat), None, dfeat.numel(), ptr(go), stream())
        return dfeat, None, None, None, None


class PrototypeContrastiveLoss(nn.Module):
    def __init__(self, temperature=8.0, ignore_label=-1):
        super().__init__()
        self.temperature, self.ignore_label = temperature, ignore_label

    def forward(self, Proto, feat, labels):
        """Proto (C, A) no grad; feat (B, A, h, w) or (N, A) with grad; labels (B, 1, h, w) / (N,) int64."""
        assert not Proto.requires_grad and not labels.requires_grad and feat.requires_grad
        ops.need_gpu(Proto, feat, labels)
        if feat.dim() != 2:
            k = feat.size(1)
            feat = ops_as_rows(feat, k)
        labels = labels.reshape(-1).contiguous().long()
        if feat.shape[0] != labels.shape[0]:
            raise UemError("PrototypeContrastiveLoss: feat / labels row mismatch")
        return _PCLFn.apply(feat, Proto.detach().contiguous().float(), labels, self.temperature, self.ignore_label)


def ops_as_rows(feat, k):
    """(B, k, h, w) logical NCHW -> (B*h*w, k) rows; zero-copy for the channels_last tensors the model returns."""
    v = feat.permute(0, 2, 3, 1)
    if not v.is_contiguous():
        v = v.contiguous()
    return v.reshape(-1, k)
