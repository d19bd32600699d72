"""CoralLoss on the MI355X (reference uemda/gast/coral.py:15-47; used by `Aligner.align_domain`,
alignment.py:79-84).  The two d x d Gram matrices are weight-gradient GEMMs and the two backward products are
1x1 convolutions with the centring as operand prologue -- all on the f32-MFMA conv kernels."""
import torch
import torch.nn as nn
from torch.autograd import Function

from .. import ops
from ..ops import UemError, call, ptr, stream


def _gram_and_mean(x2d):
    n, d = x2d.shape
    x4 = x2d.view(1, 1, n, d)
    gram = torch.zeros((d, 1, 1, d), device=x2d.device, dtype=torch.float32)
    ops.conv2d_wgrad(x4, x4, gram)                                   # sum_m x[m][o] x[m][i]
    ones, zeros = torch.ones(d, device=x2d.device), torch.zeros(d, device=x2d.device)
    st = ops.bn_stats(x4, ones, zeros, None, None, True)
    return gram.view(d, d), st.mean


class _CoralFn(Function):
    @staticmethod
    def forward(ctx, source, target):
        ns, d = source.shape
        nt = target.shape[0]
        gs, mus = _gram_and_mean(source)
        gt, mut = _gram_and_mean(target)
        Gs, Gt = torch.empty_like(gs), torch.empty_like(gt)
        loss = torch.empty(1, device=source.device, dtype=torch.float32)
        partial = torch.empty(1024, device=source.device, dtype=torch.float32)
        call("uem_coral_finish", ptr(gs), ptr(gt), ptr(mus), ptr(mut), ns, nt, d, ptr(Gs), ptr(Gt), ptr(loss), ptr(partial), stream())
        ctx.save_for_backward(source, target, Gs, Gt, mus.clone(), mut.clone())
        return loss[0]

    @staticmethod
    def backward(ctx, go):
        source, target, Gs, Gt, mus, mut = ctx.saved_tensors
        d = source.shape[1]
        one = torch.ones(d, device=source.device, dtype=torch.float32)
        grads = []
        for x, G, mu in ((source, Gs, mus), (target, Gt, mut)):
            neg = torch.empty_like(mu)
            call("uem_negate", ptr(mu), ptr(neg), d, stream())
            n = x.shape[0]
            # d x = (x - mu) . G : 1x1 conv, weight[o][i] = G[o][i] (symmetric), prologue x' = x*1 + (-mu)
            dx = ops.conv2d(x.view(1, 1, n, d), G.view(d, 1, 1, d), in_scale=one, in_shift=neg, in_relu=False).view(n, d)
            grads.append(dx)
        go = go.detach().reshape(1).float().contiguous()
        call("uem_scale_by_scalar", ptr(grads[0]), None, grads[0].numel(), ptr(go), stream())
        call("uem_scale_by_scalar", ptr(grads[1]), None, grads[1].numel(), ptr(go), stream())
        return grads[0], grads[1]


class CoralLoss(nn.Module):
    def __init__(self, is_sqrt=False):
        super().__init__()
        if is_sqrt:
            raise UemError("CoralLoss(is_sqrt=True) is never used by the UemDA scripts; not implemented")

    def forward(self, source, target):
        ops.need_gpu(source, target)
        if source.dim() != 2 or target.dim() != 2 or source.shape[1] != target.shape[1]:
            raise UemError("CoralLoss expects (n, d) feature matrices")
        if source.shape[1] % 128 != 0:
            raise UemError("CoralLoss: feature dimension must be a multiple of 128")
        return _CoralFn.apply(source.contiguous(), target.contiguous())
