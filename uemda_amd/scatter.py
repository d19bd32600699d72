"""`torch_scatter.scatter` replacement (reference call sites alignment.py:187,245: src (B, N, C) float32, index (B, N, 1) int64, dim=1):
any rank and `dim`, reduce in {'max', 'sum', 'add', 'mean'}, one index per position along `dim`."""
import torch

from . import ops
from .ops import UemError, call, ptr, stream

_REDUCE = {"max": 0, "sum": 1, "add": 1, "mean": 2}


def index_max(index):
    """Batch-global max of an int64 index tensor, as a 0-dim DEVICE tensor (no host sync)."""
    ops.need_gpu(index)
    idx = index.contiguous()
    out = torch.empty((), device=index.device, dtype=torch.int64)
    call("uem_index_max", ptr(idx), idx.numel(), ptr(out), stream())
    return out


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    """torch_scatter.scatter(src, index, dim, out, dim_size, reduce) for float tensors of any rank: the entries of `src` along `dim` are
    reduced into `dim_size` (default index.max() + 1) segments; untouched segments are 0.  `index` is broadcast to `src` as in
    torch_scatter, with one restriction that comes from the kernel's layout -- one index per position along `dim`, shared by everything
    behind it: its dimensions after `dim` must be 1 (or absent).  The call sites of the path (alignment.py:187,245) are
    src (B, N, C), index (B, N, 1), dim=1.  `out=` (reduce into a caller's tensor) is not implemented."""
    if reduce not in _REDUCE:
        raise UemError(f"scatter: reduce={reduce!r} not supported (max / sum / mean)")
    if out is not None:
        raise UemError("scatter: out= is not implemented (the result is a fresh tensor with untouched segments at 0)")
    ops.need_gpu(src, index)
    if src.dim() == 0:
        raise UemError("scatter: src needs at least one dimension")
    d = dim if dim >= 0 else dim + src.dim()
    if not 0 <= d < src.dim():
        raise UemError(f"scatter: dim={dim} out of range for a {src.dim()}-d tensor")
    lead, N, trail = tuple(src.shape[:d]), src.shape[d], tuple(src.shape[d + 1:])
    # torch_scatter's broadcast rule (utils.broadcast): a 1-d index is lined up with `dim`, anything else is padded with trailing
    # singleton dimensions; then expanded to src
    idx = index
    if idx.dim() == 1:
        for _ in range(d):
            idx = idx.unsqueeze(0)
    while idx.dim() < src.dim():
        idx = idx.unsqueeze(-1)
    if idx.dim() != src.dim() or any(s != 1 for s in idx.shape[d + 1:]):
        raise UemError("scatter: index may not vary along the dimensions after `dim` (expected shape (..., N, 1, ...) or a 1-d index)")
    try:
        idx = idx.expand(lead + (N,) + (1,) * len(trail))
    except RuntimeError:
        raise UemError(f"scatter: index of shape {tuple(index.shape)} does not broadcast to src {tuple(src.shape)} along dim {dim}")
    B, C = 1, 1
    for v in lead:
        B *= v
    for v in trail:
        C *= v
    src3 = src.contiguous().float().view(B, N, C)
    idx2 = idx.contiguous().view(B, N)
    if dim_size is None:
        dim_size = int(index_max(idx2).item()) + 1         # torch_scatter does the same host sync
    res = torch.empty((B, dim_size, C), device=src.device, dtype=torch.float32)
    ws = torch.empty((B, dim_size), device=src.device, dtype=torch.float32) if reduce == "mean" else None
    call("uem_scatter", ptr(src3), ptr(idx2), ptr(res), ptr(ws), B, N, C, dim_size, _REDUCE[reduce], stream())
    return res.view(lead + (dim_size,) + trail)
