"""`torch_scatter.scatter` replacement for the call sites on the path (reference alignment.py:187,245):
src (B, N, C) float32, index (B, N, 1) int64, dim=1, reduce in {'max', 'sum', 'mean'}."""
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


def scatter(src, index, dim=1, out=None, dim_size=None, reduce="sum"):
    if reduce not in _REDUCE:
        raise UemError(f"scatter: reduce={reduce!r} not supported (max / sum / mean)")
    if src.dim() != 3 or dim not in (1, -2) or out is not None:
        raise UemError("scatter: only the (B, N, C) / dim=1 form used by UemDA is implemented")
    ops.need_gpu(src, index)
    B, N, C = src.shape
    if index.numel() != B * N:
        raise UemError("scatter: index must have shape (B, N, 1)")
    src = src.contiguous().float()
    index = index.contiguous()
    if dim_size is None:
        dim_size = int(index_max(index).item()) + 1        # torch_scatter does the same host sync
    res = torch.empty((B, dim_size, C), device=src.device, dtype=torch.float32)
    ws = torch.empty((B, dim_size), device=src.device, dtype=torch.float32) if reduce == "mean" else None
    call("uem_scatter", ptr(src), ptr(index), ptr(res), ptr(ws), B, N, C, dim_size, _REDUCE[reduce], stream())
    return res
