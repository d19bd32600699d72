#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound BatchNorm / elementwise kernels at benchmark-sized activations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uemda_amd import ops

def timeit(fn, reps=int(os.environ.get("REPS", "10"))):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

for (n, h, c) in ((32, 128, 256), (32, 128, 64), (32, 64, 512), (32, 32, 2048), (32, 32, 512)):
    z = torch.randn(n, h, h, c, device="cuda"); dy = torch.randn_like(z); out = torch.empty_like(z); res = torch.randn_like(z)
    bn = torch.nn.BatchNorm2d(c).cuda()
    st = ops.bn_stats(z, bn.weight.detach(), bn.bias.detach(), None, None, True)
    gg, gb = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
    nb = z.numel() * 4
    t_stats = timeit(lambda: ops.bn_stats(z, bn.weight.detach(), bn.bias.detach(), None, None, True))
    t_aff = timeit(lambda: ops.affine_act(z, st, res=res, relu=True, out=out))
    M = z.numel() // c
    ws = torch.empty(ops._lib.load().uem_bn_workspace_floats(M, c), device="cuda")
    tmp = torch.empty(2, c, device="cuda")
    t_red = timeit(lambda: ops.call("uem_bn_bwd_reduce", ops.ptr(z), ops.ptr(dy), None, ops.ptr(st.scale), ops.ptr(st.shift), ops.ptr(st.mean), ops.ptr(st.invstd), M, c, 1, ops.ptr(tmp[0]), ops.ptr(tmp[1]), None, None, ops.ptr(ws), ops.stream()))
    t_app = timeit(lambda: ops.call("uem_bn_bwd_apply", ops.ptr(z), ops.ptr(dy), None, ops.ptr(st.scale), ops.ptr(st.shift), ops.ptr(st.mean), ops.ptr(st.invstd), ops.ptr(tmp[0]), ops.ptr(tmp[1]), M, c, 1, ops.ptr(out), None, ops.stream()))
    print(f"({n},{h},{h},{c}) {nb/1e6:7.1f} MB/tensor | stats {t_stats*1e3:7.1f} us {nb/t_stats/1e6:6.0f} GB/s | affine+res {t_aff*1e3:7.1f} us {3*nb/t_aff/1e6:6.0f} GB/s | "
          f"bwd_reduce {t_red*1e3:7.1f} us {2*nb/t_red/1e6:6.0f} GB/s | bwd_apply {t_app*1e3:7.1f} us {3*nb/t_app/1e6:6.0f} GB/s")
