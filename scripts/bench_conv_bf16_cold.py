#!/usr/bin/env python3
"""bf16 1x1 convolutions with COLD operands: every launch gets its own input / output buffers out of a rotating pool larger
than the 256 MB infinity cache, as inside a training step (the hot-loop numbers of bench_conv_bf16.py are served from the
cache).  Prints the HBM GB/s of the forward launch next to an elementwise pass (affine_act) over the same bytes: is the conv
kernel slower than a streaming kernel on the same data, and does it depend on how many 64-channel k-steps slice a row?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uemda_amd import ops, ops_bf16


def timeit_pool(fn, pool, reps=3):
    torch.cuda.synchronize()
    for i in range(len(pool)):
        fn(*pool[i])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for i in range(len(pool)):
            fn(*pool[i])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(pool))


def main():
    import ctypes
    from uemda_amd import _lib
    dbg = int(os.environ.get("CONV_DBG", "0"))          # ablations of conv_bf16_kernel (results wrong): 1 no global stores, 2 no accumulator
    if dbg:                                             # staging through LDS, 4 no DMA loads
        _lib.load().uemdbg_conv_dbg(ctypes.c_int(dbg))
        print("ablation", dbg)
    M = 32 * 128 * 128
    print(f"{'shape':18s} {'k-steps':>7s} | {'fwd ms':>8s} {'GB/s':>6s} | {'dgrad ms':>8s} {'GB/s':>6s} | {'stream ms':>9s} {'GB/s':>6s}")
    for cin, cout, m in ((64, 64, M), (64, 256, M), (128, 128, M), (256, 64, M), (256, 128, M), (512, 128, M // 4), (128, 512, M // 4), (512, 256, M // 4),
                         (1024, 256, M // 16), (256, 1024, M // 16)):
        bytes_io = m * (cin + cout) * 2
        n = max(2, int(1.5e9 // bytes_io))
        w = (torch.randn(cout, 1, 1, cin, device="cuda") * 0.05).bfloat16()
        wt = w.permute(3, 1, 2, 0).contiguous()
        pool = [(torch.randn(m // 128, 8, 16, cin, device="cuda").bfloat16(), torch.empty(m // 128, 8, 16, cout, device="cuda", dtype=torch.bfloat16))
                for _ in range(n)]
        t_f = timeit_pool(lambda x, y: ops_bf16.conv2d(x, w, out=y), pool)
        t_d = timeit_pool(lambda x, y: ops_bf16.conv2d_dgrad(y, wt, x.shape, out=x), pool)
        # a streaming kernel over the same bytes: read x (cin) -> write same-size tensor; scaled to the conv's bytes
        st = ops.BNState()
        buf = torch.ones(4, cin, device="cuda")
        st.scale, st.shift, st.mean, st.invstd = buf[0], buf[1], buf[2], buf[3]
        outs = [torch.empty_like(p[0]) for p in pool]
        pool2 = [(p[0], o) for p, o in zip(pool, outs)]

        def stream(x, o):
            ops.call("uem_affine_act_bf16", ops.ptr(x), ops.ptr(st.scale), ops.ptr(st.shift), None, None, None, ops.ptr(o), x.numel() // cin, cin, 1, None, ops.stream())
        t_s = timeit_pool(stream, pool2)
        print(f"{cin:5d} -> {cout:5d} M={m:7d} {cin // 64:3d} | {t_f:8.3f} {bytes_io / t_f / 1e6:6.0f} | {t_d:8.3f} {bytes_io / t_d / 1e6:6.0f} | "
              f"{t_s:9.3f} {m * cin * 4 / t_s / 1e6:6.0f}   (pool of {n} buffer pairs)")
        del pool, pool2, outs


if __name__ == "__main__":
    main()
