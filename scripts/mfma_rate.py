#!/usr/bin/env python3
"""Ceiling check: sustained v_mfma_f32_32x32x2_f32 rate of this MI355X under the conv kernel's instruction mix."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uemda_amd import _lib
lib = _lib.load()
lib.uemdbg_mfma_rate.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.empty(4096 * 256, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for blocks in (512, 768, 1024, 2048):
    for mode in (0, 1, 2):
        iters = 400
        lib.uemdbg_mfma_rate(out.data_ptr(), blocks, 10, mode, st)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        lib.uemdbg_mfma_rate(out.data_ptr(), blocks, iters, mode, st)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e)
        flops = blocks * 4 * iters * 64 * 4096.0      # waves * iters * mfma/iter * flop/mfma
        print(f"blocks={blocks:5d} mode={mode}  {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s")
