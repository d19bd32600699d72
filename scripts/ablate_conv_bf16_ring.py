#!/usr/bin/env python3
"""Where does a tile of the three-stage-ring bf16 conv kernel spend its time?  DEBUG_HOOKS build only (wrong results on purpose):
    UEM_LIB_PATH=uemda_amd/libuemda_hip_dbg.so python scripts/ablate_conv_bf16_ring.py
dbg bits: 1 no global stores, 2 no LDS staging of the accumulators, 4 operand DMA dead (out-of-range pieces), 8 no epilogue at all,
16 the counted wait that spares the previous tile's stores on the FIRST k-step only (second k-step waits for their acknowledgement)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uemda_amd import _lib, ops_bf16
from sweep_conv_bf16_ring import timeit

lib = _lib.load()
for f in ("uemdbg_conv_bf16_ring", "uemdbg_conv_dbg"):
    getattr(lib, f).argtypes = [ctypes.c_int]; getattr(lib, f).restype = None
B = 32
SH = [("l3 256->1024 @64 (R101 1024^2)", 256, 1024, 1, 64), ("l3 1024->256 @64", 1024, 256, 1, 64), ("l1 64->256 @128", 64, 256, 1, 128),
      ("l1 256->64 @128", 256, 64, 1, 128), ("l3 3x3 256 @64", 256, 256, 3, 64), ("l4 3x3 512 @32", 512, 512, 3, 32), ("l3 256->1024 @32", 256, 1024, 1, 32)]
CASES = [("round-5 dispatch", 0, 0), ("ring", 1, 0), ("ring, kt==0 rule", 1, 16), ("ring no stores", 1, 1), ("ring no staging", 1, 2), ("ring no stores/staging", 1, 3),
         ("ring no epilogue", 1, 8), ("ring no DMA", 1, 4), ("ring no DMA no epilogue", 1, 12),
         ("ring odd blocks +3.4us", 1, 256), ("ring odd blocks +6.8us", 1, 512)]
print(f"{'shape':32s} {'stats':5s} | " + " | ".join(f"{c[0]:>22s}" for c in CASES))
for name, cin, cout, k, h in SH:
    x = torch.randn(B, h, h, cin, device="cuda").bfloat16()
    w = (torch.randn(cout, k, k, cin, device="cuda") * 0.05).bfloat16()
    M = B * h * h
    fl = 2.0 * M * cout * k * k * cin
    by = 2.0 * M * (cin + cout)
    for stats in (False, True):
        row = []
        for label, ring, dbg in CASES:
            lib.uemdbg_conv_bf16_ring(ring); lib.uemdbg_conv_dbg(dbg)
            t = timeit(lambda: ops_bf16.conv2d(x, w, pad=(k - 1) // 2, want_stats=stats))
            row.append(f"{t*1e3:7.1f}us {fl/t/1e9:5.0f}TF {by/t/1e9:4.1f}TB" if False else f"{t*1e3:8.1f} us {by/t/1e6:6.0f} GB/s")
        lib.uemdbg_conv_bf16_ring(-1); lib.uemdbg_conv_dbg(0)
        print(f"{name:32s} {str(stats):5s} | " + " | ".join(f"{r:>22s}" for r in row), flush=True)
