#!/usr/bin/env python3
"""Where do the blocks of a persistent launch land?  scripts/block_census.py [blocks] [lds bytes per block]: per CU, the block ids it
hosts (HW_REG_XCC_ID / HW_REG_HW_ID of every block of a grid that stays resident)."""
import ctypes
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uemda_amd import _lib

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 768
lds = int(sys.argv[2]) if len(sys.argv) > 2 else 50 * 1024
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.uemdbg_block_census.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(3 * blocks, dtype=torch.int32, device="cuda")
for rep in range(2):
    rc = lib.uemdbg_block_census(out.data_ptr(), blocks, lds, 400000, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
o = out.cpu().view(blocks, 3).numpy().astype("uint32")
cus = defaultdict(list)
for b in range(blocks):
    xcc, hw = int(o[b, 0]) & 0xF, int(o[b, 1])
    # gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 0x1, (hw >> 13) & 0x7
    cus[(xcc, se, sh, cu)].append(b)
print(f"{blocks} blocks of 256 threads, {lds} B LDS each: {len(cus)} distinct (xcc, se, sh, cu)")
hist = defaultdict(int)
for k in sorted(cus):
    hist[len(cus[k])] += 1
print("blocks per CU histogram:", dict(hist))
for k in sorted(cus)[:24]:
    print(k, cus[k])
diffs = defaultdict(int)
for v in cus.values():
    v = sorted(v)
    for a, b in zip(v[:-1], v[1:]):
        diffs[b - a] += 1
print("differences between consecutive block ids on one CU:", dict(sorted(diffs.items(), key=lambda kv: -kv[1])[:8]))
t = o[:, 2].astype("int64")
print("start-time spread (clocks of s_memtime):", int(t.max() - t.min()))
