#!/usr/bin/env python3
"""The 7x7 stem at the benchmark size (32 x 3 x 512 x 512): the stem's own kernels (csrc/stem.hip) against the generic register-staged
kernels of rounds 1-4 (UEM_STEM_KERNEL=0's path), forward with tile statistics and weight gradient, fp32 and bf16 dz."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uemda_amd import ops, ops_bf16 as ob
from bench_conv_shapes import timeit


def main():
    B, S = int(os.environ.get("B", "32")), int(os.environ.get("S", "512"))
    x = torch.randn(B, 3, S, S, device="cuda")
    w = (torch.randn(64, 3, 7, 7, device="cuda") / 12).contiguous(memory_format=torch.channels_last)
    wo = ops.weight_ohwi(w)
    w8 = torch.empty((64, 7, 8, 4), device="cuda")
    ops.call("uem_stem_pack_weight", ops.ptr(wo), ops.ptr(w8), ops.stream())
    x4 = ops.nchw3_to_nhwc4(x)
    bn = torch.nn.BatchNorm2d(64).cuda().train()
    z, _ = ops.stem_conv_bn(x4, wo, bn, w8=w8)
    dz = torch.randn_like(z)
    dzb = dz.to(torch.bfloat16)
    dw = torch.zeros(64, 7, 7, 3, device="cuda")
    gf = 2.0 * z.numel() * 147 / 1e9
    for on in (False, True):
        ops.STEM_KERNEL = on
        tf = timeit(lambda: ops.stem_conv_bn(x4, wo, bn, w8=w8), 5)
        tw = timeit(lambda: ops.stem_wgrad(x4, dz, dw), 5)
        twb = timeit(lambda: ob.stem_wgrad(x4, dzb, dw), 5)
        print(f"{'stem.hip' if on else 'generic '}: forward + statistics {tf * 1e3:7.1f} us ({gf / tf:6.1f} TFLOP/s algorithmic)   "
              f"weight gradient {tw * 1e3:7.1f} us ({gf / tw:6.1f})   bf16 dz {twb * 1e3:7.1f} us")


if __name__ == "__main__":
    main()
