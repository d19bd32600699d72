#!/usr/bin/env python3
"""Where do the torch-side launches of a step come from?  aten::copy_ / fill_ / zero_ / cat / clone / add_ ... by Python call site."""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import bench


class Counter(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("copy_", "fill_", "zero_", "cat", "clone", "add", "mul", "clamp", "zeros", "ones", "full", "contiguous", "_to_copy", "div", "sub")):
            st = traceback.extract_stack()
            frames = [f for f in st if "/uemda_amd/" in f.filename or f.filename.endswith("bench.py")]
            site = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in frames[-3:])
            self.sites[(name, site)] += 1
        return func(*args, **(kwargs or {}))


def main():
    import types
    args = types.SimpleNamespace(batch=int(os.environ.get("B", "8")), size=512, workload="ssl", model="resnet50", head=os.environ.get("HEAD", "aspp"),
                                 conv_prec="fp32", storage=os.environ.get("STORAGE", "fp32"), data_rank=None)
    s = bench.Setup(args, 0, 1, None)
    for i in range(2):
        s.one_step(i)
    torch.cuda.synchronize()
    with Counter() as c:
        s.one_step(2)
    torch.cuda.synchronize()
    tot = 0
    for (name, site), n in c.sites.most_common(60):
        print(f"{n:4d}  {name:36s} {site}")
        tot += n
    print("total", sum(c.sites.values()))


if __name__ == "__main__":
    main()
