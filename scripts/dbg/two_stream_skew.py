"""diagnostic: where do the two chains of a forked ssl_step stand at the joins?  bench-size step (B=32, 512x512); events on both streams
before the join after the forwards and before the fold after the backward passes.  usage: two_stream_skew.py [fp32|bf16]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from uemda_amd import ops, step as stepmod

storage = sys.argv[1] if len(sys.argv) > 1 else "fp32"
cfg = types.SimpleNamespace(gpus=1, steps=8, warmup=3, batch=32, size=512, workload="ssl", model="resnet50", head="aspp", backend="nccl", device=None,
                            no_overlap=False, dump_params="", data_rank=None, no_cpu_baseline=True, no_kernel_events=True, dump_conv_events=None,
                            storage=storage, unique_batches=2, no_hipgraph=True, no_other_configs=True)
s = bench.Setup(cfg, 0, 1, None)
ev = lambda: torch.cuda.Event(enable_timing=True)
rec = []
orig_pair = stepmod.forward_pair


def pair(model, a, b, join=True):
    out = orig_pair(model, a, b, join=False)
    if out[2] is not None:
        e1, e2 = ev(), ev()
        e1.record(torch.cuda.current_stream()); e2.record(ops.second_stream())
        rec[-1]["fwd"] = (e1, e2)
        out[2]()
    return out[0], out[1]


stepmod.forward_pair = pair
model = s.model
orig_fold = type(model).fold_shadow_grads


def fold(lo=0, synced=False):
    if lo == 0 and model._g2_dirty:
        e1, e2 = ev(), ev()
        e1.record(torch.cuda.current_stream()); e2.record(ops.second_stream())
        rec[-1]["bwd"] = (e1, e2)
    return orig_fold(model, lo, synced)


model.fold_shadow_grads = fold
for i in range(11):
    e0 = ev(); e0.record()
    rec.append({"e0": e0})
    s.one_step(i)
    e9 = ev(); e9.record()
    rec[-1]["e9"] = e9
torch.cuda.synchronize()
print(f"{storage}: ms from the step's start: forward end main / second | backward end main / second | step end")
for r in rec[4:]:
    if "fwd" in r and "bwd" in r:
        t = lambda e: r["e0"].elapsed_time(e)
        print(f"  fwd {t(r['fwd'][0]):7.2f} {t(r['fwd'][1]):7.2f} | bwd {t(r['bwd'][0]):7.2f} {t(r['bwd'][1]):7.2f} | {t(r['e9']):7.2f}")
