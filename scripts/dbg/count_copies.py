"""diagnostic: which host-side calls issue the ~118 device copies per step (rocprof: __amd_rocclr_copyBuffer)?  torch profiler, one step,
Memcpy / Memset events grouped by the innermost frames of this package"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
sys.path.insert(0, "tests")
from test_gpu_two_streams import _model, C
from oracle import synth
from uemda_amd import ops
from uemda_amd.gast.alignment import Aligner
from uemda_amd.optim import FusedSGD
from uemda_amd.step import HYPER, StepState, ssl_step

storage = sys.argv[1] if len(sys.argv) > 1 else "bf16"
ops.TWO_STREAM_FWD = False
batch = {k: v.cuda() for k, v in synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=9).items()}
model = _model(storage, False)
al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
al.prototypes = batch["prototypes"].clone()
opt, state = FusedSGD(model, lr=1e-2, momentum=0.9, weight_decay=5e-4), StepState(C)
for _ in range(3):
    ssl_step(model, al, opt, state, batch, 2e-3)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    ssl_step(model, al, opt, state, batch, 2e-3)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    n = e.name
    if n.startswith("aten::copy_") or n.startswith("aten::fill_") or n.startswith("aten::zero_") or n in ("aten::clone", "aten::contiguous", "aten::_to_copy", "aten::add_", "aten::add", "aten::mul_", "aten::cat"):
        st = [s for s in (e.stack or []) if "uemda_amd" in s or "bench" in s][:2]
        cnt[(n, tuple(s.split("/")[-1] for s in st))] += 1
for (n, st), c in cnt.most_common(40):
    print(f"{c:5d}  {n:18s} {' <- '.join(st)}")
names = collections.Counter(e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
print([(k[:50], v) for k, v in names.most_common(12)])
