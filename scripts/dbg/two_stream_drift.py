"""diagnostic: three ssl steps, sequential twice and forked once: per-step loss / weight differences (is the forked drift atomics' noise?)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from uemda_amd import ops
sys.path.insert(0, "tests")
from test_gpu_two_streams import _model, C
from oracle import synth
from uemda_amd.gast.alignment import Aligner
from uemda_amd.optim import FusedSGD
from uemda_amd.step import HYPER, StepState, ssl_step

storage = sys.argv[1] if len(sys.argv) > 1 else "fp32"
use_ppm = len(sys.argv) > 2 and sys.argv[2] == "ppm"
batch = {k: v.cuda() for k, v in synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=9).items()}
res = []
for name, fwd, bwd in (("seq", 0, 0), ("seq2", 0, 0), ("seq3", 0, 0), ("two", 1, 1), ("two2", 1, 1), ("two3", 1, 1), ("fwdonly", 1, 0), ("pipe", 1, 2), ("pipe2", 1, 2), ("pipe3", 1, 2), ("pipe4", 1, 2)):
    ops.TWO_STREAM_FWD, ops.TWO_STREAM_BWD, ops.TWO_PIPELINES = bool(fwd), bool(bwd), bwd == 2      # pipe*: the two-pipeline form of ssl_step
    model = _model(storage, use_ppm)
    if use_ppm:
        pass
    al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
    al.prototypes = batch["prototypes"].clone()
    opt, state = FusedSGD(model, lr=1e-2, momentum=0.9, weight_decay=5e-4), StepState(C)
    ws, ls = [], []
    for i in range(4):
        o = ssl_step(model, al, opt, state, batch, 2e-3)
        torch.cuda.synchronize()
        ws.append(model.flat_parameters()[0].clone()); ls.append((float(o["loss_source"]), float(o["loss_target"]), float(o["grad_norm"])))
    res.append((name, ws, ls))
base = res[0]
for name, ws, ls in res[1:]:
    print(name, " | ".join(f"step{i+1}: dLs={abs(ls[i][0]-base[2][i][0]):.2e} dLt={abs(ls[i][1]-base[2][i][1]):.2e} dgn={abs(ls[i][2]-base[2][i][2])/base[2][i][2]:.1e} dW={float((ws[i]-base[1][i]).norm()/base[1][i].norm()):.2e}" for i in range(4)))
print("losses seq:", base[2])
