#!/usr/bin/env python3
"""Where the HOST time of one SSL step goes (Python + ctypes + launch calls): enqueue time without any device sync and
the top functions by cumulative time (cProfile) -- the step becomes launch-bound when this approaches the device time."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uemda_amd.utils import synth
from uemda_amd.gast.alignment import Aligner
from uemda_amd.models.Encoder import Deeplabv2
from uemda_amd.optim import FusedSGD
from uemda_amd.step import HYPER, StepState, ssl_step


def main():
    C, B, S = 6, int(os.environ.get("B", 32)), 512
    cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False), multi_layer=True, cascade=False,
               use_ppm=False, ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)
    model = Deeplabv2(cfg).cuda()
    pool = synth.make_batch(B=4, H=S, W=S, C=C, k=2048, seed=1)
    batch = {k: (v.cuda().repeat((B // 4,) + (1,) * (v.dim() - 1)).contiguous() if k != "prototypes" else v.cuda()) for k, v in pool.items()}
    al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
    al.prototypes = batch["prototypes"].clone()
    al.check_superpixel_ids = lambda: None          # the step's only host wait: excluded, this measures pure enqueue
    opt = FusedSGD(model, 1e-2, 0.9, 5e-4)
    st = StepState(C)
    for _ in range(3):
        ssl_step(model, al, opt, st, batch, 1e-3, sup_ignore_id=1024)
    torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(4):
        t0 = time.perf_counter()
        ssl_step(model, al, opt, st, batch, 1e-3, sup_ignore_id=1024)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append(1e3 * (t1 - t0)), tot.append(1e3 * (t2 - t0))
    print(f"host enqueue {min(enq):.1f} ms/step (all: {[round(e, 1) for e in enq]}), device-complete {min(tot):.1f} ms/step")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        ssl_step(model, al, opt, st, batch, 1e-3, sup_ignore_id=1024)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
