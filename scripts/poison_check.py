#!/usr/bin/env python3
"""Uninitialised-read detector: fill the caching allocator's free blocks with NaN bit patterns, run SSL steps and
check every output, gradient and parameter for NaN/Inf.  A kernel that reads memory it (or a predecessor) never
wrote shows up deterministically here instead of once in a while."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uemda_amd.utils import synth
from uemda_amd.gast.alignment import Aligner
from uemda_amd.models.Encoder import Deeplabv2
from uemda_amd.optim import FusedSGD
from uemda_amd.step import HYPER, StepState, ssl_step


PATTERN = float("nan")


def poison(gib):
    blocks = []
    for size_mb in (4096, 2048, 512, 128, 32, 8, 2):
        n = max(1, int(gib * 1024 / 7 / size_mb))
        for _ in range(n):
            blocks.append(torch.full((size_mb * 1024 * 256,), PATTERN, device="cuda"))
    small = [torch.full((k,), PATTERN, device="cuda") for k in (16, 64, 256, 1024, 4096, 65536) for _ in range(64)]
    del blocks, small                       # back to the caching allocator, contents intact


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--poison-gib", type=float, default=60.0)
    ap.add_argument("--head", default="aspp")
    ap.add_argument("--pattern", type=float, default=float("nan"),
                    help="fill value: NaN is swallowed by fmaxf-style ReLUs, a huge finite value (1e30) is not")
    args = ap.parse_args()
    global PATTERN
    PATTERN = args.pattern
    C, B, S = 6, args.batch, args.size
    torch.manual_seed(0)
    cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False), multi_layer=True, cascade=False,
               use_ppm=(args.head == "ppm"), ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C,
               is_ins_norm=True)
    model = Deeplabv2(cfg).cuda()
    pool = synth.make_batch(B=min(B, 4), H=S, W=S, C=C, k=2048, seed=2333)
    rep = (B + 3) // 4
    batch = {k: (v.cuda().repeat((rep,) + (1,) * (v.dim() - 1))[:B].contiguous() if k != "prototypes" else v.cuda())
             for k, v in pool.items()}
    al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
    al.prototypes = batch["prototypes"].clone()
    opt = FusedSGD(model, 1e-2, 0.9, 5e-4)
    st = StepState(C)
    bad = 0
    for i in range(args.steps):
        poison(args.poison_gib)
        out = ssl_step(model, al, opt, st, batch, 1e-3, sup_ignore_id=(S // 16) ** 2)
        torch.cuda.synchronize()
        arena, garena, n = model.flat_parameters()
        report = {k: bool(torch.isfinite(v.float()).all()) for k, v in out.items() if torch.is_tensor(v)}
        report["params"] = bool(torch.isfinite(arena[:n]).all())
        report["grads"] = bool(torch.isfinite(garena[:n]).all())
        if not report["grads"]:
            for name, p in model.named_parameters():
                if p.grad is not None and not torch.isfinite(p.grad).all():
                    print("   non-finite grad:", name, int((~torch.isfinite(p.grad)).sum()), "of", p.grad.numel())
        ok = all(report.values())
        bad += 0 if ok else 1
        print(f"step {i}: {'ok' if ok else 'NON-FINITE ' + str([k for k, v in report.items() if not v])}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
