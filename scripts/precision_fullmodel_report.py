#!/usr/bin/env python3
"""One train_ssl_uem step of ResNet50-ASPP (B=2, 256x256, the golden-fixture configuration) in each matrix-core
precision, compared with the reference's CPU outputs in tests/golden/model_aspp_r50_b2_256.npz: the worst relative
logit error, pseudo-label agreement, loss / gradient-norm errors."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from uemda_amd.utils import synth
from oracle.weights import det_state_dict
from uemda_amd import ops
from uemda_amd.gast.alignment import Aligner
from uemda_amd.models.Encoder import Deeplabv2
from uemda_amd.optim import FusedSGD
from uemda_amd.step import HYPER, StepState, ssl_step

C = 6


def main():
    g = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(ROOT, "tests", "golden", "model_aspp_r50_b2_256.npz")).items()}
    for prec in ("fp32", "mixed", "bf16x3", "bf16"):
        ops.set_conv_precision(prec)
        cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False), multi_layer=True, cascade=False,
                   use_ppm=False, ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C,
                   is_ins_norm=True)
        m = Deeplabv2(cfg)
        m.load_state_dict(det_state_dict("resnet50", C, False, seed=2333))
        m = m.cuda()
        batch = {k: v.cuda() for k, v in synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=2333).items()}
        al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
        al.prototypes = batch["prototypes"].clone()
        opt = FusedSGD(m, lr=1e-2, momentum=0.9, weight_decay=5e-4)
        out = ssl_step(m, al, opt, StepState(C), batch, float(g["lr"]))
        logit = max(float((out[k].cpu() - g[k]).abs().max() / g[k].abs().max()) for k in ("pred_s1", "pred_s2", "pred_t1", "pred_t2"))
        agree = (out["label_t_hard"].cpu() == g["hard"].long()).float().mean().item()
        ls = abs(float(out["loss_source"]) - float(g["loss_source"])) / float(g["loss_source"])
        lt = abs(float(out["loss_target"]) - float(g["loss_target"])) / float(g["loss_target"])
        gn = abs(float(out["grad_norm"]) - float(g["grad_norm"])) / float(g["grad_norm"])
        print(f"{prec:7s} worst logit err/max {logit:.2e}  pseudo-label agreement {agree:.6f}  loss_s {ls:.1e}  loss_t {lt:.1e}  "
              f"grad_norm {gn:.1e}")
    ops.set_conv_precision("fp32")


if __name__ == "__main__":
    main()
