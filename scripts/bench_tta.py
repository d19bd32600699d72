#!/usr/bin/env python3
"""Offline pseudo-label generation: 8-view test-time augmentation of one 512x512 tile, the 8 views as
one batch (D4-symmetry batching) against 8 single-image forwards."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from uemda_amd.models.Encoder import Deeplabv2
from uemda_amd.utils.tools import tta_predict

C = 6
cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False), multi_layer=True, cascade=False,
           use_ppm=False, ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)
img = torch.randn(1, 3, 512, 512, device="cuda")
for storage in ("fp32", "bf16"):
  model = Deeplabv2(cfg).cuda().set_storage(storage).eval()
  with torch.no_grad():
    for batched in (False, True):
        for _ in range(3):
            tta_predict(model, img, batched=batched)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            tta_predict(model, img, batched=batched)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"tta_predict 512x512, {storage} storage, {'one batch of 8' if batched else '8 forwards   '}: {1e3 * dt:.2f} ms per tile ({1 / dt:.1f} tiles/s)")
