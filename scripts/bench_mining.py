#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound mining / loss kernels at the benchmark batch (B=32, C=6, 512x512):
achieved GB/s against the algorithmic bytes of SURVEY.md section 8(d) (HBM peak 8 TB/s, ~6.3 achievable)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from uemda_amd.utils import synth
from uemda_amd.gast.alignment import Aligner
from uemda_amd.gast.balance import CrossEntropy, UVEMLoss, loss_calc_uvem
from uemda_amd.gast.pseudo_generation import pseudo_selection
from uemda_amd.utils.tools import loss_calc

B, C, S, k = int(os.environ.get("B", 32)), 6, 512, 2048
h = S // 16
pool = synth.make_batch(B=4, H=S, W=S, C=C, k=k, seed=1)
rep = B // 4
b = {key: (v.cuda().repeat((rep,) + (1,) * (v.dim() - 1)).contiguous() if key != "prototypes" else v.cuda()) for key, v in pool.items()}
feat = torch.randn(B, h, h, k, device="cuda").permute(0, 3, 1, 2)
p1 = (2 * torch.randn(B, h, h, C, device="cuda")).permute(0, 3, 1, 2).requires_grad_(True)
p2 = (2 * torch.randn(B, h, h, C, device="cuda")).permute(0, 3, 1, 2).requires_grad_(True)
al = Aligner(None, k, C, -1, 0.996)
al.prototypes = b["prototypes"].clone()
ign = h * h
MB = 1e6


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


px = B * S * S
soft, hard = al.refine_and_select(b["label_t_sup"], feat, [p1, p2], b["label_t_soft"], sup_ignore_id=ign)
rows = [
    ("label_refine (+pearson, segment-max)", lambda: al.label_refine(b["label_t_sup"], feat, [p1, p2], b["label_t_soft"], sup_ignore_id=ign),
     px * (24 + 8 + 24 + 24 + 8) + B * h * h * k * 4),          # soft r (x2: segment max + refine), sup r (x2), soft' w, feat r
    ("pseudo_selection (incl. plane max)", lambda: pseudo_selection(soft, return_type="tensor", check_range=False), px * (24 + 24 + 8)),
    ("refine_and_select (fused maxima)", lambda: al.refine_and_select(b["label_t_sup"], feat, [p1, p2], b["label_t_soft"], sup_ignore_id=ign),
     px * (24 + 8 + 24 + 24 + 8 + 24 + 8) + B * h * h * k * 4),
    ("update_prototype (downscale + sums + EMA)", lambda: al.update_prototype(feat, b["label_s"]), px * 8 + B * h * h * k * 4),
    ("CE fwd+bwd, 2 heads (gather form)", lambda: loss_calc([p1, p2], b["label_s"], CrossEntropy(-1), True).backward(), px * 8),
    ("UVEM fwd+bwd, 2 heads (gather form)", lambda: loss_calc_uvem([p1, p2], hard, soft, UVEMLoss(0.2, 0.7, 4, None, C, -1), True).backward(), px * (8 + 24)),
]
print(f"B={B}  {S}x{S}  C={C}")
for name, fn, nbytes in rows:
    ms = timeit(fn)
    print(f"{name:45s} {ms:8.3f} ms  {nbytes / MB:9.1f} MB algorithmic  {nbytes / ms / 1e6:8.1f} GB/s  ({nbytes / ms / 1e6 / 8000:5.1%} of 8 TB/s)")
