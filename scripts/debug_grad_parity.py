#!/usr/bin/env python3
"""Per-parameter gradient parity of the HIP path against the CPU oracle for one SSL step (no optimizer step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import synth, gast
from oracle.model import OracleDeeplabv2
from oracle.weights import det_state_dict
from uemda_amd.models.Encoder import Deeplabv2
from uemda_amd.gast.alignment import Aligner
from uemda_amd.step import HYPER, StepState
from uemda_amd.gast.balance import loss_calc_uvem
from uemda_amd.utils.tools import loss_calc

C, B, S = 6, 2, int(os.environ.get("SIZE", "256"))
torch.set_num_threads(16)
cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False), multi_layer=True, cascade=False,
           use_ppm=False, ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)
sd = det_state_dict("resnet50", C, False, seed=2333)
bc = synth.make_batch(B=B, H=S, W=S, C=C, k=2048, seed=2333)
mode = os.environ.get("MODE", "ssl")
# ---- oracle
om = OracleDeeplabv2(sd, "resnet50", C, False)
ps1, ps2, fs = om(bc["images_s"])
if mode == "ssl":
    pt1, pt2, ft = om(bc["images_t"])
    with torch.no_grad():
        soft = gast.label_refine(bc["label_t_sup"], ft.detach(), [pt1.detach(), pt2.detach()], bc["label_t_soft"], bc["prototypes"])
        hard = gast.pseudo_selection(soft)
    loss = gast.loss_calc([ps1, ps2], bc["label_s"]) + gast.loss_calc_uvem([pt1, pt2], hard, soft)
else:
    loss = gast.loss_calc([ps1, ps2], bc["label_s"])
loss.backward()
og = {k: v.grad.clone() for k, v in om.named_parameters()}
# ---- HIP
m = Deeplabv2(cfg); m.load_state_dict(sd); m = m.cuda(); m.train()
b = {k: v.cuda() for k, v in bc.items()}
st = StepState(C)
a1, a2, f1 = m(b["images_s"])
if mode == "ssl":
    t1, t2, f2 = m(b["images_t"])
    # use the ORACLE's soft/hard so that only the network gradient path is compared
    l = loss_calc([a1, a2], b["label_s"], st.loss_fn_s, True) + loss_calc_uvem([t1, t2], hard.cuda(), soft.cuda(), st.loss_fn_t, True)
else:
    l = loss_calc([a1, a2], b["label_s"], st.loss_fn_s, True)
m.zero_grad(); l.backward(); torch.cuda.synchronize()
print("loss", float(loss), float(l))
print("fwd pred_s1 rel err", float((a1.detach().cpu() - ps1.detach()).abs().max() / ps1.abs().max()))
for k, p in m.named_parameters():
    g = p.grad.cpu().contiguous(); r = og[k]
    e = float((g - r).norm() / (r.norm() + 1e-20))
    if e > float(os.environ.get("THR", "2e-3")) or "conv1.weight" in k and "layer" not in k:
        print(f"{e:9.2e}  |ref|={float(r.norm()):9.3e}  {k}")
