#!/usr/bin/env python3
"""Golden vectors for the reference's SECOND dataset: 7-class LoveDA (reference configs/st/uemda/2urban.py:11,
uemda/datasets/loveda.py:18-27, tools/train_ssl_uem.py:80).  Same method as make_golden.py -- the REFERENCE itself, imported from
/root/reference under the third-party stubs, CPU only, build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_c7.py

writes ops_c7.npz (label_refine in every mode, pseudo_selection, DownscaleLabel, update_prototype, Pearson distance, CE / UVEM with
gradients, PrototypeContrastiveLoss, ClassBalance) and model_ppm_r50_b2_256_c7.npz (one full train_ssl_uem step, R50-PPM,
num_classes = 7, with the per-tensor update samples and their fp32 noise floor).  Fixtures are data; no reference source is copied."""
import logging
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg        # noqa: E402  (stubs, import_reference, save, model_cfg)

C = 7


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = mg.import_reference()
    import importlib
    from oracle import synth
    from oracle.weights import det_state_dict
    from oracle.step import HYPER
    logger = logging.getLogger("golden-c7")
    g = torch.Generator().manual_seed(1107)

    # ---------------- G-ops, C = 7 --------------------------------------------------------------------------------
    peaked = torch.softmax(6 * torch.randn(2, C, 48, 48, generator=g), 1)
    low = torch.softmax(0.5 * torch.randn(2, C, 48, 48, generator=g), 1)
    tie = torch.zeros(2, C, 48, 48)
    tie[:, 5] = 0.5
    tie[:, 6] = 0.5                                                   # a tie between the two classes a C = 6 build never sees
    tie[:, :, :8] = peaked[:, :, :8]
    masks = torch.stack([peaked, low, tie])
    hards = torch.stack([ref.pg.pseudo_selection(m.clone(), 0.8, 0.6, "tensor", -1) for m in masks])

    small = synth.make_batch(B=2, H=64, W=64, C=C, k=64, seed=75)
    feat = torch.randn(2, 64, 4, 4, generator=g)
    p1 = torch.randn(2, C, 4, 4, generator=g)
    p2 = torch.randn(2, C, 4, 4, generator=g)
    al = ref.alignment.Aligner(logger, feat_channels=64, class_num=C, ignore_label=-1, decay=0.996)
    al.prototypes = small["prototypes"].clone()
    outs = {}
    for mode in ("all", "s", "p", "l"):
        outs["refine_" + mode] = al.label_refine(small["label_t_sup"], feat, [p1, p2], small["label_t_soft"].clone(), True, mode, 2.0)
    irr = synth.irregular_superpixels(2, 64, 64, 19, seed=77)
    outs["refine_all_irregular"] = al.label_refine(irr, feat, [p1, p2], small["label_t_soft"].clone(), True, "all", 2.0)
    x = torch.randn(29, 64, generator=g)
    dist = al._pearson_dist(x, small["prototypes"])

    lab = small["label_s"].clone()
    lab[0, :16, :16] = 6
    lab[0, :4, :16] = 3                                               # 192 / 256 = 0.75 exactly of class 6: kept
    lab[1, :16, :16] = -1
    ds = ref.alignment.DownscaleLabel(16, C, -1, 0.75)(lab.clone())
    lab2 = lab.clone()
    lab2[lab2 == 4] = 0                                               # class 4 empty
    featp = torch.randn(2, 64, 4, 4, generator=g)
    al2 = ref.alignment.Aligner(logger, feat_channels=64, class_num=C, ignore_label=-1, decay=0.996)
    al2.prototypes = small["prototypes"].clone()
    ds2 = al2.update_prototype(featp, lab2.clone())

    logits = (2 * torch.randn(2, C, 4, 4, generator=g)).requires_grad_(True)
    logits2 = (2 * torch.randn(2, C, 4, 4, generator=g)).requires_grad_(True)
    soft_ref = outs["refine_all"].detach()
    hard = ref.pg.pseudo_selection(soft_ref.clone(), 0.8, 0.6, "tensor", -1)
    uv = ref.balance.UVEMLoss(m=0.2, threshold=0.7, gamma=4, class_num=C, ignore_label=-1)
    lt = ref.balance.loss_calc_uvem([logits, logits2], hard, soft_ref, uv, multi=True)
    lt.backward()
    ce = ref.balance.CrossEntropy(ignore_label=-1)
    lg3 = logits.detach().clone().requires_grad_(True)
    lg4 = logits2.detach().clone().requires_grad_(True)
    ls = ref.tools.loss_calc([lg3, lg4], small["label_s"], ce, multi=True)
    ls.backward()
    # class-balanced CE (--bcs 1, train_ssl_uem.py:129): ClassBalance weights inside the loss, C = 7
    cb = ref.balance.ClassBalance(class_num=C, ignore_label=-1, decay=0.99, temperature=0.5)
    cbw = cb.get_class_weight_4pixel(small["label_s"])

    loss_mod = importlib.import_module("uemda.loss")
    featq = (torch.randn(2, 128, 6, 5, generator=g) * 1.5 + 0.3).requires_grad_(True)
    protosq = torch.randn(C, 128, generator=g)
    labelsq = torch.randint(-1, C, (2, 1, 6, 5), generator=g)
    lp = loss_mod.PrototypeContrastiveLoss(temperature=8.0, ignore_label=-1)(protosq, featq, labelsq)
    lp.backward()
    mg.save("ops_c7", masks=masks, hards=hards, sup=small["label_t_sup"], sup_irregular=irr, feat=feat, p1=p1, p2=p2,
            soft=small["label_t_soft"], protos=small["prototypes"], pearson_x=x, pearson_dist=dist, ds_label=lab, ds_out=ds,
            up_feat=featp, up_label=lab2, up_protos_out=al2.prototypes, up_label_ds=ds2, logits1=logits, logits2=logits2,
            loss_soft=soft_ref, loss_hard=hard, uvem=lt, uvem_g1=logits.grad, uvem_g2=logits2.grad, label_s=small["label_s"], ce=ls,
            ce_g1=lg3.grad, ce_g2=lg4.grad, cb_weights=cbw, cb_freq=cb.freq, pcl_feat=featq, pcl_protos=protosq, pcl_labels=labelsq,
            pcl=lp, pcl_gfeat=featq.grad, **outs)

    # ---------------- one full train_ssl_uem step, R50-PPM, 7 classes ------------------------------------------------------------
    def step(mkldnn=True, ulp_noise=False):
        torch.backends.mkldnn.enabled = mkldnn
        sd = det_state_dict("resnet50", C, True, seed=2333)
        model = ref.Encoder.Deeplabv2(mg.model_cfg(True, C))
        model.load_state_dict(sd, strict=True)
        model.layer5.conv_last[3].p = 0.0
        model.layer6.conv_last[3].p = 0.0
        batch = synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=2333)
        if ulp_noise:
            gn = torch.Generator().manual_seed(77)
            for k in ("images_s", "images_t"):
                sgn = torch.randint(0, 2, batch[k].shape, generator=gn).float() * 2 - 1
                batch[k] = batch[k] * (1.0 + sgn * 2.0 ** -23)
        model.train()
        aln = ref.alignment.Aligner(logger, feat_channels=2048, class_num=C, ignore_label=-1, decay=HYPER["proto_decay"])
        aln.prototypes = batch["prototypes"].clone()
        opt = torch.optim.SGD(model.parameters(), lr=1e-2, momentum=0.9, weight_decay=5e-4)
        cel = ref.balance.CrossEntropy(ignore_label=-1)
        uvl = ref.balance.UVEMLoss(m=0.2, threshold=0.7, gamma=4, class_num=C, ignore_label=-1)
        lr = 3e-3
        opt.param_groups[0]["lr"] = lr
        ps1, ps2, feat_s = model(batch["images_s"])
        pt1, pt2, feat_t = model(batch["images_t"])
        soft = aln.label_refine(batch["label_t_sup"], feat_t, [pt1, pt2], batch["label_t_soft"], True, "all", 2.0)
        hrd = ref.pg.pseudo_selection(soft, 0.8, 0.6, "tensor", -1)
        aln.update_prototype(feat_s, batch["label_s"])
        loss_s = ref.tools.loss_calc([ps1, ps2], batch["label_s"], cel, multi=True)
        loss_t = ref.balance.loss_calc_uvem([pt1, pt2], hrd, soft, uvl, multi=True)
        opt.zero_grad()
        (loss_s + loss_t).backward()
        named = dict(model.named_parameters())
        gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=32, norm_type=2)
        upd = {n: (-lr * (p.grad.detach().double() + 5e-4 * p.detach().double())).reshape(-1)[:: max(1, p.numel() // 256)][:256].clone()
               for n, p in named.items()}
        opt.step()
        torch.backends.mkldnn.enabled = True
        return dict(ps1=ps1, ps2=ps2, pt1=pt1, pt2=pt2, soft=soft, hard=hrd, loss_s=loss_s, loss_t=loss_t, protos=aln.prototypes,
                    gnorm=gnorm, lr=lr, upd=upd)

    print("G-model ppm, 7 classes")
    r, r2, r3 = step(), step(mkldnn=False), step(ulp_noise=True)
    names = list(r["upd"].keys())
    noise = np.array([max(float((r["upd"][n] - q["upd"][n]).norm() / (r["upd"][n].norm() + 1e-30)) for q in (r2, r3)) for n in names])
    mg.save("model_ppm_r50_b2_256_c7", pred_s1=r["ps1"], pred_s2=r["ps2"], pred_t1=r["pt1"], pred_t2=r["pt2"],
            soft_sample=r["soft"][:, :, ::4, ::4], hard=r["hard"].to(torch.int8), loss_source=r["loss_s"], loss_target=r["loss_t"],
            prototypes=r["protos"], grad_norm=r["gnorm"], lr=r["lr"], upd_names=np.array(names),
            upd_offsets=np.cumsum([0] + [r["upd"][n].numel() for n in names]), upd_samples=torch.cat([r["upd"][n] for n in names]),
            upd_noise_floor=noise)
    print("done")


if __name__ == "__main__":
    main()
