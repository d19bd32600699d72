#!/usr/bin/env python3
"""Per-tensor update error of one train_ssl_uem step under the optional ResNetEncoder modes (frozen / cp) against the
reference fixture, with the fixture's own fp32 noise floor beside it (diagnostic for tests/test_gpu_model.py)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "frozen"
    from conftest import load_golden
    from oracle import synth
    from oracle.weights import det_state_dict
    from test_gpu_model import ENCODER_OPTIONS, _model, C
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import HYPER, StepState, ssl_step
    g = load_golden(f"model_aspp_r50_b2_256_{tag}")
    from conftest import golden_initial_state
    model = _model(False, sd=golden_initial_state(g, det_state_dict("resnet50", C, False, seed=2333)), **ENCODER_OPTIONS[tag])
    model.train()
    batch = {k: v.cuda() for k, v in synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=2333).items()}
    al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
    al.prototypes = batch["prototypes"].clone()
    opt = FusedSGD(model, lr=1e-2, momentum=0.9, weight_decay=5e-4)
    out = ssl_step(model, al, opt, StepState(C), batch, float(g["lr"]))
    hard = out["label_t_hard"].cpu()
    ref_hard = g["hard"].long()
    print("hard pseudo-label pixels that differ from the reference:", int((hard != ref_hard).sum()), "of", hard.numel(),
          "| logits max rel err", max(float((out[k].cpu() - g[k]).abs().max() / g[k].abs().max()) for k in ("pred_s1", "pred_s2", "pred_t1", "pred_t2")),
          "| grad norm", float(out["grad_norm"]), float(g["grad_norm"]))
    w0 = det_state_dict("resnet50", C, False, seed=2333)          # weights: the calibration touches running statistics only
    lr, wd = float(g["lr"]), 5e-4
    names, off, ref, floor = [str(n) for n in g["upd_names"]], g["upd_offsets"], g["upd_samples"], g["upd_noise_floor"]
    named = dict(model.named_parameters())
    for i, n in enumerate(names):
        p = named[n]
        if p.grad is None:
            continue
        st = max(1, p.numel() // 256)
        w_pre = w0[n].reshape(-1)[::st][:256].double()
        grad = p.grad.detach().cpu().reshape(-1)[::st][:256].double()
        upd = -lr * (grad + wd * w_pre)
        r = ref[int(off[i]):int(off[i + 1])].double()
        gref = -r / lr - wd * w_pre
        err = float((upd - r).norm() / (r.norm() + 1e-300))
        gerr = float((grad - gref).norm() / (gref.norm() + 1e-300))
        print(f"{n:55s} upd err {err:.3e} grad err {gerr:.3e} floor {float(floor[i]):.3e} |g| {float(gref.norm()):.3e} wd|w| {float(wd * w_pre.norm()):.3e}")


if __name__ == "__main__" and len(sys.argv) <= 2:
    main()


def full_compare(tag="frozen", names=("encoder.resnet.layer4.0.conv2.weight", "encoder.resnet.layer3.0.conv1.weight",
                                      "encoder.resnet.layer4.0.conv1.weight", "encoder.resnet.layer4.1.conv2.weight")):
    """Whole-tensor gradient comparison against the oracle run in the same mode, per filter tap."""
    from oracle import synth
    from oracle.model import OracleDeeplabv2
    from oracle.step import HYPER as OH, SGDState, ssl_step as oracle_ssl
    from oracle.weights import det_state_dict
    from test_gpu_model import ENCODER_OPTIONS, _model, C
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import HYPER, StepState, ssl_step
    sd = det_state_dict("resnet50", C, False, seed=2333)
    bc = synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=2333)
    om = OracleDeeplabv2(sd, "resnet50", C, False, **ENCODER_OPTIONS[tag])
    oracle_ssl(om, SGDState(om.parameters(), 0.9, 5e-4), bc["prototypes"], bc, 3e-3, OH, dropout=False)
    model = _model(False, **ENCODER_OPTIONS[tag])
    model.train()
    batch = {k: v.cuda() for k, v in bc.items()}
    al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
    al.prototypes = batch["prototypes"].clone()
    ssl_step(model, al, FusedSGD(model, 1e-2, 0.9, 5e-4), StepState(C), batch, 3e-3)
    named = dict(model.named_parameters())
    for n in names:
        a, b = named[n].grad.detach().cpu().double(), om.p[n].grad.double()
        print(n, "rel L2", float((a - b).norm() / b.norm()), "max abs", float((a - b).abs().max()), "ref max", float(b.abs().max()))
        if a.dim() == 4 and a.shape[-1] == 3:
            for ky in range(3):
                print("   tap row", ky, [f"{float((a[:, :, ky, kx] - b[:, :, ky, kx]).norm() / b[:, :, ky, kx].norm()):.2e}" for kx in range(3)])
        d = (a - b).abs().reshape(a.shape[0], -1)
        print("   per-out-channel err (top 5):", torch.topk(d.norm(dim=1) / b.reshape(b.shape[0], -1).norm(dim=1), 5))


if len(sys.argv) > 2 and sys.argv[2] == "full":
    full_compare(sys.argv[1])


def trace_block_grads(tag="frozen"):
    """Gradient wrt every bottleneck OUTPUT, ours against the oracle run in the same mode: where does the backward drift?"""
    from oracle import synth
    from oracle.model import OracleDeeplabv2
    from oracle.step import HYPER as OH, SGDState, ssl_step as oracle_ssl
    from oracle.weights import det_state_dict
    from test_gpu_model import ENCODER_OPTIONS, _model, C
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.optim import FusedSGD
    from uemda_amd.resnet import Bottleneck
    from uemda_amd.step import HYPER, StepState, ssl_step
    sd = det_state_dict("resnet50", C, False, seed=2333)
    bc = synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=2333)
    om = OracleDeeplabv2(sd, "resnet50", C, False, **ENCODER_OPTIONS[tag])
    ograds, oacts = {}, {}
    orig = om._bottleneck

    def wrapped(x, prefix, stride, dilation, has_ds):
        y = orig(x, prefix, stride, dilation, has_ds)
        key = (prefix, len([k for k in oacts if k[0] == prefix]))
        oacts[key] = y.detach()
        if y.requires_grad:
            y.register_hook(lambda g, key=key: ograds.__setitem__(key, g.detach().clone()))
        return y
    om._bottleneck = wrapped
    oracle_ssl(om, SGDState(om.parameters(), 0.9, 5e-4), bc["prototypes"], bc, 3e-3, OH, dropout=False)
    model = _model(False, **ENCODER_OPTIONS[tag])
    model.train()
    mgrads, macts, counts = {}, {}, {}
    names = {m: n for n, m in model.named_modules()}

    def fhook(mod, inp, out):
        prefix = names[mod]
        key = (prefix, counts.get(prefix, 0))
        counts[prefix] = key[1] + 1
        macts[key] = out.detach().permute(0, 3, 1, 2).cpu()
        if out.requires_grad:
            out.register_hook(lambda g, key=key: mgrads.__setitem__(key, g.detach().permute(0, 3, 1, 2).cpu().clone()))
    for m in model.modules():
        if isinstance(m, Bottleneck):
            m.register_forward_hook(fhook)
    batch = {k: v.cuda() for k, v in bc.items()}
    al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
    al.prototypes = batch["prototypes"].clone()
    ssl_step(model, al, FusedSGD(model, 1e-2, 0.9, 5e-4), StepState(C), batch, 3e-3)
    for key in sorted(ograds, key=lambda k: (k[1], k[0]), reverse=True):
        if key in mgrads:
            a, b = mgrads[key].double(), ograds[key].double()
            fa, fb = macts[key].double(), oacts[key].double()
            print(f"{key[0]:32s} pass {key[1]}  activation rel L2 {float((fa - fb).norm() / fb.norm()):.2e}  gradient rel L2 {float((a - b).norm() / b.norm()):.2e} "
                  f" max|dg| {float((a - b).abs().max()):.2e} / max|g| {float(b.abs().max()):.2e}  elements off by > 1e-3 max|g|: {int(((a - b).abs() > 1e-3 * b.abs().max()).sum())}")


if len(sys.argv) > 2 and sys.argv[2] == "trace":
    trace_block_grads(sys.argv[1])
