"""GPU parity of the stage-2 alignment losses and step."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
C = 6


def test_pcl_and_coral_golden():
    from uemda_amd.gast.coral import CoralLoss
    from uemda_amd.loss import PrototypeContrastiveLoss
    g = load_golden("align_losses")
    f = g["feat"].cuda().requires_grad_(True)
    l = PrototypeContrastiveLoss(8.0, -1)(g["protos"].cuda(), f, g["labels"].cuda())
    (l * 3.0).backward()
    torch.testing.assert_close(l.detach().cpu(), g["pcl"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(f.grad.cpu(), g["pcl_gfeat_x3"], rtol=1e-4, atol=1e-7)
    s, t = g["src"].cuda().requires_grad_(True), g["tgt"].cuda().requires_grad_(True)
    lc = CoralLoss()(s, t)
    lc.backward()
    torch.testing.assert_close(lc.detach().cpu(), g["coral"], rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(s.grad.cpu(), g["coral_gsrc"], rtol=1e-3, atol=1e-8)
    torch.testing.assert_close(t.grad.cpu(), g["coral_gtgt"], rtol=1e-3, atol=1e-8)


def test_align_step_vs_oracle():
    from oracle import synth
    from oracle.model import OracleDeeplabv2
    from oracle.step import HYPER as OH, SGDState, align_step as oracle_align
    from oracle.weights import det_state_dict
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.models.Encoder import Deeplabv2
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import HYPER, StepState, align_step
    cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False), multi_layer=True, cascade=False,
               use_ppm=False, ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)
    sd = det_state_dict("resnet50", C, False, seed=2333)
    for head in ("layer5", "layer6"):          # confident predictions, so that some on-the-fly pseudo labels survive
        for i in range(4):                     # (with random heads every target label is ignored and PCL is 0/0 = NaN
            sd[f"{head}.conv2d_list.{i}.bias"][0] += 1.5   #  in the reference too)
    bc = synth.make_batch(B=2, H=128, W=128, C=C, k=2048, seed=12)
    om = OracleDeeplabv2(sd, "resnet50", C, False)
    ref = oracle_align(om, SGDState(om.parameters(), 0.9, 5e-4), bc["prototypes"], bc, 3e-3, OH, C)
    model = Deeplabv2(cfg)
    model.load_state_dict(sd)
    model = model.cuda()
    b = {k: v.cuda() for k, v in bc.items()}
    al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
    al.prototypes = b["prototypes"].clone()
    out = align_step(model, al, FusedSGD(model, 1e-2, 0.9, 5e-4), StepState(C), b, 3e-3)
    for k in ("loss_seg", "loss_domain", "loss_align"):
        torch.testing.assert_close(out[k].cpu().reshape(()), ref[k].reshape(()).float(), rtol=2e-3, atol=1e-6)
    assert (out["label_t_hard"].cpu() == ref["label_t_hard"]).float().mean() > 0.999
    assert (ref["label_t_hard"] >= 0).float().mean() > 0.2 and torch.isfinite(ref["loss_align"])
    torch.testing.assert_close(al.prototypes.cpu(), ref["prototypes"], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(out["grad_norm"].cpu().reshape(()), ref["grad_norm"].reshape(()), rtol=2e-2, atol=1e-4)


def test_src_step_with_domain_alignment_vs_oracle():
    """Stage 1 with --align-domain (tools/train_src.py:126-135): source CE + CORAL(feat_s, feat_t), one step."""
    from oracle import synth
    from oracle.model import OracleDeeplabv2
    from oracle.step import HYPER as OH, SGDState, src_step as oracle_src
    from oracle.weights import det_state_dict
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.models.Encoder import Deeplabv2
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import HYPER, StepState, src_step
    cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False), multi_layer=True, cascade=False,
               use_ppm=False, ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)
    sd = det_state_dict("resnet50", C, False, seed=2333)
    bc = synth.make_batch(B=2, H=128, W=128, C=C, k=2048, seed=21)
    om = OracleDeeplabv2({k: v.clone() for k, v in sd.items()}, "resnet50", C, False)
    ref = oracle_src(om, SGDState(om.parameters(), 0.9, 5e-4), bc, 3e-3, OH, align_domain=True)
    model = Deeplabv2(cfg)
    model.load_state_dict(sd)
    model = model.cuda()
    al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
    out = src_step(model, FusedSGD(model, 1e-2, 0.9, 5e-4), StepState(C), {k: v.cuda() for k, v in bc.items()}, 3e-3,
                   aligner=al, align_domain=True)
    torch.testing.assert_close(out["loss_source"].cpu().reshape(()), ref["loss_source"].reshape(()), rtol=1e-3, atol=1e-6)
    torch.testing.assert_close(out["loss_domain"].cpu().reshape(()), ref["loss_domain"].reshape(()).float(), rtol=2e-3, atol=1e-8)
    torch.testing.assert_close(out["grad_norm"].cpu().reshape(()), ref["grad_norm"].reshape(()), rtol=2e-2, atol=1e-4)
