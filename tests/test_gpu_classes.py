"""Class counts other than 6.  The reference's second dataset is 7-class LoveDA (configs/st/uemda/2urban.py:11,
uemda/datasets/loveda.py:18-27, tools/train_ssl_uem.py:80): every kernel that dispatches on the class count is pinned by reference
goldens at C = 7 (tests/golden/make_golden_c7.py: G-ops and one full R50-PPM train_ssl_uem step) and run against the oracle at
C = 12, which takes the <16> instantiations and the c < C tails of the <8> ones (mining, losses, alignment, evaluation, ASPP gather)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def dev(t):
    return t.cuda()


# ------------------------------------------------------------------------------------------------------------------------------
# C = 7: reference goldens
# ------------------------------------------------------------------------------------------------------------------------------
def test_c7_mining_ops_golden():
    from uemda_amd.gast.alignment import Aligner, DownscaleLabel
    from uemda_amd.gast.pseudo_generation import pseudo_selection
    C = 7
    g = load_golden("ops_c7")
    for m, h in zip(g["masks"], g["hards"]):
        assert torch.equal(pseudo_selection(dev(m), 0.8, 0.6, "tensor", -1).cpu(), h)
    al = Aligner(None, feat_channels=64, class_num=C, ignore_label=-1, decay=0.996)
    al.prototypes = dev(g["protos"]).contiguous()
    for mode in ("all", "s", "p", "l"):
        out = al.label_refine(dev(g["sup"]), dev(g["feat"]), [dev(g["p1"]), dev(g["p2"])], dev(g["soft"]), True, mode, 2.0)
        torch.testing.assert_close(out.cpu(), g["refine_" + mode], rtol=2e-5, atol=1e-6)
    out = al.label_refine(dev(g["sup_irregular"]), dev(g["feat"]), [dev(g["p1"]), dev(g["p2"])], dev(g["soft"]), True, "all", 2.0)
    torch.testing.assert_close(out.cpu(), g["refine_all_irregular"], rtol=2e-5, atol=1e-6)
    # refine + select in one pass == the two reference calls
    soft, hard = al.refine_and_select(dev(g["sup"]), dev(g["feat"]), [dev(g["p1"]), dev(g["p2"])], dev(g["soft"]))
    torch.testing.assert_close(soft.cpu(), g["refine_all"], rtol=2e-5, atol=1e-6)
    assert torch.equal(hard.cpu(), pseudo_selection(soft, return_type="tensor").cpu())
    torch.testing.assert_close(al._pearson_dist(dev(g["pearson_x"]), dev(g["protos"])).cpu(), g["pearson_dist"], rtol=1e-5, atol=1e-6)
    assert torch.equal(DownscaleLabel(16, C, -1, 0.75)(dev(g["ds_label"])).cpu(), g["ds_out"])
    al.prototypes = dev(g["protos"]).contiguous().clone()
    ds = al.update_prototype(dev(g["up_feat"]), dev(g["up_label"]))
    assert torch.equal(ds.cpu(), g["up_label_ds"])
    torch.testing.assert_close(al.prototypes.cpu(), g["up_protos_out"], rtol=1e-6, atol=1e-7)


def test_c7_losses_golden_forward_and_backward():
    from uemda_amd.gast.balance import ClassBalance, CrossEntropy, UVEMLoss, loss_calc_uvem
    from uemda_amd.loss import PrototypeContrastiveLoss
    from uemda_amd.utils.tools import loss_calc
    C = 7
    g = load_golden("ops_c7")
    l1, l2 = dev(g["logits1"]).requires_grad_(True), dev(g["logits2"]).requires_grad_(True)
    uv = UVEMLoss(m=0.2, threshold=0.7, gamma=4, class_num=C, ignore_label=-1)
    loss = loss_calc_uvem([l1, l2], dev(g["loss_hard"]), dev(g["loss_soft"]), uv, multi=True)
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), g["uvem"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(l1.grad.cpu(), g["uvem_g1"], rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(l2.grad.cpu(), g["uvem_g2"], rtol=1e-4, atol=1e-8)
    l3, l4 = dev(g["logits1"]).requires_grad_(True), dev(g["logits2"]).requires_grad_(True)
    ce = loss_calc([l3, l4], dev(g["label_s"]), CrossEntropy(ignore_label=-1), multi=True)
    ce.backward()
    torch.testing.assert_close(ce.detach().cpu(), g["ce"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(l3.grad.cpu(), g["ce_g1"], rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(l4.grad.cpu(), g["ce_g2"], rtol=1e-4, atol=1e-8)
    cb = ClassBalance(C, -1, 0.99, 0.5)
    torch.testing.assert_close(cb.get_class_weight_4pixel(dev(g["label_s"])).cpu(), g["cb_weights"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(cb.freq.cpu(), g["cb_freq"], rtol=1e-5, atol=1e-8)
    f = dev(g["pcl_feat"]).requires_grad_(True)
    lp = PrototypeContrastiveLoss(8.0, -1)(dev(g["pcl_protos"]), f, dev(g["pcl_labels"]))
    lp.backward()
    torch.testing.assert_close(lp.detach().cpu(), g["pcl"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(f.grad.cpu(), g["pcl_gfeat"], rtol=1e-4, atol=1e-7)


def test_c7_full_model_ppm_ssl_step_matches_reference_golden():
    """One train_ssl_uem step of R50-PPM with num_classes = 7 (the LoveDA configuration) against the reference: logits within 1e-3,
    pseudo-labels >= 99.95 %, losses, prototypes, gradient norm, and the update of EVERY parameter tensor against its noise floor."""
    from oracle import synth
    from oracle.weights import det_state_dict
    from test_gpu_model import _check_updates
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.models.Encoder import Deeplabv2
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import HYPER, StepState, ssl_step
    C = 7
    g = load_golden("model_ppm_r50_b2_256_c7")
    cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False), multi_layer=True, cascade=False,
               use_ppm=True, ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)
    model = Deeplabv2(cfg)
    sd = det_state_dict("resnet50", C, True, seed=2333)
    assert list(model.state_dict().keys()) == list(sd.keys())
    assert tuple(sd["layer5.conv_last.4.weight"].shape) == (7, 512, 1, 1)
    model.load_state_dict(sd)
    model = model.cuda()
    model.layer5.conv_last[3].p = 0.0
    model.layer6.conv_last[3].p = 0.0
    batch = {k: v.cuda() for k, v in synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=2333).items()}
    al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
    al.prototypes = batch["prototypes"].clone()
    out = ssl_step(model, al, FusedSGD(model, 1e-2, 0.9, 5e-4), StepState(C), batch, float(g["lr"]))
    for k in ("pred_s1", "pred_s2", "pred_t1", "pred_t2"):
        err = (out[k].cpu() - g[k]).abs().max() / g[k].abs().max()
        assert err < 1e-3, (k, float(err))
    assert out["pred_t1"].shape[1] == 7
    torch.testing.assert_close(out["label_t_soft"][:, :, ::4, ::4].cpu(), g["soft_sample"], rtol=1e-3, atol=1e-5)
    assert (out["label_t_hard"].cpu() == g["hard"].long()).float().mean().item() >= 0.9995
    torch.testing.assert_close(out["loss_source"].cpu(), g["loss_source"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(out["loss_target"].cpu(), g["loss_target"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(al.prototypes.cpu(), g["prototypes"], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(out["grad_norm"].cpu().reshape(()), g["grad_norm"], rtol=5e-3, atol=1e-4)
    _check_updates(model, g, True, num_classes=C)


# ------------------------------------------------------------------------------------------------------------------------------
# C = 12: the <16> instantiations against the oracle
# ------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C", [12, 16, 3])
def test_wide_class_counts_mining_vs_oracle(C):
    from oracle import gast, synth
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.gast.pseudo_generation import pseudo_selection
    b = synth.make_batch(B=2, H=96, W=64, C=C, k=64, seed=40 + C)
    gen = torch.Generator().manual_seed(C)
    feat = torch.randn(2, 64, 6, 4, generator=gen)
    p1, p2 = 2 * torch.randn(2, C, 6, 4, generator=gen), 2 * torch.randn(2, C, 6, 4, generator=gen)
    al = Aligner(None, 64, C, -1, 0.996)
    al.prototypes = dev(b["prototypes"]).contiguous()
    for mode in ("all", "s", "p", "l"):
        ref = gast.label_refine(b["label_t_sup"], feat, [p1, p2], b["label_t_soft"], b["prototypes"], mode=mode)
        out = al.label_refine(dev(b["label_t_sup"]), dev(feat), [dev(p1), dev(p2)], dev(b["label_t_soft"]), True, mode, 2.0)
        torch.testing.assert_close(out.cpu(), ref, rtol=3e-5, atol=1e-6)
    soft, hard = al.refine_and_select(dev(b["label_t_sup"]), dev(feat), [dev(p1), dev(p2)], dev(b["label_t_soft"]))
    assert torch.equal(hard.cpu(), gast.pseudo_selection(soft.cpu()))
    # sharpened maps so that classes 8..C-1 actually get selected
    sharp = torch.softmax(12 * torch.randn(2, C, 96, 64, generator=gen), 1)
    hs = pseudo_selection(dev(sharp), 0.8, 0.6, "tensor", -1).cpu()
    assert torch.equal(hs, gast.pseudo_selection(sharp))
    if C > 8:
        assert (hs >= 8).any()
    x = torch.randn(50, 64, generator=gen)
    torch.testing.assert_close(al._pearson_dist(dev(x), dev(b["prototypes"])).cpu(), gast.pearson_dist(x, b["prototypes"]), rtol=1e-5, atol=1e-6)
    protos_ref, ds_ref = gast.update_prototype(feat, b["label_s"], b["prototypes"], C, 0.996)
    ds = al.update_prototype(dev(feat), dev(b["label_s"]))
    assert torch.equal(ds.cpu(), ds_ref)
    torch.testing.assert_close(al.prototypes.cpu(), protos_ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("C", [12, 16, 3])
def test_wide_class_counts_losses_vs_oracle(C):
    from oracle import gast, synth
    from uemda_amd.gast.balance import ClassBalance, CrossEntropy, UVEMLoss, loss_calc_uvem
    from uemda_amd.loss import PrototypeContrastiveLoss
    from uemda_amd.utils.tools import loss_calc
    b = synth.make_batch(B=2, H=64, W=64, C=C, k=8, seed=60 + C)
    gen = torch.Generator().manual_seed(100 + C)
    lg1, lg2 = 2 * torch.randn(2, C, 4, 4, generator=gen), 2 * torch.randn(2, C, 4, 4, generator=gen)
    soft = torch.softmax(4 * torch.randn(2, C, 64, 64, generator=gen), 1)          # entropies on both sides of the 0.7 gate
    hard = gast.pseudo_selection(soft)
    r1, r2 = lg1.clone().requires_grad_(True), lg2.clone().requires_grad_(True)
    ref = gast.loss_calc_uvem([r1, r2], hard, soft, 0.2, 0.7, 4.0, -1, C)
    ref.backward()
    l1, l2 = dev(lg1).requires_grad_(True), dev(lg2).requires_grad_(True)
    loss = loss_calc_uvem([l1, l2], dev(hard), dev(soft), UVEMLoss(0.2, 0.7, 4.0, None, C, -1), multi=True)
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), ref.detach(), rtol=2e-5, atol=1e-7)
    torch.testing.assert_close(l1.grad.cpu(), r1.grad, rtol=2e-4, atol=1e-8)
    torch.testing.assert_close(l2.grad.cpu(), r2.grad, rtol=2e-4, atol=1e-8)
    r3, r4 = lg1.clone().requires_grad_(True), lg2.clone().requires_grad_(True)
    refc = gast.loss_calc([r3, r4], b["label_s"])
    refc.backward()
    l3, l4 = dev(lg1).requires_grad_(True), dev(lg2).requires_grad_(True)
    ce = loss_calc([l3, l4], dev(b["label_s"]), CrossEntropy(ignore_label=-1), multi=True)
    ce.backward()
    torch.testing.assert_close(ce.detach().cpu(), refc.detach(), rtol=2e-5, atol=1e-7)
    torch.testing.assert_close(l3.grad.cpu(), r3.grad, rtol=2e-4, atol=1e-8)
    torch.testing.assert_close(l4.grad.cpu(), r4.grad, rtol=2e-4, atol=1e-8)
    # class-balanced CE (--bcs 1)
    ocb, cb = gast.ClassBalance(C, -1, 0.99, 0.5), ClassBalance(C, -1, 0.99, 0.5)
    r5 = lg1.clone().requires_grad_(True)
    refb = gast.loss_calc([r5], b["label_s"], class_balancer=ocb)
    refb.backward()
    l5 = dev(lg1).requires_grad_(True)
    ceb = loss_calc([l5], dev(b["label_s"]), CrossEntropy(ignore_label=-1, class_balancer=cb), multi=True)
    ceb.backward()
    torch.testing.assert_close(ceb.detach().cpu(), refb.detach(), rtol=2e-5, atol=1e-7)
    torch.testing.assert_close(l5.grad.cpu(), r5.grad, rtol=2e-4, atol=1e-8)
    # prototype-contrastive loss (stage 2)
    feat = torch.randn(2, 128, 6, 5, generator=gen) * 1.5
    protos = torch.randn(C, 128, generator=gen)
    labels = torch.randint(-1, C, (2, 1, 6, 5), generator=gen)
    rf = feat.clone().requires_grad_(True)
    refp = gast.pcl_loss(protos, rf, labels)
    refp.backward()
    f = dev(feat).requires_grad_(True)
    lp = PrototypeContrastiveLoss(8.0, -1)(dev(protos), f, dev(labels))
    lp.backward()
    torch.testing.assert_close(lp.detach().cpu(), refp.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(f.grad.cpu(), rf.grad, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("C", [12, 7])
def test_wide_class_counts_evaluation_and_aspp_head_vs_oracle(C):
    from oracle import infer
    from oracle.model import OracleDeeplabv2
    from oracle.weights import det_state_dict
    from uemda_amd.models.Encoder import Deeplabv2
    from uemda_amd.utils.eval import ConfusionMeter
    gen = torch.Generator().manual_seed(7 + C)
    prob = torch.rand(2, C, 40, 56, generator=gen)
    gt = torch.randint(-1, C, (2, 40, 56), generator=gen)
    meter = ConfusionMeter(C, ignore_labels=[0])
    pred = meter.update(prob.cuda(), gt.cuda())
    assert torch.equal(pred.cpu(), prob.argmax(1))
    assert np.array_equal(meter.cm.cpu().numpy(), infer.confusion(prob, gt, C))
    # ASPP heads: 2 heads x 4 dilations x 9 taps x C columns in ONE GEMM (864 columns at C = 12) + the gather, forward and backward
    cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False), multi_layer=True, cascade=False,
               use_ppm=False, ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)
    sd = det_state_dict("resnet50", C, False, seed=2333)
    model = Deeplabv2(cfg)
    model.load_state_dict(sd)
    model = model.cuda().train()
    om = OracleDeeplabv2({k: v.clone() for k, v in sd.items()}, "resnet50", C, False)
    x = torch.randn(2, 3, 128, 128, generator=gen)
    p1, p2, feat = model(x.cuda())
    om.train()
    r1, r2, rfeat = om(x)
    assert p1.shape == (2, C, 8, 8)
    for a, r in ((p1, r1), (p2, r2)):
        assert (a.detach().cpu() - r.detach()).abs().max() / r.detach().abs().max() < 1e-3
    gy = torch.randn(2, C, 8, 8, generator=gen)
    (p1 * gy.cuda()).sum().backward()
    (r1 * gy).sum().backward()
    named = dict(model.named_parameters())
    for name in ("layer5.conv2d_list.0.weight", "layer5.conv2d_list.3.bias", "layer5.conv2d_list.2.weight"):
        got, want = named[name].grad.cpu(), om.p[name].grad
        assert float((got - want).norm() / (want.norm() + 1e-12)) < 2e-3, name
