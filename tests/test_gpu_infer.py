"""GPU parity of the inference-side rows: sliding window, TTA, evaluation, prototype init."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
C = 6


def fake_model(x):
    return torch.stack([x[:, 0] * 0.5 + x[:, 1], x[:, 2] - x[:, 0], x.sum(1) * 0.25], dim=1)


def test_pre_slide_golden_and_tta_vs_oracle():
    from oracle import infer
    from uemda_amd.utils.tools import pre_slide, tta_predict
    g = load_golden("pre_slide")
    out = pre_slide(fake_model, g["image"].cuda(), num_classes=3, tile_size=(32, 32))
    torch.testing.assert_close(out.cpu(), g["out"], rtol=1e-6, atol=1e-6)
    one = pre_slide(fake_model, g["image"][:, :, :32, :32].cuda(), num_classes=3, tile_size=(32, 32))
    torch.testing.assert_close(one.cpu(), g["out_one_tile"], rtol=1e-6, atol=1e-6)

    def asym(x):                                       # position dependent: exercises the de-augmentation order
        h, w = x.shape[-2:]
        ramp = torch.arange(h * w, dtype=torch.float32, device=x.device).view(1, 1, h, w) / (h * w)
        return torch.cat([x[:, :2] * ramp, x[:, 2:] + ramp], 1)
    img = g["image"][:1, :, :40, :40]
    ref = infer.tta_predict(asym, img)
    got = tta_predict(asym, img.cuda())
    torch.testing.assert_close(got.cpu(), ref, rtol=1e-5, atol=1e-6)
    ref = infer.pre_slide(asym, g["image"][:1, :, :64, :64], 3, (32, 32), tta=True)
    got = pre_slide(asym, g["image"][:1, :, :64, :64].cuda(), 3, (32, 32), tta=True)
    torch.testing.assert_close(got.cpu(), ref, rtol=1e-5, atol=1e-6)


def test_confusion_matrix_and_metrics_exact():
    from oracle import infer
    from uemda_amd.utils.eval import ConfusionMeter
    gen = torch.Generator().manual_seed(3)
    prob = torch.rand(3, C, 50, 70, generator=gen)
    prob[0, :, :5] = 0.5                               # ties -> first maximum
    gt = torch.randint(-1, C, (3, 50, 70), generator=gen)
    meter = ConfusionMeter(C, ignore_labels=[0])
    pred = meter.update(prob.cuda(), gt.cuda())
    assert torch.equal(pred.cpu(), prob.argmax(1))
    cm = infer.confusion(prob, gt, C)
    assert np.array_equal(meter.cm.cpu().numpy(), cm)
    res, ref = meter.summary(), infer.metrics(cm, ignore_labels=[0])
    np.testing.assert_allclose(res["iou"], np.round(ref["iou"], 5))
    assert res["miou"] == pytest.approx(ref["miou"], abs=1e-5)


def test_init_prototypes_and_pseudo_generation_roundtrip(tmp_path):
    from oracle import gast, synth
    from oracle.weights import det_state_dict
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.gast.pseudo_generation import gener_target_pseudo
    from uemda_amd.models.Encoder import Deeplabv2
    from uemda_amd.utils.eval import evaluate
    b = synth.make_batch(B=4, H=64, W=64, C=C, k=64, seed=9)
    gen = torch.Generator().manual_seed(1)
    feats = [torch.randn(2, 64, 4, 4, generator=gen) for _ in range(2)]
    al = Aligner(None, 64, C, -1, 0.996)
    sums, cnts = torch.zeros(C, 64), torch.zeros(C)
    for i, f in enumerate(feats):                      # tools/init_prototypes.py:101-111
        lab = b["label_s"][2 * i:2 * i + 2]
        al.update_avg(f.cuda(), lab.cuda())
        ds = gast.downscale_label(lab, C)
        _, s, n = gast.local_prototypes(f, ds, torch.zeros(C, 64), C)
        sums, cnts = sums + s, cnts + n
    al.init_avg()
    torch.testing.assert_close(al.prototypes.cpu(), sums / (cnts.unsqueeze(1) + 1e-7), rtol=1e-5, atol=1e-6)
    # offline pseudo labels: <name>.pt holds the (C,H,W) fp32 probability map the target loader reads back
    cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False), multi_layer=True, cascade=False,
               use_ppm=False, ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)
    model = Deeplabv2(cfg)
    model.load_state_dict(det_state_dict("resnet50", C, False, seed=2333))
    model = model.cuda()
    # 512x768: two overlapping 512-px windows (images smaller than the tile give 0/0 in the reference too).  The call is
    # the reference's: (_cfg, model, pseudo_loader, path, slide, save_prob, size, ignore_label), train_ssl_uem.py:186-187
    from types import SimpleNamespace
    from uemda_amd.gast.pseudo_generation import pseudo_selection
    from uemda_amd.utils.tools import pre_slide
    imgs = [torch.randn(1, 3, 512, 768, generator=gen) for _ in range(2)]
    loader = [(imgs[0], {"fname": ["a.tif"]}), (imgs[1], {"fname": ["b.tif"]})]
    cfg_run = SimpleNamespace(DATASETS="IsprsDA", SNAPSHOT_DIR=str(tmp_path), CUTOFF_TOP=0.8, CUTOFF_LOW=0.6, PSEUDO_SELECT=True)
    out_dir = os.path.join(str(tmp_path), "pseudo")
    gener_target_pseudo(cfg_run, model, loader, out_dir, size=(512, 768), save_prob=True, slide=True, ignore_label=-1)
    prob = torch.load(os.path.join(out_dir, "a.tif.pt"))
    assert prob.shape == (C, 512, 768) and prob.dtype == torch.float32
    assert torch.allclose(prob.sum(0), torch.ones(512, 768), atol=1e-4)
    with torch.no_grad():
        direct = pre_slide(model, imgs[0].cuda(), num_classes=C, tta=True)
    assert torch.equal(prob, direct[0].cpu())
    from PIL import Image
    import numpy as np
    prev = np.array(Image.open(os.path.join(out_dir + "_color", "a.png")))           # colour preview of the selected labels
    assert prev.shape == (512, 768)
    sel = pseudo_selection(direct, 0.8, 0.6, "ndarray", -1)[0]
    # (ignored pixels are written as uint8(-1) like the reference's VisualizeSegmm; Pillow folds that index into the
    # palette's bit depth, so only the labelled pixels are compared)
    assert (prev[sel >= 0] == sel[sel >= 0]).all()
    # a different `size`: the map is resized with bilinear align_corners=True (pseudo_generation.py:135)
    small_dir = os.path.join(str(tmp_path), "small")
    gener_target_pseudo(cfg_run, model, loader[:1], small_dir, size=(256, 384), save_prob=True)
    small = torch.load(os.path.join(small_dir, "a.tif.pt"))
    ref_small = torch.nn.functional.interpolate(direct.cpu(), (256, 384), mode="bilinear", align_corners=True)[0]
    torch.testing.assert_close(small, ref_small, rtol=1e-5, atol=1e-6)
    # class-id branch (save_prob=False): uint8 image of id + 1, 0 = ignored; selection or plain argmax
    ids_dir = os.path.join(str(tmp_path), "ids")
    gener_target_pseudo(cfg_run, model, loader[:1], ids_dir, size=(512, 768), save_prob=False)
    ids = np.array(Image.open(os.path.join(ids_dir, "a.tif")))
    assert (ids == pseudo_selection(direct, return_type="ndarray")[0] + 1).all()
    cfg_run.PSEUDO_SELECT = False
    gener_target_pseudo(cfg_run, model, loader[:1], ids_dir, size=(512, 768), save_prob=False)
    ids = np.array(Image.open(os.path.join(ids_dir, "a.tif")))
    am = direct.argmax(dim=1)[0].cpu().numpy()
    assert (ids == am + 1).all() and len(np.unique(am)) > 1
    assert (np.array(Image.open(os.path.join(ids_dir + "_color", "a.png"))) == am).all()       # the colour preview of those labels
    # fp16 files (half the bytes) load back as fp32 within half precision
    from uemda_amd.gast.pseudo_generation import load_target_pseudo
    half_dir = os.path.join(str(tmp_path), "half")
    gener_target_pseudo(cfg_run, model, loader[:1], half_dir, size=(512, 768), save_prob=True, save_dtype=torch.float16)
    assert torch.load(os.path.join(half_dir, "a.tif.pt")).dtype == torch.float16
    back = load_target_pseudo(os.path.join(half_dir, "a.tif.pt"))
    assert back.dtype == torch.float32 and back.is_cuda
    torch.testing.assert_close(back.cpu(), prob, rtol=2e-3, atol=1e-3)
    imgs = [i.cuda() for i in imgs]
    res, miou = evaluate(model, [(imgs[0], torch.randint(-1, C, (1, 512, 768)).cuda())], C, ignore_labels=[0])
    assert 0.0 <= miou <= 1.0 and res["confusion"].sum() > 0


@pytest.mark.parametrize("hw", [(128, 128), (96, 160)])
def test_tta_d4_batched_equals_sequential(hw):
    """The 8 D4 views as one batch (square tile) or two batches of 4 (non-square) == 8 single-image forwards."""
    from oracle.weights import det_state_dict
    from uemda_amd.models.Encoder import Deeplabv2
    from uemda_amd.utils.tools import tta_predict
    cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False), multi_layer=True, cascade=False,
               use_ppm=False, ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)
    model = Deeplabv2(cfg)
    model.load_state_dict(det_state_dict("resnet50", C, False, seed=2333))
    model = model.cuda().eval()
    img = torch.randn(1, 3, *hw, generator=torch.Generator().manual_seed(9)).cuda()
    with torch.no_grad():
        a = tta_predict(model, img, batched=True)
        b = tta_predict(model, img, batched=False)
    assert a.shape == (1, C, *hw)
    # round 3: at batch 8 the 3x3 convs of layer3/4 take the Winograd path, at batch 1 (16 tiles) the direct kernels: the same
    # fp32 products summed in another order (1e-6 on a probability), no longer the very same kernels on the very same rows
    torch.testing.assert_close(a, b, rtol=5e-5, atol=5e-6)
    assert (a.argmax(1) == b.argmax(1)).float().mean() > 0.9999
    assert ((a.sum(1) - 1).abs() < 1e-5).all()


# ---------------- round 4: the "next" rows against outputs of the reference itself (tests/golden/make_golden_r4.py) ------------
CLOSED_FORM_W = [[0.5, 1.0, 0.0], [-1.0, 0.0, 1.0], [0.25, 0.25, 0.25], [0.0, -0.75, 0.5], [1.0, -1.0, 0.3], [-0.2, 0.6, -0.9]]


class ClosedFormModel(torch.nn.Module):
    """the deterministic stand-in network of the inference fixtures: the caller's model, so plain torch on the device"""

    def forward(self, x):
        w = torch.tensor(CLOSED_FORM_W, dtype=x.dtype, device=x.device)
        return torch.softmax(torch.einsum("kc,bchw->bkhw", w, x), dim=1)


def test_aligner_update_avg_init_avg_reference_golden():
    """Aligner.update_avg x2 + init_avg (reference uemda/gast/alignment.py:107-126) on the HIP path against the reference's own
    running sums, counts and prototypes (an empty class, a class only the second batch sees)"""
    from uemda_amd.gast.alignment import Aligner
    g = load_golden("aligner_avg")
    al = Aligner(None, 64, C, -1, 0.996)
    for feat, lab in zip(g["feats"], g["labels"]):
        al.update_avg(feat.cuda(), lab.cuda())
    torch.testing.assert_close(al._data_sum.cpu(), g["data_sum"], rtol=1e-5, atol=1e-5)
    assert torch.equal(al._data_cnt.cpu().reshape(-1), g["data_cnt"].reshape(-1))
    al.init_avg()
    torch.testing.assert_close(al.prototypes.cpu(), g["prototypes"], rtol=1e-5, atol=1e-6)
    assert float(al.prototypes[5].abs().max()) == 0.0


def test_gener_target_pseudo_pt_file_reference_golden(tmp_path):
    """gener_target_pseudo(slide=False, save_prob=True, size=...) writes the tensor the reference writes for the same model and
    image (uemda/gast/pseudo_generation.py:128-136); the sliding-window map of a two-window image equals the reference's"""
    from types import SimpleNamespace
    from uemda_amd.gast.pseudo_generation import gener_target_pseudo
    from uemda_amd.utils.tools import pre_slide
    g = load_golden("gener_pseudo")
    model = ClosedFormModel()
    cfg = SimpleNamespace(DATASETS="IsprsDA", SNAPSHOT_DIR=None, CUTOFF_TOP=0.8, CUTOFF_LOW=0.6, PSEUDO_SELECT=True)
    out_dir = os.path.join(str(tmp_path), "pseudo")
    gener_target_pseudo(cfg, model, [(g["image"], {"fname": ["tile_a"]})], out_dir, slide=False, save_prob=True, size=(64, 96))
    assert sorted(os.listdir(out_dir)) == ["tile_a.pt"]
    saved = torch.load(os.path.join(out_dir, "tile_a.pt"))
    assert saved.shape == (C, 64, 96) and saved.dtype == torch.float32 and not saved.is_cuda
    torch.testing.assert_close(saved, g["pt_file"], rtol=1e-5, atol=1e-6)
    slide = pre_slide(model, g["image2"].cuda(), num_classes=C, tile_size=(32, 32), tta=False)
    torch.testing.assert_close(slide.cpu(), g["slide2"], rtol=1e-5, atol=1e-6)


def test_evaluate_confusion_counts_reference_golden(monkeypatch):
    """`evaluate` (uemda/utils/eval.py:14-56): the device argmax + confusion matrix holds exactly the (label, prediction) pairs the
    reference's loop hands its metric; per-class formulas are `ever`'s and stay unpinned"""
    import uemda_amd.utils.eval as ev
    g = load_golden("evaluate_pairs")
    orig = ev.pre_slide
    monkeypatch.setattr(ev, "pre_slide", lambda m, x, num_classes, tta=False: orig(m, x, num_classes=num_classes, tile_size=(32, 32), tta=tta))
    batches = [(g[f"image{k}"].cuda(), g[f"label{k}"].cuda()) for k in range(int(g["n_images"]))]
    res, miou = ev.evaluate(ClosedFormModel(), batches, C, ignore_labels=[0], slide=True, tta=False)
    assert np.array_equal(res["confusion"].astype(np.int64), g["confusion"].numpy())
    # and per image: the prediction on the labelled pixels
    for k, (img, lab) in enumerate(batches):
        meter = ev.ConfusionMeter(C)
        pred = meter.update(orig(ClosedFormModel(), img, num_classes=C, tile_size=(32, 32)), lab)
        mask = lab.cpu().numpy() >= 0
        assert np.array_equal(pred.cpu().numpy()[mask].ravel(), g[f"y_pred{k}"].numpy())
