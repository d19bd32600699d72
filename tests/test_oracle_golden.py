"""The oracle (CPU restatement) is pinned against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU-only; runs in the build container and on the GPU box."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden_initial_state, load_golden
from oracle import gast, synth
from oracle.model import OracleDeeplabv2, param_shapes
from oracle.step import HYPER, SGDState, ssl_step
from oracle.weights import checksum, det_state_dict, fill_like, subsample

C = 6


def test_pseudo_selection_bit_exact():
    g = load_golden("pseudo_selection")
    for m, h in zip(g["masks"], g["hards"]):
        out = gast.pseudo_selection(m, 0.8, 0.6, -1)
        assert out.dtype == torch.int64
        assert torch.equal(out, h)


@pytest.mark.parametrize("mode", ["all", "s", "p", "l"])
def test_label_refine_modes(mode):
    g = load_golden("label_refine")
    out = gast.label_refine(g["sup"], g["feat"], [g["p1"], g["p2"]], g["soft"], g["protos"], True, mode, 2.0)
    torch.testing.assert_close(out, g["out_" + mode], rtol=1e-5, atol=2e-7)


def test_label_refine_irregular_and_single_pred():
    g = load_golden("label_refine")
    out = gast.label_refine(g["sup_irregular"], g["feat"], [g["p1"], g["p2"]], g["soft"], g["protos"], True, "all", 2.0)
    torch.testing.assert_close(out, g["out_all_irregular"], rtol=1e-5, atol=2e-7)
    out = gast.label_refine(g["sup"], g["feat"], g["p1"], g["soft"], g["protos"], True, "l", 1.5)
    torch.testing.assert_close(out, g["out_single_pred"], rtol=1e-5, atol=2e-7)


def test_pearson_gemm_form():
    g = load_golden("pearson")
    torch.testing.assert_close(gast.pearson_dist(g["x"], g["protos"]), g["dist"], rtol=1e-5, atol=1e-6)


def test_downscale_label_exact():
    g = load_golden("downscale_label")
    out = gast.downscale_label(g["label"], C)
    assert torch.equal(out, g["out"])
    assert out[0, 0, 0, 0] == 2 and out[0, 0, 1, 0] == -1 and out[1, 0, 0, 0] == -1


def test_update_prototype_with_empty_class():
    g = load_golden("update_prototype")
    new, ds = gast.update_prototype(g["feat"], g["label"], g["protos_in"], C, 0.996)
    assert torch.equal(ds, g["label_ds"])
    torch.testing.assert_close(new, g["protos_out"], rtol=1e-6, atol=1e-7)


def test_losses_and_grads():
    g = load_golden("losses")
    l1 = g["logits1"].clone().requires_grad_(True)
    l2 = g["logits2"].clone().requires_grad_(True)
    loss = gast.loss_calc_uvem([l1, l2], g["hard"], g["soft"], 0.2, 0.7, 4.0, -1, C)
    loss.backward()
    torch.testing.assert_close(loss.detach(), g["uvem"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(l1.grad, g["uvem_g1"], rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(l2.grad, g["uvem_g2"], rtol=1e-4, atol=1e-8)
    l3 = g["logits1"].clone().requires_grad_(True)
    l4 = g["logits2"].clone().requires_grad_(True)
    ce = gast.loss_calc([l3, l4], g["label_s"], -1)
    ce.backward()
    torch.testing.assert_close(ce.detach(), g["ce"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(l3.grad, g["ce_g1"], rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(gast.uvem_weight(g["u"]), g["uvem_w"], rtol=1e-6, atol=1e-7)


def test_class_balance_three_steps():
    g = load_golden("class_balance")
    cb = gast.ClassBalance(C, -1, 0.99, 2.0)
    for lab, w in zip(g["labels"], g["weights"]):
        torch.testing.assert_close(cb.get_class_weight_4pixel(lab), w, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(cb.freq, g["freq"], rtol=1e-6, atol=1e-8)


def test_lr_schedule():
    g = load_golden("lr_schedule")
    for i, lr in zip(g["iters"].tolist(), g["lrs"].tolist()):
        assert gast.learning_rate(i, 1e-2, int(6000 / 20), 6000 * 1.5, 0.9) == pytest.approx(lr, rel=1e-12)


def test_scatter_semantics():
    src = torch.tensor([[[1.0, -2.0], [3.0, -1.0], [0.5, -4.0]]])
    idx = torch.tensor([[[2], [0], [2]]])
    out = gast.scatter(src, idx, 1, "max")
    assert out.shape == (1, 3, 2)
    assert out[0, 1].tolist() == [0.0, 0.0]              # untouched segment -> 0
    assert out[0, 2].tolist() == [1.0, -2.0] and out[0, 0].tolist() == [3.0, -1.0]
    assert gast.scatter(src, idx, 1, "sum")[0, 2].tolist() == [1.5, -6.0]


# ---------------- layers ---------------------------------------------------------------------------
def _layer_model(state):
    m = OracleDeeplabv2({}, requires_grad=False)
    m.p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
           for k, v in state.items()}
    return m


def _check_layer(name, shapes, fn, rtol=2e-4, atol=2e-5):
    g = load_golden(name)
    m = _layer_model(fill_like(shapes, name))
    x = g["x"].clone().requires_grad_(True)
    y = fn(m, x)
    torch.testing.assert_close(y, g["y"], rtol=rtol, atol=atol)
    y.backward(g["gy"])
    torch.testing.assert_close(x.grad, g["gx"], rtol=rtol * 5, atol=atol * 5)
    for k, v in g.items():
        if k.startswith("g:"):
            torch.testing.assert_close(subsample(m.p[k[2:]].grad), v, rtol=rtol * 5, atol=atol * 20)
        if k.startswith("post:"):
            torch.testing.assert_close(m.p[k[5:]], v, rtol=1e-5, atol=1e-6)


def _bn_shapes(prefix, c):
    return {prefix + ".weight": (c,), prefix + ".bias": (c,), prefix + ".running_mean": (c,),
            prefix + ".running_var": (c,), prefix + ".num_batches_tracked": ()}


def _bott_shapes(inpl, planes, ds):
    s = {"conv1.weight": (planes, inpl, 1, 1)}
    s.update(_bn_shapes("bn1", planes))
    s["conv2.weight"] = (planes, planes, 3, 3)
    s.update(_bn_shapes("bn2", planes))
    s["conv3.weight"] = (planes * 4, planes, 1, 1)
    s.update(_bn_shapes("bn3", planes * 4))
    if ds:
        s["downsample.0.weight"] = (planes * 4, inpl, 1, 1)
        s.update(_bn_shapes("downsample.1", planes * 4))
    return s


def _strip(m, prefix):
    """oracle blocks address params as '<prefix>.conv1.weight'; fixtures use bare names."""
    m.p = {prefix + "." + k: v for k, v in m.p.items()}


def test_layer_bottleneck_stride2():
    def fn(m, x):
        _strip(m, "blk")
        y = m._bottleneck(x, "blk", 2, 1, True)
        m.p = {k[4:]: v for k, v in m.p.items()}
        return y
    _check_layer("layer_bottleneck_s2", _bott_shapes(64, 32, True), fn)


def test_layer_bottleneck_dilation2():
    def fn(m, x):
        _strip(m, "blk")
        y = m._bottleneck(x, "blk", 1, 2, False)
        m.p = {k[4:]: v for k, v in m.p.items()}
        return y
    _check_layer("layer_bottleneck_d2", _bott_shapes(128, 32, False), fn)


def test_layer_aspp():
    shapes = {}
    for i in range(4):
        shapes[f"conv2d_list.{i}.weight"] = (C, 32, 3, 3)
        shapes[f"conv2d_list.{i}.bias"] = (C,)

    def fn(m, x):
        _strip(m, "hd")
        y = m.aspp_head(x, "hd")
        m.p = {k[3:]: v for k, v in m.p.items()}
        return y
    _check_layer("layer_aspp", shapes, fn)


def test_layer_ppm():
    shapes = {}
    for i in range(4):
        shapes[f"ppm.{i}.1.weight"] = (512, 32, 1, 1)
        shapes.update(_bn_shapes(f"ppm.{i}.2", 512))
    shapes["conv_last.0.weight"] = (512, 32 + 2048, 3, 3)
    shapes.update(_bn_shapes("conv_last.1", 512))
    shapes["conv_last.4.weight"] = (C, 512, 1, 1)
    shapes["conv_last.4.bias"] = (C,)

    def fn(m, x):
        _strip(m, "hd")
        y = m.ppm_head(x, "hd", dropout=False)
        m.p = {k[3:]: v for k, v in m.p.items()}
        return y
    _check_layer("layer_ppm", shapes, fn, rtol=1e-3, atol=1e-4)


def test_layer_instnorm_and_stem():
    g = load_golden("layer_instnorm")
    x = g["x"].clone().requires_grad_(True)
    y = F.instance_norm(x, eps=1e-5)
    torch.testing.assert_close(y, g["y"], rtol=1e-5, atol=1e-6)
    y.backward(g["gy"])
    torch.testing.assert_close(x.grad, g["gx"], rtol=1e-4, atol=1e-6)
    shapes = {"0.weight": (64, 3, 7, 7)}
    shapes.update(_bn_shapes("1", 64))

    def fn(m, x):
        m.p = {("encoder.resnet.conv1.weight" if k == "0.weight" else "encoder.resnet.bn1" + k[1:]): v
               for k, v in m.p.items()}
        y = F.conv2d(x, m.p["encoder.resnet.conv1.weight"], stride=2, padding=3)
        y = F.max_pool2d(F.relu(m._bn(y, "encoder.resnet.bn1")), 3, 2, 1)
        m.p = {("0.weight" if k.endswith("conv1.weight") else "1" + k[len("encoder.resnet.bn1"):]): v
               for k, v in m.p.items()}
        return y
    _check_layer("layer_stem", shapes, fn)


# ---------------- full model + one SSL step (B=2, 256^2: BASELINE config 1) ------------------------------
@pytest.mark.parametrize("tag", ["aspp", "ppm"])
def test_full_model_ssl_step(tag):
    use_ppm = tag == "ppm"
    g = load_golden(f"model_{tag}_r50_b2_256")
    sd = det_state_dict("resnet50", C, use_ppm, seed=2333)
    assert list(sd.keys()) == list(param_shapes("resnet50", C, use_ppm).keys())
    assert len(sd) == (382 if use_ppm else 334)
    model = OracleDeeplabv2(sd, "resnet50", C, use_ppm)
    batch = synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=2333)
    model.eval()
    with torch.no_grad():
        prob = model(batch["images_t"])
    torch.testing.assert_close(prob[:, :, ::8, ::8], g["eval_prob_sample"], rtol=1e-4, atol=1e-6)
    opt = SGDState(model.parameters(), HYPER["momentum"], HYPER["weight_decay"])
    out = ssl_step(model, opt, batch["prototypes"], batch, float(g["lr"]), HYPER, dropout=False)
    torch.testing.assert_close(out["pred_s1"], g["pred_s1"], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(out["pred_t2"], g["pred_t2"], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(out["feat_t"].reshape(-1)[g["feat_idx"]], g["feat_t_sample"], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(out["label_t_soft"][:, :, ::4, ::4], g["soft_sample"], rtol=1e-4, atol=1e-6)
    agree = (out["label_t_hard"] == g["hard"].long()).float().mean().item()
    assert agree >= 0.9999, agree
    torch.testing.assert_close(out["loss_source"], g["loss_source"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(out["loss_target"], g["loss_target"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(out["prototypes"], g["prototypes"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(out["grad_norm"], g["grad_norm"], rtol=1e-3, atol=1e-5)
    named = dict(model.named_parameters())
    for k, v in g.items():
        if k.startswith("grad:"):
            # fixture grads are views taken before clip_grad_norm_ scaled them in place => post-clip
            torch.testing.assert_close(subsample(named[k[5:]].grad), v, rtol=2e-2, atol=2e-5)
    s, a = checksum(model.parameters())
    assert s == pytest.approx(float(g["post_checksum"][0]), rel=1e-5, abs=1e-2)
    assert a == pytest.approx(float(g["post_checksum"][1]), rel=1e-6)
    torch.testing.assert_close(model.p["encoder.resnet.bn1.running_mean"], g["post_bn1_running_mean"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(model.p["encoder.resnet.layer4.2.bn3.running_var"], g["post_l4_bn3_running_var"], rtol=1e-4, atol=1e-6)
    assert int(model.p["encoder.resnet.bn1.num_batches_tracked"]) == int(g["nbt"]) == 2


ENCODER_OPTIONS = {"frozen": dict(freeze_at=2, batchnorm_trainable=False), "cp": dict(with_cp=(True, True, True, True))}


@pytest.mark.parametrize("tag", ["frozen", "cp"])
def test_ssl_step_encoder_options(tag):
    """The ResNetEncoder modes no UemDA script switches on but its config exposes (reference uemda/resnet.py:112-130
    freeze_at / frozen BatchNorm, :146-165 checkpointed layers): the oracle's restatement against the reference's own step."""
    g = load_golden(f"model_aspp_r50_b2_256_{tag}")
    assert str(g["options"]) == repr(ENCODER_OPTIONS[tag])
    sd = golden_initial_state(g, det_state_dict("resnet50", C, False, seed=2333))
    model = OracleDeeplabv2(sd, "resnet50", C, False, **ENCODER_OPTIONS[tag])
    batch = synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=2333)
    frozen = [str(n) for n in g["frozen_names"] if str(n)]
    assert sorted(k for k, t in model.p.items() if t.is_floating_point() and "running_" not in k and not t.requires_grad) == sorted(frozen)
    opt = SGDState(model.parameters(), HYPER["momentum"], HYPER["weight_decay"])
    out = ssl_step(model, opt, batch["prototypes"], batch, float(g["lr"]), HYPER, dropout=False)
    for k in ("pred_s1", "pred_s2", "pred_t1", "pred_t2"):
        torch.testing.assert_close(out[k], g[k], rtol=1e-3, atol=1e-4)
    assert (out["label_t_hard"] == g["hard"].long()).float().mean().item() >= 0.9999
    torch.testing.assert_close(out["loss_source"], g["loss_source"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(out["loss_target"], g["loss_target"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(out["grad_norm"], g["grad_norm"], rtol=1e-3, atol=1e-5)
    for k in frozen:                                                   # untouched: no gradient, no weight decay
        assert torch.equal(model.p[k], sd[k].float()), k
        assert model.p[k].grad is None
    for k, v in g.items():
        if k.startswith("post:"):
            got = model.p[k[5:]]
            if k.endswith("num_batches_tracked"):
                assert int(got) == int(v), (k, int(got), int(v))
            else:
                torch.testing.assert_close(got, v, rtol=1e-4, atol=1e-6)


def test_pre_slide_windowing_golden():
    from oracle import infer
    g = load_golden("pre_slide")

    def fake_model(x):
        return torch.stack([x[:, 0] * 0.5 + x[:, 1], x[:, 2] - x[:, 0], x.sum(1) * 0.25], dim=1)
    out = infer.pre_slide(fake_model, g["image"], num_classes=3, tile_size=(32, 32))
    torch.testing.assert_close(out, g["out"], rtol=1e-6, atol=1e-6)
    one = infer.pre_slide(fake_model, g["image"][:, :, :32, :32], num_classes=3, tile_size=(32, 32))
    torch.testing.assert_close(one, g["out_one_tile"], rtol=1e-6, atol=1e-6)
    # TTA of a model that commutes with the dihedral group is the model itself
    eq = infer.tta_predict(lambda x: x * 2.0, g["image"][:1])
    torch.testing.assert_close(eq, g["image"][:1] * 2.0, rtol=1e-6, atol=1e-6)


def test_stage2_alignment_losses_golden():
    g = load_golden("align_losses")
    f = g["feat"].clone().requires_grad_(True)
    l = gast.pcl_loss(g["protos"], f, g["labels"], 8.0, -1)
    (l * 3.0).backward()
    torch.testing.assert_close(l.detach(), g["pcl"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(f.grad, g["pcl_gfeat_x3"], rtol=1e-5, atol=1e-8)
    s, t = g["src"].clone().requires_grad_(True), g["tgt"].clone().requires_grad_(True)
    lc = gast.coral_loss(s, t)
    lc.backward()
    torch.testing.assert_close(lc.detach(), g["coral"], rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(s.grad, g["coral_gsrc"], rtol=1e-4, atol=1e-9)
    torch.testing.assert_close(t.grad, g["coral_gtgt"], rtol=1e-4, atol=1e-9)


def test_superpixel_edge_shrinking_golden():
    """oracle restatement == the reference's own edge_shrinking (fixture: tests/golden/make_golden_superpixels.py)."""
    from oracle import gast
    g = load_golden("superpixel_shrink")
    for k in ("a", "b", "c"):
        got = gast.edge_shrinking(g[f"in_{k}"].numpy(), 3, 16)
        assert (got == g[f"out_{k}"].numpy()).all(), k


# ---------------- 7 classes: the reference's LoveDA configuration (tests/golden/make_golden_c7.py) ------------------------------
def test_c7_ops_golden():
    """The oracle at C = 7 against the reference: label_refine in every mode, pseudo_selection, DownscaleLabel, update_prototype,
    Pearson distance, CE / UVEM with gradients, ClassBalance, PrototypeContrastiveLoss."""
    C7 = 7
    g = load_golden("ops_c7")
    for m, h in zip(g["masks"], g["hards"]):
        assert torch.equal(gast.pseudo_selection(m, 0.8, 0.6, -1), h)
    for mode in ("all", "s", "p", "l"):
        out = gast.label_refine(g["sup"], g["feat"], [g["p1"], g["p2"]], g["soft"], g["protos"], True, mode, 2.0)
        torch.testing.assert_close(out, g["refine_" + mode], rtol=1e-5, atol=2e-7)
    out = gast.label_refine(g["sup_irregular"], g["feat"], [g["p1"], g["p2"]], g["soft"], g["protos"], True, "all", 2.0)
    torch.testing.assert_close(out, g["refine_all_irregular"], rtol=1e-5, atol=2e-7)
    torch.testing.assert_close(gast.pearson_dist(g["pearson_x"], g["protos"]), g["pearson_dist"], rtol=1e-5, atol=1e-6)
    assert torch.equal(gast.downscale_label(g["ds_label"], C7), g["ds_out"])
    new, ds = gast.update_prototype(g["up_feat"], g["up_label"], g["protos"], C7, 0.996)
    assert torch.equal(ds, g["up_label_ds"])
    torch.testing.assert_close(new, g["up_protos_out"], rtol=1e-6, atol=1e-7)
    l1, l2 = g["logits1"].clone().requires_grad_(True), g["logits2"].clone().requires_grad_(True)
    loss = gast.loss_calc_uvem([l1, l2], g["loss_hard"], g["loss_soft"], 0.2, 0.7, 4.0, -1, C7)
    loss.backward()
    torch.testing.assert_close(loss.detach(), g["uvem"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(l1.grad, g["uvem_g1"], rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(l2.grad, g["uvem_g2"], rtol=1e-4, atol=1e-8)
    l3, l4 = g["logits1"].clone().requires_grad_(True), g["logits2"].clone().requires_grad_(True)
    ce = gast.loss_calc([l3, l4], g["label_s"], -1)
    ce.backward()
    torch.testing.assert_close(ce.detach(), g["ce"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(l3.grad, g["ce_g1"], rtol=1e-4, atol=1e-8)
    cb = gast.ClassBalance(C7, -1, 0.99, 0.5)
    torch.testing.assert_close(cb.get_class_weight_4pixel(g["label_s"]), g["cb_weights"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(cb.freq, g["cb_freq"], rtol=1e-6, atol=1e-8)
    f = g["pcl_feat"].clone().requires_grad_(True)
    lp = gast.pcl_loss(g["pcl_protos"], f, g["pcl_labels"], 8.0, -1)
    lp.backward()
    torch.testing.assert_close(lp.detach(), g["pcl"], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(f.grad, g["pcl_gfeat"], rtol=1e-5, atol=1e-8)


def test_c7_full_model_ppm_ssl_step():
    """One train_ssl_uem step of R50-PPM with num_classes = 7 (reference configs/st/uemda/2urban.py:11, train_ssl_uem.py:80)."""
    C7 = 7
    g = load_golden("model_ppm_r50_b2_256_c7")
    sd = det_state_dict("resnet50", C7, True, seed=2333)
    assert tuple(sd["layer6.conv_last.4.weight"].shape) == (7, 512, 1, 1)
    model = OracleDeeplabv2(sd, "resnet50", C7, True)
    batch = synth.make_batch(B=2, H=256, W=256, C=C7, k=2048, seed=2333)
    opt = SGDState(model.parameters(), HYPER["momentum"], HYPER["weight_decay"])
    out = ssl_step(model, opt, batch["prototypes"], batch, float(g["lr"]), HYPER, dropout=False, n_classes=C7)
    for k in ("pred_s1", "pred_s2", "pred_t1", "pred_t2"):
        torch.testing.assert_close(out[k], g[k], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(out["label_t_soft"][:, :, ::4, ::4], g["soft_sample"], rtol=1e-4, atol=1e-6)
    assert (out["label_t_hard"] == g["hard"].long()).float().mean().item() >= 0.9999
    torch.testing.assert_close(out["loss_source"], g["loss_source"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(out["loss_target"], g["loss_target"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(out["prototypes"], g["prototypes"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(out["grad_norm"], g["grad_norm"], rtol=1e-3, atol=1e-5)


# ---------------- round 4: prototype initialisation, offline pseudo labels, evaluation, the 512x512 operating point ------------
CLOSED_FORM_W = [[0.5, 1.0, 0.0], [-1.0, 0.0, 1.0], [0.25, 0.25, 0.25], [0.0, -0.75, 0.5], [1.0, -1.0, 0.3], [-0.2, 0.6, -0.9]]


def closed_form_model(x):
    """the deterministic stand-in network of the inference fixtures (tests/golden/make_golden_r4.py ClosedFormModel)"""
    w = torch.tensor(CLOSED_FORM_W, dtype=x.dtype, device=x.device)
    return torch.softmax(torch.einsum("kc,bchw->bkhw", w, x), dim=1)


def test_aligner_update_avg_init_avg_golden():
    """Aligner.update_avg x2 + init_avg (reference uemda/gast/alignment.py:107-126, tools/init_prototypes.py:101-111)"""
    g = load_golden("aligner_avg")
    s, n = torch.zeros(C, 64), torch.zeros(C, 1)
    for feat, lab in zip(g["feats"], g["labels"]):
        s, n = gast.update_avg(feat, lab, s, n, C)
    torch.testing.assert_close(s, g["data_sum"], rtol=1e-5, atol=1e-5)
    assert torch.equal(n, g["data_cnt"])
    assert float(n[5]) == 0.0 and float(n[4]) > 0.0                      # an empty class, and one only the second batch sees
    protos = gast.init_avg(s, n)
    torch.testing.assert_close(protos, g["prototypes"], rtol=1e-5, atol=1e-6)
    assert float(protos[5].abs().max()) == 0.0


def test_gener_target_pseudo_pt_tensor_golden():
    """the `<fname>.pt` tensor of gener_target_pseudo(save_prob=True) (reference uemda/gast/pseudo_generation.py:128-136) and the
    sliding-window map of a two-window image"""
    from oracle import infer
    g = load_golden("gener_pseudo")
    assert np.allclose(g["model_w"].numpy(), np.array(CLOSED_FORM_W, dtype=np.float32))
    out = infer.pseudo_prob_map(closed_form_model, g["image"], (64, 96), C, slide=False)
    assert out.shape == g["pt_file"].shape == (C, 64, 96)
    torch.testing.assert_close(out, g["pt_file"], rtol=1e-6, atol=1e-7)
    slide = infer.pre_slide(closed_form_model, g["image2"], num_classes=C, tile_size=(32, 32), tta=False)
    torch.testing.assert_close(slide, g["slide2"], rtol=1e-6, atol=1e-7)


def test_evaluate_feeds_the_metric_golden():
    """what `evaluate` hands the metric (reference uemda/utils/eval.py:39-47): argmax of the sliding-window map on the labelled pixels"""
    from oracle import infer
    g = load_golden("evaluate_pairs")
    cm = np.zeros((C, C), dtype=np.int64)
    for k in range(int(g["n_images"])):
        yt, yp = infer.evaluate_pairs(closed_form_model, g[f"image{k}"], g[f"label{k}"], C, slide=True, tta=False, tile_size=(32, 32))
        assert np.array_equal(yt, g[f"y_true{k}"].numpy()) and np.array_equal(yp, g[f"y_pred{k}"].numpy())
        cm += infer.confusion(infer.pre_slide(closed_form_model, g[f"image{k}"], C, (32, 32)), g[f"label{k}"], C)
    assert np.array_equal(cm, g["confusion"].numpy())


@pytest.mark.parametrize("name,B,S,rtype,stride", [("model_aspp_r50_b8_512", 8, 512, "resnet50", 16),
                                                   ("model_aspp_r101_b2_256", 2, 256, "resnet101", 4)])
def test_full_model_ssl_step_at(name, B, S, rtype, stride):
    """One train_ssl_uem step of R50-ASPP at the reference's own batch (8 + 8, configs/ToPotsdam.py:58) and the benchmark's tile
    (512 x 512), and of ResNet101-ASPP (BASELINE config 5's model family): the oracle against the reference's outputs."""
    g = load_golden(name)
    model = OracleDeeplabv2(det_state_dict(rtype, C, False, seed=2333), rtype, C, False)
    batch = synth.make_batch(B=B, H=S, W=S, C=C, k=2048, seed=2333)
    opt = SGDState(model.parameters(), HYPER["momentum"], HYPER["weight_decay"])
    out = ssl_step(model, opt, batch["prototypes"], batch, float(g["lr"]), HYPER, dropout=False)
    for k in ("pred_s1", "pred_s2", "pred_t1", "pred_t2"):
        err = float((out[k] - g[k]).abs().max() / g[k].abs().max())
        assert err < 1e-3, (k, err)
    torch.testing.assert_close(out["label_t_soft"][:, :, ::stride, ::stride], g["soft_sample"], rtol=1e-3, atol=1e-5)
    assert (out["label_t_hard"] == g["hard"].long()).float().mean().item() >= 0.9995
    torch.testing.assert_close(out["loss_source"], g["loss_source"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(out["loss_target"], g["loss_target"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(out["prototypes"], g["prototypes"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(out["grad_norm"], g["grad_norm"], rtol=2e-3, atol=1e-5)
    torch.testing.assert_close(model.p["encoder.resnet.bn1.running_mean"], g["post_bn1_running_mean"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(model.p["encoder.resnet.layer4.2.bn3.running_var"], g["post_l4_bn3_running_var"], rtol=1e-4, atol=1e-6)


VARIANTS = {"single_aspp": dict(multi_layer=False, cascade=False, use_ppm=False),
            "single_ppm": dict(multi_layer=False, cascade=False, use_ppm=True),
            "cascade_aspp": dict(multi_layer=True, cascade=True, use_ppm=False)}


def variant_loss(outs):
    """the seeded quadratic loss of tests/golden/make_golden_r4.py model_variants"""
    loss = 0.0
    for k, o in enumerate(outs):
        r = torch.randn(o.shape, generator=torch.Generator().manual_seed(700 + k)).to(o.device)
        loss = loss + (o * r).sum() / o.numel() ** 0.5
    return loss


@pytest.mark.parametrize("tag", list(VARIANTS))
def test_deeplabv2_variants_golden(tag):
    """The branches of Deeplabv2 no UemDA script builds (reference uemda/models/Encoder.py:93-102,111-116,129-143,156-165): the
    single-head default and the cascade branch, forward in both modes and gradients, against the reference's own outputs."""
    g = load_golden("model_variants")
    v = VARIANTS[tag]
    sd = det_state_dict("resnet50", C, v["use_ppm"], seed=2333, multi_layer=v["multi_layer"], cascade=v["cascade"])
    model = OracleDeeplabv2(sd, "resnet50", C, v["use_ppm"], multi_layer=v["multi_layer"], cascade=v["cascade"])
    model.eval()
    with torch.no_grad():
        prob = model(g["image"])
    torch.testing.assert_close(prob[:, :, ::4, ::4], g[f"{tag}:prob_sample"], rtol=1e-4, atol=1e-6)
    model.train()
    outs = model(g["image"])
    assert len(outs) == (4 if v["cascade"] else 2)
    for k, o in enumerate(outs):
        ref = g[f"{tag}:out{k}"]
        got = o.detach() if o.shape[1] == C else o.detach().reshape(-1)[:: max(1, o.numel() // 4096)][:4096]
        torch.testing.assert_close(got, ref, rtol=1e-3, atol=1e-4)
    loss = variant_loss(outs)
    loss.backward()
    torch.testing.assert_close(loss.detach(), g[f"{tag}:loss"], rtol=1e-4, atol=1e-4)
    named = dict(model.named_parameters())
    for k, ref in g.items():
        if k.startswith(f"{tag}:grad:"):
            gr = named[k.split(":", 2)[2]].grad
            got = gr if gr.numel() <= 8192 else gr.reshape(-1)[:: max(1, gr.numel() // 4096)][:4096]
            err = float((got - ref).norm() / (ref.norm() + 1e-12))
            assert err < 2e-2, (k, err)
