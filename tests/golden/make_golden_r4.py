#!/usr/bin/env python3
"""Round-4 golden vectors, produced by running the REFERENCE itself (imported from /root/reference under the stubs of
make_golden.py, CPU only, build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r4.py [step512] [r101] [avg] [pseudo] [evaluate] [variants]

  model_aspp_r50_b8_512.npz   one tools/train_ssl_uem.py iteration of R50-ASPP at the reference's own operating point -- 8 source
                              + 8 target tiles (configs/ToPotsdam.py:58, configs/st/uemda/2potsdam.py:31,43) of 512x512, the
                              benchmark's tile: logits of both heads and both domains, sampled features / soft labels, the hard
                              pseudo-labels, losses, prototypes, gradient norm, the first-step update of EVERY parameter tensor
                              (256 strided samples) with its fp32 noise floor (the same step under the other CPU conv backend and
                              with the images moved by one unit in the last place), as model_aspp_r50_b2_256.npz holds them
  model_aspp_r101_b2_256.npz  the same step for ResNet-101 (BASELINE config 5's model family; configs/st/uemda/2potsdam.py:6), B = 2, 256 x 256
  aligner_avg.npz             Aligner.update_avg x2 + init_avg (uemda/gast/alignment.py:107-126): the --ckpt-proto prototypes
  gener_pseudo.npz            gener_target_pseudo (uemda/gast/pseudo_generation.py:96-155) through a closed-form model: the
                              `<fname>.pt` wire format (slide=False: model -> bilinear align_corners=True resize -> (C,H,W) fp32),
                              and the sliding-window probability map of a two-window image (pre_slide, tta=False); the TTA leg
                              needs `ttach`, which the tree does not hold
  model_variants.npz          the Deeplabv2 branches no UemDA script builds but the class offers (uemda/models/Encoder.py:93-102,111-116,
                              129-143,156-165): the single-head default (multi_layer=False; ASPP and PPM) and the cascade branch
                              (layer5 on the layer3 output): training-mode outputs, eval-mode probabilities, gradients of a seeded
                              quadratic loss for a few tensors, B = 2 tiles of 128 x 128
  evaluate_pairs.npz          uemda/utils/eval.py:14-56 `evaluate` through a closed-form model and a recording stand-in for
                              ever's PixelMetric (absent from the tree): what the function FEEDS the metric -- argmax over the
                              sliding-window map, pixels with label >= 0 -- as confusion counts; the IoU / F1 formulas live in
                              `ever` and stay unpinned
Inputs that the tests regenerate from seeds are not stored.  Fixtures are data; no reference source or bytecode is copied.
"""
import logging
import os
import sys
import tempfile
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                     # noqa: E402  (stubs, save(), model_cfg())

C = 6


class ClosedFormModel(nn.Module):
    """A deterministic stand-in for the network in the inference fixtures: class scores are fixed linear maps of the image's
    channels followed by a softmax, elementwise per pixel, so that windowing / resizing / argmax are what the fixtures pin.  The
    tests build the same function from the same constants (tests/test_oracle_golden.py, tests/test_gpu_infer.py)."""
    W = [[0.5, 1.0, 0.0], [-1.0, 0.0, 1.0], [0.25, 0.25, 0.25], [0.0, -0.75, 0.5], [1.0, -1.0, 0.3], [-0.2, 0.6, -0.9]]

    def forward(self, x):
        w = torch.tensor(self.W, dtype=x.dtype, device=x.device)
        return torch.softmax(torch.einsum("kc,bchw->bkhw", w, x), dim=1)


def step512(ref, logger, B=8, S=512, rtype="resnet50", name="model_aspp_r50_b8_512", soft_stride=16):
    from oracle import synth
    from oracle.step import HYPER
    from oracle.weights import det_state_dict

    def run(mkldnn=True, ulp_noise=False):
        torch.backends.mkldnn.enabled = mkldnn
        sd = det_state_dict(rtype, C, False, seed=2333)
        model = ref.Encoder.Deeplabv2(mg.model_cfg(False, C, rtype))
        model.load_state_dict(sd, strict=True)
        batch = synth.make_batch(B=B, H=S, W=S, C=C, k=2048, seed=2333)
        if ulp_noise:
            gn = torch.Generator().manual_seed(77)
            for k in ("images_s", "images_t"):
                sgn = torch.randint(0, 2, batch[k].shape, generator=gn).float() * 2 - 1
                batch[k] = batch[k] * (1.0 + sgn * 2.0 ** -23)
        model.train()
        al = ref.alignment.Aligner(logger, feat_channels=2048, class_num=C, ignore_label=-1, decay=HYPER["proto_decay"])
        al.prototypes = batch["prototypes"].clone()
        opt = torch.optim.SGD(model.parameters(), lr=1e-2, momentum=0.9, weight_decay=5e-4)
        ce = ref.balance.CrossEntropy(ignore_label=-1)
        uv = ref.balance.UVEMLoss(m=0.2, threshold=0.7, gamma=4, class_num=C, ignore_label=-1)
        lr = 3e-3
        opt.param_groups[0]["lr"] = lr
        ps1, ps2, feat_s = model(batch["images_s"])
        pt1, pt2, feat_t = model(batch["images_t"])
        soft = al.label_refine(batch["label_t_sup"], feat_t, [pt1, pt2], batch["label_t_soft"], True, "all", 2.0)
        hard = ref.pg.pseudo_selection(soft, 0.8, 0.6, "tensor", -1)
        al.update_prototype(feat_s, batch["label_s"])
        loss_s = ref.tools.loss_calc([ps1, ps2], batch["label_s"], ce, multi=True)
        loss_t = ref.balance.loss_calc_uvem([pt1, pt2], hard, soft, uv, multi=True)
        opt.zero_grad()
        (loss_s + loss_t).backward()
        named = dict(model.named_parameters())
        gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=32, norm_type=2)
        upd = {n: (-lr * (p.grad.detach().double() + 5e-4 * p.detach().double())).reshape(-1)[:: max(1, p.numel() // 256)][:256].clone()
               for n, p in named.items()}
        opt.step()
        post = model.state_dict()
        torch.backends.mkldnn.enabled = True
        out = dict(ps1=ps1.detach(), ps2=ps2.detach(), pt1=pt1.detach(), pt2=pt2.detach(), feat_s=feat_s.detach(), feat_t=feat_t.detach(),
                   soft=soft.detach(), hard=hard, loss_s=loss_s.detach(), loss_t=loss_t.detach(), protos=al.prototypes.clone(),
                   gnorm=gnorm, lr=lr, upd=upd,
                   bn1_rm=post["encoder.resnet.bn1.running_mean"].clone(), bn1_rv=post["encoder.resnet.bn1.running_var"].clone(),
                   l4_rv=post["encoder.resnet.layer4.2.bn3.running_var"].clone())
        del model, opt
        return out

    print(f"G-step {rtype}-ASPP B={B}+{B} {S}x{S} (three reference steps: plain, other conv backend, inputs moved by 1 ulp)", flush=True)
    r = run()
    print("  plain step done", flush=True)
    r3 = run(ulp_noise=True)
    print("  ulp-noise step done", flush=True)
    r2 = run(mkldnn=False)
    print("  other-backend step done", flush=True)
    names = list(r["upd"].keys())
    noise = np.array([max(float((r["upd"][n] - q["upd"][n]).norm() / (r["upd"][n].norm() + 1e-30)) for q in (r2, r3)) for n in names])
    # how far the reference's own forward moves under the same two perturbations (the forward's noise floor at this size)
    logit_floor = max(float((r["pt1"] - q["pt1"]).abs().max() / r["pt1"].abs().max()) for q in (r2, r3))
    hard_floor = min(float((r["hard"] == q["hard"]).float().mean()) for q in (r2, r3))
    feat_t = r["feat_t"]
    idx = torch.from_numpy(np.random.default_rng(3).integers(0, feat_t.numel(), 4096))
    # ... and the pooled features the mining reads (absolute: they are O(1) after the last ReLU)
    feat_floor = max(float((r["feat_t"].reshape(-1)[idx] - q["feat_t"].reshape(-1)[idx]).abs().max()) for q in (r2, r3))
    mg.save(name, pred_s1=r["ps1"], pred_s2=r["ps2"], pred_t1=r["pt1"], pred_t2=r["pt2"], feat_idx=idx,
            feat_t_sample=feat_t.reshape(-1)[idx], feat_s_sample=r["feat_s"].reshape(-1)[idx],
            feat_t_chmean=feat_t.mean(dim=(0, 2, 3)), feat_t_chvar=feat_t.var(dim=(0, 2, 3)),
            soft_sample=r["soft"][:, :, ::soft_stride, ::soft_stride], hard=r["hard"].to(torch.int8), loss_source=r["loss_s"], loss_target=r["loss_t"],
            prototypes=r["protos"], grad_norm=r["gnorm"], lr=r["lr"], post_bn1_running_mean=r["bn1_rm"],
            post_bn1_running_var=r["bn1_rv"], post_l4_bn3_running_var=r["l4_rv"],
            upd_names=np.array(names), upd_offsets=np.cumsum([0] + [r["upd"][n].numel() for n in names]),
            upd_samples=torch.cat([r["upd"][n] for n in names]), upd_noise_floor=noise,
            ref_logit_floor=np.array(logit_floor), ref_hard_agreement_floor=np.array(hard_floor), ref_feat_floor=np.array(feat_floor))
    print(f"  reference against itself: logits {logit_floor:.2e}, features {feat_floor:.2e}, hard labels {hard_floor:.6f}, median update floor "
          f"{np.median(noise):.3e}")


def aligner_avg(ref, logger):
    from oracle import synth
    print("Aligner.update_avg x2 + init_avg")
    g = torch.Generator().manual_seed(404)
    al = ref.alignment.Aligner(logger, feat_channels=64, class_num=C, ignore_label=-1, decay=0.996)
    feats, labels = [], []
    for i in range(2):
        lab = synth.make_batch(B=2, H=64, W=64, C=C, k=8, seed=31 + i)["label_s"]
        if i == 0:
            lab[lab == 4] = 1                           # class 4 is seen by the second batch only
        lab[lab == 5] = 0                               # class 5 stays empty: prototype 0 / (0 + eps)
        feat = torch.randn(2, 64, 4, 4, generator=g) * (1.0 + i) + 0.25 * i
        al.update_avg(feat, lab.clone())
        feats.append(feat)
        labels.append(lab)
    al.init_avg()
    mg.save("aligner_avg", feats=torch.stack(feats), labels=torch.stack(labels), data_sum=al._data_sum, data_cnt=al._data_cnt,
            prototypes=al.prototypes)


def gener_pseudo(ref, logger):
    print("gener_target_pseudo (.pt wire format) + two-window pre_slide")
    model = ClosedFormModel()
    g = torch.Generator().manual_seed(909)
    img = torch.randn(1, 3, 40, 56, generator=g)
    cfg = types.SimpleNamespace(DATASETS="IsprsDA", SNAPSHOT_DIR=None, CUTOFF_TOP=0.8, CUTOFF_LOW=0.6, PSEUDO_SELECT=True)
    with tempfile.TemporaryDirectory() as tmp:
        out_dir = os.path.join(tmp, "pseudo")
        ref.pg.gener_target_pseudo(cfg, model, [(img, {"fname": ["tile_a"]})], out_dir, slide=False, save_prob=True, size=(64, 96))
        files = sorted(os.listdir(out_dir))
        assert files == ["tile_a.pt"], files
        saved = torch.load(os.path.join(out_dir, "tile_a.pt"))
    assert saved.shape == (C, 64, 96) and saved.dtype == torch.float32
    # the sliding-window map of an image two windows wide (tile 32, stride 16: 1 x 2 windows after the reference's re-alignment)
    img2 = torch.randn(1, 3, 32, 44, generator=g)
    slide = ref.tools.pre_slide(model, img2, num_classes=C, tile_size=(32, 32), tta=False)
    mg.save("gener_pseudo", image=img, pt_file=saved, model_w=np.array(ClosedFormModel.W, dtype=np.float32), image2=img2, slide2=slide)


def evaluate_pairs(ref, logger):
    print("evaluate: what it feeds the metric")
    import importlib

    class _Stop(Exception):
        pass

    class _Total:
        def toarray(self):
            raise _Stop()

    class RecordingPixelMetric:
        """stand-in for ever.api.metric.pixel.PixelMetric (absent from the tree): records every forward() call"""
        calls = []

        def __init__(self, num_classes, logdir=None, logger=None, class_names=None):
            self.num_classes, self._class_names, self._total = num_classes, class_names, _Total()

        def forward(self, y_true, y_pred):
            RecordingPixelMetric.calls.append((np.asarray(y_true).copy(), np.asarray(y_pred).copy()))

    pixel = types.ModuleType("ever.api.metric.pixel")
    pixel.PixelMetric = RecordingPixelMetric
    sys.modules["ever.api.metric.pixel"] = pixel
    sys.modules["ever.util.param_util"].count_model_parameters = lambda *a, **k: None
    # daLoader.py:18 derives its loader from ever's ConfigurableMixin (config plumbing, no arithmetic); the loader itself is
    # replaced below by the fixture's two (image, label) pairs
    sys.modules["ever.interface"].ConfigurableMixin = type("ConfigurableMixin", (), {"__init__": lambda self, config=None: None})
    ev = importlib.import_module("uemda.utils.eval")
    model = ClosedFormModel()
    g = torch.Generator().manual_seed(1212)
    imgs = [torch.randn(1, 3, 32, 44, generator=g), torch.randn(1, 3, 48, 32, generator=g)]
    gts = [torch.randint(-1, C, (1,) + tuple(i.shape[2:]), generator=g) for i in imgs]
    loader = [(im, {"cls": gt, "fname": [f"t{k}.tif"]}) for k, (im, gt) in enumerate(zip(imgs, gts))]
    ev.DALoader = lambda *a, **k: loader
    # pre_slide's default window is 512: the function under test takes no tile size, so the fixture's images go through one
    # re-aligned window each only if they are padded -- instead run it at the module's tile size with images that are one or
    # two windows of 32 (the reference's pre_slide is patched in its default argument only)
    orig = ev.pre_slide
    ev.pre_slide = lambda m, x, num_classes, tta=False: orig(m, x, num_classes=num_classes, tile_size=(32, 32), tta=tta)
    with tempfile.TemporaryDirectory() as tmp:
        cfg = types.SimpleNamespace(DATASETS="IsprsDA", SNAPSHOT_DIR=tmp, EVAL_DATA_CONFIG=None, TEST_DATA_CONFIG=None)
        try:
            ev.evaluate(model, cfg, is_training=True, ckpt_path=os.path.join(tmp, "none.pth"), logger=logger, slide=True, tta=False)
            raise AssertionError("the recording metric should have stopped summary_all()")
        except _Stop:
            pass
    calls = RecordingPixelMetric.calls
    assert len(calls) == len(imgs)
    cm = np.zeros((C, C), dtype=np.int64)                 # counts of the recorded (y_true, y_pred) pairs: bookkeeping of the generator
    for yt, yp in calls:
        np.add.at(cm, (yt.astype(np.int64), yp.astype(np.int64)), 1)
    arrays = {f"image{k}": im for k, im in enumerate(imgs)}
    arrays.update({f"label{k}": gt for k, gt in enumerate(gts)})
    arrays.update({f"y_pred{k}": yp.astype(np.int8) for k, (_, yp) in enumerate(calls)})
    arrays.update({f"y_true{k}": yt.astype(np.int8) for k, (yt, _) in enumerate(calls)})
    mg.save("evaluate_pairs", confusion=cm, model_w=np.array(ClosedFormModel.W, dtype=np.float32), n_images=np.array(len(imgs)), **arrays)


VARIANTS = {"single_aspp": dict(multi_layer=False, cascade=False, use_ppm=False),
            "single_ppm": dict(multi_layer=False, cascade=False, use_ppm=True),
            "cascade_aspp": dict(multi_layer=True, cascade=True, use_ppm=False)}


def model_variants(ref, logger):
    from oracle.weights import det_state_dict
    print("Deeplabv2 variants: single head (ASPP, PPM), cascade")
    g = torch.Generator().manual_seed(606)
    x = torch.randn(2, 3, 128, 128, generator=g)
    arrays = dict(image=x)
    for tag, v in VARIANTS.items():
        sd = det_state_dict("resnet50", C, v["use_ppm"], seed=2333, multi_layer=v["multi_layer"], cascade=v["cascade"])
        cfg = mg.model_cfg(v["use_ppm"], C)
        cfg.update(multi_layer=v["multi_layer"], cascade=v["cascade"])
        model = ref.Encoder.Deeplabv2(cfg)
        model.load_state_dict(sd, strict=True)
        assert list(model.state_dict().keys()) == list(sd.keys()), tag
        if v["use_ppm"]:
            model.cls_pred.conv_last[3].p = 0.0                       # Dropout2d off for parity
        model.eval()
        with torch.no_grad():
            prob = model(x)
        model.train()
        outs = model(x)
        # a seeded quadratic loss over every output: sum_k <out_k, r_k> with fixed random r_k
        loss = 0.0
        for k, o in enumerate(outs):
            r = torch.randn(o.shape, generator=torch.Generator().manual_seed(700 + k))
            loss = loss + (o * r).sum() / o.numel() ** 0.5
        loss.backward()
        named = dict(model.named_parameters())
        head = "cls_pred" if not v["multi_layer"] else "layer5"
        gnames = ["encoder.resnet.conv1.weight", "encoder.resnet.layer3.0.conv1.weight", "encoder.resnet.layer4.2.bn3.bias",
                  f"{head}.conv_last.4.weight" if v["use_ppm"] else f"{head}.conv2d_list.2.weight"]
        arrays[f"{tag}:prob_sample"] = prob[:, :, ::4, ::4]
        arrays[f"{tag}:loss"] = loss.detach()
        for k, o in enumerate(outs):
            o = o.detach()
            arrays[f"{tag}:out{k}"] = o if o.shape[1] == C else o.reshape(-1)[:: max(1, o.numel() // 4096)][:4096]
        for n in gnames:
            gr = named[n].grad
            arrays[f"{tag}:grad:{n}"] = gr if gr.numel() <= 8192 else gr.reshape(-1)[:: max(1, gr.numel() // 4096)][:4096]
        arrays[f"{tag}:grad_norm"] = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters())).float()
    mg.save("model_variants", **arrays)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = mg.import_reference()
    logger = logging.getLogger("golden-r4")
    what = set(sys.argv[1:]) or {"avg", "pseudo", "evaluate", "variants", "step512", "r101"}
    if "avg" in what:
        aligner_avg(ref, logger)
    if "pseudo" in what:
        gener_pseudo(ref, logger)
    if "evaluate" in what:
        evaluate_pairs(ref, logger)
    if "variants" in what:
        model_variants(ref, logger)
    if "step512" in what:
        step512(ref, logger)
    if "r101" in what:                                   # BASELINE config 5's model family: ResNet-101 (23 blocks in layer3), B = 2, 256 x 256
        step512(ref, logger, B=2, S=256, rtype="resnet101", name="model_aspp_r101_b2_256", soft_stride=4)
    print("done")


if __name__ == "__main__":
    main()
