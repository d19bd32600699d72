#!/usr/bin/env python3
"""Golden-vector generator: runs the REFERENCE itself (imported from /root/reference under
third-party stubs, CPU only) on seeded inputs and stores inputs/expected outputs as small .npz
fixtures next to this script.  Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Neither reference source nor bytecode is copied; the fixtures are data.
The stubs below replace third-party packages the image lacks (`ever`, `torch_scatter`, cv2,
ttach, skimage, albumentations, prettytable, torchvision); they carry no arithmetic except
`torch_scatter.scatter`, restated from its documented semantics.
"""
import os
import sys
import types
import logging

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("UEMDA_REFERENCE", "/root/reference")


# ------------------------------------------------------------------------------------------------
# stubs
# ------------------------------------------------------------------------------------------------
class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def update(self, other=(), **kw):
        for k, v in dict(other, **kw).items():
            if isinstance(v, dict):
                cur = self.get(k)
                if isinstance(cur, _AttrDict):
                    cur.update(v)
                else:
                    nd = _AttrDict()
                    nd.update(v)
                    dict.__setitem__(self, k, nd)
            else:
                dict.__setitem__(self, k, v)


class _ERModule(nn.Module):
    def __init__(self, config=None):
        super().__init__()
        self.__dict__["config"] = _AttrDict()
        self.set_default_config()
        if config:
            self.config.update(config)

    def set_default_config(self):
        pass


class _Registry(dict):
    def register(self, name=None, obj=None):
        if obj is not None:
            self[name] = obj
            return obj

        def deco(o):
            self[name or o.__name__] = o
            return o
        return deco


class _Catch(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        sub = _Catch(self.__name__ + "." + name)
        setattr(self, name, sub)
        return sub

    def __call__(self, *a, **k):
        return None


def _scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    index = index.expand_as(src)
    n = int(index.max()) + 1 if dim_size is None else dim_size
    shape = list(src.shape)
    shape[dim] = n
    res = torch.zeros(shape, dtype=src.dtype)
    red = {"max": "amax", "min": "amin", "sum": "sum", "add": "sum", "mean": "mean"}[reduce]
    return res.scatter_reduce(dim, index, src, reduce=red, include_self=False)


def install_stubs():
    ever = types.ModuleType("ever")
    ever.ERModule = _ERModule
    interface = types.ModuleType("ever.interface")
    interface.ERModule = _ERModule
    core = types.ModuleType("ever.core")
    registry = types.ModuleType("ever.core.registry")
    registry.MODEL = _Registry()
    logger_m = types.ModuleType("ever.core.logger")
    logger_m.get_logger = lambda *a, **k: logging.getLogger("ever-stub")
    util = types.ModuleType("ever.util")
    param_util = types.ModuleType("ever.util.param_util")

    def freeze_params(m):
        for p in m.parameters():
            p.requires_grad = False

    def freeze_modules(m, cls):
        for mod in m.modules():
            if isinstance(mod, cls):
                freeze_params(mod)
    param_util.freeze_params, param_util.freeze_modules = freeze_params, freeze_modules
    core.registry, core.logger = registry, logger_m
    util.param_util = param_util
    ever.interface, ever.core, ever.util = interface, core, util
    mods = {"ever": ever, "ever.interface": interface, "ever.core": core,
            "ever.core.registry": registry, "ever.core.logger": logger_m, "ever.util": util,
            "ever.util.param_util": param_util}
    ts = types.ModuleType("torch_scatter")
    ts.scatter = _scatter
    mods["torch_scatter"] = ts
    for name in ("cv2", "ttach", "skimage", "skimage.io", "albumentations", "albumentations.pytorch",
                 "prettytable", "torchvision", "torchvision.transforms", "ever.core.iterator",
                 "ever.api", "ever.api.metric", "ever.api.data", "matplotlib", "matplotlib.pyplot",
                 "tqdm", "pandas_stub"):
        if name in ("matplotlib", "matplotlib.pyplot", "tqdm"):
            try:
                __import__(name)
                continue
            except Exception:
                pass
        mods[name] = _Catch(name)
    for k, v in mods.items():
        sys.modules.setdefault(k, v)
    # constructors hard-code .cuda() (alignment.py:48,56,60,76-77; balance.py:25)
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)


def import_reference():
    install_stubs()
    import importlib
    ref = types.SimpleNamespace()
    ref.resnets = importlib.import_module("uemda._resnets")
    ref.balance = importlib.import_module("uemda.gast.balance")
    ref.Encoder = importlib.import_module("uemda.models.Encoder")
    # pseudo_generation star-imports datasets/viz; pull only what the path needs
    try:
        ref.pg = importlib.import_module("uemda.gast.pseudo_generation")
        ref.alignment = importlib.import_module("uemda.gast.alignment")
        ref.tools = importlib.import_module("uemda.utils.tools")
    except Exception as e:                                              # pragma: no cover
        raise RuntimeError(f"reference import failed under stubs: {e!r}")
    return ref


# ------------------------------------------------------------------------------------------------
def _np(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (_np(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()})
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KB)")


def model_cfg(use_ppm, C=6, rtype="resnet50"):
    return dict(backbone=dict(resnet_type=rtype, output_stride=16, pretrained=False),
                multi_layer=True, cascade=False, use_ppm=use_ppm,
                ppm=dict(num_classes=C, use_aux=False, fc_dim=2048),
                inchannels=2048, num_classes=C, is_ins_norm=True)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = import_reference()
    from oracle import synth
    from oracle.weights import det_state_dict, checksum
    from oracle.step import HYPER
    logger = logging.getLogger("golden")
    C = 6

    # ---------------- G-ops ---------------------------------------------------------------------
    print("G-ops")
    g = torch.Generator().manual_seed(11)
    # pseudo_selection: peaked / uniform / tie / all-below-0.6
    peaked = torch.softmax(6 * torch.randn(2, C, 64, 64, generator=g), 1)
    uniform = torch.full((2, C, 64, 64), 1.0 / C)
    tie = torch.zeros(2, C, 64, 64)
    tie[:, 0] = 0.5
    tie[:, 1] = 0.5
    tie[:, :, :8] = peaked[:, :, :8]
    low = torch.softmax(0.5 * torch.randn(2, C, 64, 64, generator=g), 1)
    masks = torch.stack([peaked, uniform, tie, low])
    hards = torch.stack([ref.pg.pseudo_selection(m.clone(), 0.8, 0.6, "tensor", -1) for m in masks])
    save("pseudo_selection", masks=masks, hards=hards)

    # label_refine, small: B2 C6 k64 h=w=4 H=W=64
    small = synth.make_batch(B=2, H=64, W=64, C=C, k=64, seed=5)
    feat = torch.randn(2, 64, 4, 4, generator=g)
    p1 = torch.randn(2, C, 4, 4, generator=g)
    p2 = torch.randn(2, C, 4, 4, generator=g)
    al = ref.alignment.Aligner(logger, feat_channels=64, class_num=C, ignore_label=-1, decay=0.996)
    al.prototypes = small["prototypes"].clone()
    outs = {}
    for mode in ("all", "s", "p", "l"):
        outs["out_" + mode] = al.label_refine(small["label_t_sup"], feat, [p1, p2],
                                              small["label_t_soft"].clone(), True, mode, 2.0)
    irr = synth.irregular_superpixels(2, 64, 64, 23, seed=7)
    outs["out_all_irregular"] = al.label_refine(irr, feat, [p1, p2], small["label_t_soft"].clone(), True, "all", 2.0)
    outs["out_single_pred"] = al.label_refine(small["label_t_sup"], feat, p1, small["label_t_soft"].clone(), True, "l", 1.5)
    save("label_refine", sup=small["label_t_sup"], sup_irregular=irr, feat=feat, p1=p1, p2=p2,
         soft=small["label_t_soft"], protos=small["prototypes"], **outs)

    # pearson
    x = torch.randn(37, 64, generator=g)
    pr = torch.randn(C, 64, generator=g)
    save("pearson", x=x, protos=pr, dist=al._pearson_dist(x, pr))

    # DownscaleLabel: random blobs + exact 192/256 boundary + ignore majority
    lab = synth.make_batch(B=2, H=64, W=64, C=C, k=8, seed=9)["label_s"]
    lab[0, :16, :16] = 2
    lab[0, :4, :16] = 3                 # 192/256 = 0.75 exactly -> kept
    lab[0, 16:32, :16] = 1
    lab[0, 16:21, :16] = 4              # 176/256 < 0.75 -> ignored
    lab[1, :16, :16] = -1               # ignore majority
    ds = ref.alignment.DownscaleLabel(16, C, -1, 0.75)(lab.clone())
    save("downscale_label", label=lab, out=ds)

    # update_prototype incl. an empty class
    lab2 = lab.clone()
    lab2[lab2 == 5] = 0                 # class 5 empty
    featp = torch.randn(2, 64, 4, 4, generator=g)
    al2 = ref.alignment.Aligner(logger, feat_channels=64, class_num=C, ignore_label=-1, decay=0.996)
    al2.prototypes = small["prototypes"].clone()
    ds2 = al2.update_prototype(featp, lab2.clone())
    save("update_prototype", feat=featp, label=lab2, protos_in=small["prototypes"], protos_out=al2.prototypes,
         label_ds=ds2)

    # losses
    logits = (2 * torch.randn(2, C, 4, 4, generator=g)).requires_grad_(True)
    logits2 = (2 * torch.randn(2, C, 4, 4, generator=g)).requires_grad_(True)
    soft_ref = outs["out_all"].detach()
    hard = ref.pg.pseudo_selection(soft_ref.clone(), 0.8, 0.6, "tensor", -1)
    uv = ref.balance.UVEMLoss(m=0.2, threshold=0.7, gamma=4, class_num=C, ignore_label=-1)
    lt = ref.balance.loss_calc_uvem([logits, logits2], hard, soft_ref, uv, multi=True)
    lt.backward()
    ce = ref.balance.CrossEntropy(ignore_label=-1)
    lg3 = logits.detach().clone().requires_grad_(True)
    lg4 = logits2.detach().clone().requires_grad_(True)
    ls = ref.tools.loss_calc([lg3, lg4], small["label_s"], ce, multi=True)
    ls.backward()
    u = torch.linspace(0, 1.0, 101)
    save("losses", logits1=logits, logits2=logits2, soft=soft_ref, hard=hard, uvem=lt, uvem_g1=logits.grad,
         uvem_g2=logits2.grad, label_s=small["label_s"], ce=ls, ce_g1=lg3.grad, ce_g2=lg4.grad,
         u=u, uvem_w=uv.get_weight(u))

    # ClassBalance 3 EMA steps
    cb = ref.balance.ClassBalance(class_num=C, ignore_label=-1, decay=0.99, temperature=2.0)
    labs = [synth.make_batch(B=1, H=32, W=32, C=C, k=8, seed=20 + i)["label_s"] for i in range(3)]
    ws = [cb.get_class_weight_4pixel(l) for l in labs]
    save("class_balance", labels=torch.stack(labs), weights=torch.stack(ws), freq=cb.freq)

    # LR schedule (train_ssl_uem.py:82-84 with STAGE3_STEPS=6000)
    cfg = types.SimpleNamespace(LEARNING_RATE=1e-2, NUM_STEPS=6000 * 1.5, PREHEAT_STEPS=int(6000 / 20), POWER=0.9)
    opt = types.SimpleNamespace(param_groups=[{"lr": 0}])
    its = [0, 1, 299, 300, 301, 3000, 5999]
    save("lr_schedule", iters=np.array(its), lrs=np.array([ref.tools.adjust_learning_rate(opt, i, cfg) for i in its]))

    # pre_slide with a closed-form "model" (elementwise, so that the windowing is what is pinned)
    def fake_model(x):
        return torch.stack([x[:, 0] * 0.5 + x[:, 1], x[:, 2] - x[:, 0], x.sum(1) * 0.25], dim=1)
    img = torch.randn(2, 3, 72, 88, generator=torch.Generator().manual_seed(4242))
    save("pre_slide", image=img, out=ref.tools.pre_slide(fake_model, img, num_classes=3, tile_size=(32, 32), tta=False),
         out_one_tile=ref.tools.pre_slide(fake_model, img[:, :, :32, :32], num_classes=3, tile_size=(32, 32), tta=False))

    # stage-2 alignment losses
    import importlib
    loss_mod, coral_mod = importlib.import_module("uemda.loss"), importlib.import_module("uemda.gast.coral")
    ga = torch.Generator().manual_seed(77)
    featp = (torch.randn(2, 128, 6, 5, generator=ga) * 1.5 + 0.3).requires_grad_(True)
    protosp = torch.randn(6, 128, generator=ga)
    labelsp = torch.randint(-1, 6, (2, 1, 6, 5), generator=ga)
    lp = loss_mod.PrototypeContrastiveLoss(temperature=8.0, ignore_label=-1)(protosp, featp, labelsp)
    (lp * 3.0).backward()
    srcp = (torch.randn(300, 128, generator=ga) + 0.5).requires_grad_(True)
    tgtp = (torch.randn(280, 128, generator=ga) * 1.3 - 0.2).requires_grad_(True)
    lcp = coral_mod.CoralLoss()(srcp, tgtp)
    lcp.backward()
    save("align_losses", feat=featp, protos=protosp, labels=labelsp, pcl=lp, pcl_gfeat_x3=featp.grad, src=srcp, tgt=tgtp,
         coral=lcp, coral_gsrc=srcp.grad, coral_gtgt=tgtp.grad)

    # ---------------- G-layers ------------------------------------------------------------------
    print("G-layers")
    from oracle.weights import _rng, fill_like

    def sub(t):
        return t if t.numel() <= 8192 else t.reshape(-1)[:: t.numel() // 4096][:4096]

    def run_layer(name, module, x, train=True):
        # weights are NOT stored: tests regenerate them with oracle.weights.fill_like(shapes, name)
        shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
        module.load_state_dict(fill_like(shapes, name))
        module.train(train)
        x = x.clone().requires_grad_(True)
        y = module(x)
        gy = torch.from_numpy(_rng(name + "gy", 2).normal(0, 1, tuple(y.shape)).astype(np.float32))
        y.backward(gy)
        arrays = dict(x=x, y=y, gy=gy, gx=x.grad)
        for k, p in module.named_parameters():
            arrays["g:" + k] = sub(p.grad)
        for k, v in module.state_dict().items():
            if "running" in k:
                arrays["post:" + k] = v
        save(name, **arrays)

    Bott = ref.resnets.Bottleneck
    ds_mod = nn.Sequential(nn.Conv2d(64, 128, 1, 2, bias=False), nn.BatchNorm2d(128))
    run_layer("layer_bottleneck_s2", Bott(64, 32, stride=2, downsample=ds_mod), torch.randn(2, 64, 16, 16, generator=g))
    run_layer("layer_bottleneck_d2", Bott(128, 32, dilation=2), torch.randn(2, 128, 8, 8, generator=g))
    run_layer("layer_aspp", ref.Encoder.Classifier_Module(32, [6, 12, 18, 24], [6, 12, 18, 24], C),
              torch.randn(2, 32, 16, 16, generator=g))
    ppm = ref.Encoder.PPMBilinear(num_classes=C, fc_dim=32)
    ppm.conv_last[3].p = 0.0            # Dropout2d off for parity (SURVEY §7 hard parts)
    run_layer("layer_ppm", ppm, torch.randn(2, 32, 16, 16, generator=g))
    run_layer("layer_instnorm", nn.InstanceNorm2d(32), torch.randn(2, 32, 8, 8, generator=g) * 3 + 1)
    stem = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(3, 2, 1))
    run_layer("layer_stem", stem, torch.randn(2, 3, 32, 32, generator=g))

    # ---------------- G-model / G-step ---------------------------------------------------------
    def ssl_step_reference(use_ppm, mkldnn=True, ulp_noise=False, backbone=None, sd=None):
        """One train_ssl_uem.py iteration of the REFERENCE on the seeded batch; returns everything the fixture stores.
        ulp_noise: the two image batches are perturbed by one unit in the last place (x * (1 +- 2^-23), seeded).
        backbone: extra ResNetEncoder options (freeze_at, batchnorm_trainable, with_cp; reference resnet.py:170-181)."""
        torch.backends.mkldnn.enabled = mkldnn
        sd = det_state_dict("resnet50", C, use_ppm, seed=2333) if sd is None else sd
        cfg = model_cfg(use_ppm, C)
        cfg["backbone"].update(backbone or {})
        model = ref.Encoder.Deeplabv2(cfg)
        model.load_state_dict(sd, strict=True)
        assert list(model.state_dict().keys()) == list(sd.keys()), "state_dict key order mismatch"
        if use_ppm:
            model.layer5.conv_last[3].p = 0.0
            model.layer6.conv_last[3].p = 0.0
        batch = synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=2333)
        if ulp_noise:
            gn = torch.Generator().manual_seed(77)
            for k in ("images_s", "images_t"):
                sgn = torch.randint(0, 2, batch[k].shape, generator=gn).float() * 2 - 1
                batch[k] = batch[k] * (1.0 + sgn * 2.0 ** -23)
        # eval forward
        model.eval()
        with torch.no_grad():
            prob = model(batch["images_t"])
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
        loss = loss_s + loss_t
        opt.zero_grad()
        loss.backward()
        gnames = ["encoder.resnet.conv1.weight", "encoder.resnet.layer1.0.conv2.weight",
                  "encoder.resnet.layer2.0.downsample.0.weight", "encoder.resnet.layer3.2.bn2.weight",
                  "encoder.resnet.layer4.1.conv2.weight", "encoder.resnet.layer4.2.bn3.bias"]
        gnames += (["layer5.conv_last.4.weight", "layer6.ppm.2.1.weight"] if use_ppm else
                   ["layer5.conv2d_list.0.weight", "layer6.conv2d_list.3.bias"])
        named = dict(model.named_parameters())
        grads = {}          # NB: views of .grad -> they hold the POST-clip values once clip ran
        for n in gnames:
            gr = named[n].grad
            if gr is None:                                   # frozen parameter
                continue
            grads["grad:" + n] = gr if gr.numel() <= 8192 else gr.reshape(-1)[:: max(1, gr.numel() // 4096)][:4096]
        frozen = [n for n, p in named.items() if p.grad is None]
        for n in frozen:
            assert not named[n].requires_grad, n
        gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=32, norm_type=2)
        # the UPDATE of every parameter tensor, 256 strided samples each: -lr * (clipped grad + wd * w), what SGD's first
        # step applies (momentum buffer = gradient), evaluated in float64 from the reference's own fp32 gradients -- the
        # difference w_post - w_pre itself is quantised to the weights' last place (8 % of a BatchNorm gamma's update)
        upd = {n: ((-lr * (p.grad.detach().double() + 5e-4 * p.detach().double())) if p.grad is not None else
                   torch.zeros_like(p, dtype=torch.float64)).reshape(-1)[:: max(1, p.numel() // 256)][:256].clone()
               for n, p in named.items()}              # a frozen parameter (grad None) is skipped by SGD: update 0
        opt.step()
        torch.backends.mkldnn.enabled = True
        return dict(model=model, prob=prob, ps1=ps1, ps2=ps2, pt1=pt1, pt2=pt2, feat_s=feat_s, feat_t=feat_t, soft=soft, hard=hard,
                    loss_s=loss_s, loss_t=loss_t, al=al, gnorm=gnorm, lr=lr, grads=grads, upd=upd, frozen=frozen, pre=sd)

    for use_ppm in (False, True):
        tag = "ppm" if use_ppm else "aspp"
        print("G-model", tag)
        r = ssl_step_reference(use_ppm)
        model, feat_t = r["model"], r["feat_t"]
        cs = checksum(model.parameters())
        post = model.state_dict()
        idx = torch.from_numpy(np.random.default_rng(3).integers(0, feat_t.numel(), 4096))
        # fp32 noise floor of every update: the same reference step with the other CPU conv backend (different summation
        # order flips ~1e-5 of the ReLU masks per layer); stored per tensor so that the test holds each tensor
        # to ITS floor instead of one global tolerance
        # (a) the other CPU conv backend (different summation order in the conv backward passes), (b) the input images
        # perturbed by one unit in the last place: the least any independent fp32 implementation differs from this one.
        # Both are amplified by the ~50 training-mode BatchNorm backward passes (B = 2) between a tensor and the loss.
        r2 = ssl_step_reference(use_ppm, mkldnn=False)
        r3 = ssl_step_reference(use_ppm, ulp_noise=True)
        names = list(r["upd"].keys())
        noise = np.array([max(float((r["upd"][n] - q["upd"][n]).norm() / (r["upd"][n].norm() + 1e-30)) for q in (r2, r3))
                          for n in names])
        upd_flat = torch.cat([r["upd"][n] for n in names])
        upd_off = np.cumsum([0] + [r["upd"][n].numel() for n in names])
        save(f"model_{tag}_r50_b2_256", eval_prob_sample=r["prob"][:, :, ::8, ::8],
             pred_s1=r["ps1"], pred_s2=r["ps2"], pred_t1=r["pt1"], pred_t2=r["pt2"], feat_idx=idx,
             feat_t_sample=feat_t.reshape(-1)[idx], feat_s_sample=r["feat_s"].reshape(-1)[idx],
             feat_t_chmean=feat_t.mean(dim=(0, 2, 3)), feat_t_chvar=feat_t.var(dim=(0, 2, 3)),
             soft_sample=r["soft"][:, :, ::4, ::4], hard=r["hard"].to(torch.int8), loss_source=r["loss_s"], loss_target=r["loss_t"],
             prototypes=r["al"].prototypes, grad_norm=r["gnorm"], lr=r["lr"], post_checksum=np.array(cs),
             post_bn1_running_mean=post["encoder.resnet.bn1.running_mean"],
             post_bn1_running_var=post["encoder.resnet.bn1.running_var"],
             post_l4_bn3_running_var=post["encoder.resnet.layer4.2.bn3.running_var"],
             post_conv1_sample=post["encoder.resnet.conv1.weight"].reshape(-1)[::7],
             nbt=post["encoder.resnet.bn1.num_batches_tracked"],
             upd_names=np.array(names), upd_offsets=upd_off, upd_samples=upd_flat, upd_noise_floor=noise, **r["grads"])
    # ---------------- G-step with the encoder's optional modes (resnet.py:112-130,146-165,183-190) ---------------
    # the same seeded step with (a) the stem and layer1 frozen and every encoder BatchNorm frozen in eval mode, (b) every
    # residual layer under torch.utils.checkpoint.  Stored: outputs, losses, gradient norm, per-tensor update samples with
    # their fp32 noise floor (as above), which tensors were frozen, and the BatchNorm buffers that tell the modes apart.
    bn_keys = ["encoder.resnet.bn1", "encoder.resnet.layer1.0.bn1", "encoder.resnet.layer2.0.downsample.1", "encoder.resnet.layer4.2.bn3"]
    def calibrated_state_dict():
        """Frozen BatchNorm needs running statistics that normalise (with the initial mean 0 / variance 1 an eval-mode
        ResNet lets its activations collapse and its backward pass cancels to 1e-4 of the incoming gradient, so every fp32
        summation order gives another answer).  Like pretrained weights would, the state dict for the frozen step carries
        the statistics of the data: the plain reference model, training mode, momentum=None (cumulative average), one forward
        over the source and the target batch."""
        sd0 = det_state_dict("resnet50", C, False, seed=2333)
        m = ref.Encoder.Deeplabv2(model_cfg(False, C))
        m.load_state_dict(sd0, strict=True)
        m.train()
        for mod in m.modules():
            if isinstance(mod, nn.BatchNorm2d):
                mod.momentum = None
        b = synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=2333)
        with torch.no_grad():
            m(b["images_s"])
            m(b["images_t"])
        out = {k: v.clone() for k, v in m.state_dict().items()}
        for k in out:
            if k.endswith("num_batches_tracked"):
                out[k].zero_()
        return out

    for tag, bb in (("frozen", dict(freeze_at=2, batchnorm_trainable=False)), ("cp", dict(with_cp=(True, True, True, True)))):
        print("G-step encoder options", tag, bb)
        sd_in = calibrated_state_dict() if tag == "frozen" else None
        r = ssl_step_reference(False, backbone=bb, sd=sd_in)
        r2 = ssl_step_reference(False, mkldnn=False, backbone=bb, sd=sd_in)
        r3 = ssl_step_reference(False, ulp_noise=True, backbone=bb, sd=sd_in)
        names = list(r["upd"].keys())
        noise = np.array([max(float((r["upd"][n] - q["upd"][n]).norm() / (r["upd"][n].norm() + 1e-30)) for q in (r2, r3))
                          for n in names])
        post = r["model"].state_dict()
        bufs = {}
        for k in bn_keys:
            for b in ("running_mean", "running_var", "num_batches_tracked"):
                bufs[f"post:{k}.{b}"] = post[f"{k}.{b}"]
        unchanged = all(torch.equal(post[n], r["pre"][n]) for n in r["frozen"])
        if sd_in is not None:                              # the calibrated running statistics the step started from
            rs = [k for k in sd_in if k.endswith("running_mean") or k.endswith("running_var")]
            bufs["init_stat_names"] = np.array(rs)
            bufs["init_stat_offsets"] = np.cumsum([0] + [sd_in[k].numel() for k in rs])
            bufs["init_stat_values"] = torch.cat([sd_in[k].reshape(-1) for k in rs])
        save(f"model_aspp_r50_b2_256_{tag}", pred_s1=r["ps1"], pred_s2=r["ps2"], pred_t1=r["pt1"], pred_t2=r["pt2"],
             hard=r["hard"].to(torch.int8), loss_source=r["loss_s"], loss_target=r["loss_t"], prototypes=r["al"].prototypes,
             grad_norm=r["gnorm"], lr=r["lr"], upd_names=np.array(names), upd_offsets=np.cumsum([0] + [r["upd"][n].numel() for n in names]),
             upd_samples=torch.cat([r["upd"][n] for n in names]), upd_noise_floor=noise,
             frozen_names=np.array(r["frozen"] if r["frozen"] else [""]), frozen_unchanged=np.array(unchanged),
             options=np.array(repr(bb)), **bufs)
    print("done")


if __name__ == "__main__":
    main()
