"""CPU-side checks: the C ABI library builds, loads and exports every symbol include/uemda_hip.h declares;
host logic (config merge, state_dict surface, LR schedule, fail-loudly behaviour)."""
import os
import re

import pytest
import torch

from conftest import ROOT, load_golden

HEADER = os.path.join(ROOT, "include", "uemda_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(uem_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    from uemda_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 50
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in uemda_hip.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(syms)
    assert lib.uem_version() >= 100


def test_bad_arguments_return_error_codes_not_crashes():
    from uemda_amd import _lib
    lib = _lib.load()
    assert lib.uem_pseudo_select(None, None, None, None, 1, 6, 10, 0.8, 0.6, -1, None) == -1
    assert b"null" in lib.uem_last_error()
    with pytest.raises(_lib.UemError):
        _lib.call("uem_downscale_label", None, None, 1, 30, 30, 16, 6, -1, 0.75, None)


def test_product_path_has_no_cpu_fallback_and_never_imports_oracle():
    from uemda_amd import UemError
    from uemda_amd.gast.alignment import DownscaleLabel
    with pytest.raises(UemError):
        DownscaleLabel(16, 6)(torch.zeros(1, 32, 32, dtype=torch.int64))
    for dirpath, _, files in os.walk(os.path.join(ROOT, "uemda_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f


def test_state_dict_surface_matches_reference_layout():
    from oracle.model import param_shapes
    from uemda_amd.models.Encoder import Deeplabv2
    for use_ppm, n in ((False, 334), (True, 382)):
        cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False), multi_layer=True,
                   cascade=False, use_ppm=use_ppm, ppm=dict(num_classes=6, use_aux=False, fc_dim=2048),
                   inchannels=2048, num_classes=6, is_ins_norm=True)
        m = Deeplabv2(cfg)
        sd = m.state_dict()
        ref = param_shapes("resnet50", 6, use_ppm)
        assert len(sd) == n and list(sd.keys()) == list(ref.keys())
        for k, shp in ref.items():
            assert tuple(sd[k].shape) == tuple(shp), k
    # the class's other branches (reference Encoder.py:93-102,111-116): the single head `cls_pred` -- multi_layer=False is the DEFAULT
    # of the reference class -- and the cascade pair, layer5 on the 1024-channel layer3 output
    for (ml, ca, ppm), n in (((False, False, False), 326), ((False, False, True), 350), ((True, True, False), 334)):
        v = Deeplabv2(dict(cfg, multi_layer=ml, cascade=ca, use_ppm=ppm))
        ref = param_shapes("resnet50", 6, ppm, multi_layer=ml, cascade=ca)
        assert len(v.state_dict()) == n and list(v.state_dict().keys()) == list(ref.keys())
        for k, shp in ref.items():
            assert tuple(v.state_dict()[k].shape) == tuple(shp), k
    assert Deeplabv2(dict(backbone=dict(pretrained=False))).config.multi_layer is False and hasattr(Deeplabv2(dict(backbone=dict(pretrained=False))), "cls_pred")
    l4 = m.encoder.resnet.layer4
    assert l4[0].conv2.stride == (1, 1) and l4[0].conv2.dilation == (1, 1) and l4[0].downsample[0].stride == (1, 1)
    assert l4[1].conv2.dilation == (2, 2) and l4[1].conv2.padding == (2, 2)
    m101 = Deeplabv2(dict(cfg, backbone=dict(resnet_type="resnet101", output_stride=16, pretrained=False)))
    assert len(m101.encoder.resnet.layer3) == 23
    with pytest.raises(Exception):
        m(torch.zeros(2, 3, 64, 64))             # CPU tensor: loud failure, no fallback


def test_lr_schedule_matches_reference_golden():
    import types
    from uemda_amd.utils.tools import adjust_learning_rate
    g = load_golden("lr_schedule")
    cfg = types.SimpleNamespace(LEARNING_RATE=1e-2, NUM_STEPS=6000 * 1.5, PREHEAT_STEPS=int(6000 / 20), POWER=0.9)
    opt = types.SimpleNamespace(param_groups=[{"lr": 0}, {"lr": 0}])
    for i, lr in zip(g["iters"].tolist(), g["lrs"].tolist()):
        assert adjust_learning_rate(opt, i, cfg) == pytest.approx(lr, rel=1e-12)
        assert opt.param_groups[1]["lr"] == pytest.approx(10 * lr, rel=1e-12)


def test_config_merge_is_recursive():
    from uemda_amd.models.config import AttrDict
    c = AttrDict()
    c.update(dict(backbone=dict(resnet_type="resnet50", output_stride=32), x=1))
    c.update(dict(backbone=dict(output_stride=16)))
    assert c.backbone.resnet_type == "resnet50" and c.backbone.output_stride == 16 and c.x == 1


def test_committed_bench_line_obeys_the_contract():
    """The bench line under profiles/ (the last `python bench.py` of the round, copied from the GPU box) has the fields the driver
    and the judge read, fractions are fractions, and its HBM-traffic stamp belongs to the conv sources in the tree."""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = sorted(glob.glob(os.path.join(root, "profiles", "r*_bench.json")))[-1]
    line = json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert line["metric"] == base["metric"] and line["higher_is_better"] is True and line["scaling"] == "weak"
    for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["vs_baseline"] is None                               # BASELINE.md publishes no number for this metric on this hardware
    tiles = line["config"]["global_batch"] * line["n_gpus"] if "global_batch" in line["config"] else None
    assert abs(line["value"] * line["ms_per_step"] / 1e3 - (tiles or 64)) < 0.02 * (tiles or 64)     # value = tiles per step / step time
    roof = line["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert 0.0 < roof["frac"] <= 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    for fam in roof["families"].values():
        assert 0.0 < fam["frac"] <= 1.0
    for cfg in line.get("other_configs", {}).values():
        for fam in cfg["roofline"]["families"].values():
            assert 0.0 < fam["frac"] <= 1.0
    hb = line["roofline_hbm"]
    assert hb["bound"] == "hbm" and 0.0 < hb["frac"] <= 1.0
    cb = line["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    # the PMC traffic stamp in the tree was taken on THESE kernel sources (bench.py reports traffic: null otherwise)
    src = open(os.path.join(root, "bench.py")).read()
    assert "kernel_source_sha256_16" in src
    import hashlib
    h = hashlib.sha256()
    for f in ("conv.hip", "stem.hip", "wgrad.hip", "winograd.hip"):
        h.update(open(os.path.join(root, "uemda_amd", "csrc", f), "rb").read())
    stamp = json.load(open(os.path.join(root, "profiles", "traffic_latest.json")))
    assert stamp["kernel_source_sha256_16"] == h.hexdigest()[:16], "re-run scripts/measure_round.sh: the conv sources changed since the PMC pass"
    assert roof["traffic"] is not None
