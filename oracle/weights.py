"""Deterministic, name-keyed weight fill (build-side; no reference code involved).

The same filled state_dict is loaded into the reference (golden generator, via
`load_state_dict`), the oracle and the HIP model, so every implementation sees bit-identical
weights without needing pretrained checkpoints (no network: SURVEY.md §8 a2).
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch

from .model import param_shapes


def _rng(key, seed):
    return np.random.default_rng([zlib.crc32(key.encode()), seed])


def det_state_dict(resnet_type="resnet50", num_classes=6, use_ppm=False, seed=2333, fc_dim=2048,
                   shapes=None, multi_layer=True, cascade=False):
    """Return an OrderedDict of CPU tensors covering every state_dict entry.

    conv weights ~ N(0, 2/fan_out) (the reference's Kaiming fan_out init, _resnets.py:166);
    ASPP conv weights ~ N(0, 0.01) (Encoder.py:77-78); BN gamma ~ U(.5,1.5), beta ~ N(0,.1),
    running_mean ~ N(0,.1), running_var ~ U(.5,1.5) so that eval-mode BN is non-trivial.
    """
    shapes = shapes if shapes is not None else param_shapes(resnet_type, num_classes, use_ppm, fc_dim, multi_layer, cascade)
    sd = OrderedDict()
    for k, shp in shapes.items():
        g = _rng(k, seed)
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.int64)
            continue
        if len(shp) == 4:
            if "conv2d_list" in k:
                a = g.normal(0.0, 0.01, shp)
            else:
                fan_out = shp[0] * shp[2] * shp[3]
                a = g.normal(0.0, np.sqrt(2.0 / fan_out), shp)
        elif k.endswith("running_var"):
            a = g.uniform(0.5, 1.5, shp)
        elif k.endswith("running_mean"):
            a = g.normal(0.0, 0.1, shp)
        elif k.endswith(".weight"):            # BN gamma
            a = g.uniform(0.5, 1.5, shp)
        else:                                  # BN beta / conv bias
            a = g.normal(0.0, 0.1, shp)
        sd[k] = torch.from_numpy(np.asarray(a, dtype=np.float32))
    return sd


def checksum(tensors):
    """Order-sensitive float64 checksum used by the step goldens (sum, abs-sum)."""
    s = 0.0
    a = 0.0
    for t in tensors:
        t64 = t.detach().double()
        s += float(t64.sum())
        a += float(t64.abs().sum())
    return s, a


def fill_like(shapes, tag, seed=1):
    """Deterministic fill for small layer fixtures: {key: shape} -> {key: tensor}."""
    sd = OrderedDict()
    for k, shp in shapes.items():
        r = _rng(tag + k, seed)
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.int64)
        elif len(shp) == 4:
            sd[k] = torch.from_numpy(r.normal(0, 0.2, shp).astype(np.float32))
        elif k.endswith("running_var") or (k.endswith("weight") and len(shp) == 1):
            sd[k] = torch.from_numpy(r.uniform(0.5, 1.5, shp).astype(np.float32))
        else:
            sd[k] = torch.from_numpy(r.normal(0, 0.1, shp).astype(np.float32))
    return sd


def subsample(t):
    """The fixture subsampling rule used by tests/golden/make_golden.py for large tensors."""
    return t if t.numel() <= 8192 else t.reshape(-1)[:: t.numel() // 4096][:4096]
