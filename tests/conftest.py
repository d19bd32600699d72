import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    g = {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiub" else z[k]) for k in z.files}         # name lists stay numpy
    # step fixtures: the per-tensor noise floor of the first update as the MAX over many self-draws of the reference
    # (tests/golden/make_noise_floors.py -> update_noise_floors.npz), in the fixture's own tensor order
    fl = os.path.join(GOLDEN, "update_noise_floors.npz")
    if "upd_names" in g and os.path.exists(fl):
        f = np.load(fl)
        if name + ":floor_max" in f.files:
            assert [str(n) for n in f[name + ":names"]] == [str(n) for n in g["upd_names"]], "noise-floor table out of step with the fixture"
            g["upd_noise_floor_max"] = torch.from_numpy(f[name + ":floor_max"])
            g["upd_noise_floor_draws"] = int(f[name + ":draws"].shape[0])
    return g


def golden_initial_state(g, sd):
    """State dict a model-level fixture's step started from: `sd` with the running statistics the fixture stores (calibrated
    ones for the frozen-BatchNorm step, tests/golden/make_golden.py calibrated_state_dict) written over the defaults."""
    if "init_stat_names" not in g:
        return sd
    sd = {k: v.clone() for k, v in sd.items()}
    off, vals = g["init_stat_offsets"], g["init_stat_values"]
    for i, k in enumerate(str(n) for n in g["init_stat_names"]):
        sd[k] = vals[int(off[i]):int(off[i + 1])].reshape(sd[k].shape).clone()
    return sd


@pytest.fixture(scope="session")
def golden():
    return load_golden
