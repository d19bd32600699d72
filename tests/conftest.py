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
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiub" else z[k]) for k in z.files}      # name lists stay numpy


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
