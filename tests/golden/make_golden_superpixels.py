#!/usr/bin/env python3
"""Golden vectors for the superpixel edge shrinking (reference uemda/gast/superpixels.py:129-152), produced by the
reference function itself under the same stubs as make_golden.py (cv2 and skimage.io are absent here; the function
only uses them to write the result, which the stub swallows).  Run in the build container only:
    python tests/golden/make_golden_superpixels.py      ->  tests/golden/superpixel_shrink.npz
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    mg.install_stubs()
    sp = importlib.import_module("uemda.gast.superpixels")
    from uemda_amd.utils.synth import irregular_superpixels
    rng = np.random.default_rng(11)
    cases = {}
    # (a) irregular regions, 48 x 64 (ignored id = 3*4 = 12); ids collapsed to < 12 so that real ids and the ignored id differ
    lab = irregular_superpixels(1, 48, 64, 11, seed=3)[0, 0].numpy().astype(np.int32)
    lab = np.where(lab >= 11, 0, lab)
    cases["a"] = lab
    # (b) regular 16 x 16 blocks, 32 x 48
    yy, xx = np.mgrid[0:32, 0:48]
    cases["b"] = ((yy // 16) * 3 + xx // 16).astype(np.int32)
    # (c) noise: almost nothing survives
    cases["c"] = rng.integers(0, 4, size=(32, 32)).astype(np.int32)
    out = {}
    for k, lab in cases.items():
        res = sp.edge_shrinking("/tmp/_uem_golden_unused", "x.png", "png", lab.copy(), win_size=3, region_size=16)
        out[f"in_{k}"] = lab
        out[f"out_{k}"] = np.asarray(res).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "superpixel_shrink.npz"), **out)
    for k in cases:
        print(k, cases[k].shape, "kept", float((out[f"out_{k}"] == cases[k]).mean()))


if __name__ == "__main__":
    main()
