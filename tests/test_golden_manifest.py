"""A committed fixture must be what its committed generator writes (`make_golden_r4.py step512` had gained a key the
committed `.npz` lacked).  The generators cannot run on the GPU box or inside the CPU suite (they import /root/reference and take
minutes), so the check is static: every `save(<fixture>, key=..., ...)` call of tests/golden/make_golden*.py is parsed, the fixture
name resolved (literal, f-string pattern, or a parameter of the enclosing function with its default and the literals passed at its
call sites), and every explicitly named key must exist in the committed `.npz`."""
import ast
import glob
import os
import re

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _save_calls(path):
    """[(fixture-name candidates, explicit keys, line)] of one generator script"""
    tree = ast.parse(open(path).read())
    funcs = {n.name: n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)}
    parents = {}
    for node in ast.walk(tree):
        for ch in ast.iter_child_nodes(node):
            parents[ch] = node

    def enclosing_function(node):
        while node in parents:
            node = parents[node]
            if isinstance(node, ast.FunctionDef):
                return node
        return None

    def param_values(fn, pname):
        """literal values a parameter takes: its default + literals passed by keyword at the function's call sites"""
        vals = []
        args = fn.args.args
        defaults = [None] * (len(args) - len(fn.args.defaults)) + list(fn.args.defaults)
        for a, d in zip(args, defaults):
            if a.arg == pname and isinstance(d, ast.Constant) and isinstance(d.value, str):
                vals.append(d.value)
        for call in ast.walk(tree):
            if isinstance(call, ast.Call) and isinstance(call.func, ast.Name) and call.func.id == fn.name:
                for kw in call.keywords:
                    if kw.arg == pname and isinstance(kw.value, ast.Constant):
                        vals.append(kw.value.value)
        return vals

    found = []
    all_files = [os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, "*.npz"))]
    for call in ast.walk(tree):
        if not isinstance(call, ast.Call) or not call.args:
            continue
        f = call.func
        is_save = (isinstance(f, ast.Name) and f.id == "save") or (isinstance(f, ast.Attribute) and f.attr == "save" and
                                                                   isinstance(f.value, ast.Name) and f.value.id == "mg")
        if not is_save:
            continue
        keys = [kw.arg for kw in call.keywords if kw.arg is not None]
        if not keys:
            continue                                              # save(name, **arrays): nothing named in the source
        a = call.args[0]
        if isinstance(a, ast.Constant):
            names = [a.value]
        elif isinstance(a, ast.JoinedStr):
            pat = "".join(re.escape(v.value) if isinstance(v, ast.Constant) else ".+" for v in a.values)
            names = [n for n in all_files if re.fullmatch(pat, n)]
        elif isinstance(a, ast.Name) and enclosing_function(call) is not None:
            names = param_values(enclosing_function(call), a.id)
        else:
            names = []
        found.append((names, keys, call.lineno))
    return found


GENERATORS = sorted(glob.glob(os.path.join(GOLDEN, "make_golden*.py")))
CASES = [(os.path.basename(g), names, keys, line) for g in GENERATORS for names, keys, line in _save_calls(g)]


def test_generators_were_found():
    assert len(GENERATORS) >= 4 and len(CASES) >= 15
    covered = {n for _, names, _, _ in CASES for n in names}
    assert {"model_aspp_r50_b8_512", "model_aspp_r101_b2_256", "model_aspp_r50_b2_256", "model_ppm_r50_b2_256", "label_refine",
            "losses"} <= covered


@pytest.mark.parametrize("gen,names,keys,line", CASES, ids=[f"{g}:{l}" for g, _, _, l in CASES])
def test_committed_fixture_has_every_key_its_generator_writes(gen, names, keys, line):
    assert names, f"{gen}:{line}: could not resolve the fixture name of this save() call"
    for name in names:
        path = os.path.join(GOLDEN, name + ".npz")
        assert os.path.exists(path), f"{gen}:{line} writes {name}.npz, which is not committed"
        have = set(np.load(path, allow_pickle=False).files)
        missing = [k for k in keys if k not in have]
        assert not missing, f"{name}.npz lacks {missing}: re-run `python tests/golden/{gen}` (generator line {line})"


def test_update_noise_floors_cover_every_step_fixture():
    """tests/golden/update_noise_floors.npz (make_noise_floors.py: the reference's own first-step update under N rounding-level
    perturbations) has an entry for EVERY step fixture, in the fixture's tensor order, with at least 6 draws; its first two draws are
    the two the fixture's own `upd_noise_floor` was taken from, and reproduce it."""
    import glob
    import os
    import numpy as np
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    fl = np.load(os.path.join(here, "update_noise_floors.npz"))
    seen = 0
    for path in sorted(glob.glob(os.path.join(here, "model_*.npz"))):
        fx = np.load(path)
        if "upd_names" not in fx.files:
            continue
        name = os.path.splitext(os.path.basename(path))[0]
        assert name + ":floor_max" in fl.files, f"{name}: no derived noise floor (run tests/golden/make_noise_floors.py {name})"
        assert [str(n) for n in fl[name + ":names"]] == [str(n) for n in fx["upd_names"]]
        draws = fl[name + ":draws"]
        assert draws.shape[0] >= 6 and draws.shape[1] == len(fx["upd_names"]) and len(fl[name + ":perturbations"]) == draws.shape[0]
        np.testing.assert_allclose(draws[:2].max(axis=0), fx["upd_noise_floor"], rtol=1e-5, atol=1e-8)
        np.testing.assert_array_equal(draws.max(axis=0), fl[name + ":floor_max"])
        assert (fl[name + ":floor_max"] >= fx["upd_noise_floor"] * (1 - 1e-5) - 1e-8).all()
        seen += 1
    assert seen >= 7
