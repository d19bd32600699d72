"""RCCL plumbing on one GPU: the data-parallel code paths (process group over backend nccl = RCCL, parameter
broadcast, two-bucket gradient all-reduce overlapped with backward, prototype-sum all-reduce) forced on with a single
rank.  The collectives are trivial at world size 1 but real; the multi-rank arithmetic is covered by the gloo tests
(tests/test_dp_gloo.py).  Runs bench.py in a child process so that the test process keeps no process group."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, *args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", **extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "4", "--size", "256", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline", "--no-other-precisions", *args],
                         env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, out.stderr[-2000:]
    return json.loads(lines[-1])


def test_single_rank_rccl_step_matches_plain_step():
    plain = _bench({})
    forced = _bench({"UEM_DP_FORCE": "1", "MASTER_PORT": "29533"})
    assert forced["n_gpus"] == 1 and forced["value"] > 0
    # same seeds, same arithmetic: the source loss after the same number of steps agrees (atomics reorder the last bits)
    assert forced["loss_source"] == pytest.approx(plain["loss_source"], rel=1e-4)
