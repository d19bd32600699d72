"""`python bench.py --gpus N` means N ranks (without WORLD_SIZE it used to measure one GPU and print n_gpus: 1).
CPU only: `--launch-check` runs the file's rank plumbing -- self-launch through torch.distributed.run, process group (gloo), barrier,
max-over-ranks timing, rank 0's one JSON line -- without the model.  The same self-launch with the real step on one MI355X
(two ranks sharing device 0) is tests/test_gpu_dp.py::test_bench_gpus_2_launches_two_ranks_itself."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env.update(extra)
    return env


def _lines(out):
    return [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]


def test_gpus_2_without_a_torchrun_environment_starts_two_ranks():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--launch-check"],
                         env=_clean_env(), capture_output=True, text=True, timeout=300)
    lines = _lines(out)
    assert out.returncode == 0 and len(lines) == 1, (out.returncode, out.stdout[-1500:], out.stderr[-3000:])
    assert lines[0]["n_gpus"] == 2 and lines[0]["ranks_seen"] == 2 and lines[0]["steps"] == 3


def test_gpus_1_stays_in_process():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--launch-check"],
                         env=_clean_env(), capture_output=True, text=True, timeout=120)
    lines = _lines(out)
    assert out.returncode == 0 and len(lines) == 1 and lines[0]["n_gpus"] == 1, (out.stdout[-1500:], out.stderr[-3000:])
    assert "launching" not in out.stderr


def test_world_size_that_contradicts_gpus_is_refused():
    """under torchrun with another world size the line must not be printed at all (never n_gpus: 1 for --gpus 8)"""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--launch-check"], env=_clean_env(WORLD_SIZE="1", RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and not _lines(out)
    assert "--gpus 8 but WORLD_SIZE=1" in out.stderr


def test_a_failing_rank_fails_the_launcher():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=_clean_env(UEM_LAUNCH_CHECK_FAIL_RANK="0"),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and not _lines(out)
