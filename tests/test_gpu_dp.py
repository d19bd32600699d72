"""Data parallel on the GPU box (one MI355X): the REAL `uemda_amd.dp.DataParallel` object that bench.py runs under
torchrun -- parameter broadcast, forward/backward pairing, the layer3[0] bucket trigger, the asynchronous tail
all-reduce, the prototype partial-sum all-reduce -- driven by two fresh child processes that
share device 0 (backend gloo: RCCL refuses two ranks on one device).  Plus the single-rank RCCL plumbing smoke.
Every run is `bench.py` in child processes, so the test process keeps no process group and never re-execs."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--batch", "4", "--size", "256", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-other-precisions",
          "--no-kernel-events"]


def _report(rc, stdout, stderr, head=2500, tail=2500):
    """What an assertion on a child process shows: the return code (negative = killed by that signal), the tail of stdout, and
    BOTH ends of stderr -- the first lines carry the c10 / HIP message that names the failing call, the last ones the abort's frames
    (a SIGABRT in the captured data-parallel step left only libc frames in the record because the tail alone was kept)."""
    err = stderr if len(stderr) <= head + tail else stderr[:head] + f"\n... [{len(stderr) - head - tail} chars cut] ...\n" + stderr[-tail:]
    return f"child rc={rc}\n--- stdout (tail) ---\n{stdout[-1500:]}\n--- stderr (head + tail) ---\n{err}"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(extra_env, *args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", **extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *COMMON, *args],
                         env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, _report(out.returncode, out.stdout, out.stderr)
    return json.loads(lines[-1])


def _two_ranks(path, *args, env=None):
    """bench.py as 2 ranks on device 0; returns (rank-0 JSON line, [rank dumps])."""
    port = _free_port()
    procs = []
    for r in range(2):
        e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r),
                 LOCAL_RANK=str(r), WORLD_SIZE="2", **(env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                                       "--device", "0", *COMMON, "--dump-params", path, *args],
                                      env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, _report(p.returncode, so, se)
    line = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    return line, [torch.load(f"{path}.rank{r}.pt") for r in range(2)]


def test_bench_gpus_2_launches_two_ranks_itself():
    """`python bench.py --gpus 2` with no torchrun environment starts its two ranks itself (fresh children of a parent that made no GPU
    call) and rank 0's line says n_gpus: 2 -- here both ranks share device 0 over gloo, the only two-rank form one MI355X allows."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--device", "0", *COMMON],
                         env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, _report(out.returncode, out.stdout, out.stderr)
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["replicas_identical"] and line["value"] > 0
    assert line["config"]["global_batch"] == 2 * 2 * 4            # (source + target) x per-rank batch x ranks


def test_single_rank_rccl_step_matches_plain_step():
    plain = _bench({})
    forced = _bench({"UEM_DP_FORCE": "1", "MASTER_PORT": str(_free_port())})
    assert forced["n_gpus"] == 1 and forced["value"] > 0
    # same seeds, same arithmetic: the source loss after the same number of steps agrees (atomics reorder the last bits)
    assert forced["loss_source"] == pytest.approx(plain["loss_source"], rel=1e-4)


def test_single_rank_rccl_allreduce_through_the_c_abi():
    """uem_comm_unique_id / uem_comm_init / uem_allreduce_flat / uem_comm_destroy (SURVEY 8b `allreduce_flat`), the C ABI's own
    collective for hosts without torch.distributed: one rank (RCCL refuses two on one device), so the sum is the buffer itself, but it
    is RCCL on a stream of the caller's choosing, loaded from the host process.  In a child process: the test process keeps no
    communicator."""
    code = (
        "import ctypes, torch\n"
        "from uemda_amd._lib import call\n"
        "ident = ctypes.create_string_buffer(128)\n"
        "call('uem_comm_unique_id', ident)\n"
        "h = ctypes.c_void_p()\n"
        "call('uem_comm_init', ctypes.byref(h), ident.raw, 0, 1)\n"
        "x = torch.randn(1 << 20, device='cuda'); ref = x.clone()\n"
        "st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())\n"
        "call('uem_allreduce_flat', h, x.data_ptr(), x.numel(), st.cuda_stream)\n"
        "st.synchronize()\n"
        "assert torch.equal(x, ref)\n"
        "call('uem_comm_destroy', h)\n"
        "print('ALLREDUCE_OK')\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ALLREDUCE_OK" in out.stdout, _report(out.returncode, out.stdout, out.stderr)


def test_single_rank_rccl_graphed_data_parallel_step():
    """GraphedStep(dp=wrapper) over the nccl (= RCCL) backend: the early tail-bucket all-reduce and the head all-reduce are captured
    with the step.  One rank (one device), real RCCL launches inside the graph; each replay against an eager data-parallel step
    from the same complete state (scripts/dp_graph_check.py)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), UEM_DP_FORCE="1",
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dp_graph_check.py")], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "DP_GRAPH_OK" in out.stdout, _report(out.returncode, out.stdout, out.stderr)
    # and the gloo group (host round trip) is refused up front, not captured wrongly
    env["UEM_DP_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dp_graph_check.py")], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0 and "DP_GRAPH_REFUSED" in out.stdout, _report(out.returncode, out.stdout, out.stderr)


def test_two_rank_data_parallel_object(tmp_path):
    line, (r0, r1) = _two_ranks(str(tmp_path / "ov"))
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2"
    # replicas: parameters (3 optimizer steps) and prototypes bit-identical on both ranks
    assert r0["params_sum"] == r1["params_sum"] and torch.equal(r0["params_sample"], r1["params_sample"])
    assert torch.equal(r0["prototypes"], r1["prototypes"])
    assert r0["unpaired_forwards"] == 0
    # the reduced gradient every rank steps with is the same one
    assert torch.equal(r0["first_grad_sample"], r1["first_grad_sample"])
    # overlap (early tail bucket) vs one all-reduce after backward: the same sums (two RUNS differ in the last bits only
    # because the split-K weight gradients are accumulated with fp32 atomics)
    def close(a, b, tol):
        return float((a.double() - b.double()).norm() / b.double().norm()) < tol
    _, (n0, n1) = _two_ranks(str(tmp_path / "nov"), "--no-overlap")
    assert torch.equal(n0["params_sample"], n1["params_sample"]) and torch.equal(n0["prototypes"], n1["prototypes"])
    assert close(n0["first_grad_sample"], r0["first_grad_sample"], 1e-5)
    assert close(n0["params_sample"], r0["params_sample"], 1e-5) and close(n0["prototypes"], r0["prototypes"], 1e-5)
    # the reduced gradient = mean of what each rank computes alone on its own tiles from the same weights
    alone = []
    for r in range(2):
        _bench({}, "--data-rank", str(r), "--dump-params", str(tmp_path / f"solo{r}"))
        alone.append(torch.load(str(tmp_path / f"solo{r}") + ".rank0.pt"))
    mean = 0.5 * (alone[0]["first_grad_sample"] + alone[1]["first_grad_sample"])
    rel = float((r0["first_grad_sample"] - mean).norm() / mean.norm())
    assert rel < 1e-4, rel                      # fp32 atomics / summation order only
    assert float((alone[0]["first_grad_sample"] - alone[1]["first_grad_sample"]).norm() / mean.norm()) > 1e-2   # the ranks' tiles differ


def test_forward_backward_pairing_guard():
    """A train-mode forward that never gets a backward must not desynchronise the bucket trigger."""
    from uemda_amd import dp as udp
    from uemda_amd.models.Encoder import Deeplabv2
    from uemda_amd.ops import UemError
    cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False), multi_layer=True, cascade=False,
               use_ppm=False, ppm=dict(num_classes=6, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=6,
               is_ins_norm=True)
    model = Deeplabv2(cfg).cuda().train()
    wrap = udp.DataParallel(model)
    x = torch.randn(2, 3, 64, 64, device="cuda")
    model(x)                                    # e.g. a validation pass left in .train(): no backward follows
    p1, p2, _ = model(x)
    (p1.sum() + p2.sum()).backward()
    assert wrap._fwd_calls == 2 and wrap._bwd_calls == 1 and wrap._pending is None
    assert wrap.reduce_gradients() == 1.0
    assert wrap.unpaired_forwards == 1 and wrap._fwd_calls == 0 and wrap._bwd_calls == 0
    p1, p2, _ = model(x)                        # the next step pairs up again
    (p1.sum() + p2.sum()).backward()
    assert wrap._fwd_calls == 1 and wrap._bwd_calls == 1
    wrap.reduce_gradients()
    with pytest.raises(UemError, match="backward passes"):
        wrap._on_trigger_backward()             # a backward the wrapper saw no forward for


def test_dropout_masks_differ_between_ranks():
    from uemda_amd import ops
    from uemda_amd.models.ppm import dropout_seed
    a = torch.ones(8, 4, 4, 512, device="cuda")
    masks = []
    for rank in (0, 1):
        m = torch.empty(8, 512, device="cuda")
        ops.call("uem_dropout2d", ops.ptr(a), ops.ptr(a.clone()), ops.ptr(m), 8, 16, 512, 0.1, dropout_seed(1, rank), ops.stream())
        masks.append(m.cpu())
    assert not torch.equal(masks[0], masks[1])
    assert dropout_seed(1, 0) != dropout_seed(2, 0)
