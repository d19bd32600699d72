"""N > 1 path on CPU: two `gloo` ranks exercise uemda_amd.dp (process-group init from the torchrun
environment, parameter broadcast, flat-gradient all-reduce with the 1/world prescale) and the prototype
partial-sum reduction contract (sum BEFORE division/EMA)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from uemda_amd import dp
    from oracle import gast
    r, w, _ = dp.init("gloo")
    assert (r, w) == (rank, world) and dp.world_size() == world
    # C0: replicas start from rank 0's parameters
    arena = torch.full((1003,), float(rank + 1))
    dp.broadcast_flat(arena)
    assert torch.equal(arena, torch.ones(1003))
    # C1: gradient all-reduce(sum); the average is folded into the optimizer as grad_prescale = 1/world
    g = torch.arange(1003, dtype=torch.float32) * (rank + 1)
    dp.allreduce_flat_(g)
    prescale = 1.0 / world
    ref = torch.arange(1003, dtype=torch.float32) * sum(range(1, world + 1)) * prescale
    torch.testing.assert_close(g * prescale, ref)
    # clip coefficient is computed from the REDUCED gradient => identical on every rank
    norm = (g * prescale).norm()
    norms = [torch.zeros(()) for _ in range(world)]
    dist.all_gather(norms, norm)
    assert all(torch.equal(n, norms[0]) for n in norms)
    # C2: prototype update from rank-local features == single-process update over the concatenated batch
    gen = torch.Generator().manual_seed(5)
    feat = torch.randn(4, 16, 2, 2, generator=gen)
    lab = torch.randint(-1, 3, (4, 32, 32), generator=gen)
    lab = lab // 1
    protos = torch.randn(3, 16, generator=gen)
    half = slice(rank * 2, rank * 2 + 2)
    ds = gast.downscale_label(lab[half], 3)
    _, sums, cnts = gast.local_prototypes(feat[half], ds, protos, 3)
    packed = torch.cat([sums.reshape(-1), cnts])
    dp.allreduce_flat_(packed)
    sums, cnts = packed[:48].view(3, 16), packed[48:]
    local = torch.where((cnts < 1).unsqueeze(1), protos, sums / (cnts.unsqueeze(1) + 1e-7))
    new = 0.004 * local + 0.996 * protos
    full, _ = gast.update_prototype(feat, lab, protos, 3, 0.996)
    torch.testing.assert_close(new, full, rtol=1e-6, atol=1e-6)
    out.put((rank, float(new.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_data_parallel_contract():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert res[0][1] == pytest.approx(res[1][1], rel=0, abs=0)         # replicas agree bit-for-bit


def test_weak_scaling_accounting():
    """bench.py counts source + target tiles of every rank per step (value = whole-job tiles/s)."""
    B, world, steps, elapsed = 32, 8, 5, 2.0
    assert (2 * B) * world * steps / elapsed == 1280.0
