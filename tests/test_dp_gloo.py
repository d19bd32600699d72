"""N > 1 path on CPU: `gloo` ranks (world size 2 and 8 -- the node's size) exercise uemda_amd.dp (process-group
init from the torchrun environment, parameter broadcast, flat-gradient all-reduce with the 1/world prescale, the bucket split /
trigger pairing of the real DataParallel object, bench.py's `replicas_identical` check) and the prototype partial-sum and
class-count reduction contracts (sum BEFORE division/EMA)."""
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
    feat = torch.randn(2 * world, 16, 2, 2, generator=gen)
    lab = torch.randint(-1, 3, (2 * world, 32, 32), generator=gen)
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


WORLDS = [2, 8]          # 8 = the ranks of one MI355X node (BASELINE config 4), rehearsed on the CPU


def _spawn(target, world, timeout=240):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    import time
    res, t_end = [], time.time() + timeout
    while len(res) < world:                                       # drained BEFORE the joins: a child blocks in exit on a full pipe
        try:
            res.append(q.get(timeout=1.0))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() > t_end:                       # a failed rank leaves the others in a collective: end them all
                for p in procs:
                    p.kill()
                raise AssertionError(f"ranks exited with {dead}" if dead else f"no result from every rank within {timeout} s")
    for p in procs:
        p.join(timeout=timeout)
        assert p.exitcode == 0
    return sorted(res, key=lambda t: t[0])


@pytest.mark.parametrize("world", WORLDS)
def test_gloo_data_parallel_contract(world):
    res = _spawn(_worker, world)
    assert all(r[1] == res[0][1] for r in res)                         # replicas agree bit-for-bit


class _Trig(torch.autograd.Function):
    """stands in for the END of blocks.BottleneckFn.backward: identity forward, calls the block's data-parallel trigger in backward.
    It sits on the block's INPUT, so that autograd runs it after the block's own weight gradient has been accumulated -- as the real
    callback runs after the block's weight-gradient launches.  (Round 3 had it on the block's output: the early bucket could leave
    before layer3's gradient of the last graph had landed, a race the test lost under CPU contention.)"""

    @staticmethod
    def forward(ctx, x, blk):
        ctx.blk = blk
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        cb = getattr(ctx.blk, "_uem_after_backward", None)
        if cb is not None:
            cb()
        return g, None


def _toy_model():
    """The attributes DataParallel touches (flat_parameters, encoder.resnet.layer3[0], forward pre-hook) around a
    three-layer network whose parameters are views of one flat arena, in forward order like Deeplabv2's."""
    import torch.nn as nn

    class Toy(nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = nn.Module()
            self.encoder.resnet = nn.Module()
            self.encoder.resnet.stem = nn.Linear(8, 8, bias=False)
            self.encoder.resnet.layer3 = nn.ModuleList([nn.Linear(8, 8, bias=False)])
            self.head = nn.Linear(8, 4, bias=False)
            ps = list(self.parameters())
            n = sum(p.numel() for p in ps)
            self._arena, self._garena, self._n = torch.zeros(n), torch.zeros(n), n
            off = 0
            g = torch.Generator().manual_seed(3)
            for p in ps:
                v = self._arena[off:off + p.numel()].view_as(p)
                v.copy_(torch.randn(p.shape, generator=g))
                p.data = v
                p.grad = self._garena[off:off + p.numel()].view_as(p)
                off += p.numel()

        def flat_parameters(self):
            return self._arena, self._garena, self._n

        def forward(self, x):
            r = self.encoder.resnet
            return self.head(r.layer3[0](_Trig.apply(torch.relu(r.stem(x)), r.layer3[0])))

    return Toy()


def _dp_object_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from uemda_amd import dp
    from uemda_amd.ops import UemError
    dp.init("gloo")
    torch.manual_seed(100 + rank)                      # every rank would start from different weights ...
    model = _toy_model()
    model._arena.add_(float(rank))
    wrap = dp.DataParallel(model)                      # ... until the broadcast
    assert wrap._split == 64 and wrap._active
    w0 = model._arena.clone()
    g = torch.Generator().manual_seed(rank)
    xs, xt = torch.randn(5, 8, generator=g), torch.randn(5, 8, generator=g)
    model.train()
    model._garena.zero_()
    loss = model(xs).square().mean() + model(xt).square().mean()       # two forwards, ONE backward (the SSL step)
    loss.backward()
    assert wrap._pending is not None and wrap._bwd_calls == 2          # the tail bucket left during backward
    prescale = wrap.reduce_gradients()
    reduced = model._garena.clone() * prescale
    # reference: every rank's local gradient, recomputed without the wrapper, averaged
    # (the ring sums the ranks' gradients in another order than this loop: a few ulp at 8 ranks)
    torch.testing.assert_close(reduced, sum(_solo_grad(w0, r, (0, 1)) for r in range(world)) / world, rtol=1e-5, atol=1e-6)
    # a forward that never sees a backward: no early bucket, full all-reduce, counters reset, replicas still agree
    model._garena.zero_()
    model(xs)
    model(xt).square().mean().backward()
    assert wrap._pending is None
    wrap.reduce_gradients()
    assert wrap.unpaired_forwards == 1 and wrap._fwd_calls == 0
    torch.testing.assert_close(model._garena * prescale, sum(_solo_grad(w0, r, (1,)) for r in range(world)) / world,
                               rtol=1e-5, atol=1e-6)
    # a train-mode forward after the early bucket went out must not pass silently
    model._garena.zero_()
    model(xs).square().mean().backward()
    try:
        model(xt)
        raised = False
    except UemError:
        raised = True
    assert raised
    wrap.reduce_gradients()
    # bench.py's end-of-run check under N > 1 (`replicas_identical`): step every replica with its reduced gradient, gather the
    # parameter checksums, compare -- and see the check FAIL for a replica that is nudged by one ulp
    model._arena.copy_(w0 - 0.1 * reduced)
    def gathered(t):
        mine = torch.tensor([float(t.double().sum()), float(t.double().abs().sum())], dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        return torch.stack(allr)
    allr = gathered(model._arena)
    assert bool((allr == allr[0]).all())
    if rank == world - 1:
        model._arena[0] = torch.nextafter(model._arena[0], torch.tensor(float("inf")))
    allr = gathered(model._arena)
    assert not bool((allr == allr[0]).all())
    out.put((rank, float(w0.sum()), float(reduced.sum())))
    dist.barrier()
    dist.destroy_process_group()


def _solo_grad(w0, r, which):
    """gradient of rank r's loss over its inputs `which` (0 = first, 1 = second tensor of its seeded pair), no wrapper"""
    m = _toy_model()
    m._arena.copy_(w0)
    g = torch.Generator().manual_seed(r)
    xs = [torch.randn(5, 8, generator=g), torch.randn(5, 8, generator=g)]
    m._garena.zero_()
    sum(m(xs[i]).square().mean() for i in which).backward()
    return m._garena.clone()


@pytest.mark.parametrize("world", WORLDS)
def test_gloo_data_parallel_object(world):
    """uemda_amd.dp.DataParallel itself with 2 and 8 ranks: broadcast, bucket split at layer3[0], trigger counting over two
    forwards and one backward, asynchronous tail all-reduce + head all-reduce, the pairing guards, the replica check."""
    res = _spawn(_dp_object_worker, world)
    assert all(r[1:] == res[0][1:] for r in res)                      # same start weights, same reduced gradient


def _class_balance_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from uemda_amd import dp
    from uemda_amd.gast.balance import ClassBalance
    dp.init("gloo")
    gen = torch.Generator().manual_seed(11)
    C = 7
    labels = torch.randint(-1, C, (3, 2 * world, 24, 24), generator=gen)      # three steps of a global batch of 2 tiles per rank
    labels[:, :2][labels[:, :2] == 3] = -1                                    # rank 0's share never sees class 3, more ignored pixels
    cb = ClassBalance(class_num=C, ignore_label=-1, decay=0.9, temperature=0.5, device="cpu")
    for step in range(3):
        mine = labels[step, rank * 2: rank * 2 + 2]
        counts = torch.stack([(mine == c).sum() for c in list(range(C)) + [-1]]).float()      # what uem_class_count returns
        cb.freq = (1.0 - cb.decay) * cb._freq_from_counts(counts) + cb.decay * cb.freq         # ema_update on rank-local counts
    out.put((rank, cb.freq.clone(), cb._get_class_wight().clone()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", WORLDS)
def test_class_balance_counts_are_all_reduced(world):
    """SURVEY 8e collective (3), reference uemda/gast/balance.py:45-61: with the batch split over the ranks the class-frequency
    EMA (and the per-pixel weights built from it) must equal the single-process ones on the whole batch."""
    res = {r: (freq, cw) for r, freq, cw in _spawn(_class_balance_worker, world)}
    gen = torch.Generator().manual_seed(11)
    C = 7
    labels = torch.randint(-1, C, (3, 2 * world, 24, 24), generator=gen)
    labels[:, :2][labels[:, :2] == 3] = -1
    freq = torch.ones(C) / C
    for step in range(3):                                                     # the reference formula on the whole batch
        lab = labels[step]
        cls = torch.stack([(lab == c).sum() for c in range(C)]).float()
        freq = 0.1 * (cls / ((lab != -1).sum().float() + 1e-7)) + 0.9 * freq
    prob = torch.softmax((1.0 - freq) / 0.5, dim=0)
    cw = prob / (prob.max() + 1e-7)
    for r in range(world):
        torch.testing.assert_close(res[r][0], freq, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(res[r][1], cw, rtol=1e-6, atol=1e-7)
        assert torch.equal(res[0][0], res[r][0])                              # replicas bit-identical
    # and the rank-local frequencies would NOT have been the same (the test can fail)
    l0, l1 = labels[0, :2], labels[0, 2:]
    f0 = torch.stack([(l0 == c).sum() for c in range(C)]).float() / (l0 != -1).sum()
    f1 = torch.stack([(l1 == c).sum() for c in range(C)]).float() / (l1 != -1).sum()
    assert (f0 - f1).abs().max() > 1e-2


def test_dropout_seed_is_per_rank():
    from uemda_amd.models.ppm import dropout_seed
    assert len({dropout_seed(1, r) for r in range(8)}) == 8 and dropout_seed(1, 0) != dropout_seed(2, 0)


def test_weak_scaling_accounting():
    """bench.py counts source + target tiles of every rank per step (value = whole-job tiles/s)."""
    B, world, steps, elapsed = 32, 8, 5, 2.0
    assert (2 * B) * world * steps / elapsed == 1280.0


def _agree_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from uemda_amd import dp
    from uemda_amd.step import GraphedStep
    dp.init("gloo")
    every = GraphedStep.all_ranks_ok(True)
    one_failed = GraphedStep.all_ranks_ok(rank != world - 1)          # the last rank's capture "raised"
    none = GraphedStep.all_ranks_ok(False)
    out.put((rank, every, one_failed, none))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", WORLDS)
def test_graph_capture_agreement_across_ranks(world):
    """GraphedStep.all_ranks_ok: a captured data-parallel step may be replayed only if EVERY rank holds the graph;
    one rank whose capture failed turns the answer to False on all of them, so they fall back to the eager step together."""
    res = _spawn(_agree_worker, world)
    assert all(r[1] is True and r[2] is False and r[3] is False for r in res)


def test_graph_capture_agreement_without_a_process_group():
    from uemda_amd.step import GraphedStep
    assert GraphedStep.all_ranks_ok(True) is True and GraphedStep.all_ranks_ok(False) is False
