#!/usr/bin/env python3
"""Benchmark of the UemDA hot path on MI355X: one `train_ssl_uem.py`-style iteration per step
(2 forwards, label_refine + pseudo_selection, prototype EMA, CE + UVEM, backward, clip + SGD) on synthetic
512x512 tiles, ResNet50-ASPP, per-GPU batch 32 source + 32 target tiles (BASELINE.json configs[2] = configs[1]
plus the mining the metric names).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` = tiles (source + target) per second over all ranks, inputs resident in
HBM when the timed region starts.  `roofline` is for the dominant kernel family (the f32-MFMA implicit-GEMM
convolution), from HIP events around every one of its launches inside the timed region: `achieved` / `frac` count the multiplies
the matrix pipe EXECUTES (a Winograd conv issues 2.25x / 4x fewer than its algorithmic count), `algorithmic_tflops` is the speed
figure beside it.  `roofline_hbm` prices the HBM-bound mining phase (label_refine + pseudo_selection).  `cpu_baseline` is the
oracle (CPU restatement of the reference, `kind: "port"`) timed on the host cores at BASELINE config 1.
`other_configs` (N = 1): the same step in the other configurations BASELINE.json names (bf16 storage, the PPM head,
ResNet-101 on 1024x1024 tiles; BASELINE config 2's source-only step), 6 timed steps each AFTER the headline plus one untimed step
with per-launch events for its own roofline entry; they never touch `value`.
"""
import argparse
import gc
import hashlib
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

F32_MATRIX_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, spec
BF16_MATRIX_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA (the peak the bf16-storage configurations are priced against)
HBM_PEAK_GBPS = 8000.0               # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s is what a streaming copy reaches)


def kernel_source_hash():
    """sha256 (16 hex digits) over the conv kernel sources: profiles/traffic_latest.json is valid for ONE build of them."""
    h = hashlib.sha256()
    for f in ("conv.hip", "stem.hip", "wgrad.hip", "winograd.hip"):
        with open(os.path.join(ROOT, "uemda_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pin_cpus(local_rank, local_world):
    """One slice of the host's cores per rank (eight ranks enqueue ~1400 launches per step each; left to the scheduler they migrate
    across sockets).  Returns the number of cores this rank may use."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
        if local_world > 1 and len(cpus) >= local_world:
            per = len(cpus) // local_world
            mine = cpus[local_rank * per:(local_rank + 1) * per]
            os.sched_setaffinity(0, mine)
            return len(mine)
        return len(cpus)
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def note(msg):
    """progress line on stderr (a long silent run looks hung to the GPU harness)"""
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def usable_cpus():
    """CPUs this process can actually run on: the affinity mask capped by the cgroup CPU quota (a one-GPU box shows all 256 host
    CPUs in the mask and grants 16 of them; 256 torch threads on 16 cores take minutes per step)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def _all_threads_leg(nthreads, q):
    """child process: the R50-ASPP ssl step of BASELINE config 1 on `nthreads` torch threads (killed by the parent on time-out)"""
    from oracle import synth
    from oracle.model import OracleDeeplabv2
    from oracle.step import HYPER, SGDState, ssl_step
    from oracle.weights import det_state_dict
    torch.set_num_threads(nthreads)
    model = OracleDeeplabv2(det_state_dict("resnet50", 6, False, seed=2333), "resnet50", 6, False)
    opt = SGDState(model.parameters(), 0.9, 5e-4)
    batch = synth.make_batch(B=2, H=256, W=256, C=6, k=2048, seed=2333)
    protos, times = batch["prototypes"], []
    for i in range(3):
        t0 = time.time()
        protos = ssl_step(model, opt, protos, batch, 1e-3, HYPER)["prototypes"]
        if i >= 1:
            times.append(time.time() - t0)
    q.put(dict(tiles_per_s=round(4.0 / min(times), 3), s_per_step=round(min(times), 3), timed_steps=len(times), threads=nthreads))


def cpu_baseline(seconds_budget=30.0):
    """The oracle (CPU restatement of the reference, kind "port") on the host cores, as SURVEY 8(d) specifies: BASELINE
    config 1 (B=2, 256x256, fp32), R50-ASPP and R50-PPM, (i) train_src-style and (ii) train_ssl_uem-style steps, 1 warm-up
    + 3 timed steps each on the box's CPU share, one timed step of the ASPP pair pinned to ONE thread, and the ASPP ssl step on
    ALL the host threads the process may use.  `value` is the leg that matches the metric (R50-ASPP ssl step on 512x512 tiles,
    counting source + target)."""
    from oracle import synth
    from oracle.model import OracleDeeplabv2
    from oracle.step import HYPER, SGDState, src_step, ssl_step
    from oracle.weights import det_state_dict
    try:
        visible = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        visible = os.cpu_count() or 1
    avail = usable_cpus()
    threads = min(16, avail)                          # the one-GPU box's CPU share
    t_stop = time.time() + seconds_budget

    def leg(head, style, size, nthreads, warm, timed):
        torch.set_num_threads(nthreads)
        ppm = head == "ppm"
        sd = det_state_dict("resnet50", 6, ppm, seed=2333)
        model = OracleDeeplabv2(sd, "resnet50", 6, ppm)
        opt = SGDState(model.parameters(), 0.9, 5e-4)
        batch = synth.make_batch(B=2, H=size, W=size, C=6, k=2048, seed=2333)
        protos = batch["prototypes"]
        times = []
        for i in range(warm + timed):
            t0 = time.time()
            if style == "ssl":
                protos = ssl_step(model, opt, protos, batch, 1e-3, HYPER)["prototypes"]
            else:
                src_step(model, opt, batch, 1e-3, HYPER)
            if i >= warm:
                times.append(time.time() - t0)
            if time.time() > t_stop and times:
                break
        best = min(times)
        return dict(tiles_per_s=round((4.0 if style == "ssl" else 2.0) / best, 3), s_per_step=round(best, 3),
                    timed_steps=len(times), threads=nthreads)

    headline = leg("aspp", "ssl", 512, threads, 1, 3)
    cfg1 = {}
    for head in ("aspp", "ppm"):
        for style in ("src", "ssl"):
            if time.time() < t_stop:
                cfg1[f"r50-{head} {style} {threads} threads"] = leg(head, style, 256, threads, 1, 3)
    for style in ("src", "ssl"):
        if time.time() < t_stop:
            cfg1[f"r50-aspp {style} 1 thread"] = leg("aspp", style, 256, 1, 0, 1)       # no warm-up: one cold step
    # every host thread the affinity mask shows ("all host threads"): only where the cgroup grants them -- on a one-GPU box
    # the mask shows 256 CPUs and the cgroup grants 16; 256 torch threads on 16 cores take minutes per step (round 3 spent 40 s of
    # the run finding that out again), so that leg is declined up front and says why
    if visible > threads:
        name = f"r50-aspp ssl all {visible} visible host threads"
        if avail >= visible:
            import multiprocessing as mp
            ctx = mp.get_context("spawn")
            q = ctx.Queue()
            p = ctx.Process(target=_all_threads_leg, args=(visible, q), daemon=True)
            p.start()
            p.join(40.0)
            if p.is_alive():
                p.terminate()
                p.join(5.0)
                cfg1[name] = dict(skipped="no result within 40 s")
            else:
                try:
                    cfg1[name] = q.get(timeout=2.0)
                except Exception:                         # noqa: BLE001
                    cfg1[name] = dict(error=f"child exited with code {p.exitcode}")
        else:
            cfg1[name] = dict(skipped=f"the affinity mask shows {visible} CPUs, the cgroup grants {avail}: not run")
    torch.set_num_threads(threads)
    return dict(value=headline["tiles_per_s"], unit="tiles/s", cores=threads, kind="port",
                sample=f"oracle (CPU restatement) ssl_step, R50-ASPP, 2 source + 2 target 512x512 tiles per step, fp32, best of "
                       f"{headline['timed_steps']} steps ({headline['s_per_step']:.3f} s/step), {visible} host CPUs in the affinity mask, "
                       f"{avail} granted by the cgroup; config1 = BASELINE config 1 (B=2, 256x256 tiles: tiles/s counts 256x256 "
                       f"tiles, src = 2 per step, ssl = 4 per step) on {threads} threads, 1 thread and all usable threads",
                config1=cfg1)


class Setup:
    """Model + synthetic batch + optimizer for one configuration; `one_step(i)` runs one iteration."""

    def __init__(self, cfg, rank, world, wrapper_factory):
        from uemda_amd import ops
        from uemda_amd.utils import synth             # seeded synthetic tiles
        from uemda_amd.gast.alignment import Aligner
        from uemda_amd.models.Encoder import Deeplabv2
        from uemda_amd.optim import FusedSGD
        from uemda_amd.step import HYPER, StepState, src_step, ssl_step
        from uemda_amd.utils.tools import lr_poly, lr_warmup, seed_torch
        self.cfg, self.world = cfg, world
        C, B, S = 6, cfg.batch, cfg.size
        seed_torch(2333)
        mcfg = dict(backbone=dict(resnet_type=cfg.model, output_stride=16, pretrained=False), multi_layer=True,
                    cascade=False, use_ppm=(cfg.head == "ppm"), ppm=dict(num_classes=C, use_aux=False, fc_dim=2048),
                    inchannels=2048, num_classes=C, is_ins_norm=True)
        self.model = Deeplabv2(mcfg).cuda().set_storage(cfg.storage)   # random init of the reference's architecture (no checkpoints)
        self.wrapper = wrapper_factory(self.model) if wrapper_factory is not None else None
        # synthetic tiles: every tile of a batch is its own seeded tile (no repeats), generated on the host (2 s per 32
        # tiles) and resident in HBM before the timed region; `nbatches` distinct batches alternate step by step, so that no two
        # consecutive steps see the same data
        data_rank = rank if cfg.data_rank is None else cfg.data_rank
        nb = max(1, int(getattr(cfg, "unique_batches", 2)))
        self.batches = []
        for j in range(nb):
            pool = synth.make_batch(B=B, H=S, W=S, C=C, k=2048, seed=2333 + data_rank + 1000 * j)
            self.batches.append({k: v.cuda().contiguous() for k, v in pool.items()})
        self.batch = self.batches[0]
        self.aligner = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
        # replicas start from the same prototypes: seeded independently of the rank, then broadcast
        self.aligner.prototypes = synth.make_batch(B=1, H=32, W=32, C=C, k=2048, seed=2333)["prototypes"].cuda().contiguous()
        from uemda_amd import dp as udp
        udp.broadcast_flat(self.aligner.prototypes)
        self.opt = FusedSGD(self.model, lr=HYPER["lr"], momentum=HYPER["momentum"], weight_decay=HYPER["weight_decay"])
        self.state = StepState(C)
        self.sup_ignore = (S // 16) * (S // 16)            # explicit ignored superpixel id (DP-safe)
        self.tiles_per_step = (2 * B if cfg.workload == "ssl" else B) * world
        stop_steps = 6000

        def lr_at(i):                                 # train_ssl_uem.py:82-84 + tools.py:191-207
            pre = int(stop_steps / 20)
            return lr_warmup(HYPER["lr"], i, pre) if i < pre else lr_poly(HYPER["lr"], i, stop_steps * 1.5, 0.9)
        self.lr_at = lr_at
        self._ssl, self._src = ssl_step, src_step

    def one_step(self, i, mark=None):
        batch = self.batches[i % len(self.batches)]
        if self.cfg.workload == "ssl":
            return self._ssl(self.model, self.aligner, self.opt, self.state, batch, self.lr_at(i + 1), dp=self.wrapper,
                             sup_ignore_id=self.sup_ignore, mark=mark)
        return self._src(self.model, self.opt, self.state, batch, self.lr_at(i + 1), dp=self.wrapper)


def roofline_of(prof, peak, traffic_for=None):
    """bench `roofline` object from the per-launch HIP events of one step: the family with the most time.  `achieved` and `frac` are
    what the matrix pipe executes (a fraction of its peak: never above 1); the algorithmic rate -- what the step gets done per second,
    Winograd's saved multiplies counted -- stands beside it as `algorithmic_tflops`."""
    if not prof:
        return None
    fam, agg = max(prof.items(), key=lambda kv: kv[1]["ms"])
    exe = agg["executed"] / (agg["ms"] * 1e-3) / 1e12
    alg = agg["flops"] / (agg["ms"] * 1e-3) / 1e12
    frac = exe / peak
    assert frac <= 1.0, f"roofline fraction {frac} of {fam}: executed flops cannot exceed the peak"
    return dict(bound="mfma", kernel=fam, achieved=round(exe, 2), peak=peak, unit="TFLOP/s", frac=round(frac, 4),
                algorithmic_tflops=round(alg, 2), traffic=traffic_for(fam) if traffic_for else None,
                launches_per_step=agg["launches"], avg_launch_ms=round(agg["ms"] / agg["launches"], 4),
                executed_gflop_per_launch=round(agg["executed"] / agg["launches"] / 1e9, 3),
                algorithmic_gflop_per_launch=round(agg["flops"] / agg["launches"] / 1e9, 3),
                families={k: dict(executed_tflops=round(v["executed"] / (v["ms"] * 1e-3) / 1e12, 2), frac=round(v["executed"] / (v["ms"] * 1e-3) / 1e12 / peak, 4),
                                  algorithmic_tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2), ms_per_step=round(v["ms"], 2)) for k, v in prof.items()},
                measured="HIP events around every conv launch of the last timed step.  achieved / frac = EXECUTED flops (the multiplies "
                         "issued to the matrix pipe) over the event time: a 3x3 layer on Winograd F(2x2,3x3) / F(4x4,3x3) (input transform + "
                         "16- / 36-position GEMM + output transform inside one event pair) executes 2*M*Cout*Cin*16/4 resp. *36/16 where "
                         "the direct conv executes 2*M*Cout*9*Cin; algorithmic_tflops counts the latter for every conv")


def replay_leg(s, workload, step0, tiles_per_step, nrep=5):
    """The same step captured in ONE hipGraph and replayed (uemda_amd.step.GraphedStep): device time per step and what the host
    spends per step.  An extra leg: never the headline `value`, and never allowed to cost the line."""
    from uemda_amd.step import GraphedStep, ssl_step as _ssl, src_step as _src
    gs, err = None, None
    try:
        kw = dict(sup_ignore_id=s.sup_ignore) if workload == "ssl" else {}
        if s.wrapper is not None:
            kw["dp"] = s.wrapper                   # the data-parallel step: RCCL's all-reduces are captured with it
        # data parallel: NO eager warm-up step inside the constructor (a rank that failed there would leave its peers inside a
        # real gradient all-reduce; the timed steps above were the warm-up -- same shapes, the optimizer has stepped); the capture itself
        # executes no collective, and the ranks agree on its success right after
        gs = GraphedStep(_ssl if workload == "ssl" else _src, s.model, s.aligner if workload == "ssl" else None, s.opt, s.state,
                         s.batch, warmup=0 if s.wrapper is not None else 1, lr=s.lr_at(step0), **kw)
    except Exception as e:                    # noqa: BLE001
        err = repr(e)[:300]
    # data parallel: replay only a graph that EVERY rank holds (a rank replaying alone would launch collectives nobody joins); the
    # ranks agree through one eager all-reduce of a flag, and fall back to "no graph leg" together
    if s.wrapper is not None and not GraphedStep.all_ranks_ok(gs is not None):
        s.opt.lr_device = None
        del gs
        gc.collect()
        torch.cuda.empty_cache()
        return dict(skipped="the capture did not succeed on every rank: no rank replays", error_on_this_rank=err)
    if gs is None:
        s.opt.lr_device = None
        gc.collect()
        torch.cuda.empty_cache()
        return dict(error=err)
    try:
        gs(s.lr_at(step0))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(nrep):
            gs(s.lr_at(step0 + 1 + i))
        t_host = (time.perf_counter() - t1) / nrep
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / nrep
        gs.check()
        leg = dict(value=round(tiles_per_step / dt, 3), unit="tiles/s", ms_per_step=round(1e3 * dt, 2), steps=nrep,
                   host_ms_per_step=round(1e3 * t_host, 3),
                   note="the whole step (2 forwards, mining, losses, backward, " + ("gradient all-reduce, " if s.wrapper is not None else "") +
                        "clip + SGD) as ONE hipGraph launch per step; lr travels as a device scalar")
        del gs
    except Exception as e:                    # noqa: BLE001
        leg = dict(error=repr(e)[:300])
    s.opt.lr_device = None
    gc.collect()
    torch.cuda.empty_cache()
    return leg


def short_leg(cfg, steps=6, warmup=2):
    """One of the `other_configs`: fresh model, `warmup` untimed + `steps` timed steps as shipped (weight gradients on the side stream,
    the two graphs on two streams), then ONE more step, outside the timed region, with per-launch events for the family fractions
    (that step runs one kernel at a time -- ops.in_backward, step.forward_pair -- so that the fractions mean something; until
    round 6 it was the last of the timed steps and put a sixth of its serial time into ms_per_step)."""
    from uemda_amd import ops
    s = Setup(cfg, 0, 1, None)
    for i in range(warmup):
        s.one_step(i)
    ops.PROF.enabled = False
    ops.PROF.records = []
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for i in range(steps):
        s.one_step(warmup + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ops.PROF.enabled = True
    s.one_step(warmup + steps)
    torch.cuda.synchronize()
    ops.PROF.enabled = False
    bf16 = cfg.storage == "bf16"
    out = dict(value=round(s.tiles_per_step / dt, 2), unit=f"tiles({cfg.size}x{cfg.size})/s", ms_per_step=round(1e3 * dt, 2), steps=steps,
               warmup=warmup, dtype="bf16 storage (fp32 accumulate / statistics / master weights)" if cfg.storage == "bf16" else "f32",
               workload=f"{cfg.model}-{cfg.head} {cfg.workload} step, {cfg.batch} source + {cfg.batch if cfg.workload == 'ssl' else 0} target {cfg.size}x{cfg.size} tiles",
               roofline=roofline_of(ops.PROF.summary(), BF16_MATRIX_PEAK_TFLOPS if bf16 else F32_MATRIX_PEAK_TFLOPS),
               peak_allocated_GB=round(torch.cuda.max_memory_allocated() / 1e9, 1),
               peak_reserved_GB=round(torch.cuda.max_memory_reserved() / 1e9, 1),
               profiled_step="one extra step after the timed ones (per-launch events, one kernel at a time): not in ms_per_step")
    ops.PROF.records = []
    if cfg.storage == "bf16" and cfg.size <= 512 and not getattr(cfg, "no_hipgraph", False):
        out["hipgraph"] = replay_leg(s, cfg.workload, warmup + steps + 1, s.tiles_per_step)     # where the host's share was largest (16 of 45 ms)
    del s
    gc.collect()
    torch.cuda.empty_cache()
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` outside torchrun: start the N rank processes -- fresh children of a parent that has made no GPU
    call -- through `torch.distributed.run` (one rank per GPU, rendezvous on 127.0.0.1), forward their output (rank 0 prints the
    JSON line) and return the launcher's exit code, which is non-zero when any rank failed."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver supports dmabuf IPC only (RCCL across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    note(f"--gpus {n} without a torchrun environment: launching {n} ranks ({' '.join(cmd[1:8])} ...)")
    return subprocess.call(cmd, env=env)


def launch_check(args):
    """`--launch-check`: the rank plumbing of this file WITHOUT the model -- process group, barrier, K timed no-op steps, MAX of the
    elapsed time over the ranks, rank 0's one JSON line -- so that `--gpus N` meaning N ranks can be tested where there is no GPU
    (tests/test_dp_gloo.py, gloo).  It measures nothing: `value` is null."""
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if os.environ.get("UEM_LAUNCH_CHECK_FAIL_RANK") == str(rank):          # the test of "a failed rank fails the launcher"
        raise SystemExit(f"launch check: rank {rank} was told to fail")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    ranks = torch.zeros(world, dtype=torch.float64)
    ranks[rank] = 1.0
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(ranks)
    if rank == 0:
        print(json.dumps({"metric": "launch check (no model, no GPU)", "value": None, "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(1e3 * float(t) / max(1, args.steps), 3),
                          "ranks_seen": int(ranks.sum()), "launch_check": True}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU source batch (= target batch)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--workload", default="ssl", choices=["ssl", "src"])
    ap.add_argument("--model", default="resnet50", choices=["resnet50", "resnet101"])
    ap.add_argument("--head", default="aspp", choices=["aspp", "ppm"])
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--device", type=int, default=None, help="force the CUDA device index (rehearsal on one GPU)")
    ap.add_argument("--no-overlap", action="store_true", help="all-reduce the whole gradient arena after backward")
    ap.add_argument("--dump-params", default="", help="write a parameter checksum after the run (DP rehearsals); with "
                    "<path>.rank<r>.pt: prototypes, post-run parameter samples and the FIRST step's reduced gradient")
    ap.add_argument("--data-rank", type=int, default=None, help="use this rank's synthetic tiles (single-process leg of the "
                    "data-parallel test: the gradient rank r contributes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--dump-conv-events", default=None, help="write the per-launch conv timings of the last timed step to this JSON file")
    ap.add_argument("--storage", default="fp32", choices=["fp32", "bf16"],
                    help="bf16: activations / weight copies of the encoder stored in bf16, bf16 matrix cores (BASELINE config 5)")
    ap.add_argument("--no-other-precisions", action="store_true", help="accepted and ignored (the operand-precision legs are retired)")
    ap.add_argument("--unique-batches", type=int, default=2, help="distinct seeded batches alternating step by step")
    ap.add_argument("--no-hipgraph", action="store_true", help="skip the short leg that replays the step as one hipGraph")
    ap.add_argument("--hipgraph-dp", action="store_true", help="accepted and ignored: the hipGraph leg now runs under data parallel by default "
                    "(nccl backend: the all-reduces are captured; the ranks agree on the capture's success before anyone replays)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short legs for the other BASELINE configurations (bf16 storage, PPM head, R101 1024^2; N=1 only)")
    ap.add_argument("--launch-check", action="store_true", help="rank plumbing only (process group, barrier, max-over-ranks timing, one "
                    "JSON line from rank 0) without the model or a GPU: the CPU test of `--gpus N`")
    args = ap.parse_args()

    # `--gpus N` means N ranks.  Under torchrun (the driver's launch line) WORLD_SIZE says so already; a bare `python bench.py --gpus N`
    # starts the ranks itself, as fresh children, BEFORE this process makes any GPU call -- and never prints an n_gpus: 1 line for it.
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        sys.exit(launch_ranks(args.gpus))
    if int(env_world or "1") != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world or 1}: launch one rank per GPU "
                         f"(python bench.py --gpus N starts them itself when no torchrun environment is set)")
    if args.launch_check:
        return launch_check(args)

    from uemda_amd import dp as udp, ops
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    cores = pin_cpus(int(os.environ.get("LOCAL_RANK", "0")), local_world)     # before any thread pool or GPU call
    # what the rank can actually run on: its slice of the affinity mask capped by the cgroup's CPU grant (a one-GPU box shows 256 CPUs
    # in the mask and grants 16 -- the figure cpu_baseline.cores reports too)
    cores = max(1, min(cores, usable_cpus() // max(1, local_world)))
    if args.device is not None:
        torch.cuda.set_device(args.device)
    rank, world, local = udp.init(args.backend, device=args.device)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the process group has {world} ranks")
    torch.cuda.set_device(local if args.device is None else args.device)

    make_wrapper = (lambda m: udp.DataParallel(m, overlap=not args.no_overlap)) if (world > 1 or udp.FORCE) else None
    s = Setup(args, rank, world, make_wrapper)
    model, wrapper, aligner = s.model, s.wrapper, s.aligner
    B, S = args.batch, args.size

    marks = []                                    # (phase name, HIP event) of the last timed step

    def mark(name):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        marks.append((name, ev))

    first_grad = {}

    def grab_first_grad(name):          # --dump-params: the reduced gradient of the very first step, before the optimizer
        if name == "grad_allreduce_wait" and not first_grad:
            _, garena, n = model.flat_parameters()
            scale = 1.0 / world
            first_grad["sample"] = (garena[:n:101].double() * scale).cpu()
            first_grad["norm"] = float((garena[:n].double() * scale).norm())

    def one_step(i, marked=False):
        return s.one_step(i, mark=mark if marked else (grab_first_grad if args.dump_params and not first_grad else None))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(i)
    ops.PROF.enabled = False
    ops.PROF.records = []
    # Host hygiene: the interpreter's cyclic collector runs a full (generation-2) pass every so many allocations; over the
    # model's ~10^5 long-lived objects that pass takes ~60 ms and, landing in the first step after the barrier (empty device
    # queue), stalled the device for as long (steps_ms[0] = 180-220 ms against 134).  Collect now and move the survivors
    # out of the collector's reach; the steady-state loop is unaffected either way (the host runs half a step ahead).
    gc.collect()
    gc.freeze()
    barrier()
    step_events = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    step_host = []
    t0 = time.perf_counter()
    step_events[0].record()
    for i in range(args.steps):
        # per-launch HIP events (the `roofline` leg) are recorded during the LAST timed step only: 660 event
        # records per step put ~10 ms of command-processor bubbles into a 160 ms step when left on throughout
        ops.PROF.enabled = (not args.no_kernel_events) and i == args.steps - 1
        if i == 0 and os.environ.get("UEM_BENCH_PROFILE_FIRST"):
            import cProfile, pstats
            pr = cProfile.Profile(); pr.enable()
            out = one_step(args.warmup + i, marked=False)
            pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(14)
        else:
            out = one_step(args.warmup + i, marked=(i == args.steps - 1))
        step_events[i + 1].record()
        step_host.append(time.perf_counter() - t0)
    barrier()
    elapsed = time.perf_counter() - t0
    ops.PROF.enabled = False
    if args.workload == "ssl":
        aligner.check_superpixel_ids()            # the last step's report (inside the step the check never waits)
    replicas = None
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # are the replicas still one model?  every rank's (parameter, prototype) checksums, gathered; and the slowest rank's wait
        # for the gradient collective in the last step
        arena, _, n = model.flat_parameters()
        wait_ms = 0.0
        ph_local = {b[0]: a[1].elapsed_time(b[1]) for a, b in zip(marks[:-1], marks[1:])} if marks else {}
        wait_ms = ph_local.get("grad_allreduce_wait", 0.0)
        mine = torch.tensor([float(arena[:n].double().sum()), float(arena[:n].double().abs().sum()),
                             float(aligner.prototypes.double().sum()), float(aligner.prototypes.double().abs().sum()), wait_ms],
                            device="cuda", dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu()
        replicas = dict(identical=bool((allr[:, :4] == allr[0, :4]).all()), param_sum=float(allr[0, 0]), proto_sum=float(allr[0, 2]),
                        grad_allreduce_wait_ms_max=round(float(allr[:, 4].max()), 3),
                        grad_allreduce_wait_ms_per_rank=[round(float(v), 3) for v in allr[:, 4]])
    tiles_per_step = s.tiles_per_step
    value = tiles_per_step * args.steps / elapsed
    if rank == 0:
        note(f"headline: {value:.1f} tiles/s, {1e3 * elapsed / args.steps:.2f} ms/step")
    headline_prof = ops.PROF.summary()
    if args.dump_conv_events and rank == 0:
        with open(args.dump_conv_events, "w") as f:
            json.dump(ops.PROF.per_call(), f)
    ops.PROF.records = []

    # everything the line reads off the device is read NOW, before the optional legs: a leg that fails badly (a capture left
    # invalidated took a rehearsal of round 5 down at `float(out["loss_source"])`) must never cost the measured line
    loss_source_f = float(out["loss_source"])
    steps_ms_f = [round(a.elapsed_time(b), 2) for a, b in zip(step_events[:-1], step_events[1:])]
    phases_f = {b[0]: round(a[1].elapsed_time(b[1]), 3) for a, b in zip(marks[:-1], marks[1:])} if marks else {}
    mem_f = torch.cuda.memory_stats()

    graph_leg = None
    # data parallel: RCCL's all-reduces are captured with the step (GraphedStep(dp=...)).  Every rank attempts the capture, the ranks
    # agree on its success (GraphedStep.all_ranks_ok: one eager MIN all-reduce) and replay only a graph all of them hold, so the leg is
    # on by default under N > 1 too (round 4: opt-in); gloo groups (host round trip) have nothing to capture and skip it
    if not args.no_hipgraph and (wrapper is None or wrapper.capturable):
        try:
            graph_leg = replay_leg(s, args.workload, args.warmup + args.steps, tiles_per_step)
        except Exception as e:                    # noqa: BLE001  (the leg's own clean-up failed: report, keep the line)
            graph_leg = dict(error=repr(e)[:300])
        if rank == 0:
            note(f"hipGraph replay: {graph_leg}")

    peak = BF16_MATRIX_PEAK_TFLOPS if args.storage == "bf16" else F32_MATRIX_PEAK_TFLOPS
    prec_text = "fp32 (f32 MFMA)"
    line = None
    if rank == 0:
        # HBM bytes per launch of a family from the PMC passes (scripts/pmc_traffic.py), valid ONLY for the configuration AND the
        # build of the conv kernels they were collected on: anything else reports null
        tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
        key = f"{args.model}-{args.head} {args.workload} B={B} size={S} prec=fp32" + ("" if args.storage == "fp32" else " storage=bf16")
        tj = json.load(open(tfile)) if os.path.exists(tfile) else {}
        traffic_ok = tj.get("config") == key and tj.get("kernel_source_sha256_16") == kernel_source_hash()
        def traffic_of(fam):               # HBM bytes per launch (= per convolution) of the family: PMC bytes per step / launches per step
            per_step = (tj.get("per_step") or {}).get(fam)
            n = headline_prof.get(fam, {}).get("launches")
            return round(per_step / n) if per_step and n else None
        roof = roofline_of(headline_prof, peak, traffic_of if traffic_ok else None)
        if roof is not None and not traffic_ok:
            roof["traffic_note"] = "profiles/traffic_latest.json was taken on another configuration or another build of the conv kernels"
        try:
            metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        except Exception:
            metric = "512x512 tiles/sec (fwd+bwd+pseudo-label) ResNet50-ASPP bs=32, 1/2/4/8 GPU"
        arena_bytes = 4 * model.flat_parameters()[2]
        line = {
            "metric": metric,
            "value": round(value, 3), "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16 storage (fp32 accumulate / statistics / master weights)" if args.storage == "bf16" else "f32",
            "data": "synthetic",
            "config": {"workload": f"train_ssl_uem step ({args.workload}): {args.model}-{args.head} 6-class, per-GPU {B} source + "
                                   f"{B if args.workload == 'ssl' else 0} target {S}x{S} tiles, {prec_text if args.storage == 'fp32' else 'bf16 storage in the encoder (bf16 MFMA, fp32 accumulate)'}, "
                                   f"random init; tiles counted = source + target; every tile of a batch is its own seeded tile, "
                                   f"{len(s.batches)} distinct batches alternate step by step; the last timed step also records "
                                   f"per-launch HIP events and therefore runs one kernel at a time (no side stream, no second graph stream: "
                                   f"about 4 ms longer than the others)",
                       "global_batch": tiles_per_step, "tile": S, "parallelism": f"dp{world}",
                       "collective": None if wrapper is None else f"torch.distributed {args.backend}",
                       "collective_bytes_per_step": None if wrapper is None else arena_bytes + 4 * (6 * 2048 + 6),
                       "host_cores_per_rank": cores},
            "loss_source": round(loss_source_f, 5),
            "roofline": roof,
        }
        if replicas is not None:
            line["replicas_identical"] = replicas["identical"]
            line["replicas"] = replicas
        # device time of every timed step (one HIP event per step boundary) and when the host had finished enqueuing it
        line["steps_ms"] = steps_ms_f
        line["host_enqueued_at_ms"] = [round(1e3 * t, 1) for t in step_host]
        if marks:
            # per-phase wall time of the last timed step (HIP events on the compute stream) and, for the HBM-bound
            # phases, algorithmic bytes (SURVEY 8d per-tile figures x tiles) over that time
            ph = phases_f
            mb = {"label_refine_select": 25.2 * B, "prototype_update": 10.5 * B, "losses_forward": (8.4 + 2.1) * B,
                  "clip_sgd": 5 * 4 * model.flat_parameters()[2] / 1e6}
            line["phases_ms"] = ph
            line["phases_hbm_GBps"] = {k: round(v / 1e3 / (ph[k] * 1e-3), 1) for k, v in mb.items() if ph.get(k, 0) > 0}
            if ph.get("label_refine_select", 0) > 0:
                # the HBM-bound phase the path is named after: Pearson similarity, superpixel segment-max, the fused three-view
                # refinement and pseudo_selection over the B target tiles; SURVEY 8(d): 25.2 MB of algorithmic traffic per tile
                gbps = mb["label_refine_select"] / 1e3 / (ph["label_refine_select"] * 1e-3)
                line["roofline_hbm"] = dict(bound="hbm", kernel="label_refine + pseudo_selection (pearson, segment_max, label_refine, pseudo_select)",
                                            achieved=round(gbps, 1), peak=HBM_PEAK_GBPS, unit="GB/s", frac=round(gbps / HBM_PEAK_GBPS, 4), traffic=None,
                                            algorithmic_mb_per_tile=25.2, tiles=B, ms=ph["label_refine_select"],
                                            measured="HIP events on the compute stream around the phase in the last timed step; bytes = "
                                                     "SURVEY 8(d)'s 25.2 MB per target tile (soft 6.29 + superpixels 2.10 + features 8.39 read, "
                                                     "refined soft 6.29 + hard labels 2.10 written) x tiles")
                if roof is not None:           # mirrored inside `roofline`, the object the driver's parsed record keeps
                    roof["hbm_phase"] = {k: line["roofline_hbm"][k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "ms", "tiles")}
        ms = mem_f
        line["device_memory"] = {"peak_allocated_GB": round(ms.get("allocated_bytes.all.peak", 0) / 1e9, 1),
                                 "peak_reserved_GB": round(ms.get("reserved_bytes.all.peak", 0) / 1e9, 1),
                                 "alloc_retries": int(ms.get("num_alloc_retries", 0))}
        if graph_leg:
            line["hipgraph"] = graph_leg
    if args.dump_params:
        arena, _, n = model.flat_parameters()
        with open(f"{args.dump_params}.rank{rank}", "w") as f:
            f.write(f"{float(arena[:n].double().sum()):.10e} {float(arena[:n].double().abs().sum()):.10e}\n")
        torch.save(dict(params_sample=arena[:n:101].cpu(), params_sum=float(arena[:n].double().sum()),
                        prototypes=aligner.prototypes.cpu(), first_grad_sample=first_grad.get("sample"),
                        first_grad_norm=first_grad.get("norm"), unpaired_forwards=getattr(wrapper, "unpaired_forwards", None)),
                   f"{args.dump_params}.rank{rank}.pt")
    if rank == 0 and world == 1 and not args.no_other_configs and (args.model, args.head, args.storage, S) == ("resnet50", "aspp", "fp32", 512):
        # the other configurations BASELINE.json names, after the headline and without touching it: the headline's model is
        # released first (R101 on 1024^2 tiles needs ~105 GB of the 288)
        del s, model, aligner, one_step, out
        gc.unfreeze()
        gc.collect()
        torch.cuda.empty_cache()
        oc = {}
        legs = {"bf16 storage r50-aspp 512": dict(storage="bf16"), "fp32 r50-ppm 512": dict(head="ppm"),
                "fp32 r50-aspp 512 src (BASELINE config 2: fwd+bwd only)": dict(workload="src"),
                "bf16 storage r101-aspp 1024 (BASELINE config 5, one GPU's share)": dict(storage="bf16", model="resnet101", size=1024)}
        for name, over in legs.items():
            cfg = types.SimpleNamespace(**{**vars(args), "data_rank": None, **over})
            try:
                oc[name] = short_leg(cfg)
            except Exception as e:                # noqa: BLE001  (an extra leg must never cost the headline line)
                oc[name] = dict(error=repr(e)[:300])
                gc.collect()
                torch.cuda.empty_cache()
            note(f"other config {name}: {oc[name].get('value')} {oc[name].get('unit')} {oc[name].get('ms_per_step')} ms {oc[name].get('error', '')}")
        line["other_configs"] = oc
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            note("cpu baseline (oracle on the host cores, ~45 s)")
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as e:                # noqa: BLE001  (reported, never fatal for the measured line)
                line["cpu_baseline"] = dict(error=repr(e)[:200])
        print(json.dumps(line), flush=True)
    if dist.is_available() and dist.is_initialized():
        # also the single-rank group of UEM_DP_FORCE=1: left alive, its RCCL watchdog thread can outlive the HIP context at interpreter
        # exit and abort the process after the line was printed
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
