#!/usr/bin/env python3
"""Benchmark of the UemDA hot path on MI355X: one `train_ssl_uem.py`-style iteration per step
(2 forwards, label_refine + pseudo_selection, prototype EMA, CE + UVEM, backward, clip + SGD) on synthetic
512x512 tiles, ResNet50-ASPP, per-GPU batch 32 source + 32 target tiles (BASELINE.json configs[2] = configs[1]
plus the mining the metric names).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` = tiles (source + target) per second over all ranks, inputs resident in
HBM when the timed region starts.  `roofline` is for the dominant kernel family (the f32-MFMA implicit-GEMM
convolution), from HIP events around every one of its launches inside the timed region.  `cpu_baseline` is the
oracle (CPU restatement of the reference, `kind: "port"`) timed on the host cores at BASELINE config 1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

F32_MATRIX_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, spec
BF16_MATRIX_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA (the peak of the opt-in "bf16" operand mode)
GFLOP_PER_TILE = {("resnet50", "aspp", 512): (66.66, 198.8)}       # BASELINE.md section 3 (fwd, fwd+bwd)


def cpu_baseline(seconds_budget=30.0):
    """The oracle (CPU restatement of the reference, kind "port") on the host cores, as SURVEY 8(d) specifies: BASELINE
    config 1 (B=2, 256x256, fp32), R50-ASPP and R50-PPM, (i) train_src-style and (ii) train_ssl_uem-style steps, 1 warm-up
    + 3 timed steps each on all of the box's CPU share, plus one timed step of the ASPP pair pinned to ONE thread.
    `value` is the leg that matches the metric (R50-ASPP ssl step on 512x512 tiles, counting source + target)."""
    from oracle import synth
    from oracle.model import OracleDeeplabv2
    from oracle.step import HYPER, SGDState, src_step, ssl_step
    from oracle.weights import det_state_dict
    threads = min(16, os.cpu_count() or 1)            # the one-GPU box's CPU share
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
    torch.set_num_threads(threads)
    return dict(value=headline["tiles_per_s"], unit="tiles/s", cores=threads, kind="port",
                sample=f"oracle (CPU restatement) ssl_step, R50-ASPP, 2 source + 2 target 512x512 tiles per step, fp32, best of "
                       f"{headline['timed_steps']} steps ({headline['s_per_step']:.3f} s/step), {os.cpu_count()} host CPUs visible; "
                       f"config1 = BASELINE config 1 (B=2, 256x256 tiles: tiles/s counts 256x256 tiles, src = 2 per step, "
                       f"ssl = 4 per step)",
                config1=cfg1)


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
    ap.add_argument("--conv-prec", default="fp32", choices=["fp32", "mixed", "bf16x3", "bf16"],
                    help="matrix-core operand precision of the convolutions (fp32 = exact f32 MFMA, the parity default)")
    ap.add_argument("--storage", default="fp32", choices=["fp32", "bf16"],
                    help="bf16: activations / weight copies of the encoder stored in bf16, bf16 matrix cores (BASELINE config 5)")
    ap.add_argument("--no-other-precisions", action="store_true",
                    help="skip the short extra legs that time the same step in the two opt-in precisions (N=1 only)")
    args = ap.parse_args()

    from uemda_amd import dp as udp, ops
    if args.device is not None:
        torch.cuda.set_device(args.device)
    rank, world, local = udp.init(args.backend, device=args.device)
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local if args.device is None else args.device)
    from uemda_amd.utils import synth             # seeded synthetic tiles
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.models.Encoder import Deeplabv2
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import HYPER, StepState, src_step, ssl_step
    from uemda_amd.utils.tools import lr_poly, lr_warmup, seed_torch

    ops.set_conv_precision(args.conv_prec)
    C, B, S = 6, args.batch, args.size
    seed_torch(2333)
    cfg = dict(backbone=dict(resnet_type=args.model, output_stride=16, pretrained=False), multi_layer=True,
               cascade=False, use_ppm=(args.head == "ppm"), ppm=dict(num_classes=C, use_aux=False, fc_dim=2048),
               inchannels=2048, num_classes=C, is_ins_norm=True)
    model = Deeplabv2(cfg).cuda().set_storage(args.storage)   # random init of the reference's architecture (no checkpoints)
    wrapper = udp.DataParallel(model, overlap=not args.no_overlap) if (world > 1 or udp.FORCE) else None
    # synthetic tiles: a small seeded pool generated on the host, tiled to the batch on the device
    data_rank = rank if args.data_rank is None else args.data_rank
    pool = synth.make_batch(B=min(B, 4), H=S, W=S, C=C, k=2048, seed=2333 + data_rank)
    rep = (B + pool["images_s"].shape[0] - 1) // pool["images_s"].shape[0]
    batch = {k: (v.cuda().repeat((rep,) + (1,) * (v.dim() - 1))[:B].contiguous() if k != "prototypes" else v.cuda())
             for k, v in pool.items()}
    aligner = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
    # replicas start from the same prototypes: seeded independently of the rank, then broadcast
    aligner.prototypes = synth.make_batch(B=1, H=32, W=32, C=C, k=2048, seed=2333)["prototypes"].cuda().contiguous()
    udp.broadcast_flat(aligner.prototypes)
    opt = FusedSGD(model, lr=HYPER["lr"], momentum=HYPER["momentum"], weight_decay=HYPER["weight_decay"])
    state = StepState(C)
    sup_ignore = (S // 16) * (S // 16)            # explicit ignored superpixel id (DP-safe)
    stop_steps = 6000

    def lr_at(i):                                 # train_ssl_uem.py:82-84 + tools.py:191-207
        pre = int(stop_steps / 20)
        return lr_warmup(HYPER["lr"], i, pre) if i < pre else lr_poly(HYPER["lr"], i, stop_steps * 1.5, 0.9)

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
        if args.workload == "ssl":
            return ssl_step(model, aligner, opt, state, batch, lr_at(i + 1), dp=wrapper, sup_ignore_id=sup_ignore,
                            mark=mark if marked else (grab_first_grad if args.dump_params and not first_grad else None))
        return src_step(model, opt, state, batch, lr_at(i + 1), dp=wrapper)

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
    import gc
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
    prof_steps = 1
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tiles_per_step = (2 * B if args.workload == "ssl" else B) * world
    value = tiles_per_step * args.steps / elapsed

    others = None
    if world == 1 and not args.no_other_precisions and args.storage == "fp32":
        # the same step in the other matrix-core precisions (2 untimed + 3 timed steps each); `value` above is untouched
        others = {}
        for prec in ("fp32", "mixed", "bf16x3", "bf16"):
            if prec == args.conv_prec:
                continue
            try:                                  # an opt-in leg must never cost the headline line
                ops.set_conv_precision(prec)
                for i in range(2):
                    one_step(args.warmup + args.steps + i)
                barrier()
                t1 = time.perf_counter()
                for i in range(3):
                    one_step(args.warmup + args.steps + 2 + i)
                barrier()
                dt = (time.perf_counter() - t1) / 3
                others[prec] = dict(value=round(tiles_per_step / dt, 3), unit="tiles/s", ms_per_step=round(1e3 * dt, 2), steps=3)
            except Exception as e:                # noqa: BLE001
                others[prec] = dict(error=repr(e)[:200])
        ops.set_conv_precision(args.conv_prec)

    peak = BF16_MATRIX_PEAK_TFLOPS if (args.conv_prec == "bf16" or args.storage == "bf16") else F32_MATRIX_PEAK_TFLOPS
    prec_text = {"fp32": "fp32 (f32 MFMA)", "bf16x3": "fp32 storage, 3xbf16 split MFMA with fp32 accumulate",
                 "mixed": "fp32 (f32 MFMA) forward, 3xbf16 split MFMA data/weight gradients",
                 "bf16": "fp32 storage, bf16 MFMA operands with fp32 accumulate"}[args.conv_prec]
    if rank == 0:
        roof = None
        prof = ops.PROF.summary()
        if prof:
            fam, agg = max(prof.items(), key=lambda kv: kv[1]["ms"])
            ach = agg["flops"] / (agg["ms"] * 1e-3) / 1e12
            # HBM bytes per launch of that family from the PMC passes (scripts/pmc_traffic.py), valid ONLY for the
            # configuration they were collected on: any other run reports null
            traffic = None
            tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
            key = f"{args.model}-{args.head} {args.workload} B={B} size={S} prec={args.conv_prec}" + ("" if args.storage == "fp32" else " storage=bf16")
            if os.path.exists(tfile):
                tj = json.load(open(tfile))
                if tj.get("config") == key:
                    traffic = tj.get(fam)
            roof = dict(bound="mfma", kernel=fam, achieved=round(ach, 2), peak=peak, unit="TFLOP/s",
                        frac=round(ach / peak, 4), traffic=traffic,
                        launches_per_step=agg["launches"] // prof_steps,
                        avg_launch_ms=round(agg["ms"] / agg["launches"], 4),
                        algorithmic_gflop_per_launch=round(agg["flops"] / agg["launches"] / 1e9, 3),
                        families={k: dict(tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                          ms_per_step=round(v["ms"] / prof_steps, 2)) for k, v in prof.items()},
                        measured="HIP events around every conv launch of the last timed step")
        try:
            metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        except Exception:
            metric = "512x512 tiles/sec (fwd+bwd+pseudo-label) ResNet50-ASPP bs=32, 1/2/4/8 GPU"
        line = {
            "metric": metric,
            "value": round(value, 3), "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16 storage (fp32 accumulate / statistics / master weights)" if args.storage == "bf16" else {"fp32": "f32", "mixed": "f32 fwd / 3xbf16 split bwd", "bf16x3": "f32 (3xbf16 split)", "bf16": "bf16"}[args.conv_prec],
            "data": "synthetic",
            "config": {"workload": f"train_ssl_uem step ({args.workload}): {args.model}-{args.head} 6-class, per-GPU {B} source + "
                                   f"{B if args.workload == 'ssl' else 0} target {S}x{S} tiles, {prec_text if args.storage == 'fp32' else 'bf16 storage in the encoder (bf16 MFMA, fp32 accumulate)'}, "
                                   f"random init; tiles counted = source + target; the batch repeats {min(B, 4)} unique seeded "
                                   f"tiles x{rep} (no kernel is value-dependent); the last timed step also records per-launch HIP "
                                   f"events (660 event records: about 1 ms of command-processor bubbles)",
                       "global_batch": tiles_per_step, "tile": S, "parallelism": f"dp{world}",
                       "collective": None if wrapper is None else ("uem_allreduce_flat (RCCL through the C ABI)" if wrapper.native
                                                                   is not None else f"torch.distributed {args.backend}")},
            "loss_source": round(float(out["loss_source"]), 5),
            "roofline": roof,
        }
        # device time of every timed step (one HIP event per step boundary) and when the host had finished enqueuing it
        line["steps_ms"] = [round(a.elapsed_time(b), 2) for a, b in zip(step_events[:-1], step_events[1:])]
        line["host_enqueued_at_ms"] = [round(1e3 * t, 1) for t in step_host]
        if marks:
            # per-phase wall time of the last timed step (HIP events on the compute stream) and, for the HBM-bound
            # phases, algorithmic bytes (SURVEY 8d per-tile figures x tiles) over that time
            ph = {b[0]: round(a[1].elapsed_time(b[1]), 3) for a, b in zip(marks[:-1], marks[1:])}
            mb = {"label_refine_select": 25.2 * B, "prototype_update": 10.5 * B, "losses_forward": (8.4 + 2.1) * B,
                  "clip_sgd": 5 * 4 * model.flat_parameters()[2] / 1e6}
            line["phases_ms"] = ph
            line["phases_hbm_GBps"] = {k: round(v / 1e3 / (ph[k] * 1e-3), 1) for k, v in mb.items() if ph.get(k, 0) > 0}
        ms = torch.cuda.memory_stats()
        line["device_memory"] = {"peak_allocated_GB": round(ms.get("allocated_bytes.all.peak", 0) / 1e9, 1),
                                 "peak_reserved_GB": round(ms.get("reserved_bytes.all.peak", 0) / 1e9, 1),
                                 "alloc_retries": int(ms.get("num_alloc_retries", 0))}
        if others:
            line["other_precisions"] = others
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline()
            except Exception as e:                # noqa: BLE001  (reported, never fatal for the measured line)
                line["cpu_baseline"] = dict(error=repr(e)[:200])
        print(json.dumps(line), flush=True)
    if args.dump_params:
        arena, _, n = model.flat_parameters()
        with open(f"{args.dump_params}.rank{rank}", "w") as f:
            f.write(f"{float(arena[:n].double().sum()):.10e} {float(arena[:n].double().abs().sum()):.10e}\n")
        torch.save(dict(params_sample=arena[:n:101].cpu(), params_sum=float(arena[:n].double().sum()),
                        prototypes=aligner.prototypes.cpu(), first_grad_sample=first_grad.get("sample"),
                        first_grad_norm=first_grad.get("norm"), unpaired_forwards=getattr(wrapper, "unpaired_forwards", None)),
                   f"{args.dump_params}.rank{rank}.pt")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
