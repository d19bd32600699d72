"""Data parallelism over the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" =
RCCL over xGMI; "gloo" on CPU for tests).  New relative to the reference, which is single-GPU.

Exchange steps per iteration:
  C1  all-reduce(sum) of the flat fp32 gradient arena (ONE collective over one contiguous buffer, sized for
      xGMI: 98 MB for R50-ASPP), averaged by folding 1/world into the fused clip+SGD kernel
  C2  all-reduce of the prototype partial sums + counts (inside Aligner._class_sums, 49 KB)
BatchNorm statistics stay per rank (per-GPU batch 32 >= the reference's 8).
"""
import os

import torch
import torch.distributed as dist

from . import ops


# UEM_DP_FORCE=1: take the data-parallel code paths (process group, broadcast, bucketed all-reduce) even with a single
# rank -- a one-GPU smoke test of the RCCL plumbing (the collectives are then trivial but real).
FORCE = os.environ.get("UEM_DP_FORCE", "0") != "0"
# One transport: torch.distributed (backend "nccl" IS RCCL on ROCm; "gloo" for the CPU / one-device rehearsals).  Round 3 also had
# a switch (UEM_DP_NATIVE) that sent the gradient collective through the C ABI's uem_allreduce_flat; it could only ever be rehearsed with
# one rank and was a second path to distrust -- removed.  uem_comm_* / uem_allreduce_flat stay in the C ABI for
# hosts without torch.distributed; tests/test_gpu_dp.py drives them directly.


def init(backend=None, device=None):
    """Initialise the default process group from the torchrun environment; returns (rank, world, local_rank).
    `device` overrides LOCAL_RANK as the CUDA device index (single-GPU rehearsals with the gloo backend)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or FORCE) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local if device is None else device)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def broadcast_flat(flat, src=0):
    """Make every replica start from rank `src`'s parameters / buffers."""
    if world_size() > 1 or (FORCE and dist.is_initialized()):
        dist.broadcast(flat, src=src)
    return flat


def allreduce_flat_(flat, async_op=False):
    """Sum-reduce one flat buffer across ranks (the caller folds the 1/world average into the optimizer)."""
    if world_size() > 1:
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)
    return None


class DataParallel:
    """Broadcast at construction, then per iteration `reduce_gradients()` after backward.

    hipGraph: with the "nccl" backend both all-reduces are capturable (RCCL launches kernels on its own stream; the event
    hand-shakes with the compute stream become graph edges), so `uemda_amd.step.GraphedStep(..., dp=wrapper)` captures the whole
    data-parallel step; `capturable` says whether the process group allows it ("gloo" moves the buffer through the host: it does not).

    The gradient all-reduce is split into two buckets of the flat arena and the first one is started while
    backward is still running: parameters sit in the arena in forward order, backward finishes them last-to-
    first, so once `layer3[0]` has finished its backward for every forward of the step, the tail of the arena
    (layer3 + layer4 + heads = 94 % of the R50-ASPP bytes) is final and goes out asynchronously (RCCL runs it on
    its own stream, ordered after the kernels already queued); the head of the arena follows after backward.
    """

    def __init__(self, model, overlap=True):
        self.model = model
        self.world = world_size()
        arena, _, _ = model.flat_parameters()
        broadcast_flat(arena)
        for b in model.buffers():
            if b.is_floating_point():
                broadcast_flat(b)
        self._split = None
        self._pending = None
        self._fwd_calls = 0
        self._bwd_calls = 0
        self._trigger_events = []           # (stream, event) at the trigger of every backward pass but the last (two backward streams)
        self.unpaired_forwards = 0          # train-mode forwards that never saw a backward (diagnostic)
        self._active = self.world > 1 or (FORCE and dist.is_initialized())
        if overlap:
            self._install_overlap()     # also with one rank: the forward/backward pairing is checked either way

    @property
    def capturable(self):
        """can the gradient collective be recorded into a hipGraph?  (no process group: nothing to record; RCCL: yes; gloo: no)"""
        if not self._active:
            return True
        return dist.get_backend() == "nccl"

    # ---- overlap machinery -------------------------------------------------------------------------------
    def _install_overlap(self):
        model = self.model
        try:
            trigger = model.encoder.resnet.layer3[0]
            first = next(trigger.parameters())
        except (AttributeError, StopIteration, IndexError):
            return
        arena, _, n = model.flat_parameters()
        off = (first.data_ptr() - arena.data_ptr()) // arena.element_size()
        if not (0 < off < n):
            return
        self._split = int(off) // 4 * 4
        trigger._uem_after_backward = self._on_trigger_backward
        model.register_forward_pre_hook(self._on_forward)

    def _on_forward(self, module, args):
        if module.training and torch.is_grad_enabled():
            if self._pending is not None:
                raise ops.UemError("DataParallel: train-mode forward after the early gradient bucket went out; gradients of "
                                   "this forward would miss it -- run every forward of a step before its backward, "
                                   "or construct DataParallel(model, overlap=False)")
            self._fwd_calls += 1

    def _on_trigger_backward(self):
        self._bwd_calls += 1
        if self._bwd_calls > self._fwd_calls:
            raise ops.UemError(f"DataParallel: {self._bwd_calls} backward passes through the model since the last "
                               f"reduce_gradients() but only {self._fwd_calls} train-mode forwards were counted")
        last = self._bwd_calls == self._fwd_calls
        two = bool(ops._FWD2) and torch.cuda.is_available()      # (also while capturing: events between capturing streams are graph edges)
        if two and not last:
            # the step's graphs may run their backward chains on two streams (ops, "two streams"): remember where this one stood
            ev = torch.cuda.Event()
            ev.record()
            self._trigger_events.append((torch.cuda.current_stream(), ev))
        if self._active and self._split is not None and self._pending is None and last:
            if two:
                cur = torch.cuda.current_stream()
                for s, ev in self._trigger_events:
                    if s != cur:
                        cur.wait_event(ev)          # the other graph's chain is past its trigger too ...
                fold = getattr(self.model, "fold_shadow_grads", None)
                if fold is not None:
                    fold(lo=self._split, synced=True)   # ... and its share of the bucket joins the arena before the bucket goes out
            _, garena, n = self.model.flat_parameters()
            self._pending = dist.all_reduce(garena[self._split:], op=dist.ReduceOp.SUM, async_op=True)
        if last:
            self._trigger_events.clear()

    def reduce_gradients(self):
        """all-reduce(sum) of the gradient arena; returns the prescale (1/world) for FusedSGD.step."""
        _, garena, _ = self.model.flat_parameters()
        ops.grad_join()            # side-stream / second-stream gradients land before the arena goes out (also after a backward that raised)
        if self._split is not None and self._fwd_calls != self._bwd_calls:
            # a train-mode forward whose graph never ran backward (a validation pass left in .train(), a dropped
            # output): the early bucket was (rightly) not sent, nothing is wrong with THIS step's gradients, but
            # left alone the counters would pair the next step's backward passes with stale forwards.  The whole
            # arena goes out below and the counters start afresh.
            if self._pending is not None:
                raise ops.UemError("DataParallel: the early gradient bucket was sent before every backward pass had "
                                   f"run ({self._fwd_calls} forwards, {self._bwd_calls} backwards)")
            self.unpaired_forwards += self._fwd_calls - self._bwd_calls
        if self._active:
            if self._pending is not None:
                dist.all_reduce(garena[:self._split], op=dist.ReduceOp.SUM)
                self._pending.wait()
            else:
                dist.all_reduce(garena, op=dist.ReduceOp.SUM)
        self._pending = None
        self._fwd_calls = 0
        self._bwd_calls = 0
        self._trigger_events.clear()
        return 1.0 / self.world
