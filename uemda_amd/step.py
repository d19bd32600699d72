"""One training iteration of the hot path, reproducing the caller's order and hyper-parameters:
  ssl_step : reference tools/train_ssl_uem.py:193-232
  src_step : reference tools/train_src.py:112-141
Used by bench.py, __graft_entry__.smoke() and the parity tests; a reference-style script can equally call
the same operator surface itself."""
from .gast.balance import CrossEntropy, UVEMLoss, loss_calc_uvem
from .utils.tools import loss_calc

HYPER = dict(lr=1e-2, momentum=0.9, weight_decay=5e-4, max_norm=32.0, cutoff_top=0.8, cutoff_low=0.6,
             refine_mode="all", refine_temp=2.0, uvem_m=0.2, uvem_t=0.7, uvem_g=4.0, proto_decay=0.996,
             ignore_label=-1)


def _ops():
    from . import ops
    return ops


class StepState:
    """loss objects the reference builds once before its loop (train_ssl_uem.py:129-137)."""

    def __init__(self, class_num=6, hp=HYPER):
        self.hp = hp
        self.loss_fn_s = CrossEntropy(ignore_label=hp["ignore_label"])
        self.loss_fn_t = UVEMLoss(m=hp["uvem_m"], threshold=hp["uvem_t"], gamma=hp["uvem_g"], class_num=class_num,
                                  ignore_label=hp["ignore_label"])


def _fork_wanted(model):
    """the host-side half of _may_fork: the switch, two training forwards behind the model (its derived filter banks exist), a model
    whose BatchNorm layers allow it (Deeplabv2.two_stream_ok), a graph being recorded, no per-launch event timing"""
    import torch
    from . import ops
    return bool(ops.TWO_STREAM_FWD and getattr(model, "_uem_train_forwards", 0) >= 2 and hasattr(model, "two_stream_ok") and model.two_stream_ok()
                and torch.is_grad_enabled() and not ops.PROF.enabled)


def _may_fork(model):
    """may this step's second train-mode forward run on its own stream?  (the predicate of forward_pair, without side effects)"""
    import torch
    from . import ops
    if not _fork_wanted(model):
        return False
    arena = getattr(model, "_arena", None)
    if arena is None or not arena.is_cuda:
        return False
    if torch.cuda.is_current_stream_capturing():
        # inside a hipGraph capture the fork becomes graph edges (ops.GRAPH_TWO_STREAM); the second stream, the shadow gradient arena
        # and the full two-stream backward must exist from an eager forked step (nothing may be allocated for them while capturing)
        # (not together with a captured side stream, ops.GRAPH_SIDE: ending such a capture -- side-stream forks off two branches --
        # crashed inside the HIP runtime; each form alone captures and replays)
        return bool(ops.GRAPH_TWO_STREAM and not ops.GRAPH_SIDE and ops.TWO_STREAM_BWD and ops._FWD2.get(torch.cuda.current_device()) is not None
                    and getattr(model, "_grad_arena2", None) is not None)
    return True


def forward_pair(model, images_a, images_b, join=True):
    """The two train-mode forwards of a step, `model(images_a)` then `model(images_b)` (tools/train_ssl_uem.py:205-207), the second on
    its own stream beside the first when nothing forbids it (ops.TWO_STREAM_FWD; Deeplabv2.two_stream_ok): the outputs, the BatchNorm
    running statistics and num_batches_tracked are those of the sequential pair, bit for bit.  The second graph's backward nodes run
    on the second stream too and accumulate into the model's shadow gradient arena, folded into .grad at the end of the backward
    pass (ops, "two streams"; `ops.grad_join()` is what a caller who reads .grad right after backward may call).  Sequential under per-launch
    event timing, while a hipGraph is being captured unless an eager forked step came before (ops.GRAPH_TWO_STREAM), and until the model has taken two training forwards (its derived
    filter banks exist from then on).
    join=False (ssl_step's two pipelines): returns (out_a, out_b, finish); out_b is then still in flight on `ops.second_stream()`, and
    `finish()` -- current stream behind the second stream, the second forward's running statistics applied -- is the caller's to call
    before anyone reads the statistics or touches out_b on another stream (finish is None when the pair ran sequentially)."""
    import torch
    from . import ops
    two = _may_fork(model)
    model._uem_train_forwards = getattr(model, "_uem_train_forwards", 0) + 2
    if not two:
        out = model(images_a), model(images_b)
        return out if join else out + (None,)
    main = torch.cuda.current_stream()
    second = ops.second_stream()
    ops.PREP.refresh_all()                                  # every derived filter bank fresh before the fork ...
    if getattr(model.encoder, "storage", "fp32") == "bf16":
        from . import ops_bf16
        ops_bf16.weight(next(p for p in model.encoder.parameters() if p.dim() == 4))    # ... and the bf16 copy of the weight arena
    ops.set_backward_stream(main)
    second.wait_stream(main)
    out_a = model(images_a)
    with model.shadow_running_stats():
        with torch.cuda.stream(second):
            out_b = model(images_b)

    def finish():
        torch.cuda.current_stream().wait_stream(second)
        model.apply_shadow_running_stats()
    if join:
        finish()
        return out_a, out_b
    return out_a, out_b, finish


def _ssl_step_two_pipelines(model, aligner, optimizer, state, batch, lr, dp, sup_ignore_id):
    """ssl_step with the source graph's whole pipeline (forward, loss, backward) on the caller's stream and the target graph's
    (forward, label refinement and selection, loss, backward) on the second stream (ops, "two streams").  Same kernels on the same
    inputs as the sequential step; what the two pipelines share is ordered explicitly:
      * the prototypes: refined against (second stream) BEFORE the source features update them (caller's stream) -- :209-216's order;
      * gradients: the caller's arena / the shadow arena, folded when the second backward pass ends; zero_grad comes first (it joins);
      * BatchNorm running statistics: the second forward's contribution is applied after the join, source batch first;
      * data parallel: DataParallel's early bucket waits for both graphs' triggers (dp._on_trigger_backward)."""
    import torch
    from . import ops
    hp = state.hp
    optimizer.zero_grad()                                   # before the fork: it waits for every stream the last step used
    main, second = torch.cuda.current_stream(), ops.second_stream()
    (pred_s1, pred_s2, feat_s), (pred_t1, pred_t2, feat_t), finish = forward_pair(model, batch["images_s"], batch["images_t"], join=False)
    if finish is None:
        raise ops.UemError("ssl_step: the forward pair declined to fork after _may_fork said it would")
    with torch.cuda.stream(second):
        soft, hard = aligner.refine_and_select(batch["label_t_sup"], feat_t, [pred_t1, pred_t2], batch["label_t_soft"],
                                               mode=hp["refine_mode"], temp=hp["refine_temp"], cutoff_top=hp["cutoff_top"],
                                               cutoff_low=hp["cutoff_low"], sup_ignore_id=sup_ignore_id)   # :209-214
        refined = torch.cuda.Event()
        refined.record()
        loss_target = loss_calc_uvem([pred_t1, pred_t2], hard, soft, loss_fn=state.loss_fn_t, multi=True)    # :221
    main.wait_event(refined)                                # the prototypes were read: now they may move
    label_ds = aligner.update_prototype(feat_s, batch["label_s"])               # :216
    loss_source = loss_calc([pred_s1, pred_s2], batch["label_s"], loss_fn=state.loss_fn_s, multi=True)   # :219
    # loss = loss_source + loss_target; loss.backward() (:222-227) as two passes over two disjoint graphs: the source pass is queued on
    # the caller's stream without waiting for the target pipeline; the target pass ends with the join and the fold of its gradients
    loss_source.backward()
    loss_target.backward()
    ops.grad_join()
    finish()
    prescale = dp.reduce_gradients() if dp is not None else 1.0
    optimizer.step(max_norm=hp["max_norm"], grad_prescale=prescale)             # :230-232 (clip 32 + SGD)
    aligner.check_superpixel_ids(wait=False)
    return dict(loss_source=loss_source.detach(), loss_target=loss_target.detach(), label_t_soft=soft,
                label_t_hard=hard, label_s_ds=label_ds, pred_s1=pred_s1.detach(), pred_s2=pred_s2.detach(),
                pred_t1=pred_t1.detach(), pred_t2=pred_t2.detach(), feat_s=feat_s.detach(), feat_t=feat_t.detach(),
                grad_norm=optimizer.last_grad_norm)


def ssl_step(model, aligner, optimizer, state, batch, lr, dp=None, sup_ignore_id=None, mark=None):
    """`mark(name)`, if given, is called at the phase boundaries (bench.py records a HIP event there)."""
    hp = state.hp
    model.train()
    optimizer.param_groups[0]["lr"] = lr
    if mark is None and _ops().TWO_PIPELINES and _ops().TWO_STREAM_BWD and _may_fork(model):
        return _ssl_step_two_pipelines(model, aligner, optimizer, state, batch, lr, dp, sup_ignore_id)
    mark = mark or (lambda name: None)
    mark("start")
    (pred_s1, pred_s2, feat_s), (pred_t1, pred_t2, feat_t) = forward_pair(model, batch["images_s"], batch["images_t"])   # :205-207
    mark("forward_source")
    mark("forward_target")
    soft, hard = aligner.refine_and_select(batch["label_t_sup"], feat_t, [pred_t1, pred_t2], batch["label_t_soft"],
                                           mode=hp["refine_mode"], temp=hp["refine_temp"], cutoff_top=hp["cutoff_top"],
                                           cutoff_low=hp["cutoff_low"], sup_ignore_id=sup_ignore_id)   # :209-214
    mark("label_refine_select")
    label_ds = aligner.update_prototype(feat_s, batch["label_s"])               # :216
    mark("prototype_update")
    loss_source = loss_calc([pred_s1, pred_s2], batch["label_s"], loss_fn=state.loss_fn_s, multi=True)   # :219
    loss_target = loss_calc_uvem([pred_t1, pred_t2], hard, soft, loss_fn=state.loss_fn_t, multi=True)    # :221
    loss = loss_source + loss_target
    mark("losses_forward")
    optimizer.zero_grad()
    loss.backward()
    _ops().grad_join()              # the step's stream behind the side stream and the second graph's stream (no-op when neither ran)
    mark("backward")
    prescale = dp.reduce_gradients() if dp is not None else 1.0
    mark("grad_allreduce_wait")
    optimizer.step(max_norm=hp["max_norm"], grad_prescale=prescale)             # :230-232 (clip 32 + SGD)
    mark("clip_sgd")
    # a superpixel id that did not fit the segment table raises here if the device's report is already in, else at the next step's
    # label_refine (no host wait inside the step; `aligner.check_superpixel_ids()` after the last step waits for the last report)
    aligner.check_superpixel_ids(wait=False)
    return dict(loss_source=loss_source.detach(), loss_target=loss_target.detach(), label_t_soft=soft,
                label_t_hard=hard, label_s_ds=label_ds, pred_s1=pred_s1.detach(), pred_s2=pred_s2.detach(),
                pred_t1=pred_t1.detach(), pred_t2=pred_t2.detach(), feat_s=feat_s.detach(), feat_t=feat_t.detach(),
                grad_norm=optimizer.last_grad_norm)


def src_step(model, optimizer, state, batch, lr, dp=None, aligner=None, align_domain=False):
    """Stage-1 iteration (tools/train_src.py:112-141); with `align_domain` (the script's --align-domain) the target
    tiles are pushed through the network too and CORAL between the two feature sets joins the loss (:126-135)."""
    hp = state.hp
    model.train()
    optimizer.param_groups[0]["lr"] = lr
    if align_domain:
        # :116 and :127 -- two train-mode forwards with nothing between them that the second one reads: the pair may fork (forward_pair)
        (pred_s1, pred_s2, feat_s), (_p1, _p2, feat_t) = forward_pair(model, batch["images_s"], batch["images_t"])
    else:
        pred_s1, pred_s2, feat_s = model(batch["images_s"])                     # train_src.py:116
    loss_seg = loss_calc([pred_s1, pred_s2], batch["label_s"], loss_fn=state.loss_fn_s, multi=True)   # :132
    loss_domain = None
    if align_domain:
        loss_domain = aligner.align_domain(feat_s, feat_t)                      # :134
    loss = loss_seg + loss_domain if align_domain else loss_seg
    optimizer.zero_grad()
    loss.backward()
    _ops().grad_join()
    prescale = dp.reduce_gradients() if dp is not None else 1.0
    optimizer.step(max_norm=hp["max_norm"], grad_prescale=prescale)
    out = dict(loss_source=loss_seg.detach(), pred_s1=pred_s1.detach(), pred_s2=pred_s2.detach(),
               grad_norm=optimizer.last_grad_norm)
    if align_domain:
        out["loss_domain"] = loss_domain.detach()
    return out


def align_step(model, aligner, optimizer, state, batch, lr, dp=None, sup_ignore_id=None, align_domain=True, pcl_temp=8.0):
    """One stage-2 (prototype-contrastive alignment) iteration: reference tools/train_align_uem.py:139-183."""
    import torch
    from . import ops
    from .gast.pseudo_generation import pseudo_selection
    from .loss import PrototypeContrastiveLoss
    hp = state.hp
    if not hasattr(state, "loss_fn_pcl"):
        state.loss_fn_pcl = PrototypeContrastiveLoss(temperature=pcl_temp, ignore_label=hp["ignore_label"])
    model.train()
    optimizer.param_groups[0]["lr"] = lr
    # :147 and :156 as a pair that may fork (forward_pair): the prototype update between them (:150) reads the source features only and
    # nothing of it enters the target forward
    (pred_s1, pred_s2, feat_s), (pred_t1, pred_t2, feat_t) = forward_pair(model, batch["images_s"], batch["images_t"])
    label_s_down = aligner.update_prototype(feat_s, batch["label_s"])           # :150
    # on-the-fly soft labels (:158-160): (softmax(up(x1)) + softmax(up(x2))) / 2, the eval-output kernel
    x1, x2 = ops.as_nhwc(pred_t1.detach()).contiguous(), ops.as_nhwc(pred_t2.detach()).contiguous()
    n, h, w, c = x1.shape
    H, W = batch["images_t"].shape[-2:]
    soft0 = torch.empty((n, c, H, W), device=x1.device, dtype=torch.float32)
    ops.call("uem_upsample_softmax_avg", ops.ptr(x1), ops.ptr(x2), ops.ptr(soft0), n, c, h, w, H, W, ops.stream())
    soft, hard = aligner.refine_and_select(batch["label_t_sup"], feat_t, [pred_t1, pred_t2], soft0, mode=hp["refine_mode"],
                                           temp=hp["refine_temp"], cutoff_top=hp["cutoff_top"], cutoff_low=hp["cutoff_low"],
                                           sup_ignore_id=sup_ignore_id)          # :161-165
    label_t = aligner.downscale_gt(hard)                                        # :170
    loss_seg = loss_calc([pred_s1, pred_s2], batch["label_s"], loss_fn=state.loss_fn_s, multi=True)     # :174
    loss_domain = aligner.align_domain(feat_s, feat_t) if align_domain else 0   # :175
    loss_align = (state.loss_fn_pcl(aligner.prototypes, feat_s, label_s_down) +
                  state.loss_fn_pcl(aligner.prototypes, feat_t, label_t)) * 0.5  # :176-177
    loss = loss_seg + loss_domain + loss_align
    optimizer.zero_grad()
    loss.backward()
    ops.grad_join()
    prescale = dp.reduce_gradients() if dp is not None else 1.0
    optimizer.step(max_norm=hp["max_norm"], grad_prescale=prescale)
    aligner.check_superpixel_ids(wait=False)
    return dict(loss_seg=loss_seg.detach(), loss_domain=loss_domain.detach() if torch.is_tensor(loss_domain) else loss_domain,
                loss_align=loss_align.detach(), label_t_hard=hard, pred_s1=pred_s1.detach(), pred_t1=pred_t1.detach(),
                grad_norm=optimizer.last_grad_norm)


def finish(aligner):
    """After the LAST step of a training loop, and before a checkpoint is written: wait for the last step's superpixel-range report.
    Inside ssl_step / align_step the check never waits (`check_superpixel_ids(wait=False)`: an id outside the segment table raises one
    step late), so the final step's report is only seen by this call."""
    if aligner is not None:
        aligner.check_superpixel_ids(wait=True)


class GraphedStep:
    """One whole training iteration captured in ONE hipGraph and replayed: ~1100 kernel launches per step become one graph launch
    (host time per step 0.2 ms instead of 16-30).  The device time is unchanged -- the step is device-bound at the benchmark batch --
    so this is for hosts that cannot keep up: small batches, many ranks per host, the 45 ms bf16 step.

        gs = GraphedStep(ssl_step, model, aligner, optimizer, state, batch, sup_ignore_id=...)   # warms up, then captures
        out = gs(lr)            # writes lr into the device scalar the captured optimizer launch reads, replays; `out` = the step's
                                # dict of STATIC tensors (overwritten by the next replay)

    Data parallel: pass `dp=wrapper` (uemda_amd.dp.DataParallel over the nccl / RCCL backend); the early tail-bucket all-reduce and the
    head all-reduce are captured with the step, on RCCL's stream, as graph edges -- every rank replays the same graph.
    Requirements: inputs are the same device tensors every step (copy new data INTO `batch`), FusedSGD, at least one optimizer step
    taken before (the first step initialises the momentum buffer through a by-value flag).  Host-side logic of the step function runs
    ONCE, at capture: only state that is updated IN PLACE on the device survives replay (a rebound tensor -- `x = f(x)` -- would be
    read at its capture-time address for ever; ClassBalance's frequency EMA is in place for that reason), `param_groups['lr']` is not
    read again (the learning rate travels through the device scalar), and per-launch event timing (`ops.PROF`) must be off.  The superpixel-table overflow flag stays on the device while capturing
    (`aligner.last_superpixel_range_flag`); `check()` reads it (a host sync: call it now and then, not every step).  The PPM heads'
    Dropout2d draws its masks from torch's graph-safe generator while capturing (models/ppm.py), a fresh mask per replay."""

    def __init__(self, step_fn, model, aligner, optimizer, state, batch, warmup=2, lr=1e-3, check_every=0, **kw):
        """check_every = N > 0: every N-th replay ends with `check()` (one host sync: the superpixel-table overflow flag of a replayed
        step never leaves the device otherwise -- ADVICE r5); 0: the caller calls `check()` itself."""
        import torch
        self.check_every, self._replays = int(check_every), 0
        from .optim import FusedSGD
        from .ops import UemError
        if not isinstance(optimizer, FusedSGD):
            raise UemError("GraphedStep needs uemda_amd.optim.FusedSGD (the learning rate travels as a device scalar)")
        dp = kw.get("dp")
        if dp is not None and not dp.capturable:
            raise UemError("GraphedStep: this process group's all-reduce cannot be captured (gloo moves the buffer through the host); "
                           "use the nccl (RCCL) backend or run the data-parallel step eagerly")
        from . import ops
        if ops.PROF.enabled:
            raise UemError("GraphedStep: per-launch event timing (ops.PROF.enabled) cannot be captured; switch it off first")
        self.aligner, self.optimizer = aligner, optimizer
        self.lr = torch.full((1,), float(lr), device=next(model.parameters()).device, dtype=torch.float32)
        optimizer.lr_device = self.lr

        def run():
            if aligner is None:
                return step_fn(model, optimizer, state, batch, float(lr), **kw)
            return step_fn(model, aligner, optimizer, state, batch, float(lr), **kw)
        # The weight gradients' side stream (ops.on_side) is captured with the step (round 6): the warm-up steps run it too, so that the
        # side stream exists and the Winograd weight gradients' per-shape workspaces (ops._WinoSideWs) are allocated BEFORE the capture --
        # an allocation on a second stream during a capture is what invalidated it in round 5.  Nothing of the eager steps may still be
        # in flight on another stream when the capture starts.  UEM_GRAPH_SIDE=0 keeps the side stream out of the capture as before.
        torch.cuda.synchronize()
        graph_side = ops.GRAPH_SIDE
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                    # warm-up off the default stream, as torch.cuda.graph asks
            for _ in range(max(int(warmup), 1 if optimizer._steps == 0 else 0)):
                run()
            ops.side_join()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if aligner is not None:
            # the warm-up steps' superpixel-range report is collected NOW: inside the step the check no longer waits (round 5), and a
            # report still pending at the capture would be queried there -- hipEventQuery under a capture invalidates it (this took
            # bench.py's replay leg down in a rehearsal: hipErrorStreamCaptureInvalidated)
            aligner.check_superpixel_ids()
        # the weight-preparation job table is fixed before the capture (a single warm-up step leaves it half built) and stays alive as
        # long as the graph whose refresh launch reads it -- and so does every parameter / derived bank the table points at, including
        # those of OTHER models alive on the device now (the captured launch covers the whole table)
        self._prep_tables = ops.PREP.settle()
        self._prep_hold = ops.PREP.hold()
        self._side_ws = dict(ops._WinoSideWs.sets)       # the captured side-stream launches write into these: keep them alive
        self.graph = torch.cuda.CUDAGraph()
        steps_before = optimizer._steps
        # With a process group alive, its watchdog thread polls its work events (hipEventQuery) whenever it likes: under the default
        # "global" capture mode such a call from ANOTHER thread while this one captures is an error, raised in that thread -- the
        # process aborts, now and then (seen once in four runs of the captured data-parallel step).  "thread_local" checks this
        # thread's calls only; what is recorded is the same (capture is per stream).
        mode = "thread_local" if (dp is not None or (torch.distributed.is_available() and torch.distributed.is_initialized())) else "global"
        if not graph_side:
            ops.SIDE_OFF += 1
        try:
            with torch.cuda.graph(self.graph, capture_error_mode=mode):
                self.out = run()
        finally:
            if not graph_side:
                ops.SIDE_OFF -= 1
        optimizer._steps = steps_before                  # the capture pass ran the host side of step() without taking a step

    @staticmethod
    def all_ranks_ok(ok, device=None):
        """Data parallel: did the capture succeed on EVERY rank?  One MIN all-reduce of a flag, eagerly, outside any capture.  A graph
        that exists on some ranks only must never be replayed -- its ranks would launch collectives the others never join -- so callers
        do `gs = try: GraphedStep(...)`, then `if not GraphedStep.all_ranks_ok(gs is not None): gs = None` and every rank falls back
        to the eager step together (bench.py's replay leg).  The capture itself executes no collective (RCCL's
        launches are recorded, not run), so a rank whose capture raised leaves nobody waiting."""
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return bool(ok)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        flag = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(int(flag.item()))

    def _after_replay(self):
        from . import ops
        self.optimizer._steps += 1
        ops.weights_changed()                            # cached filter banks (transposed, Winograd) follow the replayed optimizer

    def __call__(self, lr):
        self.lr.fill_(float(lr))
        self.graph.replay()
        self._after_replay()
        self._replays += 1
        if self.check_every > 0 and self._replays % self.check_every == 0:
            self.check()
        return self.out

    def check(self):
        flag = getattr(self.aligner, "last_superpixel_range_flag", None) if self.aligner is not None else None
        if flag is not None and int(flag) != 0:
            from .ops import UemError
            raise UemError("label_refine: a superpixel id was outside the segment table during a replayed step "
                           "(set aligner.sup_capacity, see Aligner._sup_table_size)")
