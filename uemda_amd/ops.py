"""Tensor-level wrappers over the C ABI (raw pointers + sizes).  torch is plumbing only: it owns the
device allocations and the stream; every arithmetic step is a `uem_*` HIP kernel.

Internal activation layout is NHWC: tensors of shape (N, H, W, C), contiguous, fp32.
"""
import ctypes
import weakref

import torch

from . import _lib
from ._lib import CONV_ACCUMULATE, CONV_IN_AFFINE, CONV_IN_RELU, CONV_TRANSPOSED, ConvShape, UemError, call

import os
import sys

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
# epilogue fusions (on by default; the switches exist for A/B measurements)
FUSE_BN_STATS = os.environ.get("UEM_FUSE_BN_STATS", "1") != "0"
FUSE_BN_BACKWARD = os.environ.get("UEM_FUSE_BN_BACKWARD", "1") != "0"
RELU_BITS = os.environ.get("UEM_RELU_BITS", "1") != "0"      # block-output ReLU mask saved as packed bits
# matrix-core operand precision of the fp32-STORAGE conv kernels.  "fp32" is the exact fp32 MFMA: the default, and what every parity
# statement is made for.  "bf16" (bf16 operands on the bf16 MFMA, fp32 tensors, fp32 accumulate) is how a bf16-storage model runs its
# fp32 islands (7x7 stem, ASPP / PPM heads): models set it per launch through `conv_precision`.  The split-operand modes of rounds
# 1-3 ("bf16x3", "mixed") are retired: they ran the round-1 register-staged kernels and none could be the headline.
_PREC_FLAGS = {"fp32": (0, 0), "bf16": (_lib.CONV_PREC_BF16,) * 2}
_RETIRED_PREC = ("bf16x3", "mixed")


def _prec_flags(name, where):
    if name in _RETIRED_PREC:
        raise UemError(f"{where}: conv precision {name!r} was retired in round 4 (DESIGN 3.2); use one of {sorted(_PREC_FLAGS)}")
    if name not in _PREC_FLAGS:
        raise UemError(f"{where}: conv precision must be one of {sorted(_PREC_FLAGS)}, got {name!r}")
    return _PREC_FLAGS[name]


CONV_PREC, CONV_PREC_BWD = _prec_flags(os.environ.get("UEM_CONV_PREC", "fp32"), "UEM_CONV_PREC")


def set_conv_precision(name):
    global CONV_PREC, CONV_PREC_BWD
    CONV_PREC, CONV_PREC_BWD = _prec_flags(name, "set_conv_precision")


class conv_precision:
    """`with ops.conv_precision("bf16"):` -- the operand precision of the fp32-storage conv kernels for the launches inside the
    block (None: leave it alone); the bf16-storage model runs its fp32 ASPP GEMMs with bf16 operands this way."""

    def __init__(self, name):
        if name is not None:
            _prec_flags(name, "conv_precision")
        self.name = name

    def __enter__(self):
        global CONV_PREC, CONV_PREC_BWD
        self.saved = (CONV_PREC, CONV_PREC_BWD)
        if self.name is not None:
            CONV_PREC, CONV_PREC_BWD = _PREC_FLAGS[self.name]
        return self

    def __exit__(self, *exc):
        global CONV_PREC, CONV_PREC_BWD
        CONV_PREC, CONV_PREC_BWD = self.saved
        return False


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """the current HIP stream's handle (what every entry point launches on).  torch.cuda.current_stream() builds a Stream object per
    call -- 10 % of the host's time per step over ~940 launches; the raw getter is one C call and follows torch.cuda.stream(...) /
    graph capture the same way."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def nbt_inc(bn):
    """num_batches_tracked += 1 of a training-mode BatchNorm, unless its counter lives in a model-level arena that
    the model's forward bumps once for all layers (Deeplabv2._nbt_step)."""
    if bn.training and not getattr(bn, "_uem_nbt_arena", False):
        bn.num_batches_tracked.add_(1)


class _Profiler:
    """Optional per-launch HIP-event timing of the conv kernels (bench.py's `roofline` leg).  Events are
    recorded on torch's current stream, which is the stream every kernel here is launched on."""

    def __init__(self):
        self.enabled = False
        self.records = []          # (family, algorithmic flops, start event, end event)

    # UEM_PROF_MARK=1: a one-element marker launch (scale_kernel) in front of every profiled conv op, so that a rocprofv3 --pmc pass of
    # the same command can be cut into ops by dispatch order (scripts/pmc_per_op.py: HBM bytes per conv launch BY SHAPE)
    MARK = os.environ.get("UEM_PROF_MARK", "0") != "0"
    _mark = None

    def run(self, family, flops, launch, executed=None, who=None):
        """flops: ALGORITHMIC flops of the op (2*M*Cout*k*k*Cin); executed: the multiplies actually issued where they differ
        (Winograd: algorithmic / 2.25); who: the op's label when the caller is not the op's own function (a launch closure)."""
        if not self.enabled:
            return launch()
        if self.MARK:
            if self._mark is None:
                _Profiler._mark = torch.ones(1, device="cuda")
            call("uem_scale", ptr(self._mark), 1, 1.0, stream())
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = launch()
        e.record()
        fr = sys._getframe(1)
        t = fr.f_locals.get("dy", fr.f_locals.get("x"))               # label for per_call(): calling op + its input's shape
        who = (who or fr.f_code.co_name) + ("" if t is None else " " + "x".join(str(d) for d in t.shape))
        if self.MARK:                                                 # ... and, for the byte table, the other side's channels and taps
            lo = fr.f_locals
            if "cout" in lo and "cin" in lo:
                who += f" ->{lo['cout'] if fr.f_code.co_name.find('dgrad') < 0 else lo['cin']} k{lo.get('kh', 3)}"
        self.records.append((family, flops, s, e, flops if executed is None else executed, who))
        return out

    def per_call(self):
        """[(family, calling op, algorithmic flops, executed flops, ms)] in launch order; call after torch.cuda.synchronize()."""
        return [(fam, who, fl, ex, s.elapsed_time(e)) for fam, fl, s, e, ex, who in self.records]

    def summary(self):
        """family -> dict(launches, flops, executed, ms); call after torch.cuda.synchronize()."""
        agg = {}
        for fam, fl, s, e, ex, _ in self.records:
            a = agg.setdefault(fam, dict(launches=0, flops=0.0, executed=0.0, ms=0.0))
            a["launches"] += 1
            a["flops"] += fl
            a["executed"] += ex
            a["ms"] += s.elapsed_time(e)
        return agg


PROF = _Profiler()


# UEM_TRACE_NONFINITE=1: debugging aid -- after every entry-point call, synchronise and report which of the float
# tensors handed to it (inputs and outputs) hold NaN/Inf.  The first call with clean inputs and a dirty output is
# where a non-finite value is born.
TRACE_NONFINITE = os.environ.get("UEM_TRACE_NONFINITE", "0") != "0"
_trace_args = []
_lib_call = call


def ptr(t):
    if TRACE_NONFINITE and t is not None:
        _trace_args.append(t)
    return None if t is None else t.data_ptr()


if TRACE_NONFINITE:
    def call(name, *args):          # noqa: F811  (debug wrapper around _lib.call)
        tensors = list(_trace_args)
        _trace_args.clear()
        rc = _lib_call(name, *args)
        torch.cuda.synchronize()
        dirty = []
        for i, t in enumerate(tensors):
            if t.is_floating_point() and t.numel() > 0 and not bool(torch.isfinite(t).all()):
                bad = (~torch.isfinite(t)).reshape(-1)
                first = int(bad.nonzero()[0])
                dirty.append(f"arg{i}{tuple(t.shape)}: {int(bad.sum())} bad, first at flat index {first}")
        if dirty:
            print(f"[uem trace] {name}: non-finite in {dirty}", flush=True)
        return rc


# ---- side stream for the weight gradients (round 5) ---------------------------------------------------------------------------
# The weight gradients are leaves of the backward graph: nothing downstream reads them before the optimizer or the gradient
# all-reduce.  ops_bf16 runs them beside the data-gradient / BatchNorm chain (every bf16 conv launch is short and latency-bound: two
# side by side fill each other's ramps and tails, -1.0 ms of 39.6); the fp32 kernels are long and matrix-bound and gain less (-0.5 ms
# of 101: the Winograd weight gradient's HBM-bound transforms run under the main chain's GEMMs); UEM_SIDE_WGRAD=0 / UEM_BF16_SIDE_WGRAD=0
# switch it off.  Ordering: the side stream waits for everything the main stream has queued at the call, the tensors the launch reads
# are recorded on it (the caching allocator will not hand their memory out again before the launch has run), and the main stream waits
# for the side stream once per backward pass -- an autograd end-of-backward callback -- and before a data-parallel bucket goes out.
SIDE_WGRAD_F32 = os.environ.get("UEM_SIDE_WGRAD", "1") != "0"
SIDE_HOLD = os.environ.get("UEM_SIDE_HOLD", "1") != "0"       # 0: record_stream instead of holding references until the join (rounds 5 / early 6)


# Stream priorities (0 = default, -1 = high).  The second graph's stream runs at HIGH priority: its chain is queued behind the first
# graph's by the host and trails it on the device, and whatever of it is left when the first chain has finished runs alone -- with
# priority it catches up and the two chains end together (same-box A/B, profiles/r06_l_two_streams.txt: bf16 36.5-36.7 -> 35.7-35.8 ms,
# fp32 98.7-99.9 -> 98.5-98.6).  A high-priority SIDE stream loses (fp32 100.0, bf16 36.8: weight gradients in front of the chain
# that produces their inputs).
SIDE_PRIORITY = int(os.environ.get("UEM_SIDE_PRIORITY", "0"))
SECOND_PRIORITY = int(os.environ.get("UEM_SECOND_PRIORITY", "-1"))


class _Side:
    by_device = {}

    def __init__(self):
        self.stream = torch.cuda.Stream(priority=SIDE_PRIORITY)
        self.task = None                 # id of the backward graph task whose end-of-backward callback will join (None: no join queued)
        self.dirty = False               # work was queued since the last join
        self.reads = []                  # (first byte, past-the-end byte, label) of what the queued launches read, since the last join
        self.held = []                   # the tensors those launches read, kept alive until the join (see on_side)
        self.users = []                  # the streams that queued launches since the last join (two when the step's graphs run on two)

    @classmethod
    def get(cls):
        dev = torch.cuda.current_device()
        st = cls.by_device.get(dev)
        if st is None:
            st = cls.by_device[dev] = _Side()
        return st

    def join(self):
        self.task = None
        self.reads.clear()
        if self.dirty:
            cur = torch.cuda.current_stream()
            cur.wait_stream(self.stream)
            for s in self.users:         # every stream that handed tensors to the side stream waits: the held ones go back to ITS pool
                if s != cur:
                    s.wait_stream(self.stream)
            self.dirty = False
        self.users.clear()
        self.held.clear()                # freed in their streams' order BEHIND the wait: reusable at once, no cross-stream bookkeeping


def side_join():
    """main stream waits for the side stream's weight gradients (no-op when none are pending).  Called by the end-of-backward callback,
    by the data-parallel bucket trigger before a slice of the gradient arena goes out, and -- ADVICE r5 -- unconditionally at the top of
    FusedSGD.step / clip_grad_norm_ / DataParallel.reduce_gradients: a backward pass that RAISED after a launch was queued never runs
    its final callbacks, and whoever reads the gradient arena next must still wait for the side stream."""
    if not _Side.by_device:           # no side stream was ever used in this process (also: CPU-only hosts)
        return
    st = _Side.by_device.get(torch.cuda.current_device())
    if st is not None:
        st.join()


def _byte_range(t):
    lo = t.data_ptr()
    return lo, lo + t.numel() * t.element_size()


def guard_write(t, what):
    """GUARD: a main-stream kernel is about to write into the EXISTING buffer `t` (an `out=` / `dx=` handed in by
    the caller, i.e. an in-place update the allocator knows nothing about).  If a side-stream launch queued since the last join reads
    any byte of it, that is a race between the two streams -- the first side-stream build lost one exactly so (DESIGN 8, round 5:
    PPM step goldens red) -- and it fails HERE, loudly, instead of racing.  Kernels write through raw pointers and never move a
    tensor's `_version`, so the guard works on byte ranges, not on versions.  Costs nothing when no side launch is pending."""
    if t is None:
        return
    st = _Side.by_device.get(t.device.index)
    if st is None or not st.reads:
        return
    lo, hi = _byte_range(t)
    for a, b, label, alive in st.reads:
        # a range whose tensor is gone no longer guards anything (on_side holds the tensors a launch reads until the join, so this is
        # the exception: a caller that passed a temporary VIEW whose base lives on elsewhere)
        if lo < b and a < hi and alive() is not None:
            raise UemError(f"{what}: writes in place into a buffer ({tuple(t.shape)} at 0x{lo:x}) that a weight gradient queued on the "
                           f"side stream still reads ({label}); the main stream may overwrite only what no pending side launch reads "
                           f"(ops.on_side / ops.side_join)")


def on_side(launch, tensors, label="weight gradient"):
    st = _Side.get()
    main = torch.cuda.current_stream()
    if main == st.stream:
        return launch()
    st.stream.wait_stream(main)
    with torch.cuda.stream(st.stream):
        launch()
    # The tensors the launch reads stay referenced until the join instead of being recorded on the side stream (measured in
    # round 6): a block freed while recorded on another stream is reusable only once the DEVICE has passed that stream's event, and the
    # host enqueues a whole backward pass in a tenth of the time the device takes -- so within a pass none of those blocks ever came back
    # and the allocator kept asking the driver for more (peak reserved 87.7 GB against 33.9 without the side stream at 33 GB allocated;
    # R101-1024 bf16: 167-255 GB against 106).  Held until the main stream has waited for the side stream, they are freed in main-stream
    # order and reusable immediately.
    for t in tensors:
        if SIDE_HOLD:
            st.held.append(t)
        else:
            t.record_stream(st.stream)
        st.reads.append(_byte_range(t) + (f"{label} reading {tuple(t.shape)}", weakref.ref(t if t._base is None else t._base)))
    st.dirty = True
    if main not in st.users:
        st.users.append(main)
    # one join per backward pass, queued in THAT pass: the flag is the graph task's id, not a boolean -- a pass that raised after
    # queueing never ran its callbacks, and a boolean left True would keep every later pass from queueing its own
    tid = torch._C._current_graph_task_id()
    if tid < 0:
        st.join()                        # not inside a backward pass (a direct call from a test or a script): join right away
    elif st.task != tid:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(st.join)
            st.task = tid
        except RuntimeError:
            st.join()



# ---- two streams for the step's two graphs (round 6) -------------------------------------------------------------------------------
# The two train-mode forwards of a train_ssl_uem step (source batch, target batch) are independent until the mining reads both, and so
# are their backward chains: the step runs the second forward on its own stream (step.forward_pair), and autograd then runs that
# graph's backward nodes on the same stream, beside the first graph's.  Every launch of the bf16 path and most of the fp32 path leaves
# part of the chip idle (latency- and L2-bound kernels, the ramp and tail of every launch), and two independent chains fill each
# other's gaps (profiles/r06_l_two_streams.txt).  What the two graphs share is kept in order by construction:
#   * BatchNorm running statistics: the second forward writes its EMA contribution into a shadow arena, applied behind the join in the
#     reference's order -- bit for bit the sequential pair's statistics; num_batches_tracked is bumped on the main stream;
#   * derived filter banks: all refreshed before the fork;
#   * parameter gradients (read-modify-write accumulations: BatchNorm gamma / beta, the stem's and the Winograd folds, every
#     dw += ...): a backward node running on the second stream accumulates into a SHADOW gradient arena (blocks.grad_buffer), which the
#     end-of-backward callback adds to the real one on the caller's stream once that stream has waited for the second (and the side)
#     stream -- no two streams ever update one address; weight gradients of both graphs still share the ONE side stream, in order;
#   * the data-parallel early bucket: the last trigger waits for the other graph's trigger event and folds the shadow slice first.
# UEM_TWO_STREAM_BWD=0 keeps the forwards on two streams and moves every backward node to the step's stream (`on_backward_stream`).
TWO_STREAM_FWD = os.environ.get("UEM_TWO_STREAM_FWD", "1") != "0"
TWO_STREAM_BWD = os.environ.get("UEM_TWO_STREAM_BWD", "1") != "0"
# step.ssl_step with the target graph's mining and loss on its stream too and two backward passes (the source pipeline never waits for the
# target's): built, tested, measured SLOWER than the joined form on one box (profiles/r06_l_two_streams.txt: bf16 36.3 against 35.6 ms,
# R101-1024 192.9 against 191.2, fp32 97.4 against 96.8) -- the two chains drift apart and stop sharing each layer's filter bank in L2.
TWO_PIPELINES = os.environ.get("UEM_TWO_PIPELINES", "0") != "0"
_FWD2 = {}                 # device index -> the second stream
_BWD_MAIN = {}             # device index -> the stream backward work is redirected to when TWO_STREAM_BWD is off
_SHADOW_OWNERS = weakref.WeakSet()      # models whose shadow gradient arena holds unfolded gradients


def second_stream():
    dev = torch.cuda.current_device()
    st = _FWD2.get(dev)
    if st is None:
        st = _FWD2[dev] = torch.cuda.Stream(priority=SECOND_PRIORITY)
    return st


def on_second_stream():
    """is the current stream this package's second stream (a backward node of the step's second graph is running)?"""
    if not _FWD2:
        return False
    st = _FWD2.get(torch.cuda.current_device())
    return st is not None and torch.cuda.current_stream() == st


def set_backward_stream(stream_):
    _BWD_MAIN[torch.cuda.current_device()] = stream_


def on_backward_stream(fn):
    """decorator of an autograd Function's backward.  A node whose forward ran on the second stream is handed that stream by the
    engine; with TWO_STREAM_BWD (default) it stays there (gradients go to the shadow arena, see above); without, its work is moved
    to the step's stream (ordered behind what the engine queued on the node's stream, and the node's stream behind it)."""
    import functools

    @functools.wraps(fn)
    def wrapper(ctx, *grads):
        if TWO_STREAM_BWD or not _BWD_MAIN:
            return fn(ctx, *grads)
        main = _BWD_MAIN.get(torch.cuda.current_device())
        if main is None:
            return fn(ctx, *grads)
        cur = torch.cuda.current_stream()
        if cur == main or cur != _FWD2.get(torch.cuda.current_device()):
            return fn(ctx, *grads)          # only a node the engine put on the SECOND stream moves (never e.g. a capturing stream)
        main.wait_stream(cur)
        with torch.cuda.stream(main):
            out = fn(ctx, *grads)
        cur.wait_stream(main)
        return out
    return wrapper


def shadow_grads_touched(owner):
    """blocks.grad_buffer handed out a view of `owner`'s shadow gradient arena: fold it at the end of this backward pass (once per
    pass; `grad_join` folds what a pass that raised left behind)"""
    _SHADOW_OWNERS.add(owner)
    tid = torch._C._current_graph_task_id()
    if tid >= 0 and getattr(owner, "_g2_task", None) != tid:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(owner.fold_shadow_grads)
            owner._g2_task = tid
        except RuntimeError:
            pass


# The two-stream form inside a captured hipGraph (step.GraphedStep): the fork and the joins become graph edges -- two long branches, a
# handful of cross edges (unlike the ~70 forks of a captured side stream, GRAPH_SIDE).  Built, tested (the replay equals the eager
# step, tests/test_gpu_model.py) and measured (profiles/r06_n_hipgraph_two_streams.txt): the two-branch graph replays SLOWER than the
# captured sequential step -- fp32 101.8 against 99.8 ms (eager 97.2-97.7), bf16 40.4 against 38.6 (eager 35.6-35.7), 2.1-2.5 ms of
# host time per replay against 0.29: the runtime's launch of a graph with parallel branches costs more than the branches return.
# Default off; UEM_GRAPH_TWO_STREAM=1 captures the fork.
GRAPH_TWO_STREAM = os.environ.get("UEM_GRAPH_TWO_STREAM", "0") != "0"


def stream_capturing(st):
    with torch.cuda.stream(st):
        return torch.cuda.is_current_stream_capturing()


def grad_join():
    """Whoever reads or clears the gradient arena next (FusedSGD.step, clip_grad_norm_, DataParallel.reduce_gradients, zero_grad, the
    step after backward) calls this first: the current stream waits for the side stream's weight gradients and for the second
    stream's backward chain, and shadow gradients not yet folded (a backward pass that raised never ran its callbacks) are folded."""
    side_join()
    for owner in list(_SHADOW_OWNERS):
        owner.fold_shadow_grads()
    if _FWD2:
        st = _FWD2.get(torch.cuda.current_device())
        # while capturing: only a second stream that is part of the capture may (and must) be joined -- a dependency on a stream
        # outside the capture would invalidate it
        if st is not None and (not torch.cuda.is_current_stream_capturing() or stream_capturing(st)):
            torch.cuda.current_stream().wait_stream(st)


SIDE_OFF = 0            # > 0: no side stream (GraphedStep raises it around its capture when UEM_GRAPH_SIDE=0)
# Round 6: the side stream CAN be captured with the step (a fork at every side launch, a join at the end of backward): the Winograd
# weight gradient no longer allocates inside its side-stream launch (_WinoSideWs), which is what had invalidated the capture in round 5;
# tests/test_gpu_model.py replays such a graph against the eager step.  Measured (profiles/r06_g_hipgraph_side_stream.txt, A/B on one
# box): the replay of the forked graph is SLOWER than the replay of the serial one -- fp32 104.0 against 100.2 ms (eager 98.9), bf16
# storage 39.7 against 39.1 (eager 38.7), 2.4 ms of host time per replay against 0.25: ~70 fork edges per backward pass cost the graph
# launch more than the overlap returns.  Default therefore: the capture keeps the side stream out (UEM_GRAPH_SIDE=1 captures it).
GRAPH_SIDE = os.environ.get("UEM_GRAPH_SIDE", "0") != "0"


def in_backward():
    """inside an autograd backward pass -- and not under per-launch event timing: `PROF` prices a family by the events around its
    launches, which means something only while the launches run one after the other, so the one profiled step of bench.py runs serially"""
    # (... and not while a hipGraph is being captured: the Winograd weight gradient allocates its temporaries inside the launch, and an
    # allocation on a second stream during a capture invalidates it -- hipErrorStreamCaptureInvalidated, which took the whole bench line
    # down in a rehearsal of this round; a replayed graph serialises the side stream's nodes anyway: bf16 39.4 ms without, 40.0 with)
    return ((not torch.is_grad_enabled()) and torch._C._current_graph_task_id() >= 0 and not PROF.enabled and SIDE_OFF == 0
            and (GRAPH_SIDE or not torch.cuda.is_current_stream_capturing()))


def need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise UemError("uemda_amd: the HIP path needs tensors on the MI355X device "
                           "(there is no CPU / PyTorch fallback)")


def _f32c(t, what):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise UemError(f"{what}: expected a contiguous float32 tensor, got {t.dtype} strides={t.stride()}")
    return t


def conv_out_size(h, k, s, p, d):
    return (h + 2 * p - d * (k - 1) - 1) // s + 1


def weight_ohwi(w):
    """Physical OHWI view of a conv weight whose logical shape is OIHW (channels_last storage)."""
    v = w.detach().permute(0, 2, 3, 1)
    if not v.is_contiguous():
        raise UemError("conv weight is not in channels_last (OHWI) storage; call model._flatten_parameters()")
    return v


def _shape(x, cout, kh, kw, stride, pad, dil, x_ld=None, y_ld=None):
    n, h, w, cin = x.shape
    s = ConvShape()
    s.N, s.H, s.W, s.Cin = n, h, w, cin
    s.Ho, s.Wo, s.Cout = conv_out_size(h, kh, stride, pad, dil), conv_out_size(w, kw, stride, pad, dil), cout
    s.KH, s.KW, s.stride, s.pad, s.dil = kh, kw, stride, pad, dil
    s.x_ld = cin if x_ld is None else x_ld
    s.y_ld = cout if y_ld is None else y_ld
    return s


# ------------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------------
def conv2d(x, w_ohwi, bias=None, stride=1, pad=0, dil=1, in_scale=None, in_shift=None, in_relu=False,
           out=None, accumulate=False, algo_cout=None):
    """y = conv(prologue(x), w) + bias.  x (N,H,W,Cin); w (Cout,KH,KW,Cin); returns (N,Ho,Wo,Cout)."""
    need_gpu(x, w_ohwi)
    _f32c(x, "conv2d x"), _f32c(w_ohwi, "conv2d w")
    cout, kh, kw, cin = w_ohwi.shape
    if cin != x.shape[3]:
        raise UemError(f"conv2d: Cin mismatch {cin} vs {x.shape[3]}")
    s = _shape(x, cout, kh, kw, stride, pad, dil)
    if out is None:
        out = torch.empty((s.N, s.Ho, s.Wo, cout), device=x.device, dtype=torch.float32)
    flags = (CONV_IN_AFFINE if in_scale is not None else 0) | (CONV_IN_RELU if in_relu else 0) | \
            (CONV_ACCUMULATE if accumulate else 0) | CONV_PREC
    flops = 2.0 * s.N * s.Ho * s.Wo * (algo_cout or cout) * kh * kw * cin
    PROF.run("conv_fwd", flops, lambda: call("uem_conv2d_fwd", ptr(x), ptr(w_ohwi), ptr(bias), ptr(in_scale),
                                             ptr(in_shift), ptr(out), ctypes.byref(s), flags, stream()))
    return out


def conv2d_bn(x, w_ohwi, bn, stride=1, pad=0, dil=1, in_scale=None, in_shift=None, in_relu=False):
    """conv followed by training-mode BatchNorm statistics.  When the GEMM tiles are full (M % 128 == 0,
    Cout % 64 == 0) the statistics come out of the conv epilogue (no extra pass over y); otherwise, and in eval
    mode, this is conv2d + bn_stats.  Returns (y, BNState)."""
    training = bn.training or bn.running_mean is None
    cout, kh, kw, cin = w_ohwi.shape
    s = _shape(x, cout, kh, kw, stride, pad, dil)
    M = s.N * s.Ho * s.Wo
    if not training or M % 128 != 0 or cout % 64 != 0 or not FUSE_BN_STATS:
        y = conv2d(x, w_ohwi, None, stride, pad, dil, in_scale, in_shift, in_relu)
        return y, bn_stats(y, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, training, bn.eps,
                           bn.momentum if bn.momentum is not None else 0.1)
    need_gpu(x, w_ohwi)
    _f32c(x, "conv2d x"), _f32c(w_ohwi, "conv2d w")
    y = torch.empty((s.N, s.Ho, s.Wo, cout), device=x.device, dtype=torch.float32)
    tiles = M // 128
    ts = torch.empty((tiles, 2, cout), device=x.device, dtype=torch.float32)
    flags = (CONV_IN_AFFINE if in_scale is not None else 0) | (CONV_IN_RELU if in_relu else 0) | CONV_PREC
    flops = 2.0 * M * cout * kh * kw * cin
    PROF.run("conv_fwd", flops, lambda: call("uem_conv2d_fwd_stats", ptr(x), ptr(w_ohwi), ptr(in_scale), ptr(in_shift),
                                             ptr(y), ctypes.byref(s), flags, ptr(ts), stream()))
    st = BNState()
    st.training = True
    buf = torch.empty((4, cout), device=x.device, dtype=torch.float32)
    st.scale, st.shift, st.mean, st.invstd = buf[0], buf[1], buf[2], buf[3]
    call("uem_bn_stats_from_tiles", ptr(ts), tiles, M, cout, ptr(bn.weight.detach()), ptr(bn.bias.detach()), float(bn.eps),
         float(bn.momentum if bn.momentum is not None else 0.1), ptr(bn.running_mean), ptr(bn.running_var),
         ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift), stream())
    return y, st


def conv2d_dgrad(dy, w_t, x_shape, stride=1, pad=0, dil=1, out=None, accumulate=False, algo_cout=None):
    """dx (N,H,W,Cin) from dy (N,Ho,Wo,Cout); w_t = weight_transpose(w) of shape (Cin,KH,KW,Cout)."""
    need_gpu(dy, w_t)
    _f32c(dy, "dgrad dy"), _f32c(w_t, "dgrad w_t")
    cin, kh, kw, cout = w_t.shape
    n, h, w, _ = x_shape
    s = ConvShape()
    s.N, s.H, s.W, s.Cin = n, h, w, cin
    s.Ho, s.Wo, s.Cout = dy.shape[1], dy.shape[2], cout
    s.KH, s.KW, s.stride, s.pad, s.dil = kh, kw, stride, pad, dil
    s.x_ld, s.y_ld = cin, cout
    if out is None:
        out = torch.empty((n, h, w, cin), device=dy.device, dtype=torch.float32)
    else:
        guard_write(out, "conv2d_dgrad(out=)")
    flags = CONV_TRANSPOSED | (CONV_ACCUMULATE if accumulate else 0) | CONV_PREC_BWD
    flops = 2.0 * n * dy.shape[1] * dy.shape[2] * (algo_cout or cout) * kh * kw * cin
    PROF.run("conv_dgrad", flops, lambda: call("uem_conv2d_fwd", ptr(dy), ptr(w_t), None, None, None, ptr(out),
                                               ctypes.byref(s), flags, stream()))
    return out


def conv2d_dgrad_bn_backward(dy, w_t, z, st, gamma_grad, beta_grad, stride=1, pad=0, dil=1):
    """dA = dgrad(dy) for the conv that consumed relu(bn(z)), then the BatchNorm+ReLU backward of that bn:
    returns dz (in dA's buffer).  When the tiles are full and the conv has stride 1, the reduction pass of the
    BN backward (sum dp, sum dp*xhat) runs inside the data-gradient epilogue."""
    cin, kh, kw, cout = w_t.shape
    n, h, w, _ = z.shape
    M = n * h * w
    if stride != 1 or M % 128 != 0 or cin % 64 != 0 or not st.training or not FUSE_BN_BACKWARD:
        da = conv2d_dgrad(dy, w_t, z.shape, stride=stride, pad=pad, dil=dil)
        return bn_backward(z, da, st, gamma_grad, beta_grad, None, True, dx=da)
    need_gpu(dy, w_t, z)
    _f32c(dy, "dgrad dy"), _f32c(w_t, "dgrad w_t"), _f32c(z, "dgrad z")
    s = ConvShape()
    s.N, s.H, s.W, s.Cin = n, h, w, cin
    s.Ho, s.Wo, s.Cout = dy.shape[1], dy.shape[2], cout
    s.KH, s.KW, s.stride, s.pad, s.dil = kh, kw, stride, pad, dil
    s.x_ld, s.y_ld = cin, cout
    da = torch.empty((n, h, w, cin), device=dy.device, dtype=torch.float32)
    tiles = M // 128
    tp = torch.empty((tiles, 2, cin), device=dy.device, dtype=torch.float32)
    vec = st.scale._base if st.scale._base is not None else None
    if vec is None or vec.shape != (4, cin):
        raise UemError("conv2d_dgrad_bn_backward: BNState vectors must live in one (4, C) buffer")
    flops = 2.0 * n * dy.shape[1] * dy.shape[2] * cout * kh * kw * cin
    PROF.run("conv_dgrad", flops, lambda: call("uem_conv2d_dgrad_bnbwd", ptr(dy), ptr(w_t), ptr(da), ctypes.byref(s), ptr(z),
                                               ptr(vec), ptr(tp), CONV_PREC_BWD, stream()))
    tmp = torch.empty((2, cin), device=dy.device, dtype=torch.float32)
    call("uem_bn_bwd_from_tiles", ptr(tp), tiles, cin, ptr(tmp[0]), ptr(tmp[1]), ptr(gamma_grad), ptr(beta_grad), stream())
    call("uem_bn_bwd_apply", ptr(z), ptr(da), None, ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd),
         ptr(tmp[0]), ptr(tmp[1]), M, cin, 1, ptr(da), None, stream())
    return da


def conv2d_dgrad_tail(dy, w_t, x_shape, acc_src=None, acc_bits=None, out=None, accumulate=False, bn_z=None, bn_vec=None,
                      bn_bits=None):
    """The data gradient that closes a bottleneck block's backward, residual bookkeeping in its epilogue:
    dx = dgrad(dy) + acc_src*[acc_bits] (identity gradient gated by the block's output ReLU bits; acc_src may be `out`), or
    out += dgrad(dy) with accumulate=True; with bn_z / bn_vec (4,C) / bn_bits also returns the per-tile partial sums of the
    PREVIOUS block's bn3 backward over the final dx (tiles, 2, C) -- see uem_conv2d_dgrad_tail.  Returns (dx, partials)."""
    need_gpu(dy, w_t)
    _f32c(dy, "dgrad dy"), _f32c(w_t, "dgrad w_t")
    cin, kh, kw, cout = w_t.shape
    n, h, w, _ = x_shape
    s = ConvShape()
    s.N, s.H, s.W, s.Cin = n, h, w, cin
    s.Ho, s.Wo, s.Cout = dy.shape[1], dy.shape[2], cout
    s.KH, s.KW, s.stride, s.pad, s.dil = kh, kw, 1, (kh - 1) // 2, 1
    s.x_ld, s.y_ld = cin, cout
    if out is None:
        out = torch.empty((n, h, w, cin), device=dy.device, dtype=torch.float32)
    else:
        guard_write(out, "conv2d_dgrad_tail(out=)")
    M = n * h * w
    tp = torch.empty((M // 128, 2, cin), device=dy.device, dtype=torch.float32) if bn_z is not None else None
    flags = CONV_PREC_BWD | (CONV_ACCUMULATE if accumulate else 0)
    flops = 2.0 * n * dy.shape[1] * dy.shape[2] * cout * kh * kw * cin
    PROF.run("conv_dgrad", flops, lambda: call("uem_conv2d_dgrad_tail", ptr(dy), ptr(w_t), ptr(out), ctypes.byref(s), ptr(acc_src),
                                               ptr(acc_bits), ptr(bn_z), ptr(bn_vec), ptr(bn_bits), ptr(tp), flags, stream()))
    return out, tp


def dgrad_tail_ok(x_shape, cin):
    n, h, w, _ = x_shape
    return (n * h * w) % 128 == 0 and cin % 64 == 0 and FUSE_BN_BACKWARD and CONV_PREC_BWD == 0


def bn_backward_from_partials(x, dy, st, tp, gamma_grad, beta_grad, ymask_bits, dx=None, dres=None):
    """BatchNorm(+ReLU via packed bits) backward whose reduction pass already ran in a dgrad epilogue (tp = its per-tile
    partial sums): finalize + apply only."""
    C = x.shape[-1]
    M = x.numel() // C
    guard_write(dx, "bn_backward_from_partials(dx=)"), guard_write(dres, "bn_backward_from_partials(dres=)")
    dx = torch.empty_like(x) if dx is None else dx
    tmp = torch.empty((2, C), device=x.device, dtype=torch.float32)
    call("uem_bn_bwd_from_tiles", ptr(tp), tp.shape[0], C, ptr(tmp[0]), ptr(tmp[1]), ptr(gamma_grad), ptr(beta_grad), stream())
    call("uem_bn_bwd_apply", ptr(x), ptr(dy), ptr(ymask_bits), ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd),
         ptr(tmp[0]), ptr(tmp[1]), M, C, 2, ptr(dx), ptr(dres), stream())
    return dx


def conv2d_wgrad(x, dy, dw_ohwi, stride=1, pad=0, dil=1, in_scale=None, in_shift=None, in_relu=False,
                 algo_cout=None, side=False):
    """dw (Cout,KH,KW,Cin) += dy^T * im2col(prologue(x)).  dw must be contiguous (atomics land in it); None (a frozen
    weight, blocks.grad_ohwi) skips the launch.  side: the launch may run on the side stream (on_side) -- ONLY for a dw that nothing on
    the main stream touches before the end of the backward pass, i.e. a view of the gradient arena (the bottleneck blocks); a
    temporary that the caller unpacks right away (the ASPP heads' dwall, the PPM classifier's dw4) must stay on the main stream."""
    if dw_ohwi is None:
        return
    need_gpu(x, dy, dw_ohwi)
    _f32c(x, "wgrad x"), _f32c(dy, "wgrad dy"), _f32c(dw_ohwi, "wgrad dw")
    cout, kh, kw, cin = dw_ohwi.shape
    s = _shape(x, cout, kh, kw, stride, pad, dil)
    if (s.Ho, s.Wo) != (dy.shape[1], dy.shape[2]):
        raise UemError("conv2d_wgrad: dy spatial size mismatch")
    flags = (CONV_IN_AFFINE if in_scale is not None else 0) | (CONV_IN_RELU if in_relu else 0) | CONV_PREC_BWD
    flops = 2.0 * s.N * s.Ho * s.Wo * (algo_cout or cout) * kh * kw * cin

    def launch():
        PROF.run("conv_wgrad", flops, lambda: call("uem_conv2d_wgrad", ptr(x), ptr(dy), ptr(in_scale), ptr(in_shift),
                                                   ptr(dw_ohwi), ctypes.byref(s), flags, stream()), who="conv2d_wgrad")
    if side and SIDE_WGRAD_F32 and in_backward():
        on_side(launch, [t for t in (x, dy, in_scale, in_shift) if t is not None])
    else:
        launch()


def weight_transpose(w_ohwi):
    cout, kh, kw, cin = w_ohwi.shape
    wt = torch.empty((cin, kh, kw, cout), device=w_ohwi.device, dtype=torch.float32)
    call("uem_weight_transpose", ptr(w_ohwi), ptr(wt), cout, kh, kw, cin, stream())
    return wt


# Layouts the kernels read their filter banks in, derived from the OHWI parameters and kept until the weights change: the
# (Cin,KH,KW,Cout) banks of the direct data gradients, the Winograd banks U / U' of either tile size, the stem's padded taps.  A step's
# two backward passes (source and target graph) share them.  Weights change through FusedSGD (bumps WEIGHT_EPOCH: its kernel writes the
# arena behind torch's back) or through torch in-place ops such as load_state_dict's copy_ (bump the tensor's _version).
WEIGHT_EPOCH = 0


def weights_changed():
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


class _WeightPrep:
    """Every derived filter bank anybody asked for since its parameter came to life, refreshed TOGETHER: the first request that finds
    its bank stale (after an optimizer step all of them are) launches uem_weight_prep once over the whole job table instead of one
    small kernel per bank (round 3: 46 weight transposes + the Winograd filter transforms per step).  A new job is served by the same
    entry point on a one-job table and joins the big one; jobs of parameters that no longer exist are dropped when the table is rebuilt.
    Under hipGraph capture the refresh launch is captured like any other (the table itself is never rebuilt there: its set of jobs is
    fixed by the warm-up steps)."""
    BATCH = os.environ.get("UEM_WEIGHT_PREP_BATCH", "1") != "0"

    def __init__(self):
        self.jobs = {}              # (param data_ptr, kind) -> job dict
        self.table = None           # (jobs tensor, starts tensor, njobs, total blocks) on the device; None = rebuild before use

    @staticmethod
    def _stamp(param):
        return (WEIGHT_EPOCH, param._version)

    def _build(self, jobs):
        n = len(jobs)
        arr = (_lib.PrepJob * n)()
        starts = [0]
        for i, j in enumerate(jobs):
            arr[i].src, arr[i].dst, arr[i].kind = j["src"], j["dst"].data_ptr(), j["kind"]
            arr[i].cout, arr[i].cin, arr[i].taps = j["cout"], j["cin"], j["taps"]
            starts.append(starts[-1] + j["blocks"])
        dev = jobs[0]["dst"].device
        jt = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        st = torch.tensor(starts, dtype=torch.int32).to(dev)
        return jt, st, n, starts[-1]

    def _run(self, table):
        jt, st, n, total = table
        call("uem_weight_prep", ptr(jt), ptr(st), n, total, stream())

    def get(self, param, kind, shape, dtype=torch.float32):
        """the derived bank of `param` (a conv weight, logical OIHW in channels_last storage) for `kind`, of shape `shape`"""
        import weakref
        key = (param.data_ptr(), kind)
        job = self.jobs.get(key)
        if job is not None and (job["ref"]() is not param or tuple(job["dst"].shape) != tuple(shape)):
            job = None                                           # the address was recycled by another parameter
        if job is None:
            w = weight_ohwi(param)
            cout, kh, kw, cin = w.shape
            blocks = _lib.load().uem_weight_prep_blocks(kind, cout, cin, kh * kw)
            if blocks <= 0:
                raise UemError(f"weight_prep: kind {kind} does not take a ({cout},{kh},{kw},{cin}) filter bank")
            job = dict(ref=weakref.ref(param), src=w.data_ptr(), dst=torch.empty(shape, device=w.device, dtype=dtype), kind=kind,
                       cout=cout, cin=cin, taps=kh * kw, blocks=blocks, stamp=None)
            self.jobs[key] = job
            self.table = None
            if torch.cuda.is_current_stream_capturing():
                raise UemError("weight_prep: a new filter bank was requested while capturing a hipGraph; run a warm-up step first")
            self._run(self._build([job]))
            job["stamp"] = self._stamp(param)
            return job["dst"]
        if job["stamp"] == self._stamp(param):
            return job["dst"]
        if not self.BATCH:
            self._run(self._build([job]))
            job["stamp"] = self._stamp(param)
            return job["dst"]
        # stale: refresh every live job in one launch
        self.settle()
        self._run(self.table)
        for j in self.live:
            j["stamp"] = self._stamp(j["ref"]())
        return job["dst"]

    def refresh_all(self):
        """every live bank fresh NOW, on the current stream (step.forward_pair: before the second forward stream forks off, so that
        neither forward finds a stale bank and refreshes it under the other's reads)"""
        if not self.jobs:
            return
        self.settle()
        if self.table is None:
            return
        if all(j["stamp"] == self._stamp(j["ref"]()) for j in self.live):
            return
        self._run(self.table)
        for j in self.live:
            j["stamp"] = self._stamp(j["ref"]())

    def settle(self):
        """(re)build the device job table if the set of live jobs changed since it was built: new banks, parameters that were freed or
        moved to another arena (their old address may be unmapped).  GraphedStep calls this before capturing, and keeps the table's
        tensors alive for the captured refresh launch."""
        def gone(j):
            p = j["ref"]()
            return p is None or p.data_ptr() != j["src"]
        if self.table is not None and any(gone(j) for j in self.live):
            self.table = None
        if self.table is None and self.jobs:
            if torch.cuda.is_current_stream_capturing():
                raise UemError("weight_prep: the set of filter banks changed while capturing a hipGraph; run a warm-up step first")
            for k in [k for k, j in self.jobs.items() if gone(j)]:
                del self.jobs[k]
            self.live = list(self.jobs.values())
            self.table = self._build(self.live) if self.live else None
        return self.table


class _WeightPrepPerDevice:
    def __init__(self):
        self.by_device = {}

    def refresh_all(self):
        prep = self.by_device.get(torch.cuda.current_device())
        if prep is not None:
            prep.refresh_all()

    def get(self, param, kind, shape, dtype=torch.float32):
        need_gpu(param)
        prep = self.by_device.get(param.device.index)
        if prep is None:
            prep = self.by_device[param.device.index] = _WeightPrep()
        return prep.get(param, kind, shape, dtype)

    def settle(self):
        """every device's job table as it stands (built now if stale): [(jobs tensor, starts tensor, njobs, blocks)]"""
        return [t for t in (p.settle() for p in self.by_device.values()) if t is not None]

    def hold(self):
        """Strong references to everything the settled tables point at: [(parameter, derived bank)] of every live job.  A captured
        `uem_weight_prep` launch reads each job's parameter and writes its bank on EVERY replay -- also the jobs of other models that
        were alive on the device at capture time -- so whoever owns the graph keeps this list for the graph's life: none
        of those parameters can then be freed, no job goes `gone`, no bank is released under the graph."""
        self.settle()
        out = []
        for p in self.by_device.values():
            for j in getattr(p, "live", []) if p.table is not None else []:
                out.append((j["ref"](), j["dst"]))
        return out


PREP = _WeightPrepPerDevice()


def weight_transpose_cached(param):
    cout, cin, kh, kw = param.shape
    return PREP.get(param, _lib.PREP_TRANSPOSE, (cin, kh, kw, cout))


def stem_weight_packed(param):
    """the 7x7 stem filter bank as the stem kernels read it: (64, 7, 8, 4), kx and the channel padded with zeros"""
    return PREP.get(param, _lib.PREP_STEM_PACK, (64, 7, 8, 4))


# ------------------------------------------------------------------------------------------------
# Winograd F(2x2, 3x3) / F(4x4, 3x3): the stride-1 3x3 convolutions of the deep layers (csrc/winograd.hip), exact fp32 arithmetic,
# 2.25x / 4x fewer multiplies
# ------------------------------------------------------------------------------------------------
WINOGRAD = os.environ.get("UEM_WINOGRAD", "1") != "0"
# Narrower layers are HBM-bound on the larger transform tensors.  F(2x2,3x3), whose V / M tensors are 4x the activation
# (profiles/r03_d_winograd_vs_direct.txt): at 128 channels (layer2) the forward still gains 7 % and the weight gradient 35 %, the data
# gradient loses; at 64 channels everything loses.  F(4x4,3x3), 2.25x the activation and 4x fewer multiplies
# (profiles/r04_a_winograd_vs_direct.txt): wins forward, data gradient and weight gradient down to the 64 channels of layer1
# (0.36 / 0.41 / 0.24 ms against 0.42 / 0.48 / 0.38 direct at B = 32).
WINOGRAD_MIN_CH = int(os.environ.get("UEM_WINOGRAD_MIN_CH", "128"))
WINOGRAD_MIN_CH_DGRAD = int(os.environ.get("UEM_WINOGRAD_MIN_CH_DGRAD", "256"))
WINOGRAD_MIN_CH_F4 = int(os.environ.get("UEM_WINOGRAD_MIN_CH_F4", "64"))
# F(4x4,3x3) where the tile count allows.  Its rounding error is 1.2-2.5e-6 per convolution against float64 where F(2x2,3x3) has
# 3-6e-7 and the direct fmaf chain 4-9e-7 (scripts/bench_winograd.py, WINO_ERR=1).  Backward: far below the 2-3 % noise floor of the
# encoder's gradients, so every eligible layer takes it.  Forward: layer2, layer3, layer4 and the heads (>= 128 channels).
# Round 4 stopped at 256 channels: with layer1 AND layer2 on it one head-side update of the B = 2, 256x256 fixtures came out at 3.06x its
# noise floor against the 3.0 bound.  Round 5 re-measured layer2 alone (UEM_WINOGRAD_MIN_CH_F4_FWD=128 against 256, the six
# reference-golden steps side by side, gpurun_out/r5g): the per-fixture maxima are the same tensors at the same values (2.78, 2.87: not
# layer2's doing), the 512x512 fixture's worst tensor goes 2.34 -> 1.15, the medians 0.95 / 0.95 / 0.93 / 0.62 / 0.96 / 0.75 ->
# 0.95 / 0.99 / 0.93 / 0.86 / 1.05 / 0.81 against the 1.5 bound, logits unchanged -- and the step gains 0.86 ms (103.03 -> 102.17 on
# one box).  layer1 (64 channels: -0.2 ms more) keeps the direct forward.
WINOGRAD_F4_BWD = os.environ.get("UEM_WINOGRAD_F4_BWD", "1") != "0"
WINOGRAD_F4_FWD = os.environ.get("UEM_WINOGRAD_F4_FWD", "1") != "0"
WINOGRAD_MIN_CH_F4_FWD = int(os.environ.get("UEM_WINOGRAD_MIN_CH_F4_FWD", "128"))
# V (the transformed input the weight gradient reduces over) is kept by the forward only up to this many bytes per convolution
# (it is 4x / 2.25x the conv input); above it, and whenever forward and backward use different tile sizes, the backward
# recomputes it from the saved conv input
WINOGRAD_SAVE_V_BYTES = int(float(os.environ.get("UEM_WINOGRAD_SAVE_V_GB", "3")) * 2 ** 30)


class WinoPlan:
    """How one stride-1 3x3 conv runs: mf / mb = output tile edge of the forward / backward pass (0 = direct kernel, 2 = F(2x2,3x3),
    4 = F(4x4,3x3)); dgrad / wgrad = which gradients are on the Winograd path; keep_v = the forward keeps its transformed input for the
    weight gradient."""
    __slots__ = ("mf", "mb", "dgrad", "wgrad", "keep_v")

    def __init__(self, mf, mb, dgrad, wgrad, keep_v):
        self.mf, self.mb, self.dgrad, self.wgrad, self.keep_v = mf, mb, dgrad, wgrad, keep_v


def _wino_edge(x_shape, cout, dil, want4, min_ch4):
    """largest usable tile edge for one direction: 4, 2 or 0 (direct)"""
    n, h, w, cin = x_shape
    lo, hi = min(cin, cout), max(cin, cout)
    t4 = n * h * w // 16
    if want4 and lo >= min_ch4 and h % (4 * dil) == 0 and w % (4 * dil) == 0 and t4 % 128 == 0 and 36 * t4 * hi < 2 ** 30:
        return 4
    t2 = n * h * w // 4
    if lo >= WINOGRAD_MIN_CH and h % (2 * dil) == 0 and w % (2 * dil) == 0 and t2 % 128 == 0 and 16 * t2 * hi < 2 ** 30:
        return 2
    return 0


def wino_plan(x_shape, cout, kh, kw, stride, pad, dil):
    """WinoPlan of a conv, or None when neither pass takes the Winograd path.  Exact fp32 only (the bf16 operand mode of the
    bf16-storage islands stays on the direct kernels)."""
    n, h, w, cin = x_shape
    if not (WINOGRAD and CONV_PREC == 0 and CONV_PREC_BWD == 0 and kh == 3 and kw == 3 and stride == 1 and pad == dil and dil in (1, 2)):
        return None
    if cin % 64 or cout % 64:
        return None
    mf = _wino_edge(x_shape, cout, dil, WINOGRAD_F4_FWD, max(WINOGRAD_MIN_CH_F4, WINOGRAD_MIN_CH_F4_FWD))
    mb = _wino_edge(x_shape, cout, dil, WINOGRAD_F4_BWD, WINOGRAD_MIN_CH_F4)
    if mf == 0 and mb == 0:
        return None
    lo = min(cin, cout)
    vbytes = 4 * (mf + 2) ** 2 * (n * h * w // (mf * mf)) * cin if mf else 0
    keep_v = mf != 0 and mf == mb and vbytes <= WINOGRAD_SAVE_V_BYTES
    dgrad = mb == 4 or (mb == 2 and lo >= WINOGRAD_MIN_CH_DGRAD)
    # the weight gradient that has to recompute V pays a second input transform: at 64 channels that eats F(4x4)'s gain over the
    # direct kernel (0.377 against 0.383 ms at B = 32), from 128 channels up it still wins (0.21 against 0.36)
    wgrad = mb != 0 and (keep_v or lo >= WINOGRAD_MIN_CH)
    if not (mf or dgrad or wgrad):
        return None
    return WinoPlan(mf, mb, dgrad, wgrad, keep_v)


def wino_ok(x_shape, cout, kh, kw, stride, pad, dil):
    """Does the forward of this conv take a Winograd path?"""
    plan = wino_plan(x_shape, cout, kh, kw, stride, pad, dil)
    return plan is not None and plan.mf != 0


def wino_filter_cached(param, transposed, m=2):
    """U = G w G^T of a 3x3 conv weight (transposed: the data gradient's flipped bank), kept until the weights change."""
    cout, cin = param.shape[0], param.shape[1]
    npos = (m + 2) ** 2
    kind = {(2, False): _lib.PREP_WINO2, (2, True): _lib.PREP_WINO2_T, (4, False): _lib.PREP_WINO4, (4, True): _lib.PREP_WINO4_T}[(m, bool(transposed))]
    return PREP.get(param, kind, (npos, cin, cout) if transposed else (npos, cout, cin))


def wino_input(x, dil, in_scale=None, in_shift=None, in_relu=False, m=2, which=0):
    """which: 0 forward, 1 data gradient (x = dY), 2 weight gradient recomputing V -- names the kernel instantiation for profiles"""
    n, h, w, c = x.shape
    v = torch.empty(((m + 2) ** 2, n * h * w // (m * m), c), device=x.device, dtype=torch.float32)
    call("uem_wino_input", ptr(x), ptr(in_scale), ptr(in_shift), 1 if in_relu else 0, ptr(v), n, h, w, c, dil, m, which, stream())
    return v


def wino_gemm(v, u, data_gradient=False):
    npos, t, k = v.shape
    nn = u.shape[1]
    m = torch.empty((npos, t, nn), device=v.device, dtype=torch.float32)
    call("uem_wino_gemm", ptr(v), ptr(u), ptr(m), t, k, nn, npos, 1 if data_gradient else 0, stream())
    return m


def _wino_flops(M, cout, cin, m):
    """(algorithmic, executed) flops of a 3x3 conv on the Winograd path: (m+2)^2 multiplies per m^2 outputs and channel pair"""
    return 2.0 * M * cout * 9 * cin, 2.0 * M * cout * cin * (m + 2) ** 2 / (m * m)


def conv3x3_wino_bn(x, param, bn, dil, in_scale=None, in_shift=None, in_relu=False, m=2):
    """3x3 stride-1 conv (pad = dil) + training-mode BatchNorm statistics on the Winograd path -> (y, BNState, V); V (the
    transformed input, 4x / 2.25x the input's bytes) is what the weight gradient reduces over."""
    need_gpu(x)
    _f32c(x, "conv3x3 x")
    n, h, w, cin = x.shape
    u = wino_filter_cached(param, False, m)
    cout = u.shape[1]
    M = n * h * w
    y = torch.empty((n, h, w, cout), device=x.device, dtype=torch.float32)
    ts = torch.empty((M // 128, 2, cout), device=x.device, dtype=torch.float32)
    box = {}

    def run():
        box["v"] = wino_input(x, dil, in_scale, in_shift, in_relu, m)
        mt = wino_gemm(box["v"], u)
        call("uem_wino_output", ptr(mt), ptr(y), n, h, w, cout, dil, m, ptr(ts), None, None, None, stream())

    alg, exe = _wino_flops(M, cout, cin, m)
    PROF.run("conv_fwd", alg, run, executed=exe)
    st = BNState()
    st.training = True
    buf = torch.empty((4, cout), device=x.device, dtype=torch.float32)
    st.scale, st.shift, st.mean, st.invstd = buf[0], buf[1], buf[2], buf[3]
    call("uem_bn_stats_from_tiles", ptr(ts), M // 128, M, cout, ptr(bn.weight.detach()), ptr(bn.bias.detach()), float(bn.eps),
         float(bn.momentum if bn.momentum is not None else 0.1), ptr(bn.running_mean), ptr(bn.running_var),
         ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift), stream())
    return y, st, box["v"]


def conv3x3_wino(x, param, dil, in_scale=None, in_shift=None, in_relu=False, want_v=False, m=2):
    """plain Winograd forward (no statistics) -> y or (y, V)"""
    need_gpu(x)
    _f32c(x, "conv3x3 x")
    n, h, w, cin = x.shape
    u = wino_filter_cached(param, False, m)
    cout = u.shape[1]
    y = torch.empty((n, h, w, cout), device=x.device, dtype=torch.float32)
    box = {}

    def run():
        box["v"] = wino_input(x, dil, in_scale, in_shift, in_relu, m)
        mt = wino_gemm(box["v"], u)
        call("uem_wino_output", ptr(mt), ptr(y), n, h, w, cout, dil, m, None, None, None, None, stream())

    alg, exe = _wino_flops(n * h * w, cout, cin, m)
    PROF.run("conv_fwd", alg, run, executed=exe)
    return (y, box["v"]) if want_v else y


def conv3x3_wino_dgrad(dy, param, dil, bn_z=None, bn_st=None, m=2):
    """dx of the 3x3 stride-1 conv on the Winograd path: dx = conv(dy, flipped transposed filters).  With bn_z / bn_st also the
    per-128-pixel partial sums of the BatchNorm+ReLU backward of the layer dx feeds -> (dx, partials or None)."""
    need_gpu(dy)
    _f32c(dy, "dgrad dy")
    n, h, w, cout = dy.shape
    ut = wino_filter_cached(param, True, m)                  # (npos, cin, cout)
    cin = ut.shape[1]
    M = n * h * w
    dx = torch.empty((n, h, w, cin), device=dy.device, dtype=torch.float32)
    tp = torch.empty((M // 128, 2, cin), device=dy.device, dtype=torch.float32) if bn_z is not None else None
    vec = None
    if bn_z is not None:
        vec = bn_st.scale._base if bn_st.scale._base is not None else None
        if vec is None or vec.shape != (4, cin):
            raise UemError("conv3x3_wino_dgrad: BNState vectors must live in one (4, C) buffer")

    def run():
        mt = wino_gemm(wino_input(dy, dil, m=m, which=1), ut, data_gradient=True)
        call("uem_wino_output", ptr(mt), ptr(dx), n, h, w, cin, dil, m, None, ptr(bn_z), ptr(vec), ptr(tp), stream())

    alg, exe = _wino_flops(M, cout, cin, m)
    PROF.run("conv_dgrad", alg, run, executed=exe)
    return dx, tp


def conv3x3_wino_dgrad_bn_backward(dy, param, z, st, gamma_grad, beta_grad, dil, m=2):
    """Winograd twin of conv2d_dgrad_bn_backward: dA = dgrad(dy), then the BatchNorm+ReLU backward of the bn that produced the
    conv's input (reduction pass inside the output transform) -> dz in dA's buffer."""
    cin = z.shape[-1]
    M = z.numel() // cin
    da, tp = conv3x3_wino_dgrad(dy, param, dil, bn_z=z, bn_st=st, m=m)
    tmp = torch.empty((2, cin), device=dy.device, dtype=torch.float32)
    call("uem_bn_bwd_from_tiles", ptr(tp), M // 128, cin, ptr(tmp[0]), ptr(tmp[1]), ptr(gamma_grad), ptr(beta_grad), stream())
    call("uem_bn_bwd_apply", ptr(z), ptr(da), None, ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd),
         ptr(tmp[0]), ptr(tmp[1]), M, cin, 1, ptr(da), None, stream())
    return da


class _WinoSideWs:
    """Temporaries of the Winograd weight gradient when it runs on the SIDE stream, owned per shape and pre-sized:
    the transformed dY (npos, T, Cout), the transformed-domain accumulator dU (npos, Cout, Cin) and, where V is recomputed, V
    (npos, T, Cin).  They used to be allocated inside the launch, i.e. on the side stream -- which kept the side stream out of a hipGraph
    capture (round 5: hipErrorStreamCaptureInvalidated) and grew the side stream's own allocator pool (peak reserved 63 -> 82 GB).
    Launches of one shape run one after the other on the side stream, so one set per shape is enough; it is allocated on the MAIN stream
    at first use (the side stream waits for the main stream before every launch) and lives as long as the process (1.2 GB for R50 at
    B = 32: 36 x T x C floats per distinct 3x3 layer shape).  Main-stream launches keep their per-call allocations."""
    sets = {}

    @classmethod
    def get(cls, device, npos, t, cout, cin, need_v):
        key = (device.index, npos, t, cout, cin)
        ws = cls.sets.get(key)
        if ws is None:
            ws = cls.sets[key] = dict(dm=torch.empty((npos, t, cout), device=device, dtype=torch.float32),
                                      du=torch.empty((npos, cout, cin), device=device, dtype=torch.float32), v=None)
        if need_v and ws["v"] is None:
            ws["v"] = torch.empty((npos, t, cin), device=device, dtype=torch.float32)
        return ws


def conv3x3_wino_wgrad(v, dy, dw_ohwi, dil, x=None, in_scale=None, in_shift=None, in_relu=False, m=None, side=False):
    """dw (Cout,3,3,Cin) += weight gradient from dy (N,H,W,Cout) and the transformed input V (npos, T, Cin) the forward saved; with
    v None, V is recomputed from the conv input x (through the producer's BatchNorm affine + ReLU) at tile edge m."""
    if dw_ohwi is None:
        return
    need_gpu(dy, dw_ohwi)
    _f32c(dy, "wgrad dy"), _f32c(dw_ohwi, "wgrad dw")
    n, h, w, cout = dy.shape
    if v is not None:
        m = {16: 2, 36: 4}[v.shape[0]]
        cin = v.shape[2]
    else:
        cin = x.shape[-1]
    npos, t = (m + 2) ** 2, n * h * w // (m * m)
    on_the_side = side and SIDE_WGRAD_F32 and in_backward()
    ws = _WinoSideWs.get(dy.device, npos, t, cout, cin, v is None) if on_the_side else None

    def run():
        if v is not None:
            vv = v
        elif ws is not None:
            vv = ws["v"]
            call("uem_wino_input", ptr(x), ptr(in_scale), ptr(in_shift), 1 if in_relu else 0, ptr(vv), n, h, w, cin, dil, m, 2, stream())
        else:
            vv = wino_input(x, dil, in_scale, in_shift, in_relu, m, which=2)
        if ws is not None:
            dm, du = ws["dm"], ws["du"]
        else:
            dm = torch.empty((npos, t, cout), device=dy.device, dtype=torch.float32)
            du = torch.empty((npos, cout, cin), device=dy.device, dtype=torch.float32)   # cleared by the dY transform's launch
        call("uem_wino_dy", ptr(dy), ptr(dm), n, h, w, cout, dil, m, ptr(du), du.numel(), stream())
        call("uem_wino_wgrad_gemm", ptr(vv), ptr(dm), ptr(du), t, cin, cout, npos, stream())
        call("uem_wino_filter_grad", ptr(du), ptr(dw_ohwi), cout, cin, m, stream())

    alg, exe = _wino_flops(n * h * w, cout, cin, m)
    if on_the_side:             # side: as in conv2d_wgrad
        on_side(lambda: PROF.run("conv_wgrad", alg, run, executed=exe, who="conv3x3_wino_wgrad"),
                [t_ for t_ in (v, dy, x, in_scale, in_shift) if t_ is not None], "Winograd weight gradient")
    else:
        PROF.run("conv_wgrad", alg, run, executed=exe)


def nchw3_to_nhwc4(x):
    need_gpu(x)
    x = _f32c(x.contiguous(), "image")
    n, c, h, w = x.shape
    if c != 3:
        raise UemError("stem expects 3 input channels")
    x4 = torch.empty((n, h, w, 4), device=x.device, dtype=torch.float32)
    call("uem_nchw3_to_nhwc4", ptr(x), ptr(x4), n, h, w, stream())
    return x4


# The stem's own kernels (csrc/stem.hip, round 5): exact fp32 operands, whole 8 x 32 output tiles.  UEM_STEM_KERNEL=0: the generic
# register-staged kernels of rounds 1-4 everywhere (they also serve the bf16-operand islands and odd sizes).
STEM_KERNEL = os.environ.get("UEM_STEM_KERNEL", "1") != "0"


def stem_tiles_ok(h, w):
    return STEM_KERNEL and conv_out_size(h, 7, 2, 3, 1) % 8 == 0 and conv_out_size(w, 7, 2, 3, 1) % 32 == 0


def stem_conv(x4, w_ohwi, w8=None):
    """w8: the packed taps of the generic kernel when the caller holds them (stem_weight_packed(param)), or a callable that makes them
    (asked only when that kernel runs); else packed here from w_ohwi"""
    n, h, w, _ = x4.shape
    if CONV_PREC == 0 and stem_tiles_ok(h, w) and w_ohwi is not None:
        y = torch.empty((n, conv_out_size(h, 7, 2, 3, 1), conv_out_size(w, 7, 2, 3, 1), 64), device=x4.device, dtype=torch.float32)
        wc = _f32c(w_ohwi, "stem filter bank")
        PROF.run("conv_fwd", 2.0 * y.numel() * 147, lambda: call("uem_stem_conv_fwd", ptr(x4), ptr(wc), ptr(y), n, h, w, None, stream()),
                 executed=2.0 * y.numel() * 148)
        return y
    if callable(w8):
        w8 = w8()
    if w8 is None:
        w8 = torch.empty((64, 7, 8, 4), device=x4.device, dtype=torch.float32)
        call("uem_stem_pack_weight", ptr(w_ohwi), ptr(w8), stream())
    y = torch.empty((n, conv_out_size(h, 7, 2, 3, 1), conv_out_size(w, 7, 2, 3, 1), 64), device=x4.device,
                    dtype=torch.float32)
    flops = 2.0 * y.numel() * 147
    PROF.run("conv_fwd", flops, lambda: call("uem_conv2d_stem_fwd", ptr(x4), ptr(w8), ptr(y), n, h, w, stream()))
    return y


def stem_conv_bn(x4, w_ohwi, bn, w8=None):
    """The stem conv followed by its BatchNorm statistics -> (z, BNState): out of the conv epilogue when the tiles are full and
    the BatchNorm is in training mode, else stem_conv + bn_stats (eval mode, odd sizes)."""
    n, h, w, _ = x4.shape
    ho, wo = conv_out_size(h, 7, 2, 3, 1), conv_out_size(w, 7, 2, 3, 1)
    M = n * ho * wo
    training = bn.training or bn.running_mean is None
    if not training or M % 128 != 0 or not FUSE_BN_STATS:
        z = stem_conv(x4, w_ohwi, w8)
        return z, bn_stats(z, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, training, bn.eps,
                           bn.momentum if bn.momentum is not None else 0.1)
    own_kernel = CONV_PREC == 0 and stem_tiles_ok(h, w) and w_ohwi is not None
    if not own_kernel:
        if callable(w8):
            w8 = w8()
        if w8 is None:
            w8 = torch.empty((64, 7, 8, 4), device=x4.device, dtype=torch.float32)
            call("uem_stem_pack_weight", ptr(w_ohwi), ptr(w8), stream())
    z = torch.empty((n, ho, wo, 64), device=x4.device, dtype=torch.float32)
    ts = torch.empty((M // 128, 2, 64), device=x4.device, dtype=torch.float32)
    if own_kernel:
        wc = _f32c(w_ohwi, "stem filter bank")
        PROF.run("conv_fwd", 2.0 * z.numel() * 147, lambda: call("uem_stem_conv_fwd", ptr(x4), ptr(wc), ptr(z), n, h, w, ptr(ts), stream()),
                 executed=2.0 * z.numel() * 148)
    else:
        PROF.run("conv_fwd", 2.0 * z.numel() * 147, lambda: call("uem_conv2d_stem_fwd_stats", ptr(x4), ptr(w8), ptr(z), n, h, w, ptr(ts), CONV_PREC, stream()))
    st = BNState()
    st.training = True
    buf = torch.empty((4, 64), device=x4.device, dtype=torch.float32)
    st.scale, st.shift, st.mean, st.invstd = buf[0], buf[1], buf[2], buf[3]
    call("uem_bn_stats_from_tiles", ptr(ts), M // 128, M, 64, ptr(bn.weight.detach()), ptr(bn.bias.detach()), float(bn.eps),
         float(bn.momentum if bn.momentum is not None else 0.1), ptr(bn.running_mean), ptr(bn.running_var),
         ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift), stream())
    return z, st


def stem_wgrad(x4, dy, dw_ohwi):
    if dw_ohwi is None:                      # frozen stem (freeze_at >= 1)
        return
    n, h, w, _ = x4.shape
    if CONV_PREC_BWD == 0 and stem_tiles_ok(h, w) and dw_ohwi.is_contiguous():
        ws = torch.empty(_lib.load().uem_stem_conv_wgrad_workspace_floats(), device=x4.device, dtype=torch.float32)
        PROF.run("conv_wgrad", 2.0 * dy.numel() * 147,
                 lambda: call("uem_stem_conv_wgrad", ptr(x4), ptr(dy), ptr(dw_ohwi), ptr(ws), n, h, w, stream()),
                 executed=2.0 * dy.numel() * 160)
        return
    dw8 = torch.zeros((64, 7, 8, 4), device=x4.device, dtype=torch.float32)
    PROF.run("conv_wgrad", 2.0 * dy.numel() * 147,
             lambda: call("uem_conv2d_stem_wgrad_prec", ptr(x4), ptr(dy), ptr(dw8), n, h, w, CONV_PREC_BWD, stream()))
    call("uem_stem_unpack_grad", ptr(dw8), ptr(dw_ohwi), stream())


def bias_grad(dy2d, db, C, ld):
    call("uem_bias_grad", ptr(dy2d), ptr(db), dy2d.numel() // ld, C, ld, stream())


# ------------------------------------------------------------------------------------------------
# normalisation
# ------------------------------------------------------------------------------------------------
class BNState:
    """Per-call BatchNorm operands: scale/shift (conv prologue form) and the saved statistics."""
    __slots__ = ("scale", "shift", "mean", "invstd", "training")


def bn_stats(x, gamma, beta, running_mean, running_var, training, eps=BN_EPS, momentum=BN_MOMENTUM):
    """Batch statistics of x (.., C) -> BNState; updates the running stats in training mode."""
    need_gpu(x)
    C = x.shape[-1]
    M = x.numel() // C
    st = BNState()
    st.training = training
    dev = x.device
    buf = torch.empty((4, C), device=dev, dtype=torch.float32)
    st.scale, st.shift, st.mean, st.invstd = buf[0], buf[1], buf[2], buf[3]
    if training:
        ws = torch.empty(_lib.load().uem_bn_workspace_floats(M, C), device=dev, dtype=torch.float32)
        call("uem_bn_stats", ptr(x), M, C, C, ptr(gamma), ptr(beta), eps, momentum, ptr(running_mean),
             ptr(running_var), ptr(st.mean), ptr(st.invstd), ptr(st.scale), ptr(st.shift), ptr(ws), stream())
    else:
        call("uem_bn_eval_affine", ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), eps,
             ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd), C, stream())
    return st


def affine_act(x, st, res=None, res_st=None, relu=True, out=None, want_bits=False):
    """y = relu?(x*scale + shift (+ res [*res_scale + res_shift])).  want_bits: also return the packed sign bits of y
    (int32 words, 1/32 of y) that bn_backward(ymask_bits=...) reads instead of y."""
    C = x.shape[-1]
    guard_write(out, "affine_act(out=)")
    out = torch.empty_like(x) if out is None else out
    bits = torch.empty(x.numel() // 32, device=x.device, dtype=torch.int32) if want_bits else None
    call("uem_affine_act", ptr(x), ptr(st.scale), ptr(st.shift), ptr(res),
         ptr(res_st.scale) if res_st is not None else None, ptr(res_st.shift) if res_st is not None else None,
         ptr(out), x.numel() // C, C, 1 if relu else 0, ptr(bits), stream())
    return (out, bits) if want_bits else out


def bn_backward(x, dy, st, gamma_grad, beta_grad, ymask=None, relu=True, dx=None, dres=None, ymask_bits=None):
    """BatchNorm(+ReLU) backward.  Accumulates into gamma_grad/beta_grad; returns dx (may alias dy)."""
    C = x.shape[-1]
    M = x.numel() // C
    guard_write(dx, "bn_backward(dx=)"), guard_write(dres, "bn_backward(dres=)")
    dx = torch.empty_like(x) if dx is None else dx
    if ymask_bits is not None:
        ymask, relu = ymask_bits, 2          # UEM_RELU_BITS
    if st.training:
        tmp = torch.empty((2, C), device=x.device, dtype=torch.float32)
        ws = torch.empty(_lib.load().uem_bn_workspace_floats(M, C), device=x.device, dtype=torch.float32)
        call("uem_bn_bwd_reduce", ptr(x), ptr(dy), ptr(ymask), ptr(st.scale), ptr(st.shift), ptr(st.mean),
             ptr(st.invstd), M, C, int(relu), ptr(tmp[0]), ptr(tmp[1]), ptr(gamma_grad), ptr(beta_grad), ptr(ws),
             stream())
        call("uem_bn_bwd_apply", ptr(x), ptr(dy), ptr(ymask), ptr(st.scale), ptr(st.shift), ptr(st.mean),
             ptr(st.invstd), ptr(tmp[0]), ptr(tmp[1]), M, C, int(relu), ptr(dx), ptr(dres), stream())
    else:
        # frozen statistics (eval-mode BatchNorm inside a training graph: ResNetEncoder batchnorm_trainable=False, reference
        # resnet.py:112-117,183-190): y = x*scale + shift with constant scale / shift, so dx = dp*scale; gamma / beta, when they
        # are still trainable, get sum dp*xhat / sum dp with xhat taken from the running statistics
        if gamma_grad is not None or beta_grad is not None:
            tmp = torch.empty((2, C), device=x.device, dtype=torch.float32)
            ws = torch.empty(_lib.load().uem_bn_workspace_floats(M, C), device=x.device, dtype=torch.float32)
            call("uem_bn_bwd_reduce", ptr(x), ptr(dy), ptr(ymask), ptr(st.scale), ptr(st.shift), ptr(st.mean),
                 ptr(st.invstd), M, C, int(relu), ptr(tmp[0]), ptr(tmp[1]), ptr(gamma_grad), ptr(beta_grad), ptr(ws),
                 stream())
        call("uem_affine_act_bwd", ptr(x), ptr(dy), ptr(ymask), ptr(st.scale), ptr(st.shift), M, C, int(relu), ptr(dx),
             ptr(dres), stream())
    return dx


BN_PAIR = os.environ.get("UEM_BN_PAIR", "1") != "0"


def bn_backward_pair(x1, x2, dy, bits, st1, st2, tiles1, gg1, gb1, gg2, gb2, dx2=None):
    """bn3's and the downsample BatchNorm's backward of a bottleneck block with a downsample branch: both read dy gated by the same
    packed ReLU bits, so their apply passes run as one (uem_bn_bwd_apply_pair).  tiles1: bn3's per-tile partial sums when its reduction
    rode in the next block's data-gradient epilogue (blocks._Link), else None (reduced here).  Returns (dx1, dx2); dx2 may be dy.
    Training-mode statistics; None when the pair kernel does not take the shape or a BatchNorm is frozen (the caller runs the two)."""
    if not (BN_PAIR and st1.training and st2.training):
        return None
    C = x1.shape[-1]
    M = x1.numel() // C
    lib = _lib.load()
    if C % 32 != 0 or x2.shape != x1.shape or bits is None:
        return None
    tmp = torch.empty((4, C), device=x1.device, dtype=torch.float32)
    dx1 = torch.empty_like(x1)
    guard_write(dx2, "bn_backward_pair(dx2=)")
    dx2 = torch.empty_like(x2) if dx2 is None else dx2
    # the reductions first (they are needed either way), then the one apply; a refusal of the pair entry is decided by shape alone
    ws = torch.empty(lib.uem_bn_workspace_floats(M, C), device=x1.device, dtype=torch.float32)
    if tiles1 is not None:
        call("uem_bn_bwd_from_tiles", ptr(tiles1), tiles1.shape[0], C, ptr(tmp[0]), ptr(tmp[1]), ptr(gg1), ptr(gb1), stream())
    else:
        call("uem_bn_bwd_reduce", ptr(x1), ptr(dy), ptr(bits), ptr(st1.scale), ptr(st1.shift), ptr(st1.mean), ptr(st1.invstd), M, C, 2,
             ptr(tmp[0]), ptr(tmp[1]), ptr(gg1), ptr(gb1), ptr(ws), stream())
    call("uem_bn_bwd_reduce", ptr(x2), ptr(dy), ptr(bits), ptr(st2.scale), ptr(st2.shift), ptr(st2.mean), ptr(st2.invstd), M, C, 2,
         ptr(tmp[2]), ptr(tmp[3]), ptr(gg2), ptr(gb2), ptr(ws), stream())
    if not _lib.try_call("uem_bn_bwd_apply_pair", ptr(x1), ptr(x2), ptr(dy), ptr(bits), ptr(st1.scale), ptr(st1.mean), ptr(st1.invstd),
                         ptr(tmp[0]), ptr(tmp[1]), ptr(st2.scale), ptr(st2.mean), ptr(st2.invstd), ptr(tmp[2]), ptr(tmp[3]), M, C,
                         ptr(dx1), ptr(dx2), stream()):
        call("uem_bn_bwd_apply", ptr(x1), ptr(dy), ptr(bits), ptr(st1.scale), ptr(st1.shift), ptr(st1.mean), ptr(st1.invstd),
             ptr(tmp[0]), ptr(tmp[1]), M, C, 2, ptr(dx1), None, stream())
        call("uem_bn_bwd_apply", ptr(x2), ptr(dy), ptr(bits), ptr(st2.scale), ptr(st2.shift), ptr(st2.mean), ptr(st2.invstd),
             ptr(tmp[2]), ptr(tmp[3]), M, C, 2, ptr(dx2), None, stream())
    return dx1, dx2


def maxpool_fwd(x, want_idx):
    n, h, w, c = x.shape
    ho, wo = conv_out_size(h, 3, 2, 1, 1), conv_out_size(w, 3, 2, 1, 1)
    y = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.float32)
    idx = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.uint8) if want_idx else None
    call("uem_maxpool3x3s2_fwd", ptr(x), ptr(y), ptr(idx), n, h, w, c, stream())
    return y, idx


def maxpool_affine_fwd(z, st, want_idx):
    """maxpool3x3s2(relu(z*scale + shift)) without materialising the normalised map."""
    n, h, w, c = z.shape
    ho, wo = conv_out_size(h, 3, 2, 1, 1), conv_out_size(w, 3, 2, 1, 1)
    y = torch.empty((n, ho, wo, c), device=z.device, dtype=torch.float32)
    idx = torch.empty((n, ho, wo, c), device=z.device, dtype=torch.uint8) if want_idx else None
    call("uem_maxpool3x3s2_affine_fwd", ptr(z), ptr(st.scale), ptr(st.shift), ptr(y), ptr(idx), n, h, w, c, stream())
    return y, idx


def bn_backward_pooled(z, dy_pool, idx, st, gamma_grad, beta_grad):
    """BatchNorm+ReLU backward of the layer in front of the 3x3/s2 max-pool, from the POOLED gradient and the argmax taps
    (no full-size gradient tensor in between) -> dz.  Training-mode statistics; eval mode goes through maxpool_bwd + bn_backward."""
    n, h, w, c = z.shape
    M = n * h * w
    tmp = torch.empty((2, c), device=z.device, dtype=torch.float32)
    ws = torch.empty(_lib.load().uem_bn_workspace_floats(M, c), device=z.device, dtype=torch.float32)
    call("uem_bn_bwd_reduce_pool", ptr(z), ptr(dy_pool), ptr(idx), ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd),
         n, h, w, c, 1, ptr(tmp[0]), ptr(tmp[1]), ptr(gamma_grad), ptr(beta_grad), ptr(ws), stream())
    dz = torch.empty_like(z)
    call("uem_bn_bwd_apply_pool", ptr(z), ptr(dy_pool), ptr(idx), ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.invstd),
         ptr(tmp[0]), ptr(tmp[1]), n, h, w, c, 1, ptr(dz), stream())
    return dz


def maxpool_bwd(dy, idx, in_shape):
    n, h, w, c = in_shape
    dx = torch.empty(in_shape, device=dy.device, dtype=torch.float32)
    call("uem_maxpool3x3s2_bwd", ptr(dy), ptr(idx), ptr(dx), n, h, w, c, stream())
    return dx


def instnorm_fwd(x, eps=1e-5):
    n, h, w, c = x.shape
    y = torch.empty_like(x)
    stats = torch.empty((2, n, c), device=x.device, dtype=torch.float32)
    call("uem_instnorm_fwd", ptr(x), ptr(y), ptr(stats[0]), ptr(stats[1]), n, h * w, c, eps, stream())
    return y, stats[1]


def instnorm_bwd(y, dy, invstd):
    n, h, w, c = y.shape
    dx = torch.empty_like(y)
    call("uem_instnorm_bwd", ptr(y), ptr(dy), ptr(invstd), ptr(dx), n, h * w, c, stream())
    return dx


def add_(a, b):
    call("uem_add_inplace", ptr(a), ptr(b), a.numel(), stream())
    return a


def nhwc_to_nchw(x):
    n, h, w, c = x.shape
    y = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
    call("uem_nhwc_to_nchw", ptr(x), ptr(y), n, h * w, c, stream())
    return y


def nchw_to_nhwc(x):
    n, c, h, w = x.shape
    y = torch.empty((n, h, w, c), device=x.device, dtype=torch.float32)
    call("uem_nchw_to_nhwc", ptr(x), ptr(y), n, h * w, c, stream())
    return y


def as_nhwc(t):
    """(N,C,H,W) logical tensor -> dense (N,H,W,C) fp32 tensor (zero-copy when already channels_last)."""
    need_gpu(t)
    if t.dtype != torch.float32:
        raise UemError("expected float32")
    v = t.permute(0, 2, 3, 1)
    if v.is_contiguous():
        return v
    return nchw_to_nhwc(t.contiguous())


def as_nchw_view(x_nhwc):
    """dense (N,H,W,C) -> logical (N,C,H,W) view (channels_last strides, zero-copy)."""
    return x_nhwc.permute(0, 3, 1, 2)
