"""The weight gradients' side stream (uemda_amd.ops.on_side): the guard that fails an in-place main-stream write into a buffer a queued
side launch still reads, and the end-of-backward join being tied to the backward pass that queued it
(a backward that raised used to leave the flag set and every later pass skipped its join)."""
import pytest
import torch

from uemda_amd import ops
from uemda_amd.ops import UemError

pytestmark = pytest.mark.gpu


class _QueueOnSide(torch.autograd.Function):
    """backward: queues a (trivial) launch on the side stream that 'reads' ctx.buf, then runs ctx.then(buf)"""

    @staticmethod
    def forward(ctx, x, buf, then):
        ctx.buf, ctx.then = buf, then
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        ran = []
        ops.on_side(lambda: ran.append(1), [ctx.buf], "test launch")
        assert ran == [1]
        ctx.then(ctx.buf)
        return g, None, None


def _st(c):
    st = ops.BNState()
    buf = torch.ones(4, c, device="cuda")
    st.scale, st.shift, st.mean, st.invstd = buf[0], buf[1], buf[2], buf[3]
    st.training = True
    return st


def test_in_place_write_into_a_buffer_a_side_launch_reads_fails_loudly():
    buf = torch.randn(2, 8, 8, 64, device="cuda")
    other = torch.randn(2, 8, 8, 64, device="cuda")
    seen = {}

    def then(b):
        # the bare guard, an elementwise op writing in place, a data gradient accumulating into it, and a view of it
        with pytest.raises(UemError, match="side stream still reads"):
            ops.guard_write(b, "test")
        with pytest.raises(UemError, match="side stream still reads"):
            ops.affine_act(other, _st(64), out=b)
        with pytest.raises(UemError, match="side stream still reads"):
            ops.bn_backward(other, other, _st(64), None, None, None, True, dx=b[1:])
        w_t = torch.randn(64, 1, 1, 64, device="cuda")
        with pytest.raises(UemError, match="side stream still reads"):
            ops.conv2d_dgrad(other, w_t, other.shape, out=b, accumulate=True)
        ops.guard_write(other, "test")                         # a buffer nobody on the side stream reads: fine
        ops.affine_act(b, _st(64), out=other)                  # READING the guarded buffer on the main stream is fine too
        seen["pending"] = len(ops._Side.get().reads)

    x = torch.randn(4, device="cuda", requires_grad=True)
    _QueueOnSide.apply(x, buf, then).sum().backward()
    assert seen["pending"] == 1
    st = ops._Side.get()
    assert not st.dirty and not st.reads and st.task is None   # the end-of-backward callback joined and dropped the guard's ranges
    ops.guard_write(buf, "after the join")                     # ... after which the buffer may be overwritten
    ops.affine_act(other, _st(64), out=buf)
    torch.cuda.synchronize()


def test_a_backward_that_raised_does_not_disarm_the_join_of_the_next_one():
    buf = torch.randn(1024, device="cuda")

    def boom(b):
        raise RuntimeError("injected failure after the side launch was queued")

    x = torch.randn(4, device="cuda", requires_grad=True)
    with pytest.raises(RuntimeError, match="injected failure"):
        _QueueOnSide.apply(x, buf, boom).sum().backward()
    st = ops._Side.get()
    assert st.dirty and st.task is not None                    # autograd dropped the final callbacks: nobody joined
    # the next clean backward queues ITS OWN join (the flag is the graph task's id) ...
    x2 = torch.randn(4, device="cuda", requires_grad=True)
    _QueueOnSide.apply(x2, buf, lambda b: None).sum().backward()
    assert not st.dirty and st.task is None and not st.reads
    # ... and whoever reads the gradient arena next joins on its own account: a failed backward followed directly by the optimizer
    with pytest.raises(RuntimeError, match="injected failure"):
        _QueueOnSide.apply(x, buf, boom).sum().backward()
    assert st.dirty
    ops.side_join()                                            # what FusedSGD.step / clip_grad_norm_ / DataParallel.reduce_gradients call first
    assert not st.dirty and st.task is None and not st.reads
    torch.cuda.synchronize()


def test_step_with_side_stream_after_a_failed_backward_matches_a_clean_run():
    """the whole path: a model step whose backward raises half way (after weight gradients were queued on the side stream), then a clean
    step with the side stream on; the clean step equals the same step taken by a fresh model that never saw the failure"""
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.models.Encoder import Deeplabv2
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import HYPER, StepState, src_step
    from uemda_amd.utils import synth
    from oracle.weights import det_state_dict

    C, B, S = 6, 2, 128
    cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False), multi_layer=True, cascade=False, use_ppm=False,
               ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)
    sd = det_state_dict("resnet50", C, False, seed=11)
    batch = {k: v.cuda() for k, v in synth.make_batch(B=B, H=S, W=S, C=C, k=2048, seed=3).items()}

    def fresh():
        m = Deeplabv2(cfg)
        m.load_state_dict(sd)
        m = m.cuda()
        return m, FusedSGD(m, lr=1e-2, momentum=0.9, weight_decay=5e-4)

    def params_after_step(m, opt):
        src_step(m, opt, StepState(C), batch, 5e-3)
        torch.cuda.synchronize()
        return m.flat_parameters()[0].clone()

    m1, o1 = fresh()
    ref = params_after_step(m1, o1)

    m2, o2 = fresh()
    # a forward/backward that fails when the gradient reaches layer1's output: by then layer4 ... layer2 have queued their weight
    # gradients on the side stream and nobody has joined them
    def boom(g):
        raise RuntimeError("injected failure at layer1's output gradient")
    h = m2.encoder.resnet.layer1[-1].register_forward_hook(lambda mod, inp, out: out.register_hook(boom) and None)
    m2.train()
    p1, p2, _ = m2(batch["images_s"])
    h.remove()
    with pytest.raises(RuntimeError, match="injected failure"):
        (p1.float().sum() + p2.float().sum()).backward()
    assert ops._Side.get().dirty                               # the failed pass left side-stream work behind, unjoined
    o2.zero_grad()
    got = params_after_step(m2, o2)
    # BatchNorm running statistics moved in the failed forward, the parameters did not: the clean step's update equals the reference's
    # up to the fp32 atomics' summation order
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err < 5e-6, err
