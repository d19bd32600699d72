"""The step's two graphs on two streams (uemda_amd.step.forward_pair, round 6): the second train-mode forward runs on its own stream and
autograd runs its backward chain there.  Whatever the two graphs share -- BatchNorm running statistics, num_batches_tracked, derived
filter banks, the gradient arena (the second graph accumulates into a shadow arena, folded at the end of backward) -- must come out as
in the sequential pair: outputs, running statistics and counters bit for bit, the step's update to the fp32 atomics' order."""
import pytest
import torch

from uemda_amd import ops

pytestmark = pytest.mark.gpu
C = 6


def _model(storage, use_ppm=False, seed=2333):
    from oracle.weights import det_state_dict
    from uemda_amd.models.Encoder import Deeplabv2
    cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False), multi_layer=True, cascade=False, use_ppm=use_ppm,
               ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)
    m = Deeplabv2(cfg)
    m.load_state_dict(det_state_dict("resnet50", C, use_ppm, seed=seed))
    m = m.cuda().set_storage(storage)
    if use_ppm:
        m.layer5.conv_last[3].p = 0.0
        m.layer6.conv_last[3].p = 0.0
    return m


def _bn_state(model):
    return {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}


@pytest.mark.parametrize("storage", ["fp32", "bf16"])
def test_forward_pair_on_two_streams_equals_the_sequential_pair(storage, monkeypatch):
    from oracle import synth
    from uemda_amd.step import forward_pair
    batch = {k: v.cuda() for k, v in synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=5).items()}
    outs = {}
    for two in (False, True):
        monkeypatch.setattr(ops, "TWO_STREAM_FWD", two)
        model = _model(storage).train()
        for _ in range(2):                                      # the second pair is the one that may fork (filter banks exist by then)
            a, b = forward_pair(model, batch["images_s"], batch["images_t"])
        torch.cuda.synchronize()
        outs[two] = ([t.detach().clone() for t in a + b], _bn_state(model))
        assert ops._FWD2 if two else True
    seq, par = outs[False], outs[True]
    for x, y in zip(seq[0], par[0]):
        assert torch.equal(x, y)
    assert seq[1].keys() == par[1].keys()
    for k in seq[1]:
        assert torch.equal(seq[1][k], par[1][k]), k             # running statistics and counters: bit for bit, in the reference's order
    # ... and the statistics did move twice per pair (source, then target): not the value of a single update
    fresh = _bn_state(_model(storage))
    k0 = "encoder.resnet.bn1.running_mean"
    assert not torch.equal(fresh[k0], par[1][k0])
    assert int(par[1]["encoder.resnet.bn1.num_batches_tracked"]) == 4


@pytest.mark.parametrize("storage,use_ppm,pipelines", [("fp32", False, True), ("bf16", False, True), ("fp32", True, True), ("fp32", False, False)])
def test_ssl_steps_with_two_forward_streams_match_the_sequential_steps(storage, use_ppm, pipelines, monkeypatch):
    """three train_ssl_uem steps (the second and third fork).  Step 2 -- same weights to the order of step 1's fp32 atomics (split-K
    weight gradients, BatchNorm partial sums) -- must agree with the sequential step to that order.  Step 3 has seen that noise through
    two clipped updates of a randomly initialised network (gradient norm ~550 against the clip's 32): two SEQUENTIAL runs differ there
    by 1e-3 in the source loss, 5e-3 in the target loss and 1.7e-4 in the weights, the forked run from either by the same amounts
    (scripts/dbg/two_stream_drift.py, profiles/r06_l_two_streams.txt) -- so weights, statistics and prototypes are compared after step 2
    and step 3 only bounds the losses' drift; the sharp statement about the gradients is
    test_gradients_of_a_forked_pair_equal_the_sequential_pair.  The shadow gradient arena is folded and clean after every step."""
    from oracle import synth
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import HYPER, StepState, ssl_step
    batch = {k: v.cuda() for k, v in synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=9).items()}
    res = {}
    monkeypatch.setattr(ops, "TWO_PIPELINES", pipelines)       # the target graph's mining and loss on its stream too / on the caller's
    for two in (False, True):
        monkeypatch.setattr(ops, "TWO_STREAM_FWD", two)
        model = _model(storage, use_ppm)
        al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
        al.prototypes = batch["prototypes"].clone()
        opt, state = FusedSGD(model, lr=1e-2, momentum=0.9, weight_decay=5e-4), StepState(C)
        outs, snap = [], None
        for i in range(3):
            outs.append(ssl_step(model, al, opt, state, batch, 2e-3))
            if two:
                assert not model._g2_dirty                                                 # every step folded ...
                if i >= 1:
                    assert model._grad_arena2 is not None and float(model._grad_arena2.abs().max()) == 0.0   # ... and cleared the shadow
            if i == 1:
                torch.cuda.synchronize()
                snap = (model.flat_parameters()[0].clone(), _bn_state(model), al.prototypes.clone())
        torch.cuda.synchronize()
        res[two] = (outs,) + snap
        assert two or model._grad_arena2 is None
    (o0, w0, b0, p0), (o1, w1, b1, p1) = res[False], res[True]
    # step 1 is sequential in both runs; the forward of step 2 sees weights that differ by atomics' order only
    assert torch.equal(o0[0]["pred_t1"], o1[0]["pred_t1"])
    for i in (1, 2):
        # step 2, fp32: two sequential runs differ by up to 8e-6 (ASPP heads) / 1.1e-4 (PPM heads, gradient norm ~5000) of the loss; the
        # target loss also moves in quanta -- one marginal pseudo label selected or not: 2.4e-4 on this batch, seen in forked and in
        # sequential runs alike (profiles/r06_l_two_streams.txt)
        for k in ("loss_source", "loss_target"):
            tol = (5e-3 if storage == "bf16" else 5e-4 if (use_ppm or k == "loss_target") else 5e-5) if i == 1 else 1e-1
            assert abs(float(o0[i][k]) - float(o1[i][k])) <= tol * max(1.0, abs(float(o0[i][k]))), (i, k)
        agree = (o0[i]["label_t_hard"] == o1[i]["label_t_hard"]).float().mean().item()
        assert agree >= ((0.99 if storage == "bf16" else 0.999) if i == 1 else 0.9), (i, agree)
    # the state after step 2 (one forked step behind one sequential step)
    rel = float((w0 - w1).norm() / w0.norm())
    assert rel < (5e-4 if storage == "bf16" else 5e-5), rel
    for k in b0:
        if "num_batches" in k:
            assert torch.equal(b0[k], b1[k]), k
        else:
            torch.testing.assert_close(b0[k], b1[k], rtol=5e-3 if storage == "bf16" else 1e-3, atol=1e-4)
    torch.testing.assert_close(p0, p1, rtol=5e-3 if storage == "bf16" else 1e-3, atol=1e-4)


def test_second_forward_is_declined_where_the_statistics_could_not_be_kept_in_order():
    """frozen BatchNorm statistics (eval-mode layers inside the training graph), a checkpointed layer, differing momenta: the pair runs
    sequentially"""
    from uemda_amd.models.Encoder import Deeplabv2
    cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False, batchnorm_trainable=False), multi_layer=True,
               cascade=False, use_ppm=False, ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)
    assert not Deeplabv2(cfg).cuda().train().two_stream_ok()
    cfg["backbone"] = dict(resnet_type="resnet50", output_stride=16, pretrained=False, with_cp=(True, False, False, False))
    assert not Deeplabv2(cfg).cuda().train().two_stream_ok()
    cfg["backbone"] = dict(resnet_type="resnet50", output_stride=16, pretrained=False)
    m = Deeplabv2(cfg).cuda().train()
    assert m.two_stream_ok()
    m.encoder.resnet.layer2[0].bn1.momentum = 0.05
    assert not m.two_stream_ok()
    m.encoder.resnet.layer2[0].bn1.momentum = 0.1
    assert not m.eval().two_stream_ok()


@pytest.mark.parametrize("storage", ["fp32", "bf16"])
def test_gradients_of_a_forked_pair_equal_the_sequential_pair(storage, monkeypatch):
    """one backward pass through both graphs of a forked pair (same weights, same inputs as the sequential pair): the gradient arena
    after the fold equals the sequential arena to the order of the fp32 atomics, per parameter tensor; with UEM_TWO_STREAM_BWD=0
    (every backward node moved to the step's stream) likewise, and the shadow is never created"""
    from oracle import synth
    from uemda_amd.step import forward_pair
    batch = {k: v.cuda() for k, v in synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=11).items()}
    grads = {}
    for mode in ("seq", "two", "fwd_only"):
        monkeypatch.setattr(ops, "TWO_STREAM_FWD", mode != "seq")
        monkeypatch.setattr(ops, "TWO_STREAM_BWD", mode == "two")
        model = _model(storage).train()
        for it in range(2):
            model.zero_grad()
            (a1, a2, _), (b1, b2, _) = forward_pair(model, batch["images_s"], batch["images_t"])
            loss = (a1.float() ** 2).mean() + (a2.float() ** 2).mean() + 0.5 * (b1.float() ** 2).mean() + 0.25 * (b2.float() ** 2).mean()
            loss.backward()
            ops.grad_join()
        torch.cuda.synchronize()
        grads[mode] = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        assert (model._grad_arena2 is not None) == (mode == "two")
    for mode in ("two", "fwd_only"):
        assert grads[mode].keys() == grads["seq"].keys()
        for n, g in grads["seq"].items():
            d = float((grads[mode][n] - g).norm()) / max(float(g.norm()), 1e-20)
            assert d < (2e-2 if storage == "bf16" else 1e-4), (mode, n, d)


@pytest.mark.parametrize("which", ["src_align_domain", "align"])
def test_the_other_two_graph_steps_fork_and_match(which, monkeypatch):
    """train_src.py --align-domain (source + target forward, CORAL between the features) and train_align_uem.py's step (source forward,
    prototype update, target forward, PCL) take their two forwards through forward_pair too: two steps, the second forked, against the
    sequential steps -- losses and weights to the order of step 1's fp32 atomics, BatchNorm counters equal"""
    from oracle import synth
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import HYPER, StepState, align_step, src_step
    batch = {k: v.cuda() for k, v in synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=13).items()}
    res = {}
    for two in (False, True):
        monkeypatch.setattr(ops, "TWO_STREAM_FWD", two)
        model = _model("fp32")
        with torch.no_grad():                     # confident heads, so that some on-the-fly pseudo labels survive (random heads: every
            for head in (model.layer5, model.layer6):        # target label ignored, PCL = 0/0 = NaN, in the reference too)
                for conv in head.conv2d_list:
                    conv.bias[0] += 1.5
        al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
        al.prototypes = batch["prototypes"].clone()
        opt, state = FusedSGD(model, lr=1e-2, momentum=0.9, weight_decay=5e-4), StepState(C)
        outs = []
        for _ in range(2):
            if which == "align":
                outs.append(align_step(model, al, opt, state, batch, 2e-3))
            else:
                outs.append(src_step(model, opt, state, batch, 2e-3, aligner=al, align_domain=True))
        torch.cuda.synchronize()
        assert (model._grad_arena2 is not None) == two
        res[two] = (outs, model.flat_parameters()[0].clone(), _bn_state(model))
    (o0, w0, b0), (o1, w1, b1) = res[False], res[True]
    for k in o0[1]:
        if k.startswith("loss"):
            assert torch.isfinite(o0[1][k]).all(), k
            assert abs(float(o0[1][k]) - float(o1[1][k])) <= 1e-4 * max(1.0, abs(float(o0[1][k]))), k
    rel = float((w0 - w1).norm() / w0.norm())
    assert rel < 5e-5, rel
    for k in b0:
        if "num_batches" in k:
            assert torch.equal(b0[k], b1[k]), k
        else:
            torch.testing.assert_close(b0[k], b1[k], rtol=1e-3, atol=1e-4)
