"""Host logic of the two-stream step (uemda_amd.step.forward_pair, round 6) that needs no GPU: when the pair may fork, and the
swap of every BatchNorm's running-statistics buffers into the shadow arena and back (also when the forward inside raises)."""
import pytest
import torch

C = 6


def _cfg(**backbone):
    return dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False, **backbone), multi_layer=True, cascade=False,
                use_ppm=False, ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)


def _model(**backbone):
    from uemda_amd.models.Encoder import Deeplabv2
    m = Deeplabv2(_cfg(**backbone))
    m._flatten_parameters()                       # what .cuda() / .to() trigger: the flat arenas (here on the CPU)
    return m


def test_two_stream_eligibility():
    m = _model().train()
    assert m.two_stream_ok()
    assert not m.eval().two_stream_ok()                                             # inference: nothing to fork
    m.train()
    m.encoder.resnet.layer3[1].bn2.momentum = 0.01                                  # one common momentum is what the shadow update assumes
    assert not m.two_stream_ok()
    m.encoder.resnet.layer3[1].bn2.momentum = 0.1
    m.encoder.resnet.layer2[0].bn1.eval()                                           # an eval-mode layer inside the training graph
    assert not m.two_stream_ok()
    assert not _model(batchnorm_trainable=False).train().two_stream_ok()            # frozen statistics (resnet.py:112-130)
    assert not _model(with_cp=(False, True, False, False)).train().two_stream_ok()  # a checkpointed layer re-runs its forward in backward


def test_may_fork_needs_two_training_forwards_and_the_switch(monkeypatch):
    from uemda_amd import ops, step
    m = _model().train()
    assert not step._may_fork(m)                      # (a model on the CPU never forks)
    assert not step._fork_wanted(m)                      # the derived filter banks do not exist before two training forwards
    m._uem_train_forwards = 2
    assert step._fork_wanted(m)
    monkeypatch.setattr(ops, "TWO_STREAM_FWD", False)
    assert not step._fork_wanted(m)
    monkeypatch.setattr(ops, "TWO_STREAM_FWD", True)
    with torch.no_grad():
        assert not step._fork_wanted(m)                  # no graph, no backward chain to overlap
    ops.PROF.enabled = True
    try:
        assert not step._fork_wanted(m)                  # per-launch event timing measures one kernel at a time
    finally:
        ops.PROF.enabled = False


def test_shadow_running_statistics_swap_and_restore():
    m = _model().train()
    bn = m.encoder.resnet.layer1[0].bn1
    real_mean, real_var = bn.running_mean, bn.running_var
    assert real_mean.data_ptr() >= m._rs.data_ptr() and real_mean.data_ptr() < m._rs.data_ptr() + m._rs.numel() * 4
    n0 = int(bn.num_batches_tracked)
    with m.shadow_running_stats():
        assert bn.running_mean.data_ptr() != real_mean.data_ptr()
        assert float(bn.running_mean.abs().max()) == 0.0 and float(bn.running_var.abs().max()) == 0.0   # zeros: the update leaves m * v there
        assert bn.running_mean.data_ptr() >= m._rs_shadow.data_ptr()
        bn.running_mean.add_(1.0)                                                   # what a forward's EMA kernel would do
        m._nbt_step()                                                               # the second forward does not count here
    assert bn.running_mean.data_ptr() == real_mean.data_ptr() and bn.running_var.data_ptr() == real_var.data_ptr()
    assert int(bn.num_batches_tracked) == n0
    assert float(m._rs_shadow.abs().max()) == 1.0                                   # the contribution waits in the shadow for the join
    m._rs_shadow.zero_()
    with pytest.raises(RuntimeError):
        with m.shadow_running_stats():
            bn.running_mean.add_(3.0)
            raise RuntimeError("a forward that dies half way")
    assert bn.running_mean.data_ptr() == real_mean.data_ptr()
    assert float(m._rs_shadow.abs().max()) == 0.0                                   # its partial contribution never reaches the statistics
    # state_dict round trip: the views are ordinary buffers
    sd = m.state_dict()
    assert sd["encoder.resnet.layer1.0.bn1.running_mean"].data_ptr() == real_mean.data_ptr()
