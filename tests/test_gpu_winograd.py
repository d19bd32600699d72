"""GPU parity of the Winograd F(2x2, 3x3) path (csrc/winograd.hip; reference call sites uemda/_resnets.py:100-103, Encoder.py:35)
against torch-CPU float64 convolutions and against the direct f32-MFMA kernels it replaces, through the C ABI."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).cpu()


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


WINO_CASES = [
    (2, 32, 32, 128, 128, 1),        # layer2 at 512^2-tile scale: forward and weight gradient take this path, the data gradient does not
    # N, H, W, Cin, Cout, dil
    (2, 16, 16, 256, 256, 1),        # layer3 at 256^2 tiles: T = 128
    (2, 16, 16, 512, 512, 2),        # layer4 dilated at 256^2
    (2, 32, 32, 512, 256, 1),        # 512^2 tiles, Cin != Cout
    (1, 32, 32, 256, 512, 2),
    (2, 16, 16, 1024, 512, 1),       # wide input (the PPM head's conv_last is 4096 -> 512)
    (4, 16, 8, 256, 320, 1),         # non-square map, Cout a multiple of 64 only
]


@pytest.mark.parametrize("case", WINO_CASES)
def test_winograd_forward_dgrad_wgrad_vs_float64(case):
    from uemda_amd import ops
    N, H, W, Cin, Cout, d = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    xa = F.relu(x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).requires_grad_(True)
    wr = w.double().requires_grad_(True)
    y_ref = F.conv2d(xa, wr, padding=d, dilation=d)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy.double())
    assert ops.wino_ok((N, H, W, Cin), Cout, 3, 3, 1, d, d)
    wp = w.cuda().contiguous(memory_format=torch.channels_last)
    # forward with the BatchNorm-affine + ReLU prologue (zero padding after it)
    y, v = ops.conv3x3_wino(nhwc(x), wp, d, in_scale=sc.cuda(), in_shift=sh.cuda(), in_relu=True, want_v=True)
    assert rel(nchw(y), y_ref.detach()) < 1.5e-6
    # the direct kernel on the same operands: both are fp32 sums of the same products
    y_dir = ops.conv2d(nhwc(x), ops.weight_ohwi(wp), None, pad=d, dil=d, in_scale=sc.cuda(), in_shift=sh.cuda(), in_relu=True)
    assert rel(y, y_dir) < 2e-6
    # data gradient (with respect to the post-activation input)
    dx, tp = ops.conv3x3_wino_dgrad(nhwc(gy), wp, d)
    assert tp is None and rel(nchw(dx), xa.grad) < 1.5e-6
    # weight gradient, accumulated into a non-zero buffer
    dw0 = torch.randn(Cout, 3, 3, Cin, generator=g).cuda()
    dw = dw0.clone()
    ops.conv3x3_wino_wgrad(v, nhwc(gy), dw, d)
    assert rel((dw - dw0).permute(0, 3, 1, 2), wr.grad) < 3e-6


@pytest.mark.parametrize("case", WINO_CASES[:4])
def test_winograd_fused_batchnorm_passes_match_the_direct_kernels(case):
    """forward: BatchNorm statistics out of the output transform; backward: the BatchNorm+ReLU reduction pass of the producer's
    bn inside the data gradient's output transform -- against the direct kernels' fused epilogues on the same tensors."""
    from uemda_amd import ops
    N, H, W, Cin, Cout, d = case
    g = torch.Generator().manual_seed(sum(case) + 7)
    z = nhwc(torch.randn(N, Cin, H, W, generator=g))
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).cuda().contiguous(memory_format=torch.channels_last)

    def bn(c):
        m = torch.nn.BatchNorm2d(c).cuda()
        with torch.no_grad():
            m.weight.copy_(torch.rand(c, generator=g) + 0.5)
            m.bias.copy_(torch.randn(c, generator=g) * 0.2)
        return m
    bn_in = bn(Cin)
    st_in = ops.bn_stats(z, bn_in.weight.detach(), bn_in.bias.detach(), None, None, True)
    bn_a, bn_b = bn(Cout), bn(Cout)
    bn_b.load_state_dict(bn_a.state_dict())
    y1, st1, v = ops.conv3x3_wino_bn(z, w, bn_a, d, in_scale=st_in.scale, in_shift=st_in.shift, in_relu=True)
    y2, st2 = ops.conv2d_bn(z, ops.weight_ohwi(w), bn_b, pad=d, dil=d, in_scale=st_in.scale, in_shift=st_in.shift, in_relu=True)
    assert rel(y1, y2) < 2e-6
    for a, b in ((st1.mean, st2.mean), (st1.invstd, st2.invstd), (st1.scale, st2.scale), (st1.shift, st2.shift),
                 (bn_a.running_mean, bn_b.running_mean), (bn_a.running_var, bn_b.running_var)):
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-6)
    # backward through the conv and the producer's BatchNorm + ReLU
    dy = nhwc(torch.randn(N, Cout, H, W, generator=g))
    gg1, gb1 = torch.zeros(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
    gg2, gb2 = torch.zeros(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
    dz1 = ops.conv3x3_wino_dgrad_bn_backward(dy, w, z, st_in, gg1, gb1, d)
    dz2 = ops.conv2d_dgrad_bn_backward(dy, ops.weight_transpose(ops.weight_ohwi(w)), z, st_in, gg2, gb2, pad=d, dil=d)
    assert rel(dz1, dz2) < 2e-5
    torch.testing.assert_close(gg1, gg2, rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(gb1, gb2, rtol=2e-4, atol=2e-4)
    dwa, dwb = torch.zeros(Cout, 3, 3, Cin, device="cuda"), torch.zeros(Cout, 3, 3, Cin, device="cuda")
    ops.conv3x3_wino_wgrad(v, dy, dwa, d)
    ops.conv2d_wgrad(z, dy, dwb, pad=d, dil=d, in_scale=st_in.scale, in_shift=st_in.shift, in_relu=True)
    assert rel(dwa, dwb) < 5e-6


def test_winograd_declines_shapes_it_does_not_take():
    from uemda_amd import ops
    assert not ops.wino_ok((2, 16, 16, 64), 64, 3, 3, 1, 1, 1)          # narrow layers stay on the direct kernels
    assert ops.wino_ok((2, 16, 16, 128), 128, 3, 3, 1, 1, 1) and not ops.wino_dgrad_ok(128, 128)   # layer2: forward + weight gradient only
    assert not ops.wino_ok((2, 16, 16, 256), 256, 3, 3, 2, 1, 1)        # stride 2
    assert not ops.wino_ok((2, 16, 16, 256), 256, 1, 1, 1, 0, 1)        # 1x1
    assert not ops.wino_ok((2, 18, 18, 256), 256, 3, 3, 1, 2, 2)        # 18 is not a multiple of 2 * dilation
    assert not ops.wino_ok((1, 8, 8, 256), 256, 3, 3, 1, 1, 1)          # T = 16 tiles: not a multiple of 128
    with pytest.raises(ops.UemError):
        ops.wino_input(torch.zeros(1, 6, 6, 64, device="cuda"), 1)           # the C ABI refuses too (T % 32)
