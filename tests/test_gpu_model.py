"""GPU parity of the convolution / normalisation kernels, the residual blocks and the full network
against torch-CPU fp32 references, the oracle and the reference goldens (through the C ABI)."""
import os

import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu
C = 6


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).cpu()


def ohwi(w):
    return w.permute(0, 2, 3, 1).contiguous().cuda()


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil
    (2, 16, 16, 64, 64, 1, 1, 0, 1),
    (2, 16, 16, 64, 256, 1, 1, 0, 1),
    (2, 17, 13, 128, 128, 3, 1, 1, 1),       # ragged M
    (2, 16, 16, 128, 128, 3, 2, 1, 1),
    (2, 16, 16, 256, 512, 1, 2, 0, 1),
    (1, 12, 12, 512, 512, 3, 1, 2, 2),
    (2, 16, 16, 64, 32, 3, 1, 6, 6),         # ASPP-like: big dilation, N-tile 32
    (2, 8, 8, 32, 64, 3, 1, 12, 12),
    (3, 2, 2, 2048, 512, 1, 1, 0, 1),        # tiny M (PPM branch)
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward_dgrad_wgrad(case):
    from uemda_amd import ops
    N, H, W, Cin, Cout, k, s, p, d = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, Cin, H, W, generator=g).requires_grad_(True)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).requires_grad_(True)
    bias = torch.randn(Cout, generator=g)
    y_ref = F.conv2d(x, w, bias, stride=s, padding=p, dilation=d)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    y = ops.conv2d(nhwc(x.detach()), ohwi(w.detach()), bias.cuda(), stride=s, pad=p, dil=d)
    torch.testing.assert_close(nchw(y), y_ref.detach(), rtol=1e-4, atol=1e-4)
    dx = ops.conv2d_dgrad(nhwc(gy), ops.weight_transpose(ohwi(w.detach())), (N, H, W, Cin), stride=s, pad=p, dil=d)
    torch.testing.assert_close(nchw(dx), x.grad, rtol=1e-4, atol=1e-4)
    # accumulate epilogue of the data gradient (residual sum; strided convs skip the parity classes no tap reaches)
    dx2 = ops.conv2d_dgrad(nhwc(gy), ops.weight_transpose(ohwi(w.detach())), (N, H, W, Cin), stride=s, pad=p, dil=d,
                           out=dx.clone(), accumulate=True)
    torch.testing.assert_close(nchw(dx2), 2 * x.grad, rtol=1e-4, atol=2e-4)
    dw = torch.zeros(Cout, k, k, Cin, device="cuda")
    ops.conv2d_wgrad(nhwc(x.detach()), nhwc(gy), dw, stride=s, pad=p, dil=d)
    torch.testing.assert_close(dw.permute(0, 3, 1, 2).cpu(), w.grad, rtol=1e-4, atol=2e-4 * gy.numel() ** 0.5 / 16)
    # accumulate epilogue
    y2 = ops.conv2d(nhwc(x.detach()), ohwi(w.detach()), None, stride=s, pad=p, dil=d, out=y.clone(), accumulate=True)
    torch.testing.assert_close(nchw(y2), 2 * y_ref.detach() - bias.view(1, -1, 1, 1), rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("prec,tol", [("bf16", 6e-3)])
@pytest.mark.parametrize("case", [CONV_CASES[1], CONV_CASES[2], CONV_CASES[3], CONV_CASES[5], CONV_CASES[6], CONV_CASES[8]])
def test_conv_bf16_operand_mode_of_the_bf16_storage_islands(case, prec, tol):
    """bf16 operands over fp32 tensors (how a bf16-storage model runs its fp32 stem and heads): relative-L2 error of forward / data
    gradient / weight gradient against fp64.  The split-operand modes of rounds 1-3 are retired (the C ABI rejects their flag)."""
    from uemda_amd import ops
    N, H, W, Cin, Cout, k, s, p, d = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = torch.randn(N, Cin, H, W, generator=g).double().requires_grad_(True)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).double().requires_grad_(True)
    y_ref = F.conv2d(x, w, None, stride=s, padding=p, dilation=d)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy.double())
    xf, wf = x.detach().float(), w.detach().float()

    def rel(a, b):
        return float((a.double().cpu() - b).norm() / b.norm())
    ops.set_conv_precision(prec)
    try:
        y = ops.conv2d(nhwc(xf), ohwi(wf), None, stride=s, pad=p, dil=d)
        dx = ops.conv2d_dgrad(nhwc(gy), ops.weight_transpose(ohwi(wf)), (N, H, W, Cin), stride=s, pad=p, dil=d)
        dw = torch.zeros(Cout, k, k, Cin, device="cuda")
        ops.conv2d_wgrad(nhwc(xf), nhwc(gy), dw, stride=s, pad=p, dil=d)
    finally:
        ops.set_conv_precision("fp32")
    assert rel(nchw(y), y_ref.detach()) < tol
    assert rel(nchw(dx), x.grad) < tol
    assert rel(dw.permute(0, 3, 1, 2), w.grad) < tol


def test_conv_operand_prologue_affine_relu():
    from uemda_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 12, 12, generator=g)
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    w = torch.randn(128, 64, 3, 3, generator=g) / 24
    xa = F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y_ref = F.conv2d(xa, wr, padding=1)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    y = ops.conv2d(nhwc(x), ohwi(w), None, pad=1, in_scale=sc.cuda(), in_shift=sh.cuda(), in_relu=True)
    torch.testing.assert_close(nchw(y), y_ref.detach(), rtol=1e-4, atol=1e-4)      # zero padding stays zero
    dw = torch.zeros(128, 3, 3, 64, device="cuda")
    ops.conv2d_wgrad(nhwc(x), nhwc(gy), dw, pad=1, in_scale=sc.cuda(), in_shift=sh.cuda(), in_relu=True)
    torch.testing.assert_close(dw.permute(0, 3, 1, 2).cpu(), wr.grad, rtol=1e-4, atol=1e-3)


WGRAD_DMA_CASES = [
    # N, H, W, Cin, Cout, k, stride, dil, affine      (the shapes the LDS-DMA weight-gradient kernel takes: wgrad.hip)
    (2, 16, 32, 128, 128, 3, 1, 1, True),       # 3 taps per block, 128x128 tiles, one 32-pixel step per row
    (2, 16, 64, 64, 64, 3, 1, 1, True),         # 64x64 tiles, two steps per row
    (1, 20, 32, 128, 64, 3, 1, 2, True),        # dilation 2 (layer4), 64 x 128 tile
    (2, 24, 64, 64, 128, 3, 2, 1, True),        # stride 2 (first block of layer2/3): 65 staged pixels per step
    (2, 16, 32, 128, 128, 3, 1, 1, False),      # no prologue (PPM 3x3): padding by the buffer bounds check alone
    (3, 7, 9, 256, 128, 1, 1, 1, True),         # 1x1, M = 189: ragged last step
    (2, 16, 16, 64, 256, 1, 1, 1, False),
    (2, 16, 64, 128, 256, 1, 2, 1, False),      # downsample 1x1 stride 2
    (2, 16, 64, 128, 64, 1, 2, 1, True),
    (8, 32, 32, 128, 128, 3, 1, 1, True),       # enough rows for several pixel slices per tile (split-K) and both stages
    (2, 16, 16, 512, 512, 3, 1, 1, True),       # 16-pixel rows (256^2 tiles at stride 16): the 16-pixel-step variant; layer4.0.conv2
    (2, 16, 16, 256, 256, 3, 1, 1, True),       # layer3 at 256^2
    (2, 16, 16, 512, 512, 3, 1, 2, True),       # layer4 dilated at 256^2
    (2, 32, 32, 128, 128, 3, 2, 1, True),       # stride 2 onto 16-pixel rows
    (2, 16, 16, 1024, 512, 1, 1, 1, False),     # layer4.0.conv1 at 256^2
    (2, 16, 16, 512, 2048, 1, 1, 1, True),
]


@pytest.mark.parametrize("case", WGRAD_DMA_CASES)
def test_wgrad_dma_kernel_vs_torch(case):
    from uemda_amd import ops
    N, H, W, Cin, Cout, k, s, d, affine = case
    pad = d * (k - 1) // 2
    g = torch.Generator().manual_seed(sum(case[:8]) + 11)
    x = torch.randn(N, Cin, H, W, generator=g)
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g)
    sc[::5] *= -1                                              # negative gammas too
    xa = (F.relu(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) if affine else x).double()
    w = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(xa, w, stride=s, padding=pad, dilation=d)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy.double())
    dw = torch.zeros(Cout, k, k, Cin, device="cuda")
    kw = dict(in_scale=sc.cuda(), in_shift=sh.cuda(), in_relu=True) if affine else {}
    ops.conv2d_wgrad(nhwc(x), nhwc(gy), dw, stride=s, pad=pad, dil=d, **kw)
    ref = w.grad.permute(0, 2, 3, 1)
    err = float((dw.double().cpu() - ref).norm() / ref.norm())
    assert err < 2e-6, err
    # the gradient ACCUMULATES (two forwards add into one arena)
    ops.conv2d_wgrad(nhwc(x), nhwc(gy), dw, stride=s, pad=pad, dil=d, **kw)
    assert float((dw.double().cpu() - 2 * ref).norm() / ref.norm()) < 4e-6


def test_stem_conv_and_wgrad():
    from uemda_amd import ops
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 3, 38, 26, generator=g)
    w = (torch.randn(64, 3, 7, 7, generator=g) / 12).requires_grad_(True)
    y_ref = F.conv2d(x, w, stride=2, padding=3)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    x4 = ops.nchw3_to_nhwc4(x.cuda())
    y = ops.stem_conv(x4, ohwi(w.detach()))
    torch.testing.assert_close(nchw(y), y_ref.detach(), rtol=1e-4, atol=1e-4)
    dw = torch.zeros(64, 7, 7, 3, device="cuda")
    ops.stem_wgrad(x4, nhwc(gy), dw)
    torch.testing.assert_close(dw.permute(0, 3, 1, 2).cpu(), w.grad, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("M,Cn", [(32768, 256), (8192, 1024), (1000, 64)])
def test_batchnorm_backward_pair_equals_the_two_passes(M, Cn):
    """uem_bn_bwd_apply_pair[_bf16] (a bottleneck block with a downsample branch: bn3 and the downsample BatchNorm read the same gated
    dy): the two outputs and the four parameter-gradient vectors are the ones two bn_backward calls give, bit for bit -- with bn3's
    reduction done here and with its per-tile partial sums handed in, with dzd written over dy, and on a map too small for the rows
    kernels (the entry declines, the host runs the two apply passes)."""
    from uemda_amd import ops, ops_bf16 as ob
    g = torch.Generator().manual_seed(M + Cn)
    z3, zd = (torch.randn(M, Cn, generator=g) * 1.3 + 0.2).cuda(), (torch.randn(M, Cn, generator=g) * 0.7 - 0.1).cuda()
    dy = torch.randn(M, Cn, generator=g).cuda()
    gam = [(torch.rand(Cn, generator=g) + 0.5).cuda() for _ in range(2)]
    bet = [(torch.randn(Cn, generator=g) * 0.3).cuda() for _ in range(2)]
    rm, rv = torch.zeros(Cn).cuda(), torch.ones(Cn).cuda()
    st3 = ops.bn_stats(z3, gam[0], bet[0], rm.clone(), rv.clone(), True)
    std = ops.bn_stats(zd, gam[1], bet[1], rm.clone(), rv.clone(), True)
    _, bits = ops.affine_act(z3, st3, res=zd, res_st=std, relu=True, want_bits=True)
    gr = [torch.zeros(Cn).cuda() for _ in range(4)]
    d3_ref = ops.bn_backward(z3, dy, st3, gr[0], gr[1], relu=True, ymask_bits=bits)
    dd_ref = ops.bn_backward(zd, dy, std, gr[2], gr[3], relu=True, ymask_bits=bits)
    gp = [torch.zeros(Cn).cuda() for _ in range(4)]
    dyc = dy.clone()
    d3, dd = ops.bn_backward_pair(z3, zd, dyc, bits, st3, std, None, gp[0], gp[1], gp[2], gp[3], dx2=dyc)
    assert dd.data_ptr() == dyc.data_ptr()
    assert torch.equal(d3, d3_ref) and torch.equal(dd, dd_ref)
    for a, b in zip(gp, gr):
        assert torch.equal(a, b)
    # bf16 storage
    z3b, zdb, dyb = z3.to(torch.bfloat16), zd.to(torch.bfloat16), dy.to(torch.bfloat16)
    st3b = ops.bn_stats(z3b.float(), gam[0], bet[0], rm.clone(), rv.clone(), True)
    stdb = ops.bn_stats(zdb.float(), gam[1], bet[1], rm.clone(), rv.clone(), True)
    grb = [torch.zeros(Cn).cuda() for _ in range(4)]
    d3b_ref = ob.bn_backward(z3b, dyb, st3b, grb[0], grb[1], relu=2, bits=bits)
    ddb_ref = ob.bn_backward(zdb, dyb, stdb, grb[2], grb[3], relu=2, bits=bits)
    gpb = [torch.zeros(Cn).cuda() for _ in range(4)]
    d3b, ddb = ob.bn_backward_pair(z3b, zdb, dyb, bits, st3b, stdb, None, gpb[0], gpb[1], gpb[2], gpb[3])
    assert torch.equal(d3b, d3b_ref) and torch.equal(ddb, ddb_ref)
    for a, b in zip(gpb, grb):
        assert torch.equal(a, b)


@pytest.mark.parametrize("size", [(2, 16, 64), (3, 64, 128), (1, 144, 192)])
def test_stem_kernels_of_round5(size):
    """csrc/stem.hip (whole 8 x 32 output tiles): forward against float64 conv2d and against the generic kernel it replaces, its
    BatchNorm tile statistics against the two-pass ones, the weight gradient against autograd in float64 -- accumulating (+=), and
    bit-identical from run to run (per-block banks summed in a fixed order: no atomics)."""
    import torch.nn as nn
    from uemda_amd import ops
    n, h, w_ = size
    g = torch.Generator().manual_seed(n * h + w_)
    x = torch.randn(n, 3, h, w_, generator=g)
    w = (torch.randn(64, 3, 7, 7, generator=g) / 12)
    wd = w.double().requires_grad_(True)
    y_ref = F.conv2d(x.double(), wd, stride=2, padding=3)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy.double())
    x4 = ops.nchw3_to_nhwc4(x.cuda())
    assert ops.stem_tiles_ok(h, w_)
    y = ops.stem_conv(x4, ohwi(w))
    err = float((nchw(y).double().cpu() - y_ref.detach()).norm() / y_ref.detach().norm())
    assert err < 2e-6, err
    ops.STEM_KERNEL = False
    try:
        y_old = ops.stem_conv(x4, ohwi(w))
    finally:
        ops.STEM_KERNEL = True
    torch.testing.assert_close(y, y_old, rtol=1e-5, atol=1e-5)
    bn1, bn2 = nn.BatchNorm2d(64).cuda(), nn.BatchNorm2d(64).cuda()
    st_ref = ops.bn_stats(y, bn1.weight.detach(), bn1.bias.detach(), bn1.running_mean, bn1.running_var, True)
    z, st = ops.stem_conv_bn(x4, ohwi(w), bn2)
    assert torch.equal(z, y)
    for a, b in ((st.mean, st_ref.mean), (st.invstd, st_ref.invstd), (st.scale, st_ref.scale), (st.shift, st_ref.shift),
                 (bn2.running_mean, bn1.running_mean), (bn2.running_var, bn1.running_var)):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
    dw = torch.zeros(64, 7, 7, 3, device="cuda")
    ops.stem_wgrad(x4, nhwc(gy), dw)
    ref = wd.grad.permute(0, 2, 3, 1)
    assert float((dw.double().cpu() - ref).norm() / ref.norm()) < 2e-6
    dw1 = dw.clone()
    ops.stem_wgrad(x4, nhwc(gy), dw)                        # accumulates
    assert float((dw.double().cpu() - 2 * ref).norm() / ref.norm()) < 4e-6
    dw2 = torch.zeros(64, 7, 7, 3, device="cuda")
    ops.stem_wgrad(x4, nhwc(gy), dw2)
    assert torch.equal(dw1, dw2)
    from uemda_amd import ops_bf16 as ob
    dwb = torch.zeros(64, 7, 7, 3, device="cuda")
    gyb = nhwc(gy).to(torch.bfloat16)
    ob.stem_wgrad(x4, gyb, dwb)                             # bf16 storage: the bf16 dz widened at the load, fp32 operands
    dwf = torch.zeros(64, 7, 7, 3, device="cuda")
    ops.stem_wgrad(x4, gyb.float(), dwf)
    assert torch.equal(dwb, dwf)


@pytest.mark.parametrize("shape", [(2, 16, 16, 64), (3, 15, 21, 64), (1, 32, 8, 128), (2, 6, 10, 64)])
def test_stem_fusions_equal_the_separate_passes(shape):
    """The stem's fused passes against the passes they replace: max-pool with BatchNorm + ReLU in its fetch == max-pool of the
    materialised relu(bn(z)), bit for bit (values AND argmax taps, ties between equal zeros included); BatchNorm backward
    reading the pooled gradient per 2x2 pixel block (even sizes) == max-pool backward followed by the BatchNorm backward, up
    to the order of the two channel sums (per-pixel gradients are the same numbers, added in the same order)."""
    from uemda_amd import ops
    n, h, w, c = shape
    g = torch.Generator().manual_seed(sum(shape))
    z = torch.randn(n, h, w, c, generator=g).cuda()
    gamma, beta = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.5).cuda()
    rm, rv = torch.zeros(c).cuda(), torch.ones(c).cuda()
    st = ops.bn_stats(z, gamma, beta, rm, rv, True)
    a = ops.affine_act(z, st, relu=True)
    y_ref, idx_ref = ops.maxpool_fwd(a, True)
    y, idx = ops.maxpool_affine_fwd(z, st, True)
    assert torch.equal(y, y_ref) and torch.equal(idx, idx_ref)
    dy = torch.randn(y.shape, generator=g).cuda()
    gg_ref, gb_ref = torch.zeros(c).cuda(), torch.zeros(c).cuda()
    da = ops.maxpool_bwd(dy, idx, z.shape)
    dz_ref = ops.bn_backward(z, da, st, gg_ref, gb_ref, None, True)
    if h % 2 or w % 2:
        return                                             # odd sizes keep the two-kernel path (blocks.StemFn)
    gg, gb = torch.zeros(c).cuda(), torch.zeros(c).cuda()
    dz = ops.bn_backward_pooled(z, dy, idx, st, gg, gb)
    torch.testing.assert_close(gg, gg_ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gb, gb_ref, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dz, dz_ref, rtol=1e-5, atol=1e-6)


def test_stem_conv_tile_statistics():
    """uem_conv2d_stem_fwd_stats: the same z as the plain stem conv and BatchNorm statistics equal to the two-pass ones."""
    import torch.nn as nn
    from uemda_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 3, 64, 64, generator=g)
    w = (torch.randn(64, 3, 7, 7, generator=g) / 12)
    x4 = ops.nchw3_to_nhwc4(x.cuda())
    bn1, bn2 = nn.BatchNorm2d(64).cuda(), nn.BatchNorm2d(64).cuda()
    z_ref = ops.stem_conv(x4, ohwi(w))
    st_ref = ops.bn_stats(z_ref, bn1.weight.detach(), bn1.bias.detach(), bn1.running_mean, bn1.running_var, True)
    z, st = ops.stem_conv_bn(x4, ohwi(w), bn2)
    assert torch.equal(z, z_ref)
    for a, b in ((st.mean, st_ref.mean), (st.invstd, st_ref.invstd), (st.scale, st_ref.scale), (st.shift, st_ref.shift),
                 (bn2.running_mean, bn1.running_mean), (bn2.running_var, bn1.running_var)):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)


# (2048, 65*65) / (2048, 3*65*65): odd pixel counts of 513x513 crops at 2048 channels, large enough for the rows-per-lane kernels --
# an odd block count makes their row stride an odd multiple of 1024 vectors' worth of channels (the second row read the
# per-channel vectors of channel c + 1024); (2048, 4224): the even neighbour, which does take two rows per lane
@pytest.mark.parametrize("Cn,M", [(64, 1000), (256, 77), (2048, 300), (128, 40000), (2048, 65 * 65), (2048, 3 * 65 * 65), (2048, 4224),
                                  (1024, 65 * 65 * 2 + 1)])
def test_batchnorm_stats_apply_backward(Cn, M):
    from uemda_amd import ops
    g = torch.Generator().manual_seed(Cn + M)
    x = (torch.randn(M, Cn, generator=g) * 2 + 5).requires_grad_(True)     # mean >> 0 stresses the variance
    gamma = (torch.rand(Cn, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(Cn, generator=g).requires_grad_(True)
    rm, rv = torch.randn(Cn, generator=g), torch.rand(Cn, generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y_ref = F.relu(F.batch_norm(x, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5))
    gy = torch.randn(M, Cn, generator=g)
    y_ref.backward(gy)
    xd, rmd, rvd = x.detach().cuda(), rm.cuda(), rv.cuda()
    st = ops.bn_stats(xd.view(1, 1, M, Cn), gamma.detach().cuda(), beta.detach().cuda(), rmd, rvd, True)
    y = ops.affine_act(xd, st, relu=True)
    torch.testing.assert_close(y.cpu(), y_ref.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rmd.cpu(), rm_ref, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rvd.cpu(), rv_ref, rtol=1e-4, atol=1e-6)
    gg, gb = torch.zeros(Cn, device="cuda"), torch.zeros(Cn, device="cuda")
    y2, bits = ops.affine_act(xd, st, relu=True, want_bits=True)          # packed sign bits (C % 32 == 0 here)
    assert torch.equal(y2, y)
    expect = (y.reshape(-1, 32) > 0).to(torch.int64) << torch.arange(32, device="cuda")
    assert torch.equal(bits.to(torch.int64) & 0xFFFFFFFF, expect.sum(1))
    # The expected gradient is built in float64 from the kernel's OWN ReLU mask: a pre-activation within rounding
    # of the kink may come out on either side of it in two implementations, and a flipped mask moves its whole channel's two sums by
    # one gradient value -- so instead of leaving such channels out (round 5: "calm" channels only), every channel and every element
    # is compared against the backward pass of the function the forward actually computed.  That the mask itself is right is the
    # forward comparison above (y against torch) plus: it differs from torch's only where the pre-activation is at rounding level.
    mask = (y > 0).cpu()
    pre = F.batch_norm(x.detach(), None, None, gamma.detach(), beta.detach(), True, 0.1, 1e-5)
    flipped = mask != (pre > 0)
    assert float(pre[flipped].abs().max()) < 1e-5 if flipped.any() else True
    x64, gy64, g64 = x.detach().double(), gy.double(), gamma.detach().double()
    mu, var = x64.mean(0), x64.var(0, unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    xhat = (x64 - mu) * invstd
    dp = gy64 * mask
    gb_ref, gg_ref = dp.sum(0), (dp * xhat).sum(0)
    dx_ref = g64 * invstd * (dp - gb_ref / M - xhat * (gg_ref / M))
    dx_scale = float(dx_ref.abs().max())
    for ymask in (None, y, bits):                         # recomputed, materialised and bit-packed masks agree
        gg.zero_(), gb.zero_()
        kw = dict(ymask_bits=bits) if ymask is bits else dict(ymask=ymask)
        dx = ops.bn_backward(xd, gy.cuda(), st, gg, gb, relu=True, **kw)
        if ymask is None:                                 # the mask recomputed from x, scale, shift is the forward's, bit for bit
            remask = (xd * st.scale + st.shift > 0).cpu()
            assert torch.equal(remask, mask)
        # every element of every channel: fp32 sums over up to 26 M elements against float64
        assert float((dx.cpu().double() - dx_ref).abs().max()) <= 1e-4 * max(1.0, dx_scale)
        torch.testing.assert_close(gg.cpu().double(), gg_ref, rtol=2e-4, atol=2e-3 * float(gg_ref.abs().max()) / max(1.0, M ** 0.5))
        torch.testing.assert_close(gb.cpu().double(), gb_ref, rtol=2e-4, atol=2e-3 * float(gb_ref.abs().max()) / max(1.0, M ** 0.5))


def test_maxpool_and_instnorm_vs_torch():
    from uemda_amd import ops
    g = torch.Generator().manual_seed(8)
    x = F.relu(torch.randn(2, 64, 19, 14, generator=g)).requires_grad_(True)    # relu => many ties at 0
    y_ref = F.max_pool2d(x, 3, 2, 1)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    y, idx = ops.maxpool_fwd(nhwc(x.detach()), True)
    assert torch.equal(nchw(y), y_ref.detach())
    dx = ops.maxpool_bwd(nhwc(gy), idx, (2, 19, 14, 64))
    torch.testing.assert_close(nchw(dx), x.grad, rtol=1e-6, atol=1e-6)
    gi = load_golden("layer_instnorm")
    xi = gi["x"]
    xi2 = torch.cat([xi, xi], 1)                           # 64 channels
    yi, inv = ops.instnorm_fwd(nhwc(xi2))
    torch.testing.assert_close(nchw(yi)[:, :32], gi["y"], rtol=1e-5, atol=1e-6)
    dxi = ops.instnorm_bwd(yi, nhwc(torch.cat([gi["gy"], gi["gy"]], 1)), inv)
    torch.testing.assert_close(nchw(dxi)[:, :32], gi["gx"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("shape", [(2, 32, 32, 256), (4, 16, 16, 64), (1, 30, 33, 256), (8, 8, 8, 128), (2, 7, 9, 128), (2, 40, 40, 128),
                                   (1, 32, 32, 64)])
def test_instnorm_one_pass_kernels_vs_torch(shape):
    """InstanceNorm2d (Encoder.py:123,147: no affine, no running statistics), forward and backward against torch in float64.  The
    shapes cover the register-resident kernels of round 5 (planes of 64 ... 1024 pixels, whole and ragged, 4 and 16 pixels per thread),
    the two-pass kernels they fall back to (larger planes, block counts that do not divide over the XCDs) -- and both give the same
    numbers up to the order of the sums."""
    from uemda_amd import ops, ops_bf16 as ob
    n, h, w, c = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(n, c, h, w, generator=g) * 1.5 + 3.0).double().requires_grad_(True)     # mean >> 0 stresses the variance
    y_ref = F.instance_norm(x, eps=1e-5)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy.double())
    y, inv = ops.instnorm_fwd(nhwc(x.detach().float()))
    assert float((nchw(y).double() - y_ref.detach()).abs().max()) < 2e-5
    dx = ops.instnorm_bwd(y, nhwc(gy), inv)
    assert float((nchw(dx).double() - x.grad).norm() / x.grad.norm()) < 2e-5
    xb = nhwc(x.detach().float()).to(torch.bfloat16)
    yb, invb = ob.instnorm_fwd(xb)
    y32, inv32 = ops.instnorm_fwd(xb.float())
    torch.testing.assert_close(yb, y32, rtol=1e-5, atol=1e-5)                  # the bf16 reader widens and does the same arithmetic
    torch.testing.assert_close(invb, inv32, rtol=1e-5, atol=1e-7)              # (its forward keeps the two-pass kernel: other sum order)
    dxb = ob.instnorm_bwd(y32, nhwc(gy), inv32)
    assert torch.equal(dxb, ops.instnorm_bwd(y32, nhwc(gy), inv32).to(torch.bfloat16))


# ---------------- layer goldens from the reference ---------------------------------------------------------
def _load_into(module, shapes_tag):
    from oracle.weights import fill_like
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    module.load_state_dict(fill_like(shapes, shapes_tag))
    return module


class _Holder(torch.nn.Module):
    """gives a lone block the flat arenas the full model normally provides"""

    def __init__(self, inner):
        super().__init__()
        self.inner = inner

    def flatten(self):
        from uemda_amd.models.Encoder import Deeplabv2
        Deeplabv2._flatten_parameters(self)
        return self


def _check_block(name, blk, rtol=2e-4, atol=5e-5):
    from oracle.weights import subsample
    g = load_golden(name)
    _load_into(blk, name)
    holder = _Holder(blk).cuda().flatten()
    blk.train()
    x = nhwc(g["x"]).requires_grad_(True)
    y = blk(x)
    torch.testing.assert_close(nchw(y.detach()), g["y"], rtol=rtol, atol=atol)
    y.backward(nhwc(g["gy"]))
    torch.testing.assert_close(nchw(x.grad), g["gx"], rtol=rtol * 10, atol=atol * 10)
    named = dict(blk.named_parameters())
    for k, v in g.items():
        if k.startswith("g:"):
            torch.testing.assert_close(subsample(named[k[2:]].grad.cpu().contiguous()), v, rtol=rtol * 10, atol=atol * 40)
    sd = blk.state_dict()
    for k, v in g.items():
        if k.startswith("post:"):
            torch.testing.assert_close(sd[k[5:]].cpu(), v, rtol=1e-4, atol=1e-5)
    return holder


def test_layer_bottleneck_stride2_golden():
    import torch.nn as nn
    from uemda_amd.resnet import Bottleneck
    ds = nn.Sequential(nn.Conv2d(64, 128, 1, 2, bias=False), nn.BatchNorm2d(128))
    _check_block("layer_bottleneck_s2", Bottleneck(64, 32, stride=2, downsample=ds))


def test_layer_bottleneck_dilation2_golden():
    from uemda_amd.resnet import Bottleneck
    _check_block("layer_bottleneck_d2", Bottleneck(128, 32, dilation=2))


def test_layer_aspp_golden():
    from uemda_amd.models import blocks
    from uemda_amd.models.Encoder import Classifier_Module
    from oracle.weights import subsample
    g = load_golden("layer_aspp")
    head = _load_into(Classifier_Module(32, [6, 12, 18, 24], [6, 12, 18, 24], C), "layer_aspp")
    holder = _Holder(head).cuda().flatten()
    x = nhwc(g["x"]).requires_grad_(True)
    x1, x2 = blocks.ASPPHeadsFn.apply(x, head, head, *list(head.parameters()))     # same head twice
    torch.testing.assert_close(nchw(x1.detach()), g["y"], rtol=2e-4, atol=5e-5)
    torch.testing.assert_close(nchw(x2.detach()), g["y"], rtol=2e-4, atol=5e-5)
    (x1 * nhwc(g["gy"])).sum().backward()
    torch.testing.assert_close(nchw(x.grad), g["gx"], rtol=2e-3, atol=5e-4)
    for k, v in g.items():
        if k.startswith("g:"):
            torch.testing.assert_close(subsample(dict(head.named_parameters())[k[2:]].grad.cpu().contiguous()), v, rtol=2e-3, atol=2e-3)


def test_layer_stem_golden():
    from uemda_amd.models import blocks
    from uemda_amd.resnet import ResNet
    from oracle.weights import fill_like, subsample
    g = load_golden("layer_stem")
    net = ResNet([1, 1, 1, 1])
    shapes = {"0.weight": (64, 3, 7, 7), "1.weight": (64,), "1.bias": (64,), "1.running_mean": (64,),
              "1.running_var": (64,), "1.num_batches_tracked": ()}
    sd = fill_like(shapes, "layer_stem")
    net.conv1.weight.data.copy_(sd["0.weight"])
    for a, b in (("weight", "1.weight"), ("bias", "1.bias"), ("running_mean", "1.running_mean"), ("running_var", "1.running_var")):
        getattr(net.bn1, a).data.copy_(sd[b])
    holder = _Holder(net).cuda().flatten()
    net.train()
    y = blocks.StemFn.apply(g["x"].cuda(), net, net.conv1.weight, net.bn1.weight, net.bn1.bias)
    torch.testing.assert_close(nchw(y.detach()), g["y"], rtol=2e-4, atol=5e-5)
    y.backward(nhwc(g["gy"]))
    torch.testing.assert_close(subsample(net.conv1.weight.grad.cpu().contiguous()), g["g:0.weight"], rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(net.bn1.weight.grad.cpu(), g["g:1.weight"], rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(net.bn1.running_var.cpu(), g["post:1.running_var"], rtol=1e-4, atol=1e-5)


# ---------------- full network + one SSL step against the reference golden (BASELINE config 1) ------------------
def _model(use_ppm=False, sd=None, resnet_type="resnet50", **backbone):
    from oracle.weights import det_state_dict
    from uemda_amd.models.Encoder import Deeplabv2
    cfg = dict(backbone=dict(resnet_type=resnet_type, output_stride=16, pretrained=False, **backbone), multi_layer=True,
               cascade=False, use_ppm=use_ppm, ppm=dict(num_classes=C, use_aux=False, fc_dim=2048),
               inchannels=2048, num_classes=C, is_ins_norm=True)
    m = Deeplabv2(cfg)
    sd = det_state_dict(resnet_type, C, use_ppm, seed=2333) if sd is None else sd
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd)
    return m.cuda()


def test_full_model_aspp_ssl_step_matches_reference_golden():
    from oracle import synth
    from oracle.weights import checksum, subsample
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import HYPER, StepState, ssl_step
    g = load_golden("model_aspp_r50_b2_256")
    model = _model(False)
    batch = {k: v.cuda() for k, v in synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=2333).items()}
    model.eval()
    with torch.no_grad():
        prob = model(batch["images_t"])
    torch.testing.assert_close(prob[:, :, ::8, ::8].cpu(), g["eval_prob_sample"], rtol=1e-3, atol=1e-5)
    al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
    al.prototypes = batch["prototypes"].clone()
    opt = FusedSGD(model, lr=1e-2, momentum=0.9, weight_decay=5e-4)
    from uemda_amd.models import blocks as _blocks
    hits0 = _blocks._Link.hits
    out = ssl_step(model, al, opt, StepState(C), batch, float(g["lr"]))
    # the fused residual path is really taken: 15 of the 16 blocks of each forward get their gradient handed over by the next one
    assert _blocks._Link.hits - hits0 == 30, _blocks._Link.hits - hits0
    # north_star: fp logits within 1e-3 rel of the reference CPU path
    for k in ("pred_s1", "pred_s2", "pred_t1", "pred_t2"):
        ref = g[k]
        err = (out[k].cpu() - ref).abs().max() / ref.abs().max()
        assert err < 1e-3, (k, float(err))
    torch.testing.assert_close(out["feat_t"].cpu().reshape(-1)[g["feat_idx"]], g["feat_t_sample"], rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(out["label_t_soft"][:, :, ::4, ::4].cpu(), g["soft_sample"], rtol=1e-3, atol=1e-5)
    agree = (out["label_t_hard"].cpu() == g["hard"].long()).float().mean().item()
    assert agree >= 0.9995, agree
    torch.testing.assert_close(out["loss_source"].cpu(), g["loss_source"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(out["loss_target"].cpu(), g["loss_target"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(al.prototypes.cpu(), g["prototypes"], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(out["grad_norm"].cpu().reshape(()), g["grad_norm"], rtol=5e-3, atol=1e-4)
    named = dict(model.named_parameters())
    for k, v in g.items():
        if k.startswith("grad:"):            # fixture grads are post-clip; so are ours (the fused step scales .grad)
            got = subsample(named[k[5:]].grad.cpu().contiguous())
            # relative L2: single ReLU-mask / max-pool-argmax flips near a kink perturb isolated elements
            # fp32 rounding flips ~1e-5 of the ReLU masks per layer; two CPU runs of the reference itself
            # (mkldnn on/off) differ by 1-2 % in the deepest gradients (scripts/grad_noise_floor.py)
            err = (got - v).norm() / (v.norm() + 1e-12)
            assert err < 5e-2, (k, float(err))
    sd = model.state_dict()
    torch.testing.assert_close(sd["encoder.resnet.bn1.running_mean"].cpu(), g["post_bn1_running_mean"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(sd["encoder.resnet.layer4.2.bn3.running_var"].cpu(), g["post_l4_bn3_running_var"], rtol=1e-3, atol=1e-5)
    assert int(sd["encoder.resnet.bn1.num_batches_tracked"]) == 2
    _check_updates(model, g, False)


@pytest.mark.parametrize("name,B,S,rtype,stride", [("model_aspp_r50_b8_512", 8, 512, "resnet50", 16),
                                                   ("model_aspp_r101_b2_256", 2, 256, "resnet101", 4)])
def test_full_model_aspp_ssl_step_matches_reference_golden_at(name, B, S, rtype, stride):
    """(1) The reference's own operating point: 8 source + 8 target tiles (configs/ToPotsdam.py:58, configs/st/uemda/2potsdam.py:31,43) of
    512 x 512, the benchmark's tile -- where layer3 / layer4 run 32 x 32 maps (Winograd on 512 / 2048 tiles per conv, F(4x4,3x3) in the
    backward pass), the persistent kernels walk several tiles per block and layer1 / layer2 take their large-map dispatch branches.
    (2) BASELINE config 5's model family, ResNet-101 (configs/st/uemda/2potsdam.py:6; 23 bottlenecks in layer3), pinned by the reference
    itself rather than through the oracle alone.
    One train_ssl_uem step against the reference's outputs (tests/golden/make_golden_r4.py step512 / r101): north_star's bars on the
    forward, every tensor's first update against ITS noise floor."""
    from oracle import synth
    from uemda_amd import ops
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import HYPER, StepState, ssl_step
    g = load_golden(name)
    model = _model(False, resnet_type=rtype)
    batch = {k: v.cuda() for k, v in synth.make_batch(B=B, H=S, W=S, C=C, k=2048, seed=2333).items()}
    if ops.WINOGRAD and S == 512:
        plan = ops.wino_plan((8, 32, 32, 512), 512, 3, 3, 1, 2, 2)                # the tile sizes this fixture exercises on layer4
        assert (plan.mf, plan.mb) == (4 if ops.WINOGRAD_F4_FWD else 2, 4 if ops.WINOGRAD_F4_BWD else 2)
    al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
    al.prototypes = batch["prototypes"].clone()
    opt = FusedSGD(model, lr=1e-2, momentum=0.9, weight_decay=5e-4)
    out = ssl_step(model, al, opt, StepState(C), batch, float(g["lr"]))
    worst = 0.0
    for k in ("pred_s1", "pred_s2", "pred_t1", "pred_t2"):
        ref = g[k]
        err = float((out[k].cpu() - ref).abs().max() / ref.abs().max())
        worst = max(worst, err)
        assert err < 1e-3, (k, err)                      # north_star: fp logits within 1e-3 rel of the reference CPU path
    agree = (out["label_t_hard"].cpu() == g["hard"].long()).float().mean().item()
    print(f"{rtype} {S}x{S} B={B}+{B}: worst logit error {worst:.2e} (the reference against itself: {float(g['ref_logit_floor']):.2e}), "
          f"hard labels {agree:.6f} (reference against itself: {float(g['ref_hard_agreement_floor']):.6f})")
    assert agree >= 0.9995, agree
    # the features: 1e-3, or 3 x how far the reference's own features move (R101 at B = 2: 23 training-mode BatchNorms deeper)
    ftol = max(1e-3, 3.0 * float(g["ref_feat_floor"])) if "ref_feat_floor" in g else 1e-3
    torch.testing.assert_close(out["feat_t"].cpu().reshape(-1)[g["feat_idx"]], g["feat_t_sample"], rtol=1e-3, atol=ftol)
    torch.testing.assert_close(out["label_t_soft"][:, :, ::stride, ::stride].cpu(), g["soft_sample"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(out["loss_source"].cpu(), g["loss_source"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(out["loss_target"].cpu(), g["loss_target"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(al.prototypes.cpu(), g["prototypes"], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(out["grad_norm"].cpu().reshape(()), g["grad_norm"], rtol=5e-3, atol=1e-4)
    sd = model.state_dict()
    torch.testing.assert_close(sd["encoder.resnet.bn1.running_mean"].cpu(), g["post_bn1_running_mean"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(sd["encoder.resnet.layer4.2.bn3.running_var"].cpu(), g["post_l4_bn3_running_var"], rtol=1e-3, atol=1e-5)
    _check_updates(model, g, False, resnet_type=rtype)


def _check_updates(model, g, use_ppm, num_classes=C, resnet_type="resnet50"):
    """One optimizer step seen through the UPDATE of EVERY parameter tensor (256 strided samples each).  The fixture holds
    the reference's update -lr * (clipped grad + wd * w) in float64 (w_post - w_pre itself is quantised to the weights' last
    place: 8 % of a BatchNorm gamma's update) and, per tensor, its fp32 noise floor: how far the reference's own step moves
    when (a) the CPU conv backend changes (mkldnn on / off) or (b) the input images move by one unit in the last place --
    ~1e-6 at the heads, 2-3 % in the encoder, whose gradients pass ~50 training-mode BatchNorm backward passes (B = 2).
    Checked: (1) every tensor's update against the reference within 3 x its floor (1e-3 where the floor is rounding level),
    and over the encoder as a whole no farther from the reference than the reference is from itself; (2) the fused optimizer
    kernel applied exactly that update to the weights.  (A Sum|w| checksum after one lr = 3e-3 step could not see a wrong
    update: VERDICT r1.)"""
    from oracle.weights import det_state_dict
    w0 = det_state_dict(resnet_type, num_classes, use_ppm, seed=2333)
    lr, wd = float(g["lr"]), 5e-4
    names, off, ref, floor = [str(n) for n in g["upd_names"]], g["upd_offsets"], g["upd_samples"], g["upd_noise_floor"]
    named = dict(model.named_parameters())
    assert set(names) == set(named), "the fixture covers every parameter tensor"
    frozen = set(str(n) for n in g["frozen_names"] if str(n)) if "frozen_names" in g else set()
    ratios, report = [], []
    for i, n in enumerate(names):
        p = named[n]
        if n in frozen:                                  # freeze_at / frozen BatchNorm: no gradient, not a bit of the weight moves
            assert not p.requires_grad and p.grad is None, n
            assert torch.equal(p.detach().cpu(), w0[n].float()), n
            assert float(ref[int(off[i]):int(off[i + 1])].abs().max()) == 0.0
            continue
        assert p.requires_grad, n
        st = max(1, p.numel() // 256)
        w_pre = w0[n].reshape(-1)[::st][:256].double()
        grad = p.grad.detach().cpu().reshape(-1)[::st][:256].double()            # post-clip (the fused step scales .grad)
        upd = -lr * (grad + wd * w_pre)
        r = ref[int(off[i]):int(off[i + 1])].double()
        err = float((upd - r).norm() / (r.norm() + 1e-300))
        fl = float(floor[i])
        report.append((err, fl, n))
        if fl < 2.5e-4:
            assert err < 1e-3, (n, err, fl)              # head side: rounding level
        else:
            ratios.append(err / fl)
        # (2) the optimizer kernel: w_post = fp32(w_pre + update), to the last place of w plus the rounding of the update
        w_post = p.detach().cpu().reshape(-1)[::st][:256].double()
        ulp = torch.maximum(w_pre.abs(), w_post.abs()).float().clamp_min(1e-30)
        ulp = (torch.nextafter(ulp, torch.full_like(ulp, float("inf"))) - ulp).double()
        assert ((w_post - (w_pre + upd)).abs() <= 1.5 * ulp + 2e-6 * upd.abs()).all(), n
    if not ratios:                                       # every trainable tensor sits at rounding level (frozen BatchNorm statistics)
        return
    ratios.sort()
    median = ratios[len(ratios) // 2]
    if "upd_noise_floor_max" in g:
        # Derived bar: the floor of a tensor is the MAXIMUM distance of its update over N perturbed runs of the
        # reference itself (other conv backend x one-ulp input nudges x thread counts; N = 24, 6 for the 512 x 512 fixture).  An
        # implementation as close to the reference as the reference is to itself exceeds the maximum of N exchangeable draws with
        # probability 1 / (N + 1) per tensor; every tensor -- none excluded, no tuned constants -- must stay within 1.5 x that maximum
        fmax = g["upd_noise_floor_max"]
        by_name = {n: float(fmax[i]) for i, n in enumerate(names)}
        derived = sorted(((e / max(by_name[n], 1e-30), e, by_name[n], n) for e, f, n in report if by_name[n] >= 2.5e-4), reverse=True)
        print(f"update error / max over {g['upd_noise_floor_draws']} reference self-draws, {len(derived)} tensors: median "
              f"{derived[len(derived) // 2][0]:.2f}, worst {[(round(r, 2), n) for r, e, f, n in derived[:3]]}; against the fixture's two-draw "
              f"floor: median {median:.2f}, max {ratios[-1]:.2f}")
        if os.environ.get("UEM_TEST_REPORT_ONLY"):
            return
        for r, e, f, n in derived:
            assert r <= 1.5, (n, e, f, r)
        for e, f, n in report:                           # tensors whose reference floor is at rounding level (head side): 1e-3 as before
            if by_name[n] < 2.5e-4:
                assert e < 1e-3, (n, e, by_name[n])
        return
    worst = sorted(((e / f, n) for e, f, n in report if f >= 2.5e-4), reverse=True)[:4]
    # fixtures without a derived table (none today): the round-5 bar
    assert worst[0][0] < 5.0 and (len(worst) < 2 or worst[1][0] < 3.0), worst
    print(f"update error / reference noise floor over {len(ratios)} encoder tensors: median {median:.2f}, max {ratios[-1]:.2f}; "
          f"worst absolute {max(report)[:2]} {max(report)[2]}")
    assert median < 1.5, median


class _ReferenceOptimizerLines:
    """The caller's own optimizer lines, reference tools/train_ssl_uem.py:169-170 and :228-232, on the arena-view parameters:
        optimizer = optim.SGD(model.parameters(), lr=..., momentum=0.9, weight_decay=5e-4)
        optimizer.zero_grad(); loss.backward()
        clip_grad.clip_grad_norm_(parameters=model.parameters(), max_norm=32, norm_type=2)
        optimizer.step()
    with `clip_grad` = uemda_amd.optim, wrapped in the step(max_norm=...) shape ssl_step calls."""

    def __init__(self, model, lr, momentum, weight_decay):
        self.model = model
        self.opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=momentum, weight_decay=weight_decay)
        self.param_groups = self.opt.param_groups
        self.last_grad_norm = None

    def zero_grad(self):
        self.opt.zero_grad()

    def step(self, max_norm=None, grad_prescale=1.0):
        from uemda_amd import optim as clip_grad
        assert grad_prescale == 1.0
        self.last_grad_norm = clip_grad.clip_grad_norm_(parameters=self.model.parameters(), max_norm=max_norm, norm_type=2).reshape(1)
        self.opt.step()


def test_reference_optimizer_lines_torch_sgd_and_clip_grad_norm_on_the_arena_model():
    """the golden train_ssl_uem step driven by torch.optim.SGD + uemda_amd.optim.clip_grad_norm_ exactly as the
    reference script writes it must give the reference's update of every tensor (the same bar as FusedSGD: _check_updates), and a
    second step must see the updated weights (the cached transposed / Winograd filter banks follow the parameters' versions)."""
    from oracle import synth
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.step import HYPER, StepState, ssl_step
    g = load_golden("model_aspp_r50_b2_256")
    model = _model(False)
    batch = {k: v.cuda() for k, v in synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=2333).items()}
    al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
    al.prototypes = batch["prototypes"].clone()
    opt = _ReferenceOptimizerLines(model, 1e-2, 0.9, 5e-4)
    out = ssl_step(model, al, opt, StepState(C), batch, float(g["lr"]))
    for k in ("pred_s1", "pred_t2"):
        assert (out[k].cpu() - g[k]).abs().max() / g[k].abs().max() < 1e-3
    torch.testing.assert_close(out["grad_norm"].cpu().reshape(()), g["grad_norm"], rtol=5e-3, atol=1e-4)
    _check_updates(model, g, False)
    # the next forward runs on the UPDATED weights: identical to a fresh model loaded with them
    model.train()
    with torch.no_grad():
        p1 = model(batch["images_s"])[0]
    fresh = _model(False, sd={k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    fresh.train()
    with torch.no_grad():
        p2 = fresh(batch["images_s"])[0]
    assert torch.equal(p1, p2)
    assert (p1.cpu() - out["pred_s1"].cpu()).abs().max() > 1e-4          # and they did move


@pytest.mark.parametrize("storage,graph_side,graph_two", [("fp32", False, True), ("bf16", False, True), ("fp32", False, False), ("fp32", True, False),
                                                          ("bf16", True, False)])
def test_graphed_step_replays_the_eager_step_with_a_moving_learning_rate(storage, graph_side, graph_two, monkeypatch):
    """graph_side: the weight gradients' side stream captured WITH the step (forks and a join inside the graph; round 6, off by default
    because the forked graph replays slower -- ops.GRAPH_SIDE) or kept out of the capture.  graph_two: the step's two graphs captured on
    two streams (the fork and the joins as graph edges, ops.GRAPH_TWO_STREAM) or the sequential step captured (a captured side stream
    keeps the step sequential: step._may_fork).
    VERDICT r2 item 7: the whole train_ssl_uem step captured in one hipGraph (uemda_amd.step.GraphedStep) must BE the eager step.
    Before each of three replays -- with a learning rate that changes 10x from step to step, so that a rate baked into the capture
    would show -- the complete training state (weights, BatchNorm buffers, momentum, prototypes) is copied into a second model that
    takes the same step eagerly: same losses, same hard labels, same updated weights to fp32-atomic order.  (Whole trajectories
    cannot be compared on this network: two EAGER runs of it drift apart by 4e-4 in the loss within three steps.)"""
    from oracle import synth
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import HYPER, GraphedStep, StepState, ssl_step
    from uemda_amd import ops as _ops
    monkeypatch.setattr(_ops, "GRAPH_SIDE", graph_side)
    monkeypatch.setattr(_ops, "GRAPH_TWO_STREAM", graph_two)
    batch = {k: v.cuda() for k, v in synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=2333).items()}

    def fresh():
        model = _model(False).set_storage(storage)                      # (bf16 storage: the 45 ms step, where the host's share was largest)
        al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
        al.prototypes = batch["prototypes"].clone()
        return model, al, FusedSGD(model, lr=1e-2, momentum=0.9, weight_decay=5e-4), StepState(C)
    m1, a1, o1, s1 = fresh()
    gs = GraphedStep(ssl_step, m1, a1, o1, s1, batch, warmup=2, lr=1e-3, sup_ignore_id=256)      # 2 eager warm-up steps, then the capture
    m2, a2, o2, s2 = fresh()
    for lr in (3e-3, 3e-4, 3e-2):
        m2.load_state_dict({k: v.detach().clone() for k, v in m1.state_dict().items()})
        o2.momentum_buffer.copy_(o1.momentum_buffer)
        o2._steps = o1._steps
        a2.prototypes = a1.prototypes.clone()
        n = m1.flat_parameters()[2]
        w_before = m1.flat_parameters()[0][:n].clone()
        out = gs(lr)
        ref = ssl_step(m2, a2, o2, s2, batch, lr, sup_ignore_id=256)
        torch.cuda.synchronize()
        assert float(out["loss_source"]) == pytest.approx(float(ref["loss_source"]), rel=2e-6)
        assert float(out["loss_target"]) == pytest.approx(float(ref["loss_target"]), rel=2e-5, abs=1e-7)
        assert torch.equal(out["label_t_hard"], ref["label_t_hard"])
        assert float(out["grad_norm"]) == pytest.approx(float(ref["grad_norm"]), rel=1e-4)
        w1, w2 = m1.flat_parameters()[0][:n], m2.flat_parameters()[0][:n]
        moved = float((w1 - w_before).norm())
        assert float((w1 - w2).norm()) < 2e-3 * moved, (float((w1 - w2).norm()), moved)            # the same update, at THIS learning rate
        torch.testing.assert_close(a1.prototypes, a2.prototypes, rtol=1e-5, atol=1e-6)
    gs.check()
    if storage == "fp32":
        # PPM heads: Dropout2d(0.1).  A host-side seed would be frozen into the captured launch; while capturing the mask comes from
        # torch's graph-safe generator instead: with the learning rate at 0 and no momentum the weights stand still, so two replays
        # differ through their masks only -- and must differ.
        ppm = _model(True)
        ap = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
        ap.prototypes = batch["prototypes"].clone()
        gp = GraphedStep(ssl_step, ppm, ap, FusedSGD(ppm, lr=0.0, momentum=0.0, weight_decay=0.0), StepState(C), batch, warmup=1, lr=0.0,
                         sup_ignore_id=256)
        p1 = gp(0.0)["pred_s1"].clone()
        p2 = gp(0.0)["pred_s1"].clone()
        p3 = gp(0.0)["pred_s1"].clone()
        assert torch.isfinite(p1).all() and not torch.equal(p1, p2) and not torch.equal(p2, p3)
        assert float((p1 - p2).norm() / p1.norm()) < 0.7            # a tenth of the channels moved (~sqrt(2 x 0.1) of the norm), not the prediction as a whole
    assert int(m1.state_dict()["encoder.resnet.bn1.num_batches_tracked"]) == int(m2.state_dict()["encoder.resnet.bn1.num_batches_tracked"]) == 10


@pytest.mark.parametrize("storage", ["fp32", "bf16"])
def test_side_stream_weight_gradients_equal_the_serial_backward(storage):
    """Round 5: the weight gradients run on a side stream beside the data-gradient chain (ops.on_side).  The same step with the side
    stream on and off, from the same state, must give the same losses, labels and update -- to the order of the fp32 atomics the
    weight-gradient kernels accumulate with, which two SERIAL runs differ by as well -- and the allocator must not hand a tensor the
    side stream still reads to the main stream (a NaN-filled allocation pattern between the steps would show)."""
    from oracle import synth
    from uemda_amd import ops, ops_bf16 as ob
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import HYPER, StepState, ssl_step
    batch = {k: v.cuda() for k, v in synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=2333).items()}

    def run(side):
        saved = ops.SIDE_WGRAD_F32, ob.SIDE_WGRAD
        ops.SIDE_WGRAD_F32 = ob.SIDE_WGRAD = side
        try:
            model = _model(False).set_storage(storage)
            al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
            al.prototypes = batch["prototypes"].clone()
            opt = FusedSGD(model, lr=1e-2, momentum=0.9, weight_decay=5e-4)
            n = model.flat_parameters()[2]
            w0 = model.flat_parameters()[0][:n].clone()
            # ONE step: its forward is deterministic, so losses and labels must be equal and the update differs by the order of the
            # weight-gradient atomics only (a second step on this B = 2 network amplifies that to percents, serial or not)
            out = ssl_step(model, al, opt, StepState(C), batch, 3e-3, sup_ignore_id=256)
            torch.cuda.synchronize()
            return out, (model.flat_parameters()[0][:n] - w0).double(), al.prototypes.clone()
        finally:
            ops.SIDE_WGRAD_F32, ob.SIDE_WGRAD = saved
    o1, u1, p1 = run(True)
    o0, u0, p0 = run(False)
    o0b, u0b, _ = run(False)                                          # how far two serial runs are from each other
    noise = float((u0 - u0b).norm() / u0.norm())
    diff = float((u1 - u0).norm() / u0.norm())
    print(f"{storage}: side-stream against serial update {diff:.2e}; serial against serial {noise:.2e}")
    assert torch.isfinite(u1).all() and diff < max(5.0 * noise, 1e-5), (diff, noise)
    assert float(o1["loss_source"]) == float(o0["loss_source"]) and float(o1["loss_target"]) == float(o0["loss_target"])
    assert torch.equal(o1["label_t_hard"], o0["label_t_hard"])
    torch.testing.assert_close(p1, p0, rtol=1e-4, atol=1e-5)


def test_graphed_step_outlives_another_model_whose_filter_banks_its_refresh_launch_covers():
    """the ONE weight-preparation launch a graph captures covers every live job on the device -- also those of a second
    model (a teacher, an evaluation copy) that was alive at capture time.  When that model is dropped afterwards and an eager request
    rebuilds the job table, its parameters and derived banks must stay valid for the replays: the graph holds them
    (`GraphedStep._prep_hold`) and releases them when it goes."""
    import gc
    import weakref
    from oracle import synth
    from uemda_amd import ops
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import HYPER, GraphedStep, StepState, ssl_step
    batch = {k: v.cuda() for k, v in synth.make_batch(B=2, H=128, W=128, C=C, k=2048, seed=2333).items()}

    def fresh():
        model = _model(False)
        al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
        al.prototypes = batch["prototypes"].clone()
        return model, al, FusedSGD(model, lr=1e-2, momentum=0.9, weight_decay=5e-4), StepState(C)
    other, ao, oo, so = fresh()
    ssl_step(other, ao, oo, so, batch, 1e-3, sup_ignore_id=64)          # the second model's backward registers its filter banks
    m1, a1, o1, s1 = fresh()
    gs = GraphedStep(ssl_step, m1, a1, o1, s1, batch, warmup=2, lr=1e-3, sup_ignore_id=64)
    held = {id(p) for p, _ in gs._prep_hold}
    assert any(id(p) in held for p in other.parameters()) and any(id(p) in held for p in m1.parameters())
    alive = weakref.ref(other.encoder.resnet.layer4[0].conv2.weight)
    del other, ao, oo, so
    gc.collect()
    torch.cuda.empty_cache()
    assert alive() is not None                                          # kept by the graph, not by the table's weak references
    ops.weights_changed()
    ops.weight_transpose_cached(m1.encoder.resnet.layer4[0].conv2.weight)       # an eager stale request: settles the table again
    junk = torch.full((48 << 20,), float("nan"), device="cuda")        # whatever was freed is recycled with poison
    m2, a2, o2, s2 = fresh()
    m2.load_state_dict({k: v.detach().clone() for k, v in m1.state_dict().items()})
    o2.momentum_buffer.copy_(o1.momentum_buffer)
    o2._steps = o1._steps
    a2.prototypes = a1.prototypes.clone()
    out = gs(3e-3)
    ref = ssl_step(m2, a2, o2, s2, batch, 3e-3, sup_ignore_id=64)
    torch.cuda.synchronize()
    assert torch.isfinite(out["pred_s1"]).all() and junk.isnan().all()
    assert float(out["loss_source"]) == pytest.approx(float(ref["loss_source"]), rel=2e-6)
    assert torch.equal(out["label_t_hard"], ref["label_t_hard"])
    n = m1.flat_parameters()[2]
    w1, w2 = m1.flat_parameters()[0][:n], m2.flat_parameters()[0][:n]
    assert float((w1 - w2).norm() / w2.norm()) < 1e-5
    del gs, out
    gc.collect()
    assert alive() is None                                              # released with the graph


@pytest.mark.parametrize("affine_trainable", [True, False])
@pytest.mark.parametrize("shape", ["stride2_downsample", "stride1_downsample", "identity_dilated"])
def test_bottleneck_eval_mode_batchnorm_backward_vs_torch(shape, affine_trainable):
    """A bottleneck whose BatchNorms run in eval mode inside a training graph (ResNetEncoder batchnorm_trainable=False,
    reference resnet.py:112-117,183-190): y, dx and every parameter gradient against torch autograd in float64 on the same
    weights and running statistics.  With constant statistics nothing amplifies rounding: 1e-5 relative."""
    import torch.nn as nn
    import torch.nn.functional as F
    from uemda_amd.resnet import Bottleneck
    torch.manual_seed(11)
    if shape.endswith("downsample"):
        sd_ = 2 if shape.startswith("stride2") else 1
        blk = Bottleneck(64, 32, stride=sd_, downsample=nn.Sequential(nn.Conv2d(64, 128, 1, sd_, bias=False), nn.BatchNorm2d(128)))
        cin = 64
    else:
        blk = Bottleneck(128, 32, dilation=2)
        cin = 128
    for m in blk.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.3)
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 2.0)
            m.eval()
            for p in m.parameters():
                p.requires_grad = affine_trainable
    sd = {k: v.clone() for k, v in blk.state_dict().items()}
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, cin, 16, 16, generator=g)
    holder = _Holder(blk).cuda().flatten()
    xg = nhwc(x).requires_grad_(True)
    y = blk(xg)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy.cuda())
    # float64 torch
    ref = Bottleneck(cin, 32, stride=blk.stride, dilation=blk.dilation,
                     downsample=None if blk.downsample is None else nn.Sequential(nn.Conv2d(64, 128, 1, blk.stride, bias=False), nn.BatchNorm2d(128))).double()
    ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()})
    ref.eval()
    x64 = x.double().requires_grad_(True)

    def bn(t, m):
        return F.batch_norm(t, m.running_mean, m.running_var, m.weight, m.bias, False, 0.1, m.eps)
    o = F.relu(bn(F.conv2d(x64, ref.conv1.weight), ref.bn1))
    o = F.relu(bn(F.conv2d(o, ref.conv2.weight, None, ref.conv2.stride, ref.conv2.padding, ref.conv2.dilation), ref.bn2))
    o = bn(F.conv2d(o, ref.conv3.weight), ref.bn3)
    idn = x64 if ref.downsample is None else bn(F.conv2d(x64, ref.downsample[0].weight, None, ref.downsample[0].stride), ref.downsample[1])
    y64 = F.relu(o + idn)
    y64.backward(gy.permute(0, 3, 1, 2).double())

    def rel(a, b):
        return float((a.double().cpu() - b).norm() / b.norm())
    assert rel(nchw(y.detach()), y64.detach()) < 1e-5
    assert rel(nchw(xg.grad), x64.grad) < 1e-5
    named, rnamed = dict(blk.named_parameters()), dict(ref.named_parameters())
    for n, p in named.items():
        if "bn" in n or "downsample.1" in n:
            if not affine_trainable:
                assert p.grad is None, n
                continue
        assert rel(p.grad, rnamed[n].grad) < 2e-5, (n, rel(p.grad, rnamed[n].grad))
    for n, b in blk.named_buffers():                                   # eval mode: statistics and counters untouched
        assert torch.equal(b.cpu(), sd[n]), n
    del holder


ENCODER_OPTIONS = {"frozen": dict(freeze_at=2, batchnorm_trainable=False), "cp": dict(with_cp=(True, True, True, True))}


@pytest.mark.parametrize("tag", ["frozen", "cp"])
def test_ssl_step_encoder_options_match_reference_golden(tag):
    """The ResNetEncoder modes the config exposes but no UemDA script switches on (reference uemda/resnet.py:112-130,146-165,
    183-190), one train_ssl_uem step each against the reference's own step:
      frozen: freeze_at=2 + batchnorm_trainable=False -- stem and layer1 frozen, every encoder BatchNorm in eval mode with frozen
              gamma / beta (backward through running statistics; no gradient, no weight decay, no momentum on frozen tensors;
              running statistics and num_batches_tracked untouched);
      cp:     every residual layer under torch.utils.checkpoint (its forward runs again inside backward, so its BatchNorm
              running statistics move twice per forward and num_batches_tracked counts 4 for the step's two forwards)."""
    from oracle import synth
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import HYPER, StepState, ssl_step
    g = load_golden(f"model_aspp_r50_b2_256_{tag}")
    assert str(g["options"]) == repr(ENCODER_OPTIONS[tag])
    from oracle.weights import det_state_dict
    from conftest import golden_initial_state
    model = _model(False, sd=golden_initial_state(g, det_state_dict("resnet50", C, False, seed=2333)), **ENCODER_OPTIONS[tag])
    model.train()
    batch = {k: v.cuda() for k, v in synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=2333).items()}
    al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
    al.prototypes = batch["prototypes"].clone()
    opt = FusedSGD(model, lr=1e-2, momentum=0.9, weight_decay=5e-4)
    out = ssl_step(model, al, opt, StepState(C), batch, float(g["lr"]))
    for k in ("pred_s1", "pred_s2", "pred_t1", "pred_t2"):
        err = (out[k].cpu() - g[k]).abs().max() / g[k].abs().max()
        assert err < 1e-3, (k, float(err))
    assert (out["label_t_hard"].cpu() == g["hard"].long()).float().mean().item() >= 0.9995
    torch.testing.assert_close(out["loss_source"].cpu(), g["loss_source"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(out["loss_target"].cpu(), g["loss_target"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(al.prototypes.cpu(), g["prototypes"], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(out["grad_norm"].cpu().reshape(()), g["grad_norm"], rtol=5e-3, atol=1e-4)
    sd = model.state_dict()
    for k, v in g.items():
        if k.startswith("post:"):
            if k.endswith("num_batches_tracked"):
                assert int(sd[k[5:]]) == int(v), (k, int(sd[k[5:]]), int(v))
            else:
                torch.testing.assert_close(sd[k[5:]].cpu(), v, rtol=1e-3, atol=1e-5)
    _check_updates(model, g, False)


def test_fused_sgd_matches_torch_sgd_with_clipping():
    """uem_sgd_clip_step over the flat arenas == clip_grad_norm_(32) + torch.optim.SGD(momentum .9, wd 5e-4) over three steps
    (train_ssl_uem.py:169-170,228-232): momentum buffer, weight decay, clip coefficient, the data-parallel prescale."""
    from uemda_amd.optim import FusedSGD
    model = _model(False)
    ref_params = [p.detach().cpu().clone().requires_grad_(True) for p in model.parameters()]
    ref_opt = torch.optim.SGD(ref_params, lr=1e-2, momentum=0.9, weight_decay=5e-4)
    opt = FusedSGD(model, lr=1e-2, momentum=0.9, weight_decay=5e-4)
    gen = torch.Generator().manual_seed(12)
    for step, (lr, scale, prescale) in enumerate([(3e-3, 40.0, 1.0), (5e-3, 0.01, 0.5), (1e-2, 3.0, 1.0)]):
        opt.zero_grad()
        for p, q in zip(model.parameters(), ref_params):
            gq = torch.randn(q.shape, generator=gen) * scale / q.numel() ** 0.5          # total norm ~ scale * sqrt(#tensors)
            q.grad = gq * prescale
            p.grad.copy_(gq.cuda() if p.grad.dim() < 4 else gq.cuda().contiguous(memory_format=torch.channels_last))
        norm_ref = torch.nn.utils.clip_grad_norm_(ref_params, max_norm=32, norm_type=2)
        ref_opt.param_groups[0]["lr"] = lr
        ref_opt.step()
        opt.param_groups[0]["lr"] = lr
        opt.step(max_norm=32.0, grad_prescale=prescale)
        torch.testing.assert_close(opt.last_grad_norm.cpu().reshape(()) * prescale, norm_ref, rtol=1e-5, atol=1e-6)
        for (n, p), q in zip(model.named_parameters(), ref_params):
            torch.testing.assert_close(p.detach().cpu(), q.detach(), rtol=2e-6, atol=2e-7, msg=lambda m, n=n, s=step: f"{n} step {s}: {m}")


def test_layer_ppm_golden():
    from uemda_amd.models import ppm
    from uemda_amd.models.Encoder import PPMBilinear
    from oracle.weights import subsample
    g = load_golden("layer_ppm")
    head = _load_into(PPMBilinear(num_classes=C, fc_dim=32), "layer_ppm")
    head.conv_last[3].p = 0.0                      # Dropout2d off for parity (stochastic in train mode)
    holder = _Holder(head).cuda().flatten()
    head.train()
    x = nhwc(g["x"]).requires_grad_(True)
    y = ppm.ppm_head(x, head)
    torch.testing.assert_close(nchw(y.detach()), g["y"], rtol=1e-3, atol=2e-4)
    y.backward(nhwc(g["gy"]))
    torch.testing.assert_close(nchw(x.grad), g["gx"], rtol=5e-3, atol=2e-3)
    named = dict(head.named_parameters())
    for k, v in g.items():
        if k.startswith("g:"):
            got = subsample(named[k[2:]].grad.cpu().contiguous())
            assert (got - v).norm() / (v.norm() + 1e-12) < 5e-3, k
    sd = head.state_dict()
    for k, v in g.items():
        if k.startswith("post:"):
            torch.testing.assert_close(sd[k[5:]].cpu(), v, rtol=1e-4, atol=1e-5)
    # dropout on: a fraction ~p of (n, channel) planes is zeroed and the rest scaled by 1/(1-p)
    head.conv_last[3].p = 0.5
    y2 = ppm.ppm_head(nhwc(g["x"]), head)
    assert torch.isfinite(y2).all() and not torch.allclose(y2, y.detach())


def test_ppm_head_does_not_reuse_pooled_maps_of_a_freed_feature_map():
    """the pooled maps shared by the two heads were cached on (address, shape); a second feature map of the same shape
    placed at the freed address got the FIRST map's pools.  Outside Deeplabv2._heads' scope nothing is shared now."""
    from uemda_amd.models import ppm
    from uemda_amd.models.Encoder import PPMBilinear
    head = _load_into(PPMBilinear(num_classes=C, fc_dim=32), "layer_ppm")
    holder = _Holder(head).cuda().flatten()         # noqa: F841  (owns the arenas the head's parameters are views of)
    head.train()
    head.conv_last[3].p = 0.0
    gen = torch.Generator().manual_seed(3)
    xa, xb = torch.randn(2, 32, 16, 16, generator=gen), torch.randn(2, 32, 16, 16, generator=gen)
    with torch.no_grad():
        ref_b = ppm.ppm_head(nhwc(xb), head).cpu()
        fa = nhwc(xa)
        addr = fa.data_ptr()
        ppm.ppm_head(fa, head)
        del fa
        fb = nhwc(xb)                                   # the caching allocator hands the freed block to the same-shape tensor
        same_address = fb.data_ptr() == addr
        got_b = ppm.ppm_head(fb, head).cpu()
    torch.testing.assert_close(got_b, ref_b, rtol=0, atol=0)
    print("second feature map reused the first one's address:", same_address)
    # inside a scope two calls on the SAME tensor object share the pools; a different object does not
    with ppm.shared_pools(), torch.no_grad():
        fa = nhwc(xa)
        ya1 = ppm.ppm_head(fa, head)
        ya2 = ppm.ppm_head(fa, head)
        yb = ppm.ppm_head(nhwc(xb), head)
    assert torch.equal(ya1, ya2)
    torch.testing.assert_close(yb.cpu(), ref_b, rtol=0, atol=0)


def test_full_model_ppm_ssl_step_matches_reference_golden():
    """the configuration every UemDA script instantiates (use_ppm=True, train_ssl_uem.py:91-108)"""
    from oracle import synth
    from oracle.weights import checksum
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import HYPER, StepState, ssl_step
    g = load_golden("model_ppm_r50_b2_256")
    model = _model(True)
    model.layer5.conv_last[3].p = 0.0
    model.layer6.conv_last[3].p = 0.0
    batch = {k: v.cuda() for k, v in synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=2333).items()}
    model.eval()
    with torch.no_grad():
        prob = model(batch["images_t"])
    torch.testing.assert_close(prob[:, :, ::8, ::8].cpu(), g["eval_prob_sample"], rtol=1e-3, atol=1e-5)
    al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
    al.prototypes = batch["prototypes"].clone()
    opt = FusedSGD(model, lr=1e-2, momentum=0.9, weight_decay=5e-4)
    out = ssl_step(model, al, opt, StepState(C), batch, float(g["lr"]))
    for k in ("pred_s1", "pred_s2", "pred_t1", "pred_t2"):
        err = (out[k].cpu() - g[k]).abs().max() / g[k].abs().max()
        assert err < 1e-3, (k, float(err))
    agree = (out["label_t_hard"].cpu() == g["hard"].long()).float().mean().item()
    assert agree >= 0.9995, agree
    torch.testing.assert_close(out["loss_source"].cpu(), g["loss_source"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(out["loss_target"].cpu(), g["loss_target"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(out["grad_norm"].cpu().reshape(()), g["grad_norm"], rtol=1e-2, atol=1e-4)
    _check_updates(model, g, True)


def test_cpu_tensors_fail_loudly():
    from uemda_amd import UemError
    from uemda_amd.gast.pseudo_generation import pseudo_selection
    with pytest.raises(UemError):
        pseudo_selection(torch.rand(1, C, 8, 8), return_type="tensor")


def test_ragged_batch_and_tile_size_ssl_step_vs_oracle():
    """Odd batch (3), non-square 144x208 tiles (9x13 feature map: no GEMM dimension is a multiple of the 128-row
    tile, the fused BatchNorm epilogues fall back to the stand-alone passes): one SSL step against the oracle."""
    from oracle import synth
    from oracle.model import OracleDeeplabv2
    from oracle.step import HYPER as OH, SGDState, ssl_step as oracle_ssl
    from oracle.weights import det_state_dict
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import HYPER, StepState, ssl_step
    sd = det_state_dict("resnet50", C, False, seed=2333)
    bc = synth.make_batch(B=3, H=144, W=208, C=C, k=2048, seed=77)
    om = OracleDeeplabv2(sd, "resnet50", C, False)
    ref = oracle_ssl(om, SGDState(om.parameters(), 0.9, 5e-4), bc["prototypes"], bc, 2e-3, OH)
    model = _model(False)
    b = {k: v.cuda() for k, v in bc.items()}
    al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
    al.prototypes = b["prototypes"].clone()
    out = ssl_step(model, al, FusedSGD(model, 1e-2, 0.9, 5e-4), StepState(C), b, 2e-3)
    for k in ("pred_s1", "pred_s2", "pred_t1", "pred_t2"):
        err = (out[k].cpu() - ref[k]).abs().max() / ref[k].abs().max()
        assert err < 1e-3, (k, float(err))
    assert (out["label_t_hard"].cpu() == ref["label_t_hard"]).float().mean() >= 0.9995
    torch.testing.assert_close(out["label_t_soft"].cpu(), ref["label_t_soft"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(out["loss_source"].cpu(), ref["loss_source"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(out["loss_target"].cpu(), ref["loss_target"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(al.prototypes.cpu(), ref["prototypes"], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(out["grad_norm"].cpu().reshape(()), ref["grad_norm"].reshape(()), rtol=2e-2, atol=1e-4)


def test_resnet101_aspp_src_step_vs_oracle():
    """BASELINE config 5's architecture (ResNet-101 encoder) on the train_src path: one source-only step at B=2,
    128x128 against the oracle (forward logits, loss, gradient norm, a deep and a shallow weight update)."""
    from oracle import synth
    from oracle.model import OracleDeeplabv2
    from oracle.step import HYPER as OH, SGDState, src_step as oracle_src
    from oracle.weights import det_state_dict
    from uemda_amd.models.Encoder import Deeplabv2
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import StepState, src_step
    cfg = dict(backbone=dict(resnet_type="resnet101", output_stride=16, pretrained=False), multi_layer=True, cascade=False,
               use_ppm=False, ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)
    sd = det_state_dict("resnet101", C, False, seed=99)
    names = ("layer5.conv2d_list.0.weight", "encoder.resnet.layer3.22.conv3.weight")
    w0 = {n: sd[n].clone() for n in names}
    bc = synth.make_batch(B=2, H=128, W=128, C=C, k=2048, seed=5)
    om = OracleDeeplabv2({k: v.clone() for k, v in sd.items()}, "resnet101", C, False)
    ref = oracle_src(om, SGDState(om.parameters(), 0.9, 5e-4), bc, 5e-3, OH)
    model = Deeplabv2(cfg)
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd)
    model = model.cuda()
    out = src_step(model, FusedSGD(model, 1e-2, 0.9, 5e-4), StepState(C), {k: v.cuda() for k, v in bc.items()}, 5e-3)
    for k in ("pred_s1", "pred_s2"):
        err = (out[k].cpu() - ref[k]).abs().max() / ref[k].abs().max()
        assert err < 1e-3, (k, float(err))
    torch.testing.assert_close(out["loss_source"].cpu(), ref["loss_source"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(out["grad_norm"].cpu().reshape(()), ref["grad_norm"].reshape(()), rtol=2e-2, atol=1e-4)
    post = dict(om.named_parameters())
    for name in names:
        got, want = dict(model.named_parameters())[name].detach().cpu(), post[name].detach()
        upd, upd_ref = got - w0[name], want - w0[name]
        assert (upd - upd_ref).norm() / upd_ref.norm() < 6e-2, name


VARIANTS = {"single_aspp": dict(multi_layer=False, cascade=False, use_ppm=False),
            "single_ppm": dict(multi_layer=False, cascade=False, use_ppm=True),
            "cascade_aspp": dict(multi_layer=True, cascade=True, use_ppm=False)}


@pytest.mark.parametrize("tag", list(VARIANTS))
def test_deeplabv2_single_head_and_cascade_match_reference_golden(tag):
    """Deeplabv2's other branches (reference uemda/models/Encoder.py:93-102,111-116,129-143,156-165; multi_layer=False is the class's
    DEFAULT): the single head `cls_pred` and the cascade pair (layer5 on the layer3 output) on the HIP path -- state_dict keys,
    training-mode outputs, eval-mode probabilities and gradients of a seeded quadratic loss against the reference's own."""
    from oracle.weights import det_state_dict
    from uemda_amd.models.Encoder import Deeplabv2
    g = load_golden("model_variants")
    v = VARIANTS[tag]
    cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False), multi_layer=v["multi_layer"], cascade=v["cascade"],
               use_ppm=v["use_ppm"], ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)
    sd = det_state_dict("resnet50", C, v["use_ppm"], seed=2333, multi_layer=v["multi_layer"], cascade=v["cascade"])
    model = Deeplabv2(cfg)
    assert list(model.state_dict().keys()) == list(sd.keys())
    model.load_state_dict(sd)
    model = model.cuda()
    if v["use_ppm"]:
        model.cls_pred.conv_last[3].p = 0.0
    x = g["image"].cuda()
    model.eval()
    with torch.no_grad():
        prob = model(x)
    torch.testing.assert_close(prob[:, :, ::4, ::4].cpu(), g[f"{tag}:prob_sample"], rtol=1e-3, atol=1e-5)
    model.train()
    outs = model(x)
    assert len(outs) == (4 if v["cascade"] else 2)
    loss = 0.0
    for k, o in enumerate(outs):
        ref = g[f"{tag}:out{k}"]
        oc = o.detach().cpu().contiguous()
        got = oc if o.shape[1] == C else oc.reshape(-1)[:: max(1, oc.numel() // 4096)][:4096]
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err < 1e-3, (tag, k, err)
        r = torch.randn(o.shape, generator=torch.Generator().manual_seed(700 + k)).cuda()
        loss = loss + (o * r).sum() / o.numel() ** 0.5
    model.zero_grad()
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), g[f"{tag}:loss"], rtol=1e-3, atol=1e-3)
    named = dict(model.named_parameters())
    for k, ref in g.items():
        if k.startswith(f"{tag}:grad:"):
            gr = named[k.split(":", 2)[2]].grad.cpu().contiguous()
            got = gr if gr.numel() <= 8192 else gr.reshape(-1)[:: max(1, gr.numel() // 4096)][:4096]
            err = float((got - ref).norm() / (ref.norm() + 1e-12))
            assert err < 5e-2, (k, err)                 # the fixtures' per-tensor bound for deep gradients at B = 2 (noise floor 2-3 %)
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters())).float().cpu()
    torch.testing.assert_close(gn, g[f"{tag}:grad_norm"], rtol=2e-2, atol=1e-4)
