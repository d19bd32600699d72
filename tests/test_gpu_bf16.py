"""bf16-storage kernels (BASELINE config 5) against fp64 references built from the SAME bf16 values: the only error left
is fp32 accumulation order (~1e-6) plus the final rounding of the output to bf16 (half a unit in the last place = 2^-9)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil
    (2, 16, 16, 64, 64, 1, 1, 0, 1),
    (2, 16, 16, 64, 256, 1, 1, 0, 1),
    (2, 17, 13, 128, 128, 3, 1, 1, 1),       # ragged M
    (2, 16, 16, 128, 128, 3, 2, 1, 1),
    (2, 16, 16, 256, 512, 1, 2, 0, 1),
    (1, 12, 12, 512, 512, 3, 1, 2, 2),
    (3, 2, 2, 2048, 512, 1, 1, 0, 1),        # tiny M
    (4, 32, 32, 256, 128, 3, 1, 1, 1),       # several k-steps per tap, full tiles: statistics path
    (8, 64, 64, 64, 512, 1, 1, 0, 1),        # 1024 tiles of one k-step: persistent blocks by rule, two tiles each
    (8, 64, 64, 64, 384, 3, 1, 1, 1),        # 768 tiles on 512 persistent blocks (forced): one or two tiles per block, nine taps
]


@pytest.fixture(params=["rule", "persistent", "big", "big-persistent", "areg"])
def persist(request):
    """the conv kernel's persistent-block form by its dispatch rule (>= 1024 full tiles), and forced onto every full-tile launch;
    "big": the 256-row tiles (eight waves, round 5) forced onto every launch whose row count is a multiple of 256 -- by rule only
    launches with >= 256 such tiles take them, which none of these small cases has"""
    from uemda_amd import _lib
    lib = _lib.load()
    lib.uemdbg_conv_bf16_persist(1 if request.param.endswith("persistent") else -1)
    lib.uemdbg_conv_bf16_big(1 if request.param.startswith("big") else -1)
    import ctypes
    lib.uemdbg_conv_bf16_areg.argtypes = [ctypes.c_int]
    lib.uemdbg_conv_bf16_areg.restype = None
    lib.uemdbg_conv_bf16_areg(1 if request.param == "areg" else -1)      # "areg": the A-stationary pointwise block wherever it is legal
    yield request.param
    lib.uemdbg_conv_bf16_persist(-1)
    lib.uemdbg_conv_bf16_big(-1)
    lib.uemdbg_conv_bf16_areg(-1)


def _bf(t):
    return t.to(torch.bfloat16)


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def _close_bf16(got, ref64, what):
    got = got.double().cpu()
    tol = ref64.abs() * 2.0 ** -8 + ref64.abs().max() * 2e-6          # one bf16 ulp of the value + fp32 accumulation noise
    bad = (got - ref64).abs() > tol
    assert not bad.any(), (what, int(bad.sum()), float((got - ref64).abs().max()))


@pytest.mark.parametrize("case", CASES)
def test_conv_bf16_forward_and_data_gradient(case, persist):
    from uemda_amd import ops_bf16
    N, H, W, Cin, Cout, k, s, p, d = case
    g = torch.Generator().manual_seed(sum(case))
    x = _bf(torch.randn(N, Cin, H, W, generator=g))
    w = _bf(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5)
    x64, w64 = x.double().requires_grad_(True), w.double()
    y64 = F.conv2d(x64, w64, None, stride=s, padding=p, dilation=d)
    gy = _bf(torch.randn(y64.shape, generator=g))
    y64.backward(gy.double())
    M = y64.numel() // Cout
    want_stats = M % 128 == 0
    out = ops_bf16.conv2d(_nhwc(x), w.permute(0, 2, 3, 1).contiguous().cuda(), stride=s, pad=p, dil=d, want_stats=want_stats)
    y = out[0] if want_stats else out
    assert y.dtype == torch.bfloat16
    _close_bf16(y.permute(0, 3, 1, 2), y64.detach(), "forward")
    if want_stats:
        ts = out[1].double().cpu()                                     # (tiles, 2, Cout) laid out [2][Cout][tiles] by the kernel
        tiles = M // 128
        ts = out[1].reshape(-1).double().cpu().view(2, Cout, tiles)
        yv = y.double().cpu().reshape(M, Cout)
        torch.testing.assert_close(ts[0].sum(1), yv.sum(0), rtol=1e-5, atol=1e-3)
        torch.testing.assert_close(ts[1].sum(1), (yv * yv).sum(0), rtol=1e-5, atol=1e-3)
    # accumulate epilogue
    y2 = ops_bf16.conv2d(_nhwc(x), w.permute(0, 2, 3, 1).contiguous().cuda(), stride=s, pad=p, dil=d, out=y.clone(), accumulate=True)
    ref2 = y.double().cpu().permute(0, 3, 1, 2) + y64.detach()
    _close_bf16(y2.permute(0, 3, 1, 2), ref2, "accumulate")
    # data gradient
    wt = w.permute(1, 2, 3, 0).contiguous().cuda()                     # (Cin, KH, KW, Cout)
    dx = ops_bf16.conv2d_dgrad(_nhwc(gy), wt, (N, H, W, Cin), stride=s, pad=p, dil=d)
    _close_bf16(dx.permute(0, 3, 1, 2), x64.grad, "data gradient")


WGRAD_CASES = [
    # N, H, W, Cin, Cout, k, stride, dil
    (2, 16, 32, 128, 128, 3, 1, 1),
    (2, 16, 64, 64, 64, 3, 1, 1),
    (1, 20, 32, 128, 64, 3, 1, 2),
    (2, 24, 64, 64, 128, 3, 2, 1),
    (3, 7, 9, 256, 128, 1, 1, 1),            # ragged last step
    (2, 16, 16, 64, 256, 1, 1, 1),
    (2, 16, 64, 128, 256, 1, 2, 1),
    (8, 32, 32, 128, 128, 3, 1, 1),          # several pixel slices per tile (split-K), both stages
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wgrad_bf16_vs_fp64(case):
    from uemda_amd import ops_bf16
    N, H, W, Cin, Cout, k, s, d = case
    pad = d * (k - 1) // 2
    g = torch.Generator().manual_seed(sum(case) + 3)
    x = _bf(torch.randn(N, Cin, H, W, generator=g))
    w = torch.zeros(Cout, Cin, k, k, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double(), w, stride=s, padding=pad, dilation=d)
    gy = _bf(torch.randn(y.shape, generator=g))
    y.backward(gy.double())
    dw = torch.zeros(Cout, k, k, Cin, device="cuda")
    ops_bf16.conv2d_wgrad(_nhwc(x), _nhwc(gy), dw, stride=s, pad=pad, dil=d)
    ref = w.grad.permute(0, 2, 3, 1)
    err = float((dw.double().cpu() - ref).norm() / ref.norm())
    assert err < 2e-6, err                                             # exact bf16 products, fp32 accumulation order only
    ops_bf16.conv2d_wgrad(_nhwc(x), _nhwc(gy), dw, stride=s, pad=pad, dil=d)       # accumulates
    assert float((dw.double().cpu() - 2 * ref).norm() / ref.norm()) < 4e-6


TAIL_CASES = [
    # N, H, W, Cin, Cout, k, dil, acc ("none" | "bits" | "accumulate"), bn ("none" | "z" | "bits")
    (2, 16, 16, 64, 256, 1, 1, "none", "z"),           # conv3's data gradient + bn2's reduction
    (2, 16, 16, 128, 128, 3, 2, "none", "z"),          # dilated conv2's data gradient + bn1's reduction
    (2, 16, 16, 256, 64, 1, 1, "bits", "bits"),        # conv1's: identity gradient through the packed mask + previous bn3
    (2, 16, 16, 256, 64, 1, 1, "bits", "none"),
    (4, 8, 8, 512, 128, 1, 1, "accumulate", "bits"),   # after a stride-1 downsample's data gradient
    (1, 16, 24, 64, 64, 3, 1, "none", "z"),            # 64-wide column tiles
    (8, 64, 64, 256, 64, 1, 1, "bits", "bits"),        # layer1's residual tail at 1024 tiles: persistent blocks by rule
    (2, 16, 16, 512, 128, 1, 1, "bits", "bits"),       # layer2's / layer3's tails: 128 / 256 reduction channels (the A-stationary block's
    (4, 16, 16, 1024, 256, 1, 1, "bits", "bits"),      # tail forms when it is forced on: one panel, two panels)
    (2, 16, 16, 512, 128, 1, 1, "bits", "none"),
    (2, 16, 16, 1024, 256, 1, 1, "none", "bits"),
]


@pytest.mark.parametrize("case", TAIL_CASES)
def test_dgrad_tail_bf16_epilogue(case, persist):
    """uem_conv2d_dgrad_tail_bf16: the data gradient with the residual tail and the BatchNorm-backward reduction in its epilogue
    against fp64 built from the same bf16 values; the partial sums against sums over the dx the kernel stored."""
    from uemda_amd import ops_bf16
    N, H, W, Cin, Cout, k, d, acc, bn = case
    g = torch.Generator().manual_seed(sum(case[:7]) + len(acc) + 7 * len(bn))
    pad = d * (k - 1) // 2
    w = _bf(torch.randn(Cout, Cin, k, k, generator=g) / (Cout * k * k) ** 0.5)
    x64 = torch.zeros(N, Cin, H, W, dtype=torch.float64, requires_grad=True)
    y64 = F.conv2d(x64, w.double(), None, padding=pad, dilation=d)
    gy = _bf(torch.randn(y64.shape, generator=g))
    y64.backward(gy.double())
    ref = x64.grad.clone()                                              # (N, Cin, H, W)
    M = N * H * W
    kw = {}
    if acc != "none":
        prev = _bf(torch.randn(N, Cin, H, W, generator=g))
        if acc == "bits":
            keep = torch.rand(N, H, W, Cin, generator=g) > 0.4
            kw.update(acc_src=_nhwc(prev), acc_bits=_pack_bits(keep).cuda())
            ref = ref + prev.double() * keep.permute(0, 3, 1, 2)
        else:
            kw.update(out=_nhwc(prev), accumulate=True)
            ref = ref + prev.double()
    if bn != "none":
        z = _bf(torch.randn(N, H, W, Cin, generator=g))
        vec = torch.stack([torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g) * 0.3, torch.randn(Cin, generator=g) * 0.1,
                           torch.rand(Cin, generator=g) + 0.5])       # scale, shift, mean, invstd
        kw.update(bn_z=z.cuda(), bn_vec=vec.cuda())
        if bn == "bits":
            on = torch.rand(N, H, W, Cin, generator=g) > 0.5
            kw.update(bn_bits=_pack_bits(on).cuda())
        else:
            on = (z.float() * vec[0] + vec[1]) > 0
    wt = w.permute(1, 2, 3, 0).contiguous().cuda()
    dx, tp = ops_bf16.conv2d_dgrad_tail(_nhwc(gy), wt, (N, H, W, Cin), pad=pad, dil=d, **kw)
    _close_bf16(dx.permute(0, 3, 1, 2), ref, "data gradient + tail")
    if bn == "none":
        assert tp is None
        return
    tiles = M // 128
    tp = tp.reshape(-1).double().cpu().view(2, Cin, tiles)
    dp = dx.double().cpu() * on
    xhat = (z.double() - vec[2].double()) * vec[3].double()
    torch.testing.assert_close(tp[0].sum(1), dp.reshape(M, Cin).sum(0), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(tp[1].sum(1), (dp * xhat).reshape(M, Cin).sum(0), rtol=1e-4, atol=1e-3)
    # per tile, not only in total
    torch.testing.assert_close(tp[0].t(), dp.reshape(tiles, 128, Cin).sum(1), rtol=1e-4, atol=1e-3)


def _pack_bits(mask):
    """bool (..., C) -> int32 words, bit i of word j = element 32*j + i of the flattened tensor (the layout of the packed ReLU masks)."""
    m = mask.reshape(-1, 32).to(torch.int64)
    words = (m << torch.arange(32, dtype=torch.int64)).sum(1)
    return (words & 0xFFFFFFFF).to(torch.int64).where(words < 2 ** 31, words - 2 ** 32).to(torch.int32)


def _block_stack():
    import torch.nn as nn
    from uemda_amd.resnet import Bottleneck
    torch.manual_seed(77)

    def ds(cin, cout, s):
        return nn.Sequential(nn.Conv2d(cin, cout, 1, s, bias=False), nn.BatchNorm2d(cout))
    net = nn.Sequential(Bottleneck(64, 64, downsample=ds(64, 256, 1)), Bottleneck(256, 64), Bottleneck(256, 64),
                        Bottleneck(256, 128, stride=2, downsample=ds(256, 512, 2)), Bottleneck(512, 128),
                        Bottleneck(512, 128, dilation=2, downsample=ds(512, 512, 1)), Bottleneck(512, 128, dilation=2))
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    return net


def _run_stack(sd, x, gy, storage, fuse, lo=0, hi=7):
    from uemda_amd import ops
    from uemda_amd.models.blocks_bf16 import CastFn
    from uemda_amd.models.Encoder import Deeplabv2

    class Holder(torch.nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner
    net = _block_stack()
    net.load_state_dict(sd)
    h = Holder(net).cuda()
    Deeplabv2._flatten_parameters(h)
    net.train()
    xi = x.clone().requires_grad_(True)
    old = ops.FUSE_BN_BACKWARD
    ops.FUSE_BN_BACKWARD = fuse
    try:
        t = CastFn.apply(xi, True) if storage == "bf16" else xi
        for blk in list(net)[lo:hi]:
            t = blk(t)
        y = CastFn.apply(t, False) if storage == "bf16" else t
        y.backward(gy)
        torch.cuda.synchronize()
    finally:
        ops.FUSE_BN_BACKWARD = old
    return y.detach().double(), xi.grad.double(), h._grad_arena[:h._n_params].double().clone(), \
        {n: (p_._uem_off, p_.numel()) for n, p_ in net.named_parameters()}


class _Rnd(torch.autograd.Function):
    """bf16 storage of an activation: the value is rounded on the way up, its gradient on the way down."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().to(g.dtype)


class _RndW(torch.autograd.Function):
    """bf16 copy of an fp32 master weight: rounded value, full-precision gradient."""

    @staticmethod
    def forward(ctx, w):
        return w.bfloat16().to(w.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def _emulated_stack(sd, x, gy, lo=0, hi=7):
    """The same seven blocks in float64 torch on the CPU with a bf16 rounding wherever the storage path keeps a bf16 tensor
    (conv outputs z, relu(bn(z)), block outputs, weight copies, and every activation gradient): the arithmetic the bf16
    path implements, with none of its kernels."""
    net = _block_stack().double()
    net.load_state_dict({k: v.double() for k, v in sd.items()})
    net.train()
    R, RW = _Rnd.apply, _RndW.apply

    def cbn(t, conv, bn):
        z = R(F.conv2d(t, RW(conv.weight), None, conv.stride, conv.padding, conv.dilation))
        return z, lambda z_: F.batch_norm(z_, None, None, bn.weight, bn.bias, True, 0.0, bn.eps)
    xi = x.detach().cpu().double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    t = R(xi)
    for blk in list(net)[lo:hi]:
        z1, bn1 = cbn(t, blk.conv1, blk.bn1)
        a1 = R(F.relu(bn1(z1)))
        z2, bn2 = cbn(a1, blk.conv2, blk.bn2)
        a2 = R(F.relu(bn2(z2)))
        z3, bn3 = cbn(a2, blk.conv3, blk.bn3)
        if blk.downsample is not None:
            zd, bnd = cbn(t, blk.downsample[0], blk.downsample[1])
            res = bnd(zd)
        else:
            res = t
        t = R(F.relu(bn3(z3) + res))
    t.backward(gy.detach().cpu().double().permute(0, 3, 1, 2))
    grads = {n: p_.grad for n, p_ in net.named_parameters() if p_.grad is not None}
    return t.detach().permute(0, 2, 3, 1), xi.grad.permute(0, 2, 3, 1), grads


_CIN = [64, 256, 256, 256, 512, 512, 512]
STACK_SLICES = [(0, 1), (1, 2), (3, 4), (4, 5), (5, 6), (6, 7), (0, 3), (2, 5), (4, 7), (0, 7)]


@pytest.mark.parametrize("lo,hi", STACK_SLICES)
def test_bf16_bottleneck_backward_vs_emulation_and_unfused(lo, hi):
    """Bottleneck blocks (every block shape of the encoder: stride-1 / stride-2 / dilated downsample, identity; alone and in
    chains, where a block's bn3 reduction rides in the NEXT block's tail epilogue) forward and backward in bf16 storage against a
    float64 torch emulation of the SAME arithmetic (a bf16 rounding wherever the path stores a bf16 tensor, _emulated_stack).
    What is left between the two: fp32 accumulation order, which flips a rounding here and there (one flip = 2^-8 relative
    on that element, and downstream every ReLU whose sign it changes), and the downsample blocks adding their two data
    gradients with one more rounding.  The flips cascade: training-mode BatchNorm blocks with branches as large as their
    trunk amplify a perturbation by ~2 per block forward and, through the ReLU masks it moves, far more backward -- the same
    float64 emulation sits 4e-2 (y) / 4e-1 (dx) from exact fp32 blocks after seven blocks, so the bounds below widen with the
    chain length and the seven-block chain is printed, not asserted.  Also: the fused backward (BatchNorm reductions and the
    residual tail in the data-gradient epilogues, gradient buffers reused in place) against the plain
    one-pass-per-operation backward of the same path on the same forward (a linear map of gy: no cascade)."""
    sd = {k: v.clone() for k, v in _block_stack().state_dict().items()}
    g = torch.Generator().manual_seed(5 + 10 * lo + hi)
    hw_in, hw_out = (32 if lo <= 3 else 16), (32 if hi <= 3 else 16)
    x = torch.randn(2, hw_in, hw_in, _CIN[lo], generator=g).bfloat16().float().cuda()
    gy = torch.randn(2, hw_out, hw_out, 256 if hi <= 3 else 512, generator=g).bfloat16().float().cuda()
    y16, dx16, g16, offs = _run_stack(sd, x, gy, "bf16", True, lo, hi)
    y16u, dx16u, g16u, _ = _run_stack(sd, x, gy, "bf16", False, lo, hi)
    ye, dxe, ge = _emulated_stack(sd, x, gy, lo, hi)

    def rel(a, b):
        return float((a.cpu() - b.cpu()).norm() / b.cpu().norm())

    def layer_grad(arena, n):
        o, c = offs[n]
        t = arena[o:o + c].cpu()
        if ge[n].dim() == 4:                                           # the arena keeps conv weights OHWI
            co, ci, kh, kw = ge[n].shape
            t = t.view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return t.reshape(ge[n].shape)
    per_layer = {n: rel(layer_grad(g16, n), ge[n]) for n in ge}
    worst = max(per_layer.items(), key=lambda kv: kv[1])
    ry, rdx, rfu = rel(y16, ye), rel(dx16, dxe), max(rel(dx16, dx16u), rel(g16, g16u))
    print(f"blocks [{lo}:{hi}) bf16 path vs float64 emulation of bf16 storage: y {ry:.2e}, dx {rdx:.2e}, worst parameter gradient {worst}; "
          f"fused vs plain backward {rfu:.2e}")
    assert torch.equal(y16, y16u)                                      # the forward does not depend on the backward's fusion
    has_ds = any(b in (0, 3, 5) for b in range(lo, hi))              # downsample blocks: two data gradients, rounded in another order
    assert rfu < (FUSED_TOL if has_ds else FUSED_TOL_IDENTITY), rfu
    n = hi - lo
    if n in CHAIN_TOL:
        ty, tdx, tdw = CHAIN_TOL[n]
        assert ry < ty and rdx < tdx and worst[1] < tdw, (ry, rdx, worst)


# measured (profiles/README.md, round 2): one block y 2-7e-4, dx 2e-3..1.2e-2, parameter gradients <= 1.5e-2; three blocks y 4-7e-3,
# dx 6e-2..1.1e-1, parameter gradients <= 1.4e-1; fused vs plain 4-6e-5 on identity blocks, 3e-3..1e-2 with downsample blocks.
# A wrong mask, a missing term or a mis-indexed partial sum is an O(1) error at the one-block level.
FUSED_TOL, FUSED_TOL_IDENTITY = 2e-2, 1e-3
CHAIN_TOL = {1: (2e-3, 3e-2, 4e-2), 3: (2e-2, 2.5e-1, 3e-1)}      # blocks in the chain -> relative L2 bounds on y, dx, parameter gradients


def _damped_sd(rtype, damp=0.2, use_ppm=False):
    from oracle.weights import det_state_dict
    sd = det_state_dict(rtype, 6, use_ppm, seed=2333)
    for k in sd:
        if k.endswith("bn3.weight"):
            sd[k] = sd[k] * damp
    return sd


@pytest.mark.parametrize("rtype,size,use_ppm,B", [("resnet50", 512, False, 1), ("resnet101", 1024, False, 1), ("resnet50", 256, True, 2)])
def test_bf16_storage_ssl_step_vs_oracle(rtype, size, use_ppm, B):
    """One train_ssl_uem step with bf16 STORAGE (activations, activation gradients and weight copies in bf16 between the
    max-pool and layer4; fp32 accumulation, statistics and master weights) against the oracle, B = 1 + 1 tiles.  Residual
    branches damped (gamma3 x 0.2) as in tests/test_gpu_config5.py -- at the default initialisation the network is chaotic
    and no reduced precision can be compared through it -- and the network's conditioning measured by an fp32-storage run
    of the same step.  Per layer the storage path rounds three tensors (conv input, conv output z, relu(bn(z))) where the
    bf16-operand mode rounds one, so its logits are allowed sqrt(3) x that mode's bound: 1.5 * sqrt(3) * (2.3e-3 / 3.3e-7) x
    the fp32 run's own error, and at most 0.2; hard pseudo-labels >= 99 % identical, losses within 1 %, gradient norm
    within 10 %."""
    from oracle import synth
    from oracle.model import OracleDeeplabv2
    from oracle.step import HYPER as OH, SGDState, ssl_step as oracle_ssl
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.models.Encoder import Deeplabv2
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import HYPER, StepState, ssl_step
    C = 6
    sd = _damped_sd(rtype, use_ppm=use_ppm)
    bc = synth.make_batch(B=B, H=size, W=size, C=C, k=2048, seed=31)
    om = OracleDeeplabv2({k: v.clone() for k, v in sd.items()}, rtype, C, use_ppm)
    ref = oracle_ssl(om, SGDState(om.parameters(), 0.9, 5e-4), bc["prototypes"], bc, 2e-3, OH, dropout=False)
    cfg = dict(backbone=dict(resnet_type=rtype, output_stride=16, pretrained=False), multi_layer=True, cascade=False,
               use_ppm=use_ppm, ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)
    res, grads = {}, {}
    for storage in ("fp32", "bf16"):
        model = Deeplabv2(cfg)
        model.load_state_dict(sd)
        if use_ppm:                                                    # Dropout2d off, as in the oracle run
            model.layer5.conv_last[3].p = 0.0
            model.layer6.conv_last[3].p = 0.0
        model = model.cuda().set_storage(storage)
        b = {k: v.cuda() for k, v in bc.items()}
        al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
        al.prototypes = b["prototypes"].clone()
        out = ssl_step(model, al, FusedSGD(model, 1e-2, 0.9, 5e-4), StepState(C), b, 2e-3, sup_ignore_id=(size // 16) ** 2)
        torch.cuda.synchronize()
        rel = max(float((out[k].cpu() - ref[k]).norm() / ref[k].norm()) for k in ("pred_s1", "pred_s2", "pred_t1", "pred_t2"))
        agree = (out["label_t_hard"].cpu() == ref["label_t_hard"]).float().mean().item()
        ls = abs(float(out["loss_source"]) / float(ref["loss_source"]) - 1.0)
        lt = abs(float(out["loss_target"]) / float(ref["loss_target"]) - 1.0)
        gn = abs(float(out["grad_norm"]) / float(ref["grad_norm"]) - 1.0)
        res[storage] = (rel, agree, ls, lt, gn)
        grads[storage] = model._grad_arena[:model._n_params].clone()
        print(f"{rtype} {size}x{size} storage={storage}: logit relative L2 {rel:.3e}, hard-label agreement {agree:.5f}, "
              f"losses off by {ls:.2e} / {lt:.2e}, grad norm off by {gn:.2e}")
        del model
    rel32 = res["fp32"][0]
    rel, agree, ls, lt, gn = res["bf16"]
    assert rel32 < 1e-4
    assert rel < 0.2 and rel < 1.5 * 3 ** 0.5 * (2.3e-3 / 3.3e-7) * rel32, (rel, rel32)
    assert agree >= 0.99, agree
    assert ls < 1e-2 and lt < 1e-2, (ls, lt)
    assert gn < 0.1, gn
    # whole-step gradients, bf16 against fp32 storage: a coarse check only (through ~50 / ~100 layers the forward already
    # differs by `rel` and the ReLU masks with it; measured cosine 0.94 on ResNet-50, 0.82 on ResNet-101).  The backward kernels themselves are
    # pinned on short chains by test_bf16_bottleneck_backward_vs_emulation_and_unfused.
    g32, g16 = grads["fp32"].double(), grads["bf16"].double()
    cos = float((g32 * g16).sum() / (g32.norm() * g16.norm()))
    print(f"{rtype} {size}x{size}: gradient arena bf16 vs fp32 storage: cosine {cos:.5f}, relative L2 {float((g32 - g16).norm() / g32.norm()):.3e}")
    # printed, not asserted (round 2 asserted cos > 0.7, which a 10 % systematic error would have passed): at B = 1 + 1 through
    # 50-100 layers this number is dominated by moved ReLU masks.  What can fail on a systematic error in the bf16 backward:
    # test_bf16_seven_block_chain_well_conditioned_vs_emulation (the gain of every gradient tensor) and
    # test_bf16_storage_training_trajectory_tracks_fp32_storage (30 steps against fp32 storage).


def test_bf16_weight_copies_follow_torch_optim_sgd_and_load_state_dict():
    """the bf16 copy of the parameter arena was keyed on the arena's version, which in-place writes THROUGH a
    parameter (torch.optim.SGD, load_state_dict's copy_) never move -- after the first bf16 forward every later forward ran on the
    initial weights while the data gradient used the current ones.  A bf16-storage step with torch.optim.SGD must (1) change the
    next forward, (2) give exactly the forward of a fresh model loaded with the updated weights; and loading the initial
    state_dict back must restore the initial forward bit for bit."""
    from oracle import synth
    from uemda_amd.models.Encoder import Deeplabv2
    from uemda_amd.utils.tools import loss_calc
    from uemda_amd.gast.balance import CrossEntropy
    C = 6
    sd0 = _damped_sd("resnet50")
    cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False), multi_layer=True, cascade=False,
               use_ppm=False, ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)

    def build(sd):
        m = Deeplabv2(cfg)
        m.load_state_dict(sd)
        return m.cuda().set_storage("bf16").train()
    model = build(sd0)
    b = {k: v.cuda() for k, v in synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=5).items()}
    opt = torch.optim.SGD(model.parameters(), lr=1e-2, momentum=0.9, weight_decay=5e-4)
    p1, p2, _ = model(b["images_s"])
    first = p1.detach().clone()
    loss = loss_calc([p1, p2], b["label_s"], loss_fn=CrossEntropy(ignore_label=-1), multi=True)
    opt.zero_grad()
    loss.backward()
    opt.step()                                                       # in place through the parameters: arena._version stays put
    with torch.no_grad():
        after = model(b["images_s"])[0]
    assert (after - first).abs().max() > 1e-4, "the forward after the step still ran on the initial bf16 weight copy"
    fresh = build({k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
    with torch.no_grad():
        ref = fresh(b["images_s"])[0]
    assert torch.equal(after, ref)
    model.load_state_dict(sd0)                                       # mid-run reload
    with torch.no_grad():
        back = model(b["images_s"])[0]
    assert torch.equal(back, first)


def test_bf16_seven_block_chain_well_conditioned_vs_emulation():
    """the whole seven-block chain (every block shape of the encoder) ASSERTED, not printed.  At the default
    initialisation a chain of training-mode BatchNorm blocks whose branches are as large as their trunk amplifies one flipped bf16
    rounding by ~2 per block forward (more backward, through the ReLU masks it moves), so seven blocks sit at 3e-1 from their own
    float64 emulation and nothing can be asserted.  With the residual branches damped the way trained networks are (gamma of
    every block's last BatchNorm x 0.2), B = 8 and 32x32 maps the forward is well conditioned (y 8e-3).  The backward keeps an unbiased noise of ~1.1e-1 from ReLU masks
    moved by flipped roundings (see BF16_CHAIN7_TOL), so besides relative-L2 bounds at ~2x measured the test asserts the least-squares
    GAIN of dx and of EVERY parameter gradient against the float64 emulation of the same arithmetic: within 8 % of 1 (dx: 3 %).  A
    10 % systematic error anywhere in the bf16 backward -- a wrong factor, a missing term, a mis-indexed partial sum in the fused
    epilogues -- moves a gain to 0.9 or worse and fails."""
    sd = {k: v.clone() for k, v in _block_stack().state_dict().items()}
    for k in sd:
        if k.endswith("bn3.weight"):
            sd[k] = sd[k] * 0.2
    g = torch.Generator().manual_seed(123)
    x = torch.randn(8, 32, 32, 64, generator=g).bfloat16().float().cuda()
    gy = torch.randn(8, 16, 16, 512, generator=g).bfloat16().float().cuda()
    y16, dx16, g16, offs = _run_stack(sd, x, gy, "bf16", True, 0, 7)
    y16u, dx16u, g16u, _ = _run_stack(sd, x, gy, "bf16", False, 0, 7)
    ye, dxe, ge = _emulated_stack(sd, x, gy, 0, 7)

    def rel(a, b):
        return float((a.cpu() - b.cpu()).norm() / b.cpu().norm())

    def layer_grad(arena, n):
        o, c = offs[n]
        t = arena[o:o + c].cpu()
        if ge[n].dim() == 4:
            co, ci, kh, kw = ge[n].shape
            t = t.view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return t.reshape(ge[n].shape)
    def gain(a, b):
        """least-squares scale of a against b: <a, b> / <b, b>.  Rounding flips and the ReLU masks they move are unbiased noise and
        average out of it; a systematic error (a wrong factor, a missing term) does not."""
        a, b = a.cpu().double().reshape(-1), b.cpu().double().reshape(-1)
        return float((a * b).sum() / (b * b).sum())
    per_layer = {n: rel(layer_grad(g16, n), ge[n]) for n in ge}
    gains = {n: gain(layer_grad(g16, n), ge[n]) for n in ge}
    worst = max(per_layer.items(), key=lambda kv: kv[1])
    worst_gain = max(gains.items(), key=lambda kv: abs(kv[1] - 1.0))
    ry, rdx, rfu = rel(y16, ye), rel(dx16, dxe), max(rel(dx16, dx16u), rel(g16, g16u))
    gdx = gain(dx16, dxe)
    srt = sorted(per_layer.values())
    print(f"damped seven-block chain, B=8: y {ry:.2e}, dx {rdx:.2e} (gain {gdx:.4f}), parameter gradients median {srt[len(srt) // 2]:.2e} "
          f"worst {worst}, worst gain {worst_gain}; fused vs plain backward {rfu:.2e}")
    ty, tdx, tdw, tfu, tg = BF16_CHAIN7_TOL
    assert ry < ty and rdx < tdx, (ry, rdx)
    assert worst[1] < tdw, worst
    assert rfu < tfu, rfu
    assert abs(gdx - 1.0) < 0.03 and abs(worst_gain[1] - 1.0) < tg, (gdx, worst_gain)


# measured round 3 (gpurun_out / profiles/README.md): see the numbers printed by the test; bounds <= 3x measured and <= 5e-2
# Why not 5e-2 on dx: each conv output is rounded to bf16, and where fp32 accumulation order differs from the float64 emulation one
# rounding in ~200 flips (2^-8 relative on that element).  Forward that is 3e-4 per layer (y: 7.9e-3 after 21 conv+BatchNorm layers).
# Backward every pre-activation within that distance of zero flips its ReLU mask -- an O(1) change of that element's gradient, on a
# fraction ~3e-4 of the elements: 1.7e-2 per layer, unbiased, growing as sqrt(layers) -- 1.1e-1 after seven blocks, damped or not.
# It is NOISE: the least-squares gain of every gradient tensor against the emulation stays within 5 % of 1 (dx: 0.7 %), and that is
# what a systematic error (wrong factor, missing term, mis-indexed partial sum: gain 0.9 or worse on some tensor) cannot do.
# measured: y 7.9e-3, dx 1.12e-1 (gain 0.9934), parameter gradients median 1.13e-1, worst 1.62e-1, worst gain 0.9466, fused vs plain 6.9e-3
BF16_CHAIN7_TOL = (2.4e-2, 2.5e-1, 3.5e-1, 2e-2, 0.08)       # y, dx, worst parameter gradient, fused vs plain, |gain - 1| (dx: 0.03)


def test_bf16_storage_training_trajectory_tracks_fp32_storage():
    """30 train_ssl_uem steps of R50-ASPP on 256x256 tiles (B = 4 + 4), bf16 storage against fp32 storage from
    identical weights, inputs and seeds (damped residual branches, as every bf16 comparison here): at EVERY step the two losses
    within 2 % and the hard pseudo-labels >= 99 % identical; after 30 steps the weights of the two runs apart by a bounded fraction
    of the distance they travelled.  Replaces the whole-arena gradient cosine > 0.7 of round 2, which a 10 % systematic error
    in the bf16 backward would have passed."""
    from oracle import synth
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.models.Encoder import Deeplabv2
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import HYPER, StepState, ssl_step
    C, B, S, STEPS = 6, 4, 256, 30
    sd = _damped_sd("resnet50")
    cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False), multi_layer=True, cascade=False,
               use_ppm=False, ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)
    bc = synth.make_batch(B=B, H=S, W=S, C=C, k=2048, seed=77)
    hist, final = {}, {}
    for storage in ("fp32", "bf16"):
        model = Deeplabv2(cfg)
        model.load_state_dict(sd)
        model = model.cuda().set_storage(storage)
        b = {k: v.cuda() for k, v in bc.items()}
        al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
        al.prototypes = b["prototypes"].clone()
        opt, st = FusedSGD(model, 1e-2, 0.9, 5e-4), StepState(C)
        rows = []
        for i in range(STEPS):
            out = ssl_step(model, al, opt, st, b, 5e-3, sup_ignore_id=(S // 16) ** 2)
            rows.append((float(out["loss_source"]), float(out["loss_target"]), out["label_t_hard"].cpu(), float(out["grad_norm"])))
        hist[storage] = rows
        arena, _, n = model.flat_parameters()
        final[storage] = arena[:n].double().cpu().clone()
        del model, opt
    model = Deeplabv2(cfg)
    model.load_state_dict(sd)
    model = model.cuda()
    arena, _, n = model.flat_parameters()
    w0 = arena[:n].double().cpu().clone()
    del model
    worst_ls = worst_lt = 0.0
    worst_agree = 1.0
    for i in range(STEPS):
        a, b_ = hist["fp32"][i], hist["bf16"][i]
        worst_ls = max(worst_ls, abs(b_[0] / a[0] - 1.0))
        worst_lt = max(worst_lt, abs(b_[1] / a[1] - 1.0))
        worst_agree = min(worst_agree, (a[2] == b_[2]).float().mean().item())
    travelled = float((final["fp32"] - w0).norm()) if w0.numel() == final["fp32"].numel() else float("nan")
    apart = float((final["bf16"] - final["fp32"]).norm())
    rel_w = apart / float(final["fp32"].norm())
    print("step: loss_s fp32/bf16, loss_t fp32/bf16, agreement")
    for i in range(STEPS):
        a, b_ = hist["fp32"][i], hist["bf16"][i]
        print(f"  {i:2d}: {a[0]:.4f} {b_[0]:.4f}   {a[1]:.4f} {b_[1]:.4f}   {(a[2] == b_[2]).float().mean().item():.4f}")
    print(f"30-step trajectory bf16 vs fp32 storage: worst loss_source deviation {worst_ls:.3e}, loss_target {worst_lt:.3e}, worst hard-label "
          f"agreement {worst_agree:.5f}; final weights apart {apart:.3e} = {rel_w:.3e} of |w|, distance travelled {travelled:.3e}; "
          f"losses first/last fp32 {hist['fp32'][0][:2]} {hist['fp32'][-1][:2]} bf16 {hist['bf16'][-1][:2]}")
    worst_total = max(abs((b_[0] + b_[1]) / (a[0] + a[1]) - 1.0) for a, b_ in zip(hist["fp32"], hist["bf16"]))
    worst_lt_of_total = max(abs(b_[1] - a[1]) / (a[0] + a[1]) for a, b_ in zip(hist["fp32"], hist["bf16"]))
    print(f"worst total-loss deviation {worst_total:.3e}; worst loss_target deviation as a fraction of the total loss {worst_lt_of_total:.3e}; "
          f"weights apart / distance travelled {apart / travelled:.3e}")
    assert hist["fp32"][-1][0] < 0.6 * hist["fp32"][0][0] and hist["fp32"][-1][1] < 0.1 * hist["fp32"][0][1]      # it trains
    # loss curves within 2 %: the total and the source loss relatively; the target (UVEM) loss falls from 1.73 to 0.02 over the run,
    # where 2 % of ITSELF is 4e-4 -- it is held to 1 % of the step's total loss instead (measured 0.34 %; total 0.8 %, source 0.8 %)
    assert worst_total < 2e-2 and worst_ls < 2e-2 and worst_lt_of_total < 1e-2, (worst_total, worst_ls, worst_lt_of_total)
    assert worst_agree >= 0.99, worst_agree                                # measured 0.9921 at the worst step
    assert rel_w < BF16_TRAJ_WEIGHT_TOL[0] and apart / travelled < BF16_TRAJ_WEIGHT_TOL[1], (rel_w, apart / travelled)


# |w_bf16 - w_fp32| / |w_fp32| after 30 steps (measured 3.8e-3), and as a fraction of the distance the fp32 run travelled from the
# initial weights
BF16_TRAJ_WEIGHT_TOL = (1.2e-2, 0.9)              # measured 3.8e-3 and 0.57


@pytest.mark.parametrize("B,H,W", [(2, 256, 256), (1, 160, 96)])
def test_bf16_storage_eval_mode_inference_tracks_fp32(B, H, W):
    """Round 3 (VERDICT r2 missing #5): eval-mode inference under bf16 storage -- the offline pseudo-label pass
    (gener_target_pseudo / tta_predict) and evaluation.  The averaged probability map (Encoder.py:156-165) of the bf16-storage
    model against the fp32-storage one on the damped network: relative L2 <= 5e-2, argmax >= 99 % identical; a tile whose layer
    outputs are not multiples of 128 pixels takes the ragged tiles; an eval-mode forward that records a graph stays on the fp32 kernels."""
    from uemda_amd.models.Encoder import Deeplabv2
    C = 6
    cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False), multi_layer=True, cascade=False,
               use_ppm=False, ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)
    sd = _damped_sd("resnet50")
    x = torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(B + H)).cuda()
    probs = {}
    for storage in ("fp32", "bf16"):
        model = Deeplabv2(cfg)
        model.load_state_dict(sd)
        model = model.cuda().set_storage(storage).eval()
        with torch.no_grad():
            probs[storage] = model(x).float()
        if storage == "bf16":
            # an eval-mode forward that records a graph (frozen-statistics training) is not an inference pass: it stays on the
            # exact fp32 kernels
            g = model(x.clone().requires_grad_(True))
            torch.testing.assert_close(g.detach().float(), probs["fp32"], rtol=1e-4, atol=1e-6)
        nbt = int(model.state_dict()["encoder.resnet.layer1.0.bn1.num_batches_tracked"])
        assert nbt == 0                                               # eval mode never counts a batch
        del model
    a, b = probs["bf16"], probs["fp32"]
    assert a.shape == (B, C, H, W) and ((a.sum(1) - 1).abs() < 1e-4).all()
    rel = float((a - b).norm() / b.norm())
    agree = (a.argmax(1) == b.argmax(1)).float().mean().item()
    # randomly initialised heads give near-uniform probabilities: the argmax of a pixel whose two best classes are 1e-3 apart is
    # decided by rounding.  Pixels with a real margin (top-1 minus top-2 above 0.02 in fp32) must agree.
    top2 = b.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 0.02
    agree_clear = (a.argmax(1) == b.argmax(1))[clear].float().mean().item()
    print(f"eval-mode bf16 vs fp32 storage {B}x{H}x{W}: probability map relative L2 {rel:.3e}, argmax agreement {agree:.5f} "
          f"({agree_clear:.5f} on the {clear.float().mean().item():.2f} of the pixels with a margin above 0.02)")
    assert rel < 5e-2 and agree >= 0.95 and agree_clear >= 0.999, (rel, agree, agree_clear)


BF16_OPTIONS = {"frozen": dict(freeze_at=2, batchnorm_trainable=False), "cp": dict(with_cp=(True, True, False, True))}


@pytest.mark.parametrize("tag", ["frozen", "cp"])
def test_bf16_storage_encoder_options(tag):
    """VERDICT r2 missing #5: the ResNetEncoder modes (reference uemda/resnet.py:112-130,146-165,183-190) under bf16 storage, one
    train_ssl_uem step on the damped network.
      frozen: freeze_at=2 + batchnorm_trainable=False -- eval-mode BatchNorm inside the bf16 training graph (backward through the
              running statistics: dx = dp * scale): against the SAME options in fp32 storage -- logits within 5e-2, hard labels
              >= 99 % identical, losses within 1 %; frozen tensors keep no gradient and do not move, running statistics untouched;
      cp:     residual layers under torch.utils.checkpoint: the re-run forward is the same arithmetic, so logits, losses and labels
              equal the un-checkpointed bf16 step bit for bit and every weight after the update to atomic-order noise, and
              num_batches_tracked counts the checkpointed layers' BatchNorms twice per forward."""
    from oracle import synth
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.models.Encoder import Deeplabv2
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import HYPER, StepState, ssl_step
    C = 6
    sd = _damped_sd("resnet50")
    bc = synth.make_batch(B=2, H=256, W=256, C=C, k=2048, seed=7)

    def run(storage, **opts):
        cfg = dict(backbone=dict(resnet_type="resnet50", output_stride=16, pretrained=False, **opts), multi_layer=True, cascade=False,
                   use_ppm=False, ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)
        model = Deeplabv2(cfg)
        model.load_state_dict(sd)
        model = model.cuda().set_storage(storage).train()
        b = {k: v.cuda() for k, v in bc.items()}
        al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
        al.prototypes = b["prototypes"].clone()
        w0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        out = ssl_step(model, al, FusedSGD(model, 1e-2, 0.9, 5e-4), StepState(C), b, 2e-3, sup_ignore_id=256)
        torch.cuda.synchronize()
        return model, w0, {k: (v.detach().float().clone() if torch.is_tensor(v) else v) for k, v in out.items()}

    if tag == "frozen":
        m16, w0, o16 = run("bf16", **BF16_OPTIONS[tag])
        m32, _, o32 = run("fp32", **BF16_OPTIONS[tag])
        rel = max(float((o16[k] - o32[k]).norm() / o32[k].norm()) for k in ("pred_s1", "pred_s2", "pred_t1", "pred_t2"))
        agree = (o16["label_t_hard"] == o32["label_t_hard"]).float().mean().item()
        ls = abs(float(o16["loss_source"]) / float(o32["loss_source"]) - 1.0)
        lt = abs(float(o16["loss_target"]) / float(o32["loss_target"]) - 1.0)
        gn = abs(float(o16["grad_norm"]) / float(o32["grad_norm"]) - 1.0)
        print(f"bf16 storage, frozen BatchNorm: logits {rel:.3e}, labels {agree:.5f}, losses {ls:.2e} / {lt:.2e}, grad norm {gn:.2e}")
        assert rel < 5e-2 and agree >= 0.99 and ls < 1e-2 and lt < 1e-2 and gn < 0.1, (rel, agree, ls, lt, gn)
        s16, s32 = m16.state_dict(), m32.state_dict()
        moved = 0
        for k, v in s16.items():
            frozen = k.startswith("encoder.resnet.") and ("bn" in k or "downsample.1" in k or ".layer1." in k or k.split(".")[2] in ("conv1", "bn1"))
            if frozen:
                assert torch.equal(v, w0[k]), k                         # statistics, gamma / beta, stem and layer1: not a bit moves
            elif v.is_floating_point():
                moved += int(not torch.equal(v, w0[k]))
                d16, d32 = (v - w0[k]).double(), (s32[k] - w0[k]).double()
                if float(d32.norm()) > 0 and v.dim() == 4 and "encoder" in k:
                    gain = float((d16 * d32).sum() / (d32 * d32).sum())  # the update of every trainable conv weight, against fp32 storage
                    assert 0.8 < gain < 1.2, (k, gain)
        assert moved >= 50, moved
        for n, p in m16.named_parameters():
            if not p.requires_grad:
                assert p.grad is None, n
    else:
        mcp, _, ocp = run("bf16", **BF16_OPTIONS[tag])
        mpl, _, opl = run("bf16")
        for k in ("pred_s1", "pred_s2", "pred_t1", "pred_t2", "loss_source", "loss_target"):
            assert torch.equal(ocp[k], opl[k]), k
        assert torch.equal(ocp["label_t_hard"], opl["label_t_hard"])
        # backward: the same kernels, except that a checkpointed layer's last bn3 reduction cannot ride in the next layer's data
        # gradient (blocks._Link stops at the checkpoint boundary) and runs as its own pass: fp32 sums in another order, which moves
        # a few bf16 roundings of dz (measured 1.7e-4 on the gradient norm)
        assert float(ocp["grad_norm"]) == pytest.approx(float(opl["grad_norm"]), rel=2e-3)
        scp, spl = mcp.state_dict(), mpl.state_dict()
        rels = {}
        for k, v in scp.items():
            if k.endswith("num_batches_tracked"):
                twice = any(f".layer{i}." in k for i in (1, 2, 4))
                assert int(v) == (4 if twice else 2), (k, int(v))
            elif "running_" in k:
                continue                                                # moved twice in the checkpointed layers (as in the reference)
            else:
                w0 = sd[k].cuda()
                dcp, dpl = (v - w0).double(), (spl[k] - w0).double()
                if float(dpl.norm()) > 0:
                    rels[k] = float((dcp - dpl).norm() / dpl.norm())
        worst = max(rels, key=rels.get)
        med = sorted(rels.values())[len(rels) // 2]
        print(f"bf16 storage, checkpointed layers: relative difference of a tensor's update: median {med:.3e}, largest {rels[worst]:.3e} ({worst})")
        # a moved bf16 rounding of dz is amplified on the way down like any other bf16 rounding (0.11 on dx through seven
        # blocks); the tensors behind the checkpoint boundaries carry it
        assert med < 2e-2 and rels[worst] < 0.2, (med, worst, rels[worst])


def test_bf16_stem_and_instnorm_ends_of_the_bf16_region():
    """Round 5: the two ends of the bf16 region in bf16.  (a) Stem with a bf16 conv output z: z = RNE(the fp32 stem conv on bf16
    operands), BatchNorm statistics of the ROUNDED values (checked against torch on the widened z), pooled map and argmax taps equal
    to the fp32 pool kernel on the widened z (bit for bit after rounding), backward (pooled BatchNorm backward reading bf16 z / bf16
    pooled gradient, bf16 dz, weight gradient from the bf16 dz) against the fp32 kernels on the same widened tensors.  (b) The
    InstanceNorm reading bf16 and its backward writing bf16: the fp32 kernels' arithmetic on the widened input, rounded once."""
    from uemda_amd import ops, ops_bf16 as ob
    g = torch.Generator().manual_seed(11)
    N, H, W = 2, 64, 64
    x = torch.randn(N, 3, H, W, generator=g).cuda()
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).cuda().contiguous(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(64).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(64, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(64, generator=g) * 0.2)
    bn2 = torch.nn.BatchNorm2d(64).cuda().train()
    bn2.load_state_dict(bn.state_dict())
    x4 = ops.nchw3_to_nhwc4(x)
    w8 = torch.empty((64, 7, 8, 4), device="cuda")
    ops.call("uem_stem_pack_weight", ops.ptr(ops.weight_ohwi(w)), ops.ptr(w8), ops.stream())
    assert ob.stem_ok(x.shape, bn)
    z, st = ob.stem_conv_bn(x4, w8, bn)
    bn3 = torch.nn.BatchNorm2d(64).cuda().train()
    with ops.conv_precision("bf16"):
        z32, _ = ops.stem_conv_bn(x4, ops.weight_ohwi(w), bn3, w8=w8)            # the fp32 tensor of the same bf16-operand conv
    assert z.dtype == torch.bfloat16 and torch.equal(z, z32.to(torch.bfloat16))
    zw = z.float()
    st_ref = ops.bn_stats(zw, bn2.weight.detach(), bn2.bias.detach(), bn2.running_mean, bn2.running_var, True)
    for a, b in ((st.mean, st_ref.mean), (st.invstd, st_ref.invstd), (st.scale, st_ref.scale), (st.shift, st_ref.shift),
                 (bn.running_mean, bn2.running_mean), (bn.running_var, bn2.running_var)):
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-6)
    y, idx = ob.maxpool_affine_fwd(z, st, True)
    y_ref, idx_ref = ops.maxpool_affine_fwd(zw, st, True)
    assert y.dtype == torch.bfloat16 and torch.equal(y, y_ref.to(torch.bfloat16)) and torch.equal(idx, idx_ref)
    dy = (torch.randn(y.shape, generator=g) * 0.5).cuda().to(torch.bfloat16)
    gg, gb = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
    dz = ob.bn_backward_pooled(z, dy, idx, st, gg, gb)
    gg2, gb2 = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
    dz_ref = ops.bn_backward_pooled(zw, dy.float(), idx, st, gg2, gb2)
    torch.testing.assert_close(gg, gg2, rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(gb, gb2, rtol=1e-5, atol=1e-4)
    assert dz.dtype == torch.bfloat16
    torch.testing.assert_close(dz.float(), dz_ref, rtol=2 ** -8, atol=1e-6)     # one bf16 rounding of the same fp32 value
    dw = torch.zeros(64, 7, 7, 3, device="cuda")
    ob.stem_wgrad(x4, dz, dw)
    dw_ref = torch.zeros(64, 7, 7, 3, device="cuda")
    ops.stem_wgrad(x4, dz.float(), dw_ref)                  # the stem's own kernel: fp32 operands, the bf16 dz widened at the load
    torch.testing.assert_close(dw, dw_ref, rtol=1e-4, atol=1e-4 * float(dw_ref.abs().max()))
    # (b) InstanceNorm
    xi = (torch.randn(2, 8, 8, 128, generator=g) * 2 + 0.5).cuda().to(torch.bfloat16)
    yi, inv = ob.instnorm_fwd(xi)
    yi_ref, inv_ref = ops.instnorm_fwd(xi.float())
    torch.testing.assert_close(yi, yi_ref, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(inv, inv_ref, rtol=1e-6, atol=1e-7)
    dyi = torch.randn(yi.shape, generator=g).cuda()
    dxi = ob.instnorm_bwd(yi, dyi, inv)
    assert dxi.dtype == torch.bfloat16
    torch.testing.assert_close(dxi.float(), ops.instnorm_bwd(yi_ref, dyi, inv_ref), rtol=2 ** -8, atol=1e-6)


@pytest.mark.parametrize("cin,cout,hw,stats", [(256, 1024, 32, True), (64, 256, 64, True), (512, 128, 32, False), (1024, 256, 16, True)])
def test_pointwise_stream_kernel_equals_the_dispatched_kernel(cin, cout, hw, stats):
    """Round 6: pw_bf16_stream_kernel (four-stage operand ring of 32-channel k-steps, loader / storer waves, the tile's output parked in
    LDS and stored under the next tile's MFMAs) is off by rule -- it ties the round-5 dispatch -- but stays in the library as the record of
    that structure: forced on, it must equal the dispatched kernel bit for bit (same accumulation order), forward with and without the
    BatchNorm tile statistics and as a plain data gradient; the statistics to their summation order."""
    import ctypes
    from uemda_amd import _lib, ops_bf16
    lib = _lib.load()
    lib.uemdbg_conv_bf16_pw.argtypes = [ctypes.c_int]
    lib.uemdbg_conv_bf16_pw.restype = None
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(8, hw, hw, cin, generator=g).cuda().bfloat16()
    w = (torch.randn(cout, 1, 1, cin, generator=g) * 0.05).cuda().bfloat16()
    dy = torch.randn(8, hw, hw, cout, generator=g).cuda().bfloat16()
    wt = w.permute(3, 1, 2, 0).contiguous()
    try:
        lib.uemdbg_conv_bf16_pw(0)
        ref = ops_bf16.conv2d(x, w, want_stats=stats)
        ref_d = ops_bf16.conv2d_dgrad(dy, wt, x.shape)
        lib.uemdbg_conv_bf16_pw(1)
        out = ops_bf16.conv2d(x, w, want_stats=stats)
        out_d = ops_bf16.conv2d_dgrad(dy, wt, x.shape)
    finally:
        lib.uemdbg_conv_bf16_pw(-1)
    torch.cuda.synchronize()
    if stats:
        assert torch.equal(out[0], ref[0])
        torch.testing.assert_close(out[1], ref[1], rtol=2e-5, atol=2e-3)
    else:
        assert torch.equal(out, ref)
    assert torch.equal(out_d, ref_d)


@pytest.mark.parametrize("cin,cout,n,hw,stats", [(256, 1024, 8, 32, True), (128, 512, 8, 64, True), (64, 256, 2, 128, True), (256, 1024, 2, 32, False),
                                                  (256, 64, 8, 32, True), (128, 1024, 10, 64, False)])
def test_a_stationary_pointwise_kernel_equals_the_dispatched_kernel(cin, cout, n, hw, stats):
    """Round 6, late: pw_bf16_areg_kernel (a block of eight waves owns 512 rows and all columns: A once into registers, B as whole 64-column
    tiles through a double buffer, wave-private output staging).  Forced on wherever it is legal (K in 64 / 128 / 256, M a multiple of 512 --
    the last two cases have several panels per block's stride or a narrow output), it must equal the dispatched kernel bit for bit (same
    accumulation order over k), forward with and without the BatchNorm tile statistics and as the plain data gradient of the mirrored
    shape; the statistics to their summation order."""
    import ctypes
    from uemda_amd import _lib, ops_bf16
    lib = _lib.load()
    lib.uemdbg_conv_bf16_areg.argtypes = [ctypes.c_int]
    lib.uemdbg_conv_bf16_areg.restype = None
    g = torch.Generator().manual_seed(cin + cout + n)
    x = torch.randn(n, hw, hw, cin, generator=g).cuda().bfloat16()
    w = (torch.randn(cout, 1, 1, cin, generator=g) * 0.05).cuda().bfloat16()
    # the data gradient with the same reduction length: dy has `cin` channels, dx `cout`
    w2 = (torch.randn(cin, 1, 1, cout, generator=g) * 0.05).cuda().bfloat16()
    dy = torch.randn(n, hw, hw, cin, generator=g).cuda().bfloat16()
    wt2 = w2.permute(3, 1, 2, 0).contiguous()
    xs = (n, hw, hw, cout)
    try:
        lib.uemdbg_conv_bf16_areg(0)
        ref = ops_bf16.conv2d(x, w, want_stats=stats)
        ref_d = ops_bf16.conv2d_dgrad(dy, wt2, xs)
        lib.uemdbg_conv_bf16_areg(1)
        out = ops_bf16.conv2d(x, w, want_stats=stats)
        out_d = ops_bf16.conv2d_dgrad(dy, wt2, xs)
    finally:
        lib.uemdbg_conv_bf16_areg(-1)
    torch.cuda.synchronize()
    if stats:
        assert torch.equal(out[0], ref[0])
        torch.testing.assert_close(out[1], ref[1], rtol=2e-5, atol=2e-3)
    else:
        assert torch.equal(out, ref)
    assert torch.equal(out_d, ref_d)
