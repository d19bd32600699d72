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
]


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
def test_conv_bf16_forward_and_data_gradient(case):
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
