"""GPU parity of the Winograd F(2x2, 3x3) / F(4x4, 3x3) paths (csrc/winograd.hip; reference call sites uemda/_resnets.py:100-103,
Encoder.py:35) against torch-CPU float64 convolutions and against the direct f32-MFMA kernels they replace, through the C ABI."""
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
# F(4x4,3x3): 4x4 output tiles, T = N*H*W/16 a multiple of 128
WINO4_CASES = [
    (2, 32, 32, 128, 128, 1),
    (2, 32, 32, 512, 256, 1),
    (2, 32, 32, 256, 512, 2),        # dilated: four interleaved 16x16 sub-images
    (8, 16, 16, 1024, 512, 1),
    (16, 16, 8, 256, 320, 1),        # non-square map, Cout a multiple of 64 only
]
# relative L2 bounds against float64 (forward, data gradient, weight gradient) and against the direct kernel, per tile edge
TOL = {2: (1.5e-6, 1.5e-6, 3e-6, 2e-6), 4: (5e-6, 5e-6, 8e-6, 6e-6)}


@pytest.mark.parametrize("m,case", [(2, c) for c in WINO_CASES] + [(4, c) for c in WINO4_CASES])
def test_winograd_forward_dgrad_wgrad_vs_float64(m, case):
    from uemda_amd import ops
    N, H, W, Cin, Cout, d = case
    tf, td, tw, tdir = TOL[m]
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
    if m == 4:
        assert (N * H * W // 16) % 128 == 0 and H % (4 * d) == 0 and W % (4 * d) == 0
    wp = w.cuda().contiguous(memory_format=torch.channels_last)
    # forward with the BatchNorm-affine + ReLU prologue (zero padding after it)
    y, v = ops.conv3x3_wino(nhwc(x), wp, d, in_scale=sc.cuda(), in_shift=sh.cuda(), in_relu=True, want_v=True, m=m)
    assert v.shape[0] == (m + 2) ** 2
    assert rel(nchw(y), y_ref.detach()) < tf
    # the direct kernel on the same operands: both are fp32 sums of the same products
    y_dir = ops.conv2d(nhwc(x), ops.weight_ohwi(wp), None, pad=d, dil=d, in_scale=sc.cuda(), in_shift=sh.cuda(), in_relu=True)
    assert rel(y, y_dir) < tdir
    # data gradient (with respect to the post-activation input)
    dx, tp = ops.conv3x3_wino_dgrad(nhwc(gy), wp, d, m=m)
    assert tp is None and rel(nchw(dx), xa.grad) < td
    # weight gradient, accumulated into a non-zero buffer
    dw0 = torch.randn(Cout, 3, 3, Cin, generator=g).cuda()
    dw = dw0.clone()
    ops.conv3x3_wino_wgrad(v, nhwc(gy), dw, d)
    assert rel((dw - dw0).permute(0, 3, 1, 2), wr.grad) < tw
    # the weight gradient that recomputes V from the conv input (forward and backward on different tile sizes)
    dw2 = dw0.clone()
    ops.conv3x3_wino_wgrad(None, nhwc(gy), dw2, d, x=nhwc(x), in_scale=sc.cuda(), in_shift=sh.cuda(), in_relu=True, m=m)
    assert rel(dw2 - dw0, dw - dw0) < 1e-6          # same products; only the split-K atomics' order differs


@pytest.mark.parametrize("m,case", [(2, c) for c in WINO_CASES[:4]] + [(4, c) for c in WINO4_CASES[:3]])
def test_winograd_fused_batchnorm_passes_match_the_direct_kernels(m, case):
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
    y1, st1, v = ops.conv3x3_wino_bn(z, w, bn_a, d, in_scale=st_in.scale, in_shift=st_in.shift, in_relu=True, m=m)
    y2, st2 = ops.conv2d_bn(z, ops.weight_ohwi(w), bn_b, pad=d, dil=d, in_scale=st_in.scale, in_shift=st_in.shift, in_relu=True)
    assert rel(y1, y2) < TOL[m][3]
    for a, b in ((st1.mean, st2.mean), (st1.invstd, st2.invstd), (st1.scale, st2.scale), (st1.shift, st2.shift),
                 (bn_a.running_mean, bn_b.running_mean), (bn_a.running_var, bn_b.running_var)):
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-6)
    # backward through the conv and the producer's BatchNorm + ReLU
    dy = nhwc(torch.randn(N, Cout, H, W, generator=g))
    gg1, gb1 = torch.zeros(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
    gg2, gb2 = torch.zeros(Cin, device="cuda"), torch.zeros(Cin, device="cuda")
    dz1 = ops.conv3x3_wino_dgrad_bn_backward(dy, w, z, st_in, gg1, gb1, d, m=m)
    dz2 = ops.conv2d_dgrad_bn_backward(dy, ops.weight_transpose(ops.weight_ohwi(w)), z, st_in, gg2, gb2, pad=d, dil=d)
    assert rel(dz1, dz2) < 2e-5
    torch.testing.assert_close(gg1, gg2, rtol=2e-4, atol=2e-4)
    torch.testing.assert_close(gb1, gb2, rtol=2e-4, atol=2e-4)
    dwa, dwb = torch.zeros(Cout, 3, 3, Cin, device="cuda"), torch.zeros(Cout, 3, 3, Cin, device="cuda")
    ops.conv3x3_wino_wgrad(v, dy, dwa, d)
    ops.conv2d_wgrad(z, dy, dwb, pad=d, dil=d, in_scale=st_in.scale, in_shift=st_in.shift, in_relu=True)
    assert rel(dwa, dwb) < (5e-6 if m == 2 else 1e-5)


def test_winograd_plans_and_declined_shapes(monkeypatch):
    from uemda_amd import ops
    for k, v in dict(WINOGRAD=True, WINOGRAD_F4_FWD=True, WINOGRAD_F4_BWD=True, WINOGRAD_MIN_CH=128, WINOGRAD_MIN_CH_DGRAD=256,
                     WINOGRAD_MIN_CH_F4=64, WINOGRAD_MIN_CH_F4_FWD=256).items():
        monkeypatch.setattr(ops, k, v)                                            # the defaults, whatever the environment says

    def plan(shape, cout, d=1):
        p = ops.wino_plan(shape, cout, 3, 3, 1, d, d)
        return None if p is None else (p.mf, p.mb, p.dgrad, p.wgrad, p.keep_v)
    assert plan((2, 16, 16, 64), 64) is None                                      # narrow layer, too few 4x4 tiles: direct kernels
    assert plan((2, 16, 16, 128), 128) == (2, 2, False, True, True)               # F(2x2) forward + weight gradient on the kept V
    # the benchmark's shapes (B = 32, 512^2 tiles): layer1 backward only (data gradient), layer2 F(2x2) forward + F(4x4) backward
    # recomputing V, layer3 / layer4 / the PPM head F(4x4) both ways sharing V
    assert plan((32, 128, 128, 64), 64) == (0, 4, True, False, False)
    assert plan((32, 64, 64, 128), 128) == (2, 4, True, True, False)
    assert plan((32, 32, 32, 256), 256) == (4, 4, True, True, True)
    assert plan((32, 32, 32, 512), 512, 2) == (4, 4, True, True, True)
    assert plan((32, 32, 32, 4096), 512) == (4, 4, True, True, True)
    assert plan((8, 32, 32, 512), 512, 2) == (4, 4, True, True, True)             # layer4 of the 512^2 reference fixture
    monkeypatch.setattr(ops, "WINOGRAD_F4_FWD", False)
    assert plan((32, 32, 32, 256), 256) == (2, 4, True, True, False)
    monkeypatch.setattr(ops, "WINOGRAD_F4_BWD", False)
    assert plan((32, 32, 32, 256), 256) == (2, 2, True, True, True) and plan((32, 128, 128, 64), 64) is None
    monkeypatch.setattr(ops, "WINOGRAD_SAVE_V_BYTES", 1 << 20)                    # V over the byte cap: recomputed in backward
    assert plan((32, 32, 32, 256), 256) == (2, 2, True, True, False)
    assert ops.wino_plan((2, 16, 16, 256), 256, 3, 3, 2, 1, 1) is None            # stride 2
    assert ops.wino_plan((2, 16, 16, 256), 256, 1, 1, 1, 0, 1) is None            # 1x1
    assert ops.wino_plan((2, 18, 18, 256), 256, 3, 3, 1, 2, 2) is None            # 18 is not a multiple of 2 * dilation
    assert ops.wino_plan((1, 8, 8, 256), 256, 3, 3, 1, 1, 1) is None              # 16 tiles: not a multiple of 128
    assert not ops.wino_ok((2, 16, 16, 256), 256, 3, 3, 2, 1, 1) and ops.wino_ok((2, 16, 16, 256), 256, 3, 3, 1, 1, 1)
    with pytest.raises(ops.UemError):
        ops.wino_input(torch.zeros(1, 6, 6, 64, device="cuda"), 1)           # the C ABI refuses too (T % 32)
    with pytest.raises(ops.UemError):
        ops.wino_input(torch.zeros(2, 18, 16, 64, device="cuda"), 1, m=4)    # 18 is not a multiple of 4


def test_weight_prep_refreshes_every_derived_bank_in_one_launch():
    """uem_weight_prep (ops.PREP): the transposed banks of the direct data gradients, the Winograd banks of both tile sizes and the
    stem's packed taps, refreshed together -- bit-equal to the one-bank-per-launch entry points, following in-place weight updates,
    FusedSGD's epoch and the death of a parameter."""
    import ctypes
    import gc
    from uemda_amd import ops, ops_bf16
    g = torch.Generator().manual_seed(5)

    def param(cout, cin, k):
        return torch.nn.Parameter((torch.randn(cout, cin, k, k, generator=g) * 0.1).cuda().contiguous(memory_format=torch.channels_last))
    p1, p3, p3b, ps = param(256, 64, 1), param(128, 128, 3), param(512, 256, 3), param(64, 3, 7)

    def singles():
        out = {}
        for name, p in (("p1", p1), ("p3", p3), ("p3b", p3b)):
            out[name, "t"] = ops.weight_transpose(ops.weight_ohwi(p))
        for name, p in (("p3", p3), ("p3b", p3b)):
            cout, cin = p.shape[:2]
            for m in (2, 4):
                for tr in (False, True):
                    u = torch.empty(((m + 2) ** 2, cin, cout) if tr else ((m + 2) ** 2, cout, cin), device="cuda")
                    ops.call("uem_wino_filter", ops.ptr(ops.weight_ohwi(p)), ops.ptr(u), cout, cin, 1 if tr else 0, m, ops.stream())
                    out[name, m, tr] = u
        w8 = torch.empty(64, 7, 8, 4, device="cuda")
        ops.call("uem_stem_pack_weight", ops.ptr(ops.weight_ohwi(ps)), ops.ptr(w8), ops.stream())
        out["stem"] = w8
        for name, p in (("p1", p1), ("p3", p3)):                       # the bf16-storage model's data-gradient banks
            cout, cin, kh, kw = p.shape
            wt = torch.empty((cin, kh, kw, cout), device="cuda", dtype=torch.bfloat16)
            ops.call("uem_weight_transpose_bf16", ops.ptr(ops.weight_ohwi(p)), ops.ptr(wt), cout, kh, kw, cin, ops.stream())
            out[name, "t16"] = wt
        return out

    def batched():
        out = {("p1", "t"): ops.weight_transpose_cached(p1), ("p3", "t"): ops.weight_transpose_cached(p3), ("p3b", "t"): ops.weight_transpose_cached(p3b)}
        for name, p in (("p3", p3), ("p3b", p3b)):
            for m in (2, 4):
                for tr in (False, True):
                    out[name, m, tr] = ops.wino_filter_cached(p, tr, m)
        out["stem"] = ops.stem_weight_packed(ps)
        out["p1", "t16"], out["p3", "t16"] = ops_bf16.weight_t(p1), ops_bf16.weight_t(p3)
        return out

    def same(a, b):
        assert a.keys() == b.keys()
        for k in a:
            assert torch.equal(a[k], b[k]), k
    same(batched(), singles())                                        # first requests: one-job tables
    first = batched()
    assert all(first[k].data_ptr() == v.data_ptr() for k, v in batched().items())       # cached: the same buffers, no refresh
    with torch.no_grad():
        p3.mul_(1.5)                                                  # in place through torch: the parameter's version moves
        p1.add_(0.25)
    same(batched(), singles())                                        # ONE launch refreshed all of them
    ops.weights_changed()                                             # what FusedSGD / a replayed graph announce
    with torch.no_grad():
        p3b.data.mul_(0.5)                                            # (a kernel wrote the arena behind torch's back)
    same(batched(), singles())
    # a parameter that dies leaves the table before its memory can be read again
    prep = ops.PREP.by_device[torch.cuda.current_device()]
    addr = p3b.data_ptr()
    assert sum(1 for (a, _k) in prep.jobs if a == addr) == 5          # p3b's transposed bank and its four Winograd banks
    del p3b, first
    gc.collect()
    ops.weights_changed()
    got = ops.weight_transpose_cached(p1)
    assert torch.equal(got, ops.weight_transpose(ops.weight_ohwi(p1)))
    assert not any(a == addr for (a, _k) in prep.jobs)                # gone with the parameter (and so is every other dead model's)
    assert all(j["ref"]() is not None for j in prep.jobs.values())
    lib = ops._lib.load()
    assert lib.uem_weight_prep_blocks(1, 100, 64, 9) == -1 and lib.uem_weight_prep_blocks(0, 256, 64, 1) == 16
