"""GPU parity of the mining / loss kernels against the oracle and the reference goldens (through the C ABI)."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
C = 6


def dev(t):
    return t.cuda()


@pytest.fixture(scope="module")
def aligner():
    from uemda_amd.gast.alignment import Aligner
    return Aligner(None, feat_channels=64, class_num=C, ignore_label=-1, decay=0.996)


def test_native_library_is_the_one_running():
    from uemda_amd import _lib
    assert _lib.load().uem_version() >= 100


def test_pseudo_selection_golden_bit_exact():
    from uemda_amd.gast.pseudo_generation import pseudo_selection
    g = load_golden("pseudo_selection")
    for m, h in zip(g["masks"], g["hards"]):
        out = pseudo_selection(dev(m), 0.8, 0.6, "tensor", -1)
        assert out.dtype == torch.int64
        assert torch.equal(out.cpu(), h)
    assert pseudo_selection(dev(g["masks"][0]), return_type="ndarray").shape == (2, 64, 64)
    with pytest.raises(AssertionError):
        pseudo_selection(dev(g["masks"][0]) * 3.0, return_type="tensor")


def test_pearson_golden(aligner):
    g = load_golden("pearson")
    d = aligner._pearson_dist(dev(g["x"]), dev(g["protos"]))
    torch.testing.assert_close(d.cpu(), g["dist"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("k,n,m", [(2048, 4097, 6), (2048, 2, 7), (1024, 1023, 6), (512, 77, 8), (256, 5, 3), (2048, 1, 1), (192, 33, 6)])
def test_pearson_register_resident_rows_vs_oracle(k, n, m):
    """The one-pass kernels (k = 256 * KV: the row lives in registers, prototypes centred in the block's prologue) at every feature
    width they are instantiated for, with odd row counts (the wave's second row missing) and class counts up to 8; k = 192 takes the
    two-pass kernel.  Against the oracle's formula (alignment.py:424-451) and against float64."""
    from oracle import gast
    from uemda_amd.gast.alignment import Aligner
    g = torch.Generator().manual_seed(k + n + m)
    x = torch.randn(n, k, generator=g) * 1.5 + 0.7
    pr = torch.randn(m, k, generator=g) + 0.2
    x[0] = pr[0] * 2.0 + 0.1                                           # a row almost perfectly correlated with a prototype: dist ~ 0
    al = Aligner(None, feat_channels=k, class_num=m, ignore_label=-1, decay=0.996)
    d = al._pearson_dist(dev(x), dev(pr)).cpu()
    torch.testing.assert_close(d, gast.pearson_dist(x, pr), rtol=1e-5, atol=2e-6)
    xd, pd = x.double(), pr.double()
    xc, pc = xd - xd.mean(1, keepdim=True), pd - pd.mean(1, keepdim=True)
    r = (xc @ pc.T) / (xc.norm(dim=1, keepdim=True) * pc.norm(dim=1).unsqueeze(0))
    torch.testing.assert_close(d.double(), 0.5 * (1 - r), rtol=1e-4, atol=2e-6)
    if m <= 8:
        al.prototypes = dev(pr).contiguous()
        sim = al._pearson_sim_map(dev(x).view(1, 1, n, k)).cpu().view(n, m)
        torch.testing.assert_close(sim[1:], 1.0 / gast.pearson_dist(x, pr)[1:], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("mode", ["all", "s", "p", "l"])
def test_label_refine_golden(aligner, mode):
    g = load_golden("label_refine")
    aligner.prototypes = dev(g["protos"]).contiguous()
    out = aligner.label_refine(dev(g["sup"]), dev(g["feat"]), [dev(g["p1"]), dev(g["p2"])], dev(g["soft"]), True, mode, 2.0)
    torch.testing.assert_close(out.cpu(), g["out_" + mode], rtol=2e-5, atol=1e-6)


def test_label_refine_irregular_single_pred_and_explicit_ignore(aligner):
    g = load_golden("label_refine")
    aligner.prototypes = dev(g["protos"]).contiguous()
    out = aligner.label_refine(dev(g["sup_irregular"]), dev(g["feat"]), [dev(g["p1"]), dev(g["p2"])], dev(g["soft"]), True, "all", 2.0)
    torch.testing.assert_close(out.cpu(), g["out_all_irregular"], rtol=2e-5, atol=1e-6)
    out = aligner.label_refine(dev(g["sup"]), dev(g["feat"]), dev(g["p1"]), dev(g["soft"]), True, "l", 1.5)
    torch.testing.assert_close(out.cpu(), g["out_single_pred"], rtol=2e-5, atol=1e-6)
    out = aligner.label_refine(dev(g["sup"]), dev(g["feat"]), [dev(g["p1"]), dev(g["p2"])], dev(g["soft"]), True, "all", 2.0,
                               sup_ignore_id=int(g["sup"].max()))
    torch.testing.assert_close(out.cpu(), g["out_all"], rtol=2e-5, atol=1e-6)
    same = aligner.label_refine(dev(g["sup"]), dev(g["feat"]), dev(g["p1"]), dev(g["soft"]), refine=False)
    assert torch.equal(same.cpu(), g["soft"])


@pytest.mark.parametrize("mode", ["all", "s", "p", "l"])
@pytest.mark.parametrize("cut", [(0.8, 0.6), (0.9, 0.5), (0.3, 0.1), (0.05, 0.01), (1.0, 0.99)])
def test_refine_and_select_equals_the_two_calls(aligner, mode, cut):
    """uem_label_refine_select (the step's call, train_ssl_uem.py:209-214): the refined map and the hard labels it returns are the
    ones label_refine followed by pseudo_selection give, bit for bit -- for the reference's cutoffs (0.8, 0.6), for cutoffs under 0.5
    where several classes pass cutoff_low and the selection pass falls back to the refined map (candidate code 255), and for cutoffs
    nothing passes."""
    from uemda_amd.gast.pseudo_generation import pseudo_selection
    g = load_golden("label_refine")
    aligner.prototypes = dev(g["protos"]).contiguous()
    args = (dev(g["sup"]), dev(g["feat"]), [dev(g["p1"]), dev(g["p2"])], dev(g["soft"]))
    top, low = cut
    soft_ref = aligner.label_refine(*args, True, mode, 2.0)
    hard_ref = pseudo_selection(soft_ref, top, low, "tensor", -1)
    soft, hard = aligner.refine_and_select(*args, mode=mode, temp=2.0, cutoff_top=top, cutoff_low=low)
    assert torch.equal(soft, soft_ref)
    assert hard.dtype == torch.int64 and torch.equal(hard, hard_ref)
    pm = aligner._last_plane_max.view(torch.float32)
    assert torch.equal(pm, soft_ref.amax(dim=(2, 3)))
    if cut == (0.3, 0.1):
        assert int((hard_ref >= 0).sum()) > 0 and int((hard_ref < 0).sum()) > 0       # both outcomes occur


def test_refine_and_select_on_a_map_whose_size_is_not_a_multiple_of_four(aligner):
    """H * W % 4 != 0: the fused entry declines and the two kernels run in sequence -- same contract."""
    from uemda_amd.gast.pseudo_generation import pseudo_selection
    g = torch.Generator().manual_seed(5)
    B, H, W, h, w = 2, 17, 19, 5, 6
    soft0 = torch.softmax(torch.randn(B, C, H, W, generator=g) * 2, 1)
    sup = torch.randint(0, 12, (B, 1, H, W), generator=g)
    feat = torch.randn(B, 64, h, w, generator=g)
    p1, p2 = torch.randn(B, C, h, w, generator=g), torch.randn(B, C, h, w, generator=g)
    aligner.prototypes = torch.randn(C, 64, generator=g).cuda()
    args = (dev(sup), dev(feat), [dev(p1), dev(p2)], dev(soft0))
    soft_ref = aligner.label_refine(*args, True, "all", 2.0)
    soft, hard = aligner.refine_and_select(*args)
    assert torch.equal(soft, soft_ref) and torch.equal(hard, pseudo_selection(soft_ref, 0.8, 0.6, "tensor", -1))


def test_label_refine_superpixel_table_is_sized_per_call_and_reports_overflow():
    """The reference sizes the scatter from every batch (alignment.py:241-245).  A second batch whose ids exceed the
    first batch's must be refined exactly (the table used to be frozen by the first call, later ids read
    segment 0); an id beyond the table's capacity raises instead of borrowing another segment's maxima."""
    from oracle import gast
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.ops import UemError
    g = load_golden("label_refine")
    al = Aligner(None, feat_channels=64, class_num=C, ignore_label=-1, decay=0.996)
    al.prototypes = dev(g["protos"]).contiguous()
    args = (g["feat"], [g["p1"], g["p2"]], g["soft"])
    dargs = (dev(g["feat"]), [dev(g["p1"]), dev(g["p2"])], dev(g["soft"]))
    first = al.label_refine(dev(g["sup"]), *dargs, True, "all", 2.0)
    torch.testing.assert_close(first.cpu(), g["out_all"], rtol=2e-5, atol=1e-6)
    # same partition, ids relabelled far beyond the first batch's maximum (a crop of a larger image: ids up to 4096)
    sup2 = g["sup"] * 97 + 1000
    sup2[g["sup"] == g["sup"].max()] = 4096
    assert int(sup2.max()) > int(g["sup"].max())
    out2 = al.label_refine(dev(sup2), *dargs, True, "all", 2.0)
    ref2 = gast.label_refine(sup2, args[0], args[1], args[2], g["protos"])
    torch.testing.assert_close(out2.cpu(), ref2, rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(out2.cpu(), g["out_all"], rtol=2e-5, atol=1e-6)      # relabelling changes nothing
    # explicit ignored id smaller than the ids present (the hole the explicit path had: S = ignore + 1)
    out3 = al.label_refine(dev(sup2), *dargs, True, "all", 2.0, sup_ignore_id=16)
    ref3 = gast.label_refine(sup2, args[0], args[1], args[2], g["protos"], sup_ignore_id=16)
    torch.testing.assert_close(out3.cpu(), ref3, rtol=2e-5, atol=1e-6)
    al.check_superpixel_ids()
    # beyond the capacity: reported, never folded into segment 0
    sup4 = sup2.clone()
    sup4[sup4 == 4096] = 1 << 20
    al.label_refine(dev(sup4), *dargs, True, "all", 2.0)
    with pytest.raises(UemError, match="outside the segment table"):
        al.check_superpixel_ids()
    al.sup_capacity = (1 << 20) + 1                                   # the documented remedy
    out5 = al.label_refine(dev(sup4), *dargs, True, "all", 2.0)
    al.check_superpixel_ids()
    torch.testing.assert_close(out5.cpu(), g["out_all"], rtol=2e-5, atol=1e-6)
    neg = g["sup"].clone()
    neg[0, 0, 0, 0] = -3
    al.label_refine(dev(neg), *dargs, True, "s", 2.0)
    with pytest.raises(UemError):
        al.check_superpixel_ids()


def test_downscale_label_golden():
    from uemda_amd.gast.alignment import DownscaleLabel
    g = load_golden("downscale_label")
    out = DownscaleLabel(16, C, -1, 0.75)(dev(g["label"]))
    assert torch.equal(out.cpu(), g["out"])


def test_update_prototype_golden(aligner):
    g = load_golden("update_prototype")
    aligner.prototypes = dev(g["protos_in"]).contiguous().clone()
    ds = aligner.update_prototype(dev(g["feat"]), dev(g["label"]))
    assert torch.equal(ds.cpu(), g["label_ds"])
    torch.testing.assert_close(aligner.prototypes.cpu(), g["protos_out"], rtol=1e-6, atol=1e-7)


def test_losses_golden_forward_and_backward():
    from uemda_amd.gast.balance import CrossEntropy, UVEMLoss, loss_calc_uvem
    from uemda_amd.utils.tools import loss_calc
    g = load_golden("losses")
    l1 = dev(g["logits1"]).requires_grad_(True)
    l2 = dev(g["logits2"]).requires_grad_(True)
    uv = UVEMLoss(m=0.2, threshold=0.7, gamma=4, class_num=C, ignore_label=-1)
    loss = loss_calc_uvem([l1, l2], dev(g["hard"]), dev(g["soft"]), uv, multi=True)
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), g["uvem"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(l1.grad.cpu(), g["uvem_g1"], rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(l2.grad.cpu(), g["uvem_g2"], rtol=1e-4, atol=1e-8)
    l3 = dev(g["logits1"]).requires_grad_(True)
    l4 = dev(g["logits2"]).requires_grad_(True)
    ce = loss_calc([l3, l4], dev(g["label_s"]), CrossEntropy(ignore_label=-1), multi=True)
    (ce * 2.0).backward()                       # exercises the grad_output scaling path
    torch.testing.assert_close(ce.detach().cpu(), g["ce"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(l3.grad.cpu(), 2.0 * g["ce_g1"], rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(l4.grad.cpu(), 2.0 * g["ce_g2"], rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(uv.get_weight(dev(g["u"])).cpu(), g["uvem_w"], rtol=1e-5, atol=1e-6)


def test_class_balance_golden():
    from uemda_amd.gast.balance import ClassBalance
    g = load_golden("class_balance")
    cb = ClassBalance(C, -1, 0.99, 2.0)
    for lab, w in zip(g["labels"], g["weights"]):
        torch.testing.assert_close(cb.get_class_weight_4pixel(dev(lab)).cpu(), w, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(cb.freq.cpu(), g["freq"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("reduce", ["max", "sum", "mean"])
def test_scatter_matches_oracle(reduce):
    from oracle import gast
    from uemda_amd.scatter import scatter
    g = torch.Generator().manual_seed(3)
    src = torch.randn(3, 1000, 5, generator=g)            # negative values exercise the ordered-key max
    idx = torch.randint(0, 37, (3, 1000, 1), generator=g)
    idx[idx == 11] = 12                                   # leave a segment untouched -> 0
    out = scatter(dev(src), dev(idx), dim=1, reduce=reduce)
    ref = gast.scatter(src, idx, 1, reduce)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-5)


def test_scatter_other_ranks_and_dims():
    """beyond the path's (B, N, C) / dim=1 call: 1-d, 2-d with dim 0 / -1, 4-d with a middle dim, broadcast index, dim_size, refusals"""
    from oracle import gast
    from uemda_amd.ops import UemError
    from uemda_amd.scatter import scatter
    g = torch.Generator().manual_seed(8)

    def ref(src, idx, dim, reduce, dim_size=None):
        d = dim if dim >= 0 else dim + src.dim()
        ii = idx
        if ii.dim() == 1:                                  # torch_scatter's broadcast rule
            for _ in range(d):
                ii = ii.unsqueeze(0)
        while ii.dim() < src.dim():
            ii = ii.unsqueeze(-1)
        return gast.scatter(src, ii.expand_as(src), d, reduce, dim_size)
    cases = [(torch.randn(500, generator=g), torch.randint(0, 9, (500,), generator=g), 0),
             (torch.randn(300, 7, generator=g), torch.randint(0, 5, (300, 1), generator=g), 0),
             (torch.randn(4, 300, generator=g), torch.randint(0, 11, (4, 300), generator=g), -1),
             (torch.randn(2, 3, 200, 5, generator=g), torch.randint(0, 13, (2, 3, 200, 1), generator=g), 2),
             (torch.randn(2, 3, 200, 5, generator=g), torch.randint(0, 13, (200,), generator=g), -2)]        # a 1-d index lined up with dim
    for src, idx, dim in cases:
        for reduce in ("max", "sum", "mean"):
            out = scatter(dev(src), dev(idx), dim=dim, reduce=reduce)
            torch.testing.assert_close(out.cpu(), ref(src, idx, dim, reduce), rtol=1e-5, atol=1e-5)
    out = scatter(dev(cases[0][0]), dev(cases[0][1]), dim=0, reduce="sum", dim_size=20)
    assert out.shape == (20,) and float(out[9:].abs().max()) == 0.0
    with pytest.raises(UemError):
        scatter(dev(cases[3][0]), dev(torch.randint(0, 3, (2, 3, 200, 5))), dim=2)       # an index per trailing element: not this kernel's layout
    with pytest.raises(UemError):
        scatter(dev(cases[0][0]), dev(cases[0][1]), dim=0, out=torch.zeros(9, device="cuda"))


def test_ragged_and_empty_edges():
    """non-multiple-of-tile sizes, a batch with every pixel ignored, all-ignored superpixels."""
    from oracle import gast, synth
    from uemda_amd.gast.alignment import Aligner, DownscaleLabel
    from uemda_amd.gast.balance import CrossEntropy
    from uemda_amd.gast.pseudo_generation import pseudo_selection
    from uemda_amd.utils.tools import loss_calc
    b = synth.make_batch(B=3, H=80, W=48, C=C, k=64, seed=4)      # H, W not multiples of 64
    g = torch.Generator().manual_seed(1)
    feat = torch.randn(3, 64, 5, 3, generator=g)
    p1, p2 = torch.randn(3, C, 5, 3, generator=g), torch.randn(3, C, 5, 3, generator=g)
    al = Aligner(None, 64, C, -1, 0.996)
    al.prototypes = dev(b["prototypes"]).contiguous()
    soft_ref = gast.label_refine(b["label_t_sup"], feat, [p1, p2], b["label_t_soft"], b["prototypes"])
    soft, hard = al.refine_and_select(dev(b["label_t_sup"]), dev(feat), [dev(p1), dev(p2)], dev(b["label_t_soft"]))
    torch.testing.assert_close(soft.cpu(), soft_ref, rtol=2e-5, atol=1e-6)
    assert torch.equal(hard.cpu(), gast.pseudo_selection(soft.cpu()))       # bit-exact given identical soft input
    assert torch.equal(pseudo_selection(soft, return_type="tensor").cpu(), hard.cpu())
    all_ign = torch.full((2, 32, 32), -1, dtype=torch.int64)
    assert (DownscaleLabel(16, C)(dev(all_ign)).cpu() == -1).all()
    lg = torch.randn(2, C, 2, 2)
    loss = loss_calc([dev(lg).requires_grad_(True)], dev(all_ign), CrossEntropy(-1), multi=True)
    assert float(loss) == 0.0
    sup_all_ign = torch.full((3, 1, 80, 48), 7, dtype=torch.int64)
    out = al.label_refine(dev(sup_all_ign), dev(feat), [dev(p1), dev(p2)], dev(b["label_t_soft"]))
    ref = gast.label_refine(sup_all_ign, feat, [p1, p2], b["label_t_soft"], b["prototypes"])
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-5, atol=1e-6)


def test_mining_at_benchmark_tile_size_vs_oracle_and_properties():
    """512x512 tiles (BASELINE configs 2-3): B=2 against the oracle, then size-independent properties."""
    from oracle import gast, synth
    from uemda_amd.gast.alignment import Aligner
    b = synth.make_batch(B=2, H=512, W=512, C=C, k=2048, seed=11)
    g = torch.Generator().manual_seed(2)
    feat = torch.randn(2, 2048, 32, 32, generator=g)
    p1, p2 = 2 * torch.randn(2, C, 32, 32, generator=g), 2 * torch.randn(2, C, 32, 32, generator=g)
    al = Aligner(None, 2048, C, -1, 0.996)
    al.prototypes = dev(b["prototypes"]).contiguous()
    soft, hard = al.refine_and_select(dev(b["label_t_sup"]), dev(feat), [dev(p1), dev(p2)], dev(b["label_t_soft"]))
    ref = gast.label_refine(b["label_t_sup"], feat, [p1, p2], b["label_t_soft"], b["prototypes"])
    torch.testing.assert_close(soft.cpu(), ref, rtol=5e-5, atol=2e-6)
    ref_hard = gast.pseudo_selection(ref)
    # end to end a pixel within 1 ulp of its threshold may flip: compare with a margin mask
    thr = torch.maximum(ref.flatten(2).max(-1)[0] * 0.8, torch.tensor(0.6)).view(2, C, 1, 1)
    safe = ((ref - thr).abs() > 1e-5).all(dim=1)
    assert torch.equal(hard.cpu()[safe], ref_hard[safe]) and safe.float().mean() > 0.999
    assert torch.equal(hard.cpu(), gast.pseudo_selection(soft.cpu()))       # exact on identical input
    s = soft.sum(dim=1)
    assert ((s - 1).abs() < 1e-5).all() and (soft >= 0).all()               # normalised probabilities
    protos_ref, ds_ref = gast.update_prototype(feat, b["label_s"], b["prototypes"], C, 0.996)
    ds = al.update_prototype(dev(feat), dev(b["label_s"]))
    assert torch.equal(ds.cpu(), ds_ref)
    torch.testing.assert_close(al.prototypes.cpu(), protos_ref, rtol=1e-5, atol=1e-6)


def test_superpixel_edge_shrinking_golden_and_file_round_trip(tmp_path):
    """HIP edge shrinking == the reference's function (golden) and == the oracle on a 512x512 irregular map; the id
    map survives the `<name>.tif` wire format and feeds label_refine's layout (1,H,W) int64."""
    from oracle import gast, synth
    from uemda_amd.gast.superpixels import edge_shrinking, load_superpixels, save_superpixels
    g = load_golden("superpixel_shrink")
    for k in ("a", "b", "c"):
        got = edge_shrinking(g[f"in_{k}"].cuda(), 3, 16)
        assert got.dtype == torch.int32 and torch.equal(got.cpu(), g[f"out_{k}"])
    big = synth.irregular_superpixels(2, 512, 512, 900, seed=4)[:, 0].to(torch.int32)
    got = edge_shrinking(big.cuda(), 3, 16)
    for b in range(2):
        assert (got[b].cpu().numpy() == gast.edge_shrinking(big[b].numpy(), 3, 16)).all()
    assert int(got.max()) == 1024                                   # ignored id = 512/16 * 512/16
    p = str(tmp_path / "tile.tif")
    save_superpixels(p, got[0])
    back = load_superpixels(p)
    assert back.shape == (1, 512, 512) and back.dtype == torch.int64 and torch.equal(back[0].int(), got[0])
