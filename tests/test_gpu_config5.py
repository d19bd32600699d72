"""BASELINE config 5's single-GPU half: ResNet-101 + ASPP on 1024x1024 tiles, one train_ssl_uem step against the oracle,
in the exact-fp32 default and in the bf16-OPERAND mode (bf16 matrix-core operands, fp32 accumulate, fp32 tensors in memory --
DESIGN.md 3.2).  Config 5's own arithmetic, bf16 STORAGE, is tested at this shape in tests/test_gpu_bf16.py
(test_bf16_storage_ssl_step_vs_oracle[resnet101-1024]).  B = 1 source + 1 target tile keeps the oracle's CPU
step in the tens of seconds; the benchmark batch is covered by size-independent properties in test_gpu_fullsize.py.

Tolerances (stated here, derived in DESIGN.md 3.2):
  fp32 : north_star's bar -- max |dlogit| / max |logit| < 1e-3, hard pseudo-labels >= 99.95 % identical, on the default
         deterministic initialisation (Kaiming convs, BatchNorm gamma ~ U(.5, 1.5)).
  bf16 : at that initialisation a ResNet-101 in training-mode BatchNorm is chaotic -- every residual block adds a branch as
         large as its trunk, a 1e-7 (fp32 rounding) perturbation already reaches 1e-4 at the logits (x 1000), and operands
         rounded to 8 significand bits (2.3e-3 relative L2 per conv against fp64, profiles/r01_f_precision_per_conv.txt) come
         out DECORRELATED (measured: relative L2 of the logits 1.05; hard labels still 97.8 % identical, losses within 1 %,
         because the ASPP logits are small).  No reduced-precision implementation can be compared through such a network.
         The bf16 test therefore damps the residual branches the way trained (and zero-init-residual initialised) ResNets
         are: gamma of every block's last BatchNorm x 0.2, same weights for the oracle.  How much this network amplifies a
         per-conv rounding error is MEASURED by the fp32 leg of the same test: exact-fp32 convs differ from fp64 by 3.3e-7
         relative L2 (profiles/r01_f_precision_per_conv.txt) and the fp32 logits come out 9.8e-6 from the oracle's, a gain
         of 30 through the 104 convs and their training-mode BatchNorms.  bf16 operands (2.3e-3 per conv, same file) then
         predict 30 * 2.3e-3 = 6.9e-2 relative L2 at the logits (measured 6.2e-2).  Asserted for bf16: every logit map within
         1.5 x (2.3e-3 / 3.3e-7) x the fp32 leg's own error and below 0.1; hard pseudo-labels >= 99.5 % identical (measured
         99.88 %); both losses within 0.5 % (measured 4e-5 / 1.1e-3); gradient norm within 3 % (measured 0.7 %)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
C = 6


def _reference(damp):
    from oracle import synth
    from oracle.model import OracleDeeplabv2
    from oracle.step import HYPER as OH, SGDState, ssl_step as oracle_ssl
    from oracle.weights import det_state_dict
    sd = det_state_dict("resnet101", C, False, seed=2333)
    if damp != 1.0:
        for k in sd:
            if k.endswith("bn3.weight"):
                sd[k] = sd[k] * damp
    bc = synth.make_batch(B=1, H=1024, W=1024, C=C, k=2048, seed=31)
    om = OracleDeeplabv2({k: v.clone() for k, v in sd.items()}, "resnet101", C, False)
    ref = oracle_ssl(om, SGDState(om.parameters(), 0.9, 5e-4), bc["prototypes"], bc, 2e-3, OH)
    return sd, bc, ref


@pytest.fixture(scope="module")
def reference_step():
    return _reference(1.0)


@pytest.fixture(scope="module")
def reference_step_damped():
    return _reference(0.2)


def _run(sd, bc, prec):
    from uemda_amd import ops
    from uemda_amd.gast.alignment import Aligner
    from uemda_amd.models.Encoder import Deeplabv2
    from uemda_amd.optim import FusedSGD
    from uemda_amd.step import HYPER, StepState, ssl_step
    cfg = dict(backbone=dict(resnet_type="resnet101", output_stride=16, pretrained=False), multi_layer=True, cascade=False,
               use_ppm=False, ppm=dict(num_classes=C, use_aux=False, fc_dim=2048), inchannels=2048, num_classes=C, is_ins_norm=True)
    model = Deeplabv2(cfg)
    model.load_state_dict(sd)
    model = model.cuda()
    b = {k: v.cuda() for k, v in bc.items()}
    al = Aligner(None, 2048, C, -1, HYPER["proto_decay"])
    al.prototypes = b["prototypes"].clone()
    ops.set_conv_precision(prec)
    try:
        out = ssl_step(model, al, FusedSGD(model, 1e-2, 0.9, 5e-4), StepState(C), b, 2e-3, sup_ignore_id=4096)
        torch.cuda.synchronize()
    finally:
        ops.set_conv_precision("fp32")
    return out, al


def test_r101_aspp_1024_ssl_step_fp32_vs_oracle(reference_step):
    sd, bc, ref = reference_step
    out, al = _run(sd, bc, "fp32")
    for k in ("pred_s1", "pred_s2", "pred_t1", "pred_t2"):
        assert out[k].shape == (1, C, 64, 64)
        err = (out[k].cpu() - ref[k]).abs().max() / ref[k].abs().max()
        assert err < 1e-3, (k, float(err))
    agree = (out["label_t_hard"].cpu() == ref["label_t_hard"]).float().mean().item()
    assert agree >= 0.9995, agree
    torch.testing.assert_close(out["loss_source"].cpu(), ref["loss_source"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(out["loss_target"].cpu(), ref["loss_target"], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(al.prototypes.cpu(), ref["prototypes"], rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(out["grad_norm"].cpu().reshape(()), ref["grad_norm"].reshape(()), rtol=3e-2, atol=1e-4)


_FP32_LEG = {}
EPS_FP32, EPS_BF16 = 3.3e-7, 2.3e-3          # relative L2 of one conv against fp64 (profiles/r01_f_precision_per_conv.txt)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_r101_aspp_1024_ssl_step_damped_residuals_vs_oracle(reference_step_damped, prec):
    sd, bc, ref = reference_step_damped
    out, al = _run(sd, bc, prec)
    rels = {}
    for k in ("pred_s1", "pred_s2", "pred_t1", "pred_t2"):
        rels[k] = float((out[k].cpu() - ref[k]).norm() / ref[k].norm())
    agree = (out["label_t_hard"].cpu() == ref["label_t_hard"]).float().mean().item()
    ls = abs(float(out["loss_source"]) / float(ref["loss_source"]) - 1.0)
    lt = abs(float(out["loss_target"]) / float(ref["loss_target"]) - 1.0)
    gn = abs(float(out["grad_norm"]) / float(ref["grad_norm"]) - 1.0)
    print(f"{prec} operands, R101-ASPP 1024x1024, damped residual branches: logit relative L2 {rels}, hard-label agreement "
          f"{agree:.5f}, loss_source off by {ls:.2e}, loss_target off by {lt:.2e}, grad norm off by {gn:.2e}")
    if prec == "fp32":
        assert max(rels.values()) < 1e-4 and agree >= 0.9999 and ls < 1e-4 and lt < 1e-4 and gn < 1e-2, (rels, agree, ls, lt, gn)
        _FP32_LEG.update(rels)
        return
    for k, v in rels.items():
        assert v < 0.1, (k, v)
        if k in _FP32_LEG:              # operand rounding alone, at this network's measured conditioning
            assert v < 1.5 * (EPS_BF16 / EPS_FP32) * _FP32_LEG[k], (k, v, _FP32_LEG[k])
    assert agree >= 0.995, agree
    assert ls < 5e-3 and lt < 5e-3, (ls, lt)
    assert gn < 3e-2, gn
