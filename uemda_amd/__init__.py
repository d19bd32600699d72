"""uemda_amd: MI355X-native (gfx950) implementation of the UemDA hot path.

Same operator surface as the reference's `uemda` package for that path:
    uemda_amd.models.Encoder.Deeplabv2
    uemda_amd.gast.alignment.Aligner / DownscaleLabel
    uemda_amd.gast.pseudo_generation.pseudo_selection
    uemda_amd.gast.balance.{UVEMLoss, CrossEntropy, ClassBalance, loss_calc_uvem}
    uemda_amd.utils.tools.{loss_calc, adjust_learning_rate, lr_poly, lr_warmup, seed_torch}
    uemda_amd.scatter.scatter            (torch_scatter.scatter replacement)
plus what the reference lacks: uemda_amd.optim.FusedSGD and uemda_amd.dp (RCCL data parallel).
All arithmetic runs in hand-written HIP kernels behind the C ABI of include/uemda_hip.h.
"""
from ._lib import UemError, build, load  # noqa: F401

__version__ = "0.1.0"
