"""CPU oracle for the UemDA hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This package is a CPU restatement (plain PyTorch-CPU / numpy) of the reference
algorithm on the path named by BASELINE.json `north_star`:

  * `oracle.model`   Deeplabv2 = ResNet encoder + InstanceNorm + ASPP / PPM heads
                     (reference uemda/models/Encoder.py, uemda/resnet.py, uemda/_resnets.py)
  * `oracle.gast`    label_refine / pearson / scatter / pseudo_selection / DownscaleLabel /
                     prototype EMA / CrossEntropy / UVEMLoss / ClassBalance / LR schedule
                     (reference uemda/gast/{alignment,pseudo_generation,balance}.py,
                      uemda/utils/tools.py)
  * `oracle.step`    one `train_ssl_uem.py` / `train_src.py` iteration (reference
                     tools/train_ssl_uem.py:193-232, tools/train_src.py:112-141)
  * `oracle.synth`   the synthetic tiles of SURVEY.md §8(d)
  * `oracle.weights` deterministic weight fill keyed on state_dict names (build-side)

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it,
and only as the checker / the CPU baseline.  Nothing under `uemda_amd/` imports it.

Parity pinning: the reference tree holds no tests or fixtures for this path.
The oracle is pinned instead by golden vectors produced by the reference itself, imported in
the build container under third-party stubs (`tests/golden/make_golden.py`, committed together
with the vectors in `tests/golden/*.npz`) and checked by `tests/test_oracle_golden.py`.
`torch_scatter` (unpinned third-party wheel, absent from /root/reference) is restated from its
documented semantics; that one boundary is "parity unpinned" (see DESIGN.md).
"""
