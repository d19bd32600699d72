"""Synthetic tiles of SURVEY.md section 8(d).  The generator is plain input plumbing (no reference algorithm), so it
lives in the package (`uemda_amd/utils/synth.py`: bench.py and the scripts use it without touching `oracle/`);
this module re-exports it for the tests and smoke(), which pair it with the oracle."""
from uemda_amd.utils.synth import irregular_superpixels, make_batch  # noqa: F401
