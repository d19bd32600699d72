"""Host-side superpixel file format (`<name>.tif`, int32 ids): round trip of the minimal TIFF codec and a cross-check
against an independent implementation (Pillow) in both directions when it is installed."""
import numpy as np
import pytest

from uemda_amd.utils import tiff


@pytest.mark.parametrize("dtype", ["int32", "uint16", "uint8", "float32"])
def test_round_trip(tmp_path, dtype):
    rng = np.random.default_rng(0)
    a = (rng.integers(0, 2000, size=(37, 53)) - (500 if dtype == "int32" else 0)).astype(dtype)
    if dtype == "uint8":
        a = (a % 251).astype(dtype)
    p = str(tmp_path / "x.tif")
    tiff.write_tiff(p, a)
    b = tiff.read_tiff(p)
    assert b.dtype == a.dtype and b.shape == a.shape and (a == b).all()


def test_against_pillow(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(1)
    a = rng.integers(0, 1025, size=(64, 48)).astype(np.int32)
    p1, p2 = str(tmp_path / "ours.tif"), str(tmp_path / "theirs.tif")
    tiff.write_tiff(p1, a)
    assert (np.array(Image.open(p1)) == a).all()                 # they read what we write
    Image.fromarray(a, mode="I").save(p2)                        # uncompressed, multi-strip, little-endian
    got = tiff.read_tiff(p2)
    assert got.shape == a.shape and (got == a).all()             # we read what they write


def test_rejects_what_it_cannot_read(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    p = str(tmp_path / "rgb.tif")
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(p)
    with pytest.raises(ValueError):
        tiff.read_tiff(p)
    q = str(tmp_path / "lzw.tif")
    Image.fromarray(np.zeros((8, 8), np.uint8)).save(q, compression="tiff_lzw")
    with pytest.raises(ValueError):
        tiff.read_tiff(q)
