"""The engine's own PNG decoder (procgen2_amd/csrc/png_decode.cpp, reached through pgv_decode_png) against
PIL on every vendored asset: the atlas must hold exactly the RGBA8 expansion the oracle is fed."""
import ctypes
import os

import numpy as np
from PIL import Image

import oracle_util


def _decode(engine_lib, path):
    w, h = ctypes.c_int32(), ctypes.c_int32()
    assert engine_lib.pgv_decode_png(path.encode(), ctypes.byref(w), ctypes.byref(h), None, 0) == 0
    buf = np.zeros(w.value * h.value * 4, np.uint8)
    assert engine_lib.pgv_decode_png(path.encode(), ctypes.byref(w), ctypes.byref(h),
                                     buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
    return buf.reshape(h.value, w.value, 4)


def test_all_vendored_pngs_decode_like_pil(engine_lib):
    seen = 0
    for root, _, files in os.walk(oracle_util.ASSETS):
        for f in sorted(files):
            if not f.endswith(".png"):
                continue
            path = os.path.join(root, f)
            want = np.asarray(Image.open(path).convert("RGBA"), dtype=np.uint8)
            got = _decode(engine_lib, path)
            assert got.shape == want.shape, path
            assert np.array_equal(got, want), path
            seen += 1
    assert seen >= 119


def test_palette_grey_and_16bit_expansion(engine_lib, tmp_path):
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, (9, 13, 3), dtype=np.uint8)
    cases = {
        "pal.png": Image.fromarray(rgb).convert("P", palette=Image.ADAPTIVE, colors=17),
        "grey.png": Image.fromarray(rgb[:, :, 0]),
        "la.png": Image.fromarray(np.dstack([rgb[:, :, 0], rgb[:, :, 1]]), "LA"),
        "bits1.png": Image.fromarray((rgb[:, :, 0] > 127)).convert("1"),
    }
    for name, im in cases.items():
        p = str(tmp_path / name)
        im.save(p)
        want = np.asarray(Image.open(p).convert("RGBA"), dtype=np.uint8)
        assert np.array_equal(_decode(engine_lib, p), want), name
    pal = cases["pal.png"].copy()
    p = str(tmp_path / "pal_trns.png")
    pal.save(p, transparency=3)
    assert np.array_equal(_decode(engine_lib, p), np.asarray(Image.open(p).convert("RGBA"), dtype=np.uint8))


def test_bad_png_is_an_error(engine_lib, tmp_path):
    p = tmp_path / "broken.png"
    p.write_bytes(b"not a png at all")
    w, h = ctypes.c_int32(), ctypes.c_int32()
    assert engine_lib.pgv_decode_png(str(p).encode(), ctypes.byref(w), ctypes.byref(h), None, 0) != 0
    assert b"PNG" in engine_lib.pgv_last_error()
