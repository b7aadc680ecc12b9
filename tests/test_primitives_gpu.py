"""Device-side sweep of the bit-exact primitives (SURVEY.md rows T1–T4, H1/H2, the libm twins, raster arithmetic).

tests/cpp/test_primitives.cpp checks the HOST compilation of procgen2_amd/csrc/pg_*.h; here the same functions run in
small gfx950 kernels (tests/hip/selftest.hip → procgen2_amd/lib/libpg_selftest.so) — including their
__HIP_DEVICE_COMPILE__ branches and the wave-cooperative forms that exist only on the device — and the raw results are
compared on the host with glibc (ctypes → libm) and the genuine libstdc++ (oracle/pgo_hooks.cpp)."""
import ctypes
import os
from ctypes import POINTER, c_float, c_int, c_int16, c_uint8, c_uint32, c_void_p

import numpy as np
import pytest

import oracle_util

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def st():
    try:
        import torch  # noqa: F401  (one HIP runtime per process: torch's first, see procgen2_amd/lib.py)
    except Exception:
        pass
    lib = ctypes.CDLL(os.path.join(ROOT, "procgen2_amd", "lib", "libpg_selftest.so"))
    V = c_void_p
    for name, args in (("pgst_sincos", [c_int, V, V, V]), ("pgst_atan2", [c_int, V, V, V, V]), ("pgst_div", [c_int, V, V, V, V]),
                       ("pgst_blend", [c_int, V, V, V, V, V]), ("pgst_box", [c_int, V, V, V, V]),
                       ("pgst_div255_pair", [c_int, V, V]),
                       ("pgst_mt", [c_int, V, c_int, c_int, V]), ("pgst_draws", [c_uint32, c_int, V, V, V, V, V, c_int, V, V]),
                       ("pgst_bulk", [c_uint32, c_int, c_int, V, V]), ("pgst_hash_script", [c_int, V, V, V]),
                       ("pgst_set_rounds", [c_int, V, V, V]), ("pgst_sort_equal", [c_int, V]),
                       ("pgst_replay", [c_int, c_int, V, c_int, V, V, c_int, V, V, V]),
                       ("pgst_rooms_40", [V, c_uint32, c_uint32, V, V, V, V, V]),
                       ("pgst_rooms_20", [V, c_uint32, c_uint32, V, V, V, V, V]),
                       ("pgst_rooms_45", [V, c_uint32, c_uint32, V, V, V, V, V])):
        getattr(lib, name).argtypes = args
        getattr(lib, name).restype = c_int
    assert lib.pgst_device_count() > 0
    return lib


def _p(a):
    return a.ctypes.data_as(c_void_p)


def _libm():
    m = ctypes.CDLL("libm.so.6")
    for name in ("sinf", "cosf", "atanf"):
        getattr(m, name).restype = c_float
        getattr(m, name).argtypes = [c_float]
    m.atan2f.restype = c_float
    m.atan2f.argtypes = [c_float, c_float]
    return m


def _host_map(fn, *arrays):
    return np.array([fn(*[float(v) for v in vals]) for vals in zip(*arrays)], np.float32)


def test_sinf_cosf_match_glibc(st):
    rng = np.random.default_rng(1)
    parts = [
        rng.uniform(-7.0, 7.0, 3_000_000).astype(np.float32),                       # the games' angles
        rng.uniform(-1000.0, 1000.0, 1_000_000).astype(np.float32),                 # accumulated rotations
        np.arange(0x39000000, 0x41000000, 41, dtype=np.uint32).view(np.float32),    # every binade up to 8.0, strided
        np.arange(0, 0x7f800000, 4099, dtype=np.uint32).view(np.float32),           # all magnitudes incl. huge
        -np.arange(0, 0x7f800000, 8191, dtype=np.uint32).view(np.float32),
        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, np.pi, np.pi / 2, np.pi / 4, 1e-40, -1e-40], np.float32),
    ]
    x = np.ascontiguousarray(np.concatenate(parts))
    assert x.size > 8_000_000
    s, c = np.zeros_like(x), np.zeros_like(x)
    assert st.pgst_sincos(x.size, _p(x), _p(s), _p(c)) == 0
    # host: glibc through numpy's float32 ufuncs is NOT glibc's sinf; call libm itself on a strided sample + all specials
    m = _libm()
    idx = np.concatenate([np.arange(0, x.size, 23), np.arange(x.size - 10, x.size)])
    want_s = _host_map(m.sinf, x[idx])
    want_c = _host_map(m.cosf, x[idx])
    assert np.array_equal(s[idx].view(np.uint32)[~np.isnan(want_s)], want_s.view(np.uint32)[~np.isnan(want_s)])
    assert np.array_equal(c[idx].view(np.uint32)[~np.isnan(want_c)], want_c.view(np.uint32)[~np.isnan(want_c)])
    assert np.array_equal(np.isnan(s[idx]), np.isnan(want_s)) and np.array_equal(np.isnan(c[idx]), np.isnan(want_c))


def test_sinf_cosf_full_sweep_against_host_twin(st, tmp_path):
    """All 8 M+ arguments, every bit: device results against the HOST compilation of the same header, which
    tests/cpp/test_primitives.cpp pins to glibc; (the libm ctypes loop above is too slow for all of them)."""
    import subprocess
    src = tmp_path / "twin.cpp"
    src.write_text('#include "pg_sincos.h"\n#include "pg_atan2.h"\nextern "C" void twin_sincos(int n, const float* x, float* s, float* c)'
                   '{ for (int i = 0; i < n; i++) { s[i] = pg::sc_sinf(x[i]); c[i] = pg::sc_cosf(x[i]); } }\n'
                   'extern "C" void twin_atan2(int n, const float* y, const float* x, float* o, float* o1)'
                   '{ for (int i = 0; i < n; i++) { o[i] = pg::at_atan2f(y[i], x[i]); o1[i] = pg::at_atanf(y[i]); } }\n')
    so = tmp_path / "twin.so"
    subprocess.run(["g++", "-std=gnu++17", "-O2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC",
                    "-I" + os.path.join(ROOT, "procgen2_amd", "csrc"), str(src), "-o", str(so)], check=True)
    twin = ctypes.CDLL(str(so))
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-10, 10, 4_000_000).astype(np.float32),
                        rng.integers(0, 1 << 32, 4_400_000, dtype=np.uint64).astype(np.uint32).view(np.float32)])
    x = np.ascontiguousarray(x)
    s, c, hs, hc = (np.zeros_like(x) for _ in range(4))
    assert st.pgst_sincos(x.size, _p(x), _p(s), _p(c)) == 0
    twin.twin_sincos(x.size, _p(x), _p(hs), _p(hc))
    ok = ~np.isnan(hs)
    assert np.array_equal(s.view(np.uint32)[ok], hs.view(np.uint32)[ok]) and np.array_equal(np.isnan(s), np.isnan(hs))
    ok = ~np.isnan(hc)
    assert np.array_equal(c.view(np.uint32)[ok], hc.view(np.uint32)[ok]) and np.array_equal(np.isnan(c), np.isnan(hc))
    # atan2f / atanf the same way: level-sized vectors and arbitrary bit patterns
    y = np.concatenate([rng.uniform(-40, 40, 4_000_000).astype(np.float32),
                        rng.integers(0, 1 << 32, 4_000_000, dtype=np.uint64).astype(np.uint32).view(np.float32)])
    xx = np.concatenate([rng.uniform(-40, 40, 4_000_000).astype(np.float32),
                         rng.integers(0, 1 << 32, 4_000_000, dtype=np.uint64).astype(np.uint32).view(np.float32)])
    y, xx = np.ascontiguousarray(y), np.ascontiguousarray(xx)
    a2, a1, h2, h1 = (np.zeros_like(y) for _ in range(4))
    assert st.pgst_atan2(y.size, _p(y), _p(xx), _p(a2), _p(a1)) == 0
    twin.twin_atan2(y.size, _p(y), _p(xx), _p(h2), _p(h1))
    for got, want in ((a2, h2), (a1, h1)):
        ok = ~np.isnan(want)
        assert np.array_equal(got.view(np.uint32)[ok], want.view(np.uint32)[ok])
        assert np.array_equal(np.isnan(got), np.isnan(want))


def test_atan2f_matches_glibc(st):
    rng = np.random.default_rng(3)
    y = rng.uniform(-40, 40, 400_000).astype(np.float32)
    x = rng.uniform(-40, 40, 400_000).astype(np.float32)
    specials = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, 1e-40, -1e-40, 3e38, 0.5], np.float32)
    sy, sx = np.meshgrid(specials, specials)
    y = np.ascontiguousarray(np.concatenate([y, sy.ravel()]))
    x = np.ascontiguousarray(np.concatenate([x, sx.ravel()]))
    a2, a1 = np.zeros_like(y), np.zeros_like(y)
    assert st.pgst_atan2(y.size, _p(y), _p(x), _p(a2), _p(a1)) == 0
    m = _libm()
    assert np.array_equal(a2.view(np.uint32), _host_map(m.atan2f, y, x).view(np.uint32))
    assert np.array_equal(a1.view(np.uint32), _host_map(m.atanf, y).view(np.uint32))


def test_mt19937_lane_and_wave_forms_match_libstdcxx(st):
    L = oracle_util.oracle()
    seeds = np.array([0, 1, 7, 123, 4294967291, 0xDEADBEEF, 5489, 65535], np.uint32)
    n = 1300  # crosses two regenerations of the 624 words
    want = np.zeros((seeds.size, n), np.uint32)
    L.pgo_hook_mt(seeds.size, _p(seeds), n, _p(want))
    assert want[3, :3].tolist() == [2991312382, 3062119789, 1228959102]  # SURVEY.md T1 known answers (seed 123)
    assert want[1, 623] == 2006116153 and want[1, 624] == 1104314680 and want[4, 0] == 1844333030
    for wave in (0, 1):
        got = np.zeros_like(want)
        assert st.pgst_mt(seeds.size, _p(seeds), n, wave, _p(got)) == 0
        assert np.array_equal(got, want), "wave form" if wave else "lane form"


def test_distributions_match_libstdcxx(st):
    L = oracle_util.oracle()
    rng = np.random.default_rng(4)
    n = 4000
    kind = (rng.random(n) < 0.4).astype(np.uint8)
    lo = rng.integers(-50, 50, n).astype(np.int32)
    span = rng.choice([0, 1, 2, 3, 9, 19, 48, 100, 1599, 65535, 1 << 20, (1 << 31) - 60], n)
    hi = (lo.astype(np.int64) + span).astype(np.int32)
    fa = rng.choice(np.array([0.0, -1.0, 0.7, -2.0], np.float32), n)
    fb = (fa + rng.choice(np.array([1.0, 2.0, 0.5, 4.0], np.float32), n)).astype(np.float32)
    for seed in (123, 7, 4294967291):
        wi, wf = np.zeros(n, np.int32), np.zeros(n, np.float32)
        L.pgo_hook_draws(seed, n, _p(kind), _p(lo), _p(hi), _p(fa), _p(fb), _p(wi), _p(wf))
        for wave in (0, 1):
            gi, gf = np.zeros(n, np.int32), np.zeros(n, np.float32)
            assert st.pgst_draws(seed, n, _p(kind), _p(lo), _p(hi), _p(fa), _p(fb), wave, _p(gi), _p(gf)) == 0
            assert np.array_equal(gi, wi) and np.array_equal(gf.view(np.uint32), wf.view(np.uint32)), (seed, wave)
    # SURVEY.md T2 known answers, seed 123, in this order
    kind = np.array([0, 0, 0, 1, 1, 1], np.uint8)
    lo, hi = np.array([1, 0, 0, 0, 0, 0], np.int32), np.array([3, 48, 19, 0, 0, 0], np.int32)
    fa, fb = np.array([0, 0, 0, 0.0, -1.0, 0.7], np.float32), np.array([0, 0, 0, 1.0, 1.0, 1.2], np.float32)
    gi, gf = np.zeros(6, np.int32), np.zeros(6, np.float32)
    assert st.pgst_draws(123, 6, _p(kind), _p(lo), _p(hi), _p(fa), _p(fb), 1, _p(gi), _p(gf)) == 0
    assert gi[:3].tolist() == [3, 34, 5] and gf[3:].view(np.uint32).tolist() == [0x3EDB608B, 0xBF0BDA20, 0x3F85D10F]
    # bulk canonical draws (wave_draws): 1 600 cells starting mid-block, stream position afterwards
    for seed, skip, count in ((9, 0, 1600), (10, 500, 1600), (11, 623, 2025), (12, 100, 64)):
        want, nxt = np.zeros(count, np.float32), c_uint32()
        L.pgo_hook_bulk(seed, skip, count, _p(want), ctypes.byref(nxt))
        got, gn = np.zeros(count, np.float32), c_uint32()
        assert st.pgst_bulk(seed, skip, count, _p(got), ctypes.byref(gn)) == 0
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and gn.value == nxt.value


def test_unordered_set_order_twins_match_libstdcxx(st):
    L = oracle_util.oracle()
    rng = np.random.default_rng(5)
    # serial twin (pg_order.h HashOrder, device hash_mod): insert / erase / clear scripts over up to 1 000 ids
    for n_ops, key_max in ((300, 40), (2000, 1000), (1500, 300), (4000, 1000)):
        ops = rng.choice([0, 0, 0, 1, 2], n_ops, p=[0.25, 0.25, 0.25, 0.245, 0.005]).astype(np.int32)
        keys = rng.integers(0, key_max, n_ops).astype(np.int32)
        want, got = np.zeros(n_ops, np.uint32), np.zeros(n_ops, np.uint32)
        L.pgo_hook_hash_script(n_ops, _p(ops), _p(keys), _p(want))
        assert st.pgst_hash_script(n_ops, _p(ops), _p(keys), _p(got)) == 0
        assert np.array_equal(got, want), (n_ops, key_max)
    # T3 known answer: fresh set, insert 0..29
    ops, keys = np.zeros(30, np.int32), np.arange(30, dtype=np.int32)
    want = np.zeros(30, np.uint32)
    L.pgo_hook_hash_script(30, _p(ops), _p(keys), _p(want))
    got = np.zeros(30, np.uint32)
    assert st.pgst_hash_script(30, _p(ops), _p(keys), _p(got)) == 0 and np.array_equal(got, want)
    # closed form (pg_setorder.h wave_set_order): rounds of clear() + n distinct inserts, bucket array carried over
    for rounds in ([30], [6, 40, 5], [1600, 900, 1600, 20, 1300], [13, 14, 29, 30, 59, 60, 127, 128, 257, 258, 541, 542, 1109, 1110],
                   [int(v) for v in rng.integers(1, 1600, 12)]):
        counts = np.array(rounds, np.int32)
        keys_in = np.concatenate([rng.permutation(2025)[:c] for c in rounds]).astype(np.int16)
        want, got = np.zeros_like(keys_in), np.zeros_like(keys_in)
        L.pgo_hook_set_rounds(counts.size, _p(counts), _p(keys_in), _p(want))
        assert st.pgst_set_rounds(counts.size, _p(counts), _p(keys_in), _p(got)) == 0
        assert np.array_equal(got, want), rounds
    first = np.arange(30, dtype=np.int16)  # 29 12 11 … 0 13 14 … 28
    got = np.zeros(30, np.int16)
    assert st.pgst_set_rounds(1, _p(np.array([30], np.int32)), _p(first), _p(got)) == 0
    assert got.tolist() == [29] + list(range(12, -1, -1)) + list(range(13, 29))


def test_equal_key_sort_permutation_matches_std_sort(st):
    L = oracle_util.oracle()
    for n in list(range(1, 72)) + [100, 128, 150, 198, 208]:
        want, got = np.zeros(n, np.int32), np.zeros(n, np.int32)
        L.pgo_hook_sort_equal(n, _p(want))
        assert st.pgst_sort_equal(n, _p(got)) == 0
        assert np.array_equal(got, want), n
    got = np.zeros(17, np.int32)
    st.pgst_sort_equal(17, _p(got))
    assert got.tolist() == [8, 16, 15, 14, 13, 12, 11, 10, 9, 0, 7, 6, 5, 4, 3, 2, 1]


def test_integer_raster_arithmetic_and_aabb(st):
    rng = np.random.default_rng(6)
    n = 4_000_000
    a = rng.integers(0, 1 << 22, n).astype(np.int32)
    b = np.where(rng.random(n) < 0.5, rng.integers(1, 700, n), rng.integers(1, 1 << 22, n)).astype(np.int32)
    q, hm = np.zeros(n, np.int32), np.zeros(n, np.int32)
    assert st.pgst_div(n, _p(a), _p(b), _p(q), _p(hm)) == 0
    assert np.array_equal(q, a // b)
    assert np.array_equal(hm, (a & 0x7FFF) % (1 + (b & 0xFFF) % 4095))
    dst = rng.integers(0, 1 << 24, n, dtype=np.int64).astype(np.uint32)
    src = rng.integers(0, 1 << 32, n, dtype=np.int64).astype(np.uint32)
    al = rng.integers(0, 256, n).astype(np.int32)
    out, d255 = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    assert st.pgst_blend(n, _p(dst), _p(src), _p(al), _p(out), _p(d255)) == 0
    want255 = (np.arange(n, dtype=np.uint32) & 0xFFFF) // 255
    bad = np.nonzero(d255 != want255)[0]
    assert bad.size == 0, "div255: %d mismatches, first at x=%d: got %d want %d" % (
        bad.size, bad[0] & 0xFFFF, d255[bad[0]], want255[bad[0]])
    want = np.zeros(n, np.uint32)
    for sh in (0, 8, 16):
        s8, d8 = (src >> sh) & 0xFF, (dst >> sh) & 0xFF
        want |= ((s8 * al.astype(np.uint32)) // 255 + (d8 * (255 - al).astype(np.uint32)) // 255) << sh
    bad = np.nonzero(out != want)[0]
    assert bad.size == 0, "blend_px: %d mismatches, first dst=%08x src=%08x a=%d: got %08x want %08x" % (
        bad.size, dst[bad[0]], src[bad[0]], al[bad[0]], out[bad[0]], want[bad[0]])
    # floor(x / 255) two halves at once (pg_geom.h div255_pair, inside blend_px): every product of two bytes, in either
    # half, beside the extremes in the other
    xs = np.arange(65026, dtype=np.uint32)
    pairs = np.concatenate([xs | np.uint32(o << 16) for o in (0, 1, 254, 255, 65024, 65025)] +
                           [np.uint32(o) | (xs << 16) for o in (0, 1, 254, 255, 65024, 65025)]).astype(np.uint32)
    got = np.zeros_like(pairs)
    assert st.pgst_div255_pair(pairs.size, _p(pairs), _p(got)) == 0
    assert np.array_equal(got, ((pairs & 0xFFFF) // 255) | (((pairs >> 16) // 255) << 16))
    # H1 / H2 on the device against the REFERENCE's helpers.cpp where oracle/_ref travelled with the snapshot, else
    # against the oracle's restatement (itself pinned to helpers.cpp by tests/test_reference_pin.py)
    import ref_util
    import ref_util
    ra, rb = ref_util.rect_pairs(rng, 1_000_000)
    side = ref_util.Side("ref") if os.path.exists(os.path.join(ref_util.REF_DIR, "libref_ecs.so")) else ref_util.Side("pgo")
    h0, o0 = side.collisions(ra, rb)
    hit, ov = np.zeros(ra.shape[0], np.uint8), np.zeros((ra.shape[0], 4), np.float32)
    assert st.pgst_box(ra.shape[0], _p(ra), _p(rb), _p(hit), _p(ov)) == 0
    assert np.array_equal(hit, h0) and np.array_equal(ov.view(np.uint32), o0.view(np.uint32))


def _sprite_scene(rng, n_draws, crowded):
    """Random textures (alpha 0 / 255 / in between), a random opaque target and a list of integer draw calls: small
    and large, clipped by the target's edges, flipped, alpha-modulated, some rotated; `crowded` keeps the small ones
    in one corner so that consecutive draws overlap pixel for pixel (what wave_order() in pg_render.h is about)."""
    n_tex = 6
    tw = rng.integers(3, 40, n_tex).astype(np.int32)
    th = rng.integers(3, 40, n_tex).astype(np.int32)
    texels, desc, at = [], [], 0
    for t in range(n_tex):
        px = rng.integers(0, 256, (th[t], tw[t], 4)).astype(np.uint8)
        kind = rng.random((th[t], tw[t]))
        px[..., 3] = np.where(kind < 0.3, 0, np.where(kind < 0.7, 255, px[..., 3]))
        texels.append(px.reshape(-1, 4))
        desc.append((at, tw[t], th[t], 0))
        at += int(tw[t]) * int(th[t])
    rgba = np.concatenate(texels).astype(np.uint8)
    words = (rgba[:, 0].astype(np.uint32) | (rgba[:, 1].astype(np.uint32) << 8) | (rgba[:, 2].astype(np.uint32) << 16)
             | (rgba[:, 3].astype(np.uint32) << 24))
    bg = rng.integers(0, 1 << 24, 64 * 64, dtype=np.int64).astype(np.uint32)
    draws, deg = np.zeros((n_draws, 12), np.int32), np.zeros(n_draws, np.float64)
    for k in range(n_draws):
        t = int(rng.integers(0, n_tex))
        small = rng.random() < 0.75
        dw, dh = (int(rng.integers(1, 9)), int(rng.integers(1, 9))) if small else (int(rng.integers(9, 70)), int(rng.integers(9, 50)))
        if crowded and small:
            dx, dy = int(rng.integers(20, 30)), int(rng.integers(26, 38))  # across the row the two waves split at
        else:
            dx, dy = int(rng.integers(-12, 64)), int(rng.integers(-12, 64))
        rotated = rng.random() < 0.3
        if rotated:
            sx, sy, sw, sh, flip = 0, 0, int(tw[t]), int(th[t]), 0
            deg[k] = float(rng.uniform(-400.0, 400.0)) if rng.random() < 0.9 else 0.0
        else:
            sx, sy = int(rng.integers(0, tw[t])), int(rng.integers(0, th[t]))
            sw, sh = int(rng.integers(1, tw[t] - sx + 1)), int(rng.integers(1, th[t] - sy + 1))
            if rng.random() < 0.4:  # the whole texture: what a sprite wholly on the screen is, and what a stamp may replace
                sx, sy, sw, sh = 0, 0, int(tw[t]), int(th[t])
            flip = int(rng.integers(0, 3))
        mod = 255 if rng.random() < 0.5 else int(rng.choice([178, int(rng.integers(0, 256))]))
        if k and rng.random() < 0.25:  # the same sprite again elsewhere (same texture, size, modulation: shares a stamp)
            t, dw, dh, sx, sy, sw, sh, mod = (int(draws[k - 1][j]) for j in (0, 3, 4, 5, 6, 7, 8, 10))
            if draws[k - 1][11] and not rotated:
                flip = 0
            if rotated and (sx, sy, sw, sh) != (0, 0, int(tw[t]), int(th[t])):
                sx, sy, sw, sh = 0, 0, int(tw[t]), int(th[t])
        draws[k] = (t, dx, dy, dw, dh, sx, sy, sw, sh, flip, mod, 1 if rotated else 0)
    return tw, th, rgba, words, np.array(desc, np.int32), bg, draws, deg


@pytest.mark.parametrize("rot_in_groups", [0, 1, 2, 3])
def test_sprite_replay_matches_the_raster_spec(st, rot_in_groups):
    """pg_render.h wave_replay_rows — groups of small draws, large ones alone, rotated ones either way, each wave on
    the rows it owns — against oracle/pgo_raster.cpp's spec_blit on the same integer draw calls.  Bit 1 of the parameter:
    the draws that take their whole texture are replaced by pre-scaled, pre-multiplied stamps first (pg_stamps.h), as a
    render pre-pass would; the oracle draws the original textures."""
    hooks = oracle_util.oracle()
    hooks.pgo_hook_raster.argtypes = [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]
    hooks.pgo_hook_raster.restype = None
    for seed in range(60):
        rng = np.random.default_rng(1000 + seed)
        n_draws = int(rng.integers(0, 65))
        tw, th, rgba, words, desc, bg, draws, deg = _sprite_scene(rng, n_draws, crowded=seed % 2 == 0)
        want = np.zeros(64 * 64 * 3, np.uint8)
        hooks.pgo_hook_raster(len(tw), _p(tw), _p(th), _p(rgba), _p(bg), n_draws, _p(draws), _p(deg), _p(want))
        got = np.zeros(64 * 64 * 3, np.uint8)
        assert st.pgst_replay(rot_in_groups, len(tw), _p(desc), len(words), _p(words), _p(bg), n_draws, _p(draws), _p(deg),
                              _p(got)) == 0
        bad = np.nonzero((got != want).reshape(-1, 3).any(axis=1))[0]
        assert bad.size == 0, "seed %d: %d pixels differ, first (y=%d, x=%d): got %s want %s" % (
            seed, bad.size, bad[0] // 64, bad[0] % 64, got.reshape(-1, 3)[bad[0]], want.reshape(-1, 3)[bad[0]])


# ------------------------------------------------------------------------------------------------------------------
# The device held to outputs of the REFERENCE's own compiled sources (tests/golden/ref_fixtures.json, ref_aabb_ecs.npz:
# recorded by tests/golden/make_ref_fixtures.py from oracle/_ref) — no oracle in between.
# ------------------------------------------------------------------------------------------------------------------
def _golden(name):
    return os.path.join(ROOT, "tests", "golden", name)


def test_room_pipeline_matches_the_reference_room_generator(st):
    """caveflyer/room_generator.cpp:4-202 (= jumper's): two automaton updates, find_best_room's ITERATION ORDER (what
    `for (int i : best_room)` hands the level), find_path, expand_room's result — pg_rooms.h on the device, for the three
    world sizes the games' modes use."""
    import json
    with open(_golden("ref_fixtures.json")) as f:
        rooms = json.load(f)["rooms"]
    assert {r["gw"] for r in rooms} == {40, 20, 45}
    for r in rooms:
        side = r["gw"]
        n = side * side
        raw = np.array([int(c) for c in r["raw"]], np.uint8)
        cave, wide = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        best, path, counts = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(2, np.int32)
        fn = getattr(st, "pgst_rooms_%d" % side)
        assert fn(_p(raw), r["src_sel"], r["dst_sel"], _p(cave), _p(best), _p(path), _p(wide), _p(counts)) == 0
        assert "".join(str(int(v)) for v in cave) == r["cave"], "automaton"
        assert best[:counts[0]].tolist() == r["best_order"], "find_best_room iteration order (side %d)" % side
        if r["path"] and r["path"][0] != r["path"][-1]:
            assert path[:counts[1]].tolist() == r["path"], "find_path"
            assert sorted(np.nonzero(wide)[0].tolist()) == sorted(r["wide_order"]), "expand_room membership"


def test_aabb_helpers_match_the_reference_on_recorded_pairs(st):
    """helpers.cpp:40-108 check_collision / get_collision_overlap, 12 000 recorded pairs, results as bit patterns."""
    z = np.load(_golden("ref_aabb_ecs.npz"))
    a, b = np.ascontiguousarray(z["a"]), np.ascontiguousarray(z["b"])
    hit, ov = np.zeros(len(a), np.uint8), np.zeros((len(a), 4), np.float32)
    assert st.pgst_box(len(a), _p(a), _p(b), _p(hit), _p(ov)) == 0
    assert 0.05 < z["hit"].mean() < 0.95
    assert np.array_equal(hit, z["hit"])
    assert np.array_equal(ov.view(np.uint32), z["overlap"].view(np.uint32))


def _fnv(values):
    h = 2166136261
    for v in values:
        h = ((h ^ (int(v) & 0xFFFFFFFF)) * 16777619) & 0xFFFFFFFF
    return h


def test_set_orders_match_the_reference_coordinator_scripts(st):
    """ecs.cpp:3-83: the iteration order of the reference's three systems' entity sets after every operation of 24
    recorded create / destroy / remove-component / clear scripts (run back to back in one process: the sets keep their
    bucket arrays across clear()).  Each system's set is replayed on the device twin of std::unordered_set<int>
    (pg_order.h HashOrder) as the inserts / erases / clears the Coordinator performs, with the entity ids the reference
    handed out, and must list the same ids in the same order after every operation."""
    z = np.load(_golden("ref_aabb_ecs.npz"))
    n_scripts = int(z["n_scripts"])
    assert n_scripts >= 20
    # per system: the primitive stream, and for every (script, op) where to look and what to see
    prim = [([], []) for _ in range(3)]      # (ops, keys): 0 insert-if-absent, 1 erase-if-present, 2 clear
    look = [[] for _ in range(3)]            # (index into the stream, expected FNV of (count, ids…))
    for k in range(n_scripts):
        ops, args, ids, orders = z["ops%d" % k], z["args%d" % k], z["ids%d" % k], z["orders%d" % k]
        for sysn in range(3):                # ref_ecs_script starts with clear_entities()
            prim[sysn][0].append(2)
            prim[sysn][1].append(0)
        has_a, has_b, at = {}, {}, 0
        for op, arg, eid in zip(ops, args, ids):
            eid = int(eid)

            def put(sysn, code):
                prim[sysn][0].append(code)
                prim[sysn][1].append(eid)
            if op == 0:                      # create, add A if arg & 1, then B if arg & 2 (each add: signature changed)
                a = b = False
                for bit in (1, 2):
                    if arg & bit:
                        a, b = a or bit == 1, b or bit == 2
                        put(0, 0 if a else 1)            # SysA wants {A}
                        put(1, 0 if (a and b) else 1)    # SysAB wants {A, B}
                        put(2, 0)                        # the empty signature matches every entity
                has_a[eid], has_b[eid] = a, b
            elif op == 1 and eid >= 0:       # destroy: erased from every system
                for sysn in range(3):
                    put(sysn, 1)
                has_a.pop(eid), has_b.pop(eid)
            elif op == 2:                    # clear_entities
                for sysn in range(3):
                    prim[sysn][0].append(2)
                    prim[sysn][1].append(0)
                has_a.clear(), has_b.clear()
            elif op == 3 and eid >= 0:       # remove component B: signature changed
                has_b[eid] = False
                put(0, 0 if has_a[eid] else 1)
                put(1, 1)
                put(2, 0)
            for sysn in range(3):
                count = int(orders[at])
                look[sysn].append((len(prim[sysn][0]) - 1, _fnv([count] + orders[at + 1:at + 1 + count].tolist())))
                at += 1 + count
        assert at == len(orders)
    for sysn in range(3):
        ops = np.array(prim[sysn][0], np.int32)
        keys = np.array(prim[sysn][1], np.int32)
        assert keys.max() < 2048
        got = np.zeros(len(ops), np.uint32)
        assert st.pgst_hash_script(len(ops), _p(ops), _p(keys), _p(got)) == 0
        bad = [i for i, (where, want) in enumerate(look[sysn]) if got[where] != want]
        assert not bad, "system %d: order differs after operation %d of %d" % (sysn, bad[0], len(look[sysn]))
