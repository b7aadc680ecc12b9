"""Test-side access to oracle/_ref — the REFERENCE's own SDL-free sources compiled where they lie (oracle/Makefile: ref,
oracle/ref_driver.cpp) — and to the oracle's hooks of the same shape (oracle/pgo_hooks.cpp).  Checker only."""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_float, c_int, c_uint8, c_uint32

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
REFERENCE = "/root/reference"

PI = POINTER(c_int)
SIGS = {
    "maze_generate": (c_int, [c_uint32, c_int, c_int, c_int, PI, PI, PI, POINTER(c_uint32)]),
    "maze_level": (c_int, [c_uint32, c_int, PI, POINTER(c_uint32)]),
    "setmaze_generate": (c_int, [c_uint32, c_int, c_int, PI, POINTER(c_uint32)]),
    "rooms_update": (None, [c_int, c_int, PI, c_int]),
    "rooms_analyse": (c_int, [c_int, c_int, PI, PI, c_uint32, c_uint32, PI, PI, c_int, PI, PI]),
    "collisions": (None, [c_int, POINTER(c_float), POINTER(c_float), POINTER(c_uint8), POINTER(c_float)]),
    "ecs_script": (c_int, [c_int, PI, PI, PI, PI, c_int]),
}
FAMILY = {"maze_generate": "maze", "maze_level": "maze", "setmaze_generate": "setmaze", "rooms_update": "rooms",
          "rooms_analyse": "rooms", "collisions": "ecs", "ecs_script": "ecs"}


def have_reference_build():
    """oracle/_ref is built from /root/reference when that tree is present (this container); elsewhere a prebuilt copy
    that travelled with the snapshot is used if it is there."""
    if os.path.isdir(REFERENCE):
        import oracle_util
        oracle_util.build_oracle()  # (under the build lock; a no-op when everything is up to date)
    return all(os.path.exists(os.path.join(REF_DIR, "libref_%s.so" % f)) for f in set(FAMILY.values()))


class Side:
    """One implementation of the seven functions: prefix 'ref_' (oracle/_ref/libref_*.so) or 'pgo_hook_' (the oracle)."""

    def __init__(self, which):
        self.fn = {}
        if which == "ref":
            libs = {f: ctypes.CDLL(os.path.join(REF_DIR, "libref_%s.so" % f)) for f in set(FAMILY.values())}
            for name, (res, args) in SIGS.items():
                f = getattr(libs[FAMILY[name]], "ref_" + name)
                f.restype, f.argtypes = res, args
                self.fn[name] = f
        else:
            import oracle_util
            lib = oracle_util.oracle()
            for name, (res, args) in SIGS.items():
                f = getattr(lib, "pgo_hook_" + name)
                f.restype, f.argtypes = res, args
                self.fn[name] = f

    @staticmethod
    def _ip(a):
        return a.ctypes.data_as(PI)

    def maze_generate(self, seed, w, h, n_objects):
        grid = np.zeros((w + 2) * (h + 2), np.int32)
        free = np.zeros(w * h + 4, np.int32)
        n_free, nxt = c_int(), c_uint32()
        self.fn["maze_generate"](seed, w, h, n_objects, self._ip(grid), self._ip(free), ctypes.byref(n_free), ctypes.byref(nxt))
        return grid, free[:n_free.value].copy(), nxt.value

    def maze_level(self, seed, world_dim):
        grid = np.zeros((world_dim + 2) ** 2, np.int32)
        nxt = c_uint32()
        dim = self.fn["maze_level"](seed, world_dim, self._ip(grid), ctypes.byref(nxt))
        return dim, grid[:(dim + 2) ** 2].copy(), nxt.value

    def setmaze_generate(self, seed, dim, no_dead_ends):
        grid = np.zeros((dim + 2) ** 2, np.int32)
        nxt = c_uint32()
        self.fn["setmaze_generate"](seed, dim, int(no_dead_ends), self._ip(grid), ctypes.byref(nxt))
        return grid, nxt.value

    def rooms_update(self, gw, gh, grid, iters):
        g = np.ascontiguousarray(grid, np.int32).copy()
        self.fn["rooms_update"](gw, gh, self._ip(g), iters)
        return g

    def rooms_analyse(self, gw, gh, grid, src_sel, dst_sel, expand_n):
        g = np.ascontiguousarray(grid, np.int32)
        best = np.zeros(gw * gh, np.int32)
        path = np.zeros(gw * gh, np.int32)
        wide = np.zeros(gw * gh, np.int32)
        n_path, n_wide = c_int(), c_int()
        n = self.fn["rooms_analyse"](gw, gh, self._ip(g), self._ip(best), src_sel, dst_sel, self._ip(path),
                                     ctypes.byref(n_path), expand_n, self._ip(wide), ctypes.byref(n_wide))
        return best[:n].copy(), path[:n_path.value].copy(), wide[:n_wide.value].copy()

    def collisions(self, a, b):
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        n = a.shape[0]
        hit = np.zeros(n, np.uint8)
        ov = np.zeros((n, 4), np.float32)
        self.fn["collisions"](n, a.ctypes.data_as(POINTER(c_float)), b.ctypes.data_as(POINTER(c_float)),
                              hit.ctypes.data_as(POINTER(c_uint8)), ov.ctypes.data_as(POINTER(c_float)))
        return hit, ov

    def ecs_script(self, ops, args, cap=1 << 22):
        ops = np.ascontiguousarray(ops, np.int32)
        args = np.ascontiguousarray(args, np.int32)
        ids = np.zeros(len(ops), np.int32)
        orders = np.zeros(cap, np.int32)
        w = self.fn["ecs_script"](len(ops), self._ip(ops), self._ip(args), self._ip(ids), self._ip(orders), cap)
        assert w >= 0, "order buffer too small"
        return ids, orders[:w].copy()


def random_cave(rng, gw, gh, p_wall=0.5):
    """A cave grid as caveflyer/tilemap.cpp seeds it (independent wall bits); numpy's own generator — this is test
    input, the same for both sides."""
    return (rng.random(gw * gh) < p_wall).astype(np.int32)


def rect_pairs(rng, n):
    """Rectangles as the games build them (positions on a 1/4 … 1/64 lattice so that exact touching is common, sizes from
    the games' hit boxes) plus raw random ones."""
    base = rng.integers(-64, 64 * 64, size=(n, 2)).astype(np.float32) / np.float32(64.0)
    size = rng.choice(np.array([0.02, 0.1, 0.15, 0.25, 0.5, 0.8, 0.95, 1.0, 2.0], np.float32), size=(n, 2))
    a = np.concatenate([base, size], axis=1)
    off = rng.integers(-96, 97, size=(n, 2)).astype(np.float32) / np.float32(64.0)
    size_b = rng.choice(np.array([0.02, 0.1, 0.25, 0.5, 1.0, 1.0, 1.0, 3.0], np.float32), size=(n, 2))
    b = np.concatenate([base + off, size_b], axis=1)
    k = n // 4
    a[:k] = rng.normal(0, 10, size=(k, 4)).astype(np.float32)
    b[:k] = rng.normal(0, 10, size=(k, 4)).astype(np.float32)
    a[:k, 2:] = np.abs(a[:k, 2:])
    b[:k, 2:] = np.abs(b[:k, 2:])
    return a.astype(np.float32), b.astype(np.float32)


def ecs_random_script(rng, n_ops):
    ops = rng.choice([0, 0, 0, 1, 1, 3, 2], size=n_ops, p=[0.2, 0.2, 0.2, 0.15, 0.15, 0.08, 0.02]).astype(np.int32)
    args = rng.integers(0, 1 << 20, size=n_ops).astype(np.int32)
    args[ops == 0] = rng.integers(1, 4, size=int((ops == 0).sum()))
    return ops, args
