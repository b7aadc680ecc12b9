"""Test-side access to the CPU oracle (oracle/libpgoracle.so) — the checker, never the product.

Textures are decoded with PIL and handed to the oracle as raw RGBA, so the oracle does not depend on
the engine's own PNG decoder.
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_uint8, c_uint32, c_void_p

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.environ.get("PGO_ORACLE_LIB") or os.path.join(ORACLE_DIR, "libpgoracle.so")  # (sanitizer builds)
ASSETS = os.path.join(ROOT, "procgen2_amd", "assets")
OBS_BYTES = 12288

_lib = None
_registered = set()


def build_oracle():
    """`make -C oracle` under a file lock: several processes ask at once (pytest workers, the spawned ranks of the
    distributed tests) and only one of them may be compiling; with everything up to date make does nothing."""
    import fcntl
    with open(os.path.join(ORACLE_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def oracle():
    global _lib
    if _lib is not None:
        return _lib
    # make's own dependency check is cheap: never run the tests against a checker older than its sources.  (The
    # default library only; a sanitizer build named by PGO_ORACLE_LIB is the caller's business.)
    if not os.environ.get("PGO_ORACLE_LIB") or not os.path.exists(ORACLE_LIB):
        build_oracle()
    L = ctypes.CDLL(ORACLE_LIB)
    L.pgo_put_texture.argtypes = [c_char_p, c_int, c_int, c_void_p]
    L.pgo_make.restype = c_void_p
    L.pgo_make.argtypes = [c_char_p, c_uint32, c_int]
    L.pgo_close.argtypes = [c_void_p]
    L.pgo_reset.argtypes = [c_void_p, c_int, c_int32]
    L.pgo_step.argtypes = [c_void_p, c_int]
    L.pgo_reward.restype = c_float
    L.pgo_reward.argtypes = [c_void_p]
    L.pgo_terminated.argtypes = [c_void_p]
    L.pgo_truncated.argtypes = [c_void_p]
    L.pgo_obs.restype = POINTER(c_uint8)
    L.pgo_obs.argtypes = [c_void_p]
    L.pgo_render_frame.argtypes = [c_void_p, c_int, c_int, c_void_p]
    L.pgo_dump_state.argtypes = [c_void_p, POINTER(c_float), c_int]
    L.pgo_dump_tiles.argtypes = [c_void_p, POINTER(c_uint8), c_int]
    L.pgo_trace.argtypes = [c_char_p, c_uint32, c_int, POINTER(c_uint32), POINTER(c_int), POINTER(c_double),
                            POINTER(c_int), c_int]
    L.pgo_trace_flags.argtypes = [c_char_p, c_uint32, c_int, c_uint32, POINTER(c_uint32), POINTER(c_int), POINTER(c_double),
                                  POINTER(c_int), c_int]
    for name, args in (("pgo_hook_mt", [c_int, c_void_p, c_int, c_void_p]),
                       ("pgo_hook_draws", [c_uint32, c_int] + [c_void_p] * 7),
                       ("pgo_hook_bulk", [c_uint32, c_int, c_int, c_void_p, c_void_p]),
                       ("pgo_hook_hash_script", [c_int, c_void_p, c_void_p, c_void_p]),
                       ("pgo_hook_set_rounds", [c_int, c_void_p, c_void_p, c_void_p]),
                       ("pgo_hook_sort_equal", [c_int, c_void_p])):
        getattr(L, name).argtypes = args
        getattr(L, name).restype = None
    L.pgo_synthetic_action.argtypes = [c_uint32, c_uint32, c_uint32]
    L.pgo_vec_make.restype = c_void_p
    L.pgo_vec_make.argtypes = [c_char_p, c_int, c_uint32, c_int, c_int]
    L.pgo_vec_make_levels.restype = c_void_p
    L.pgo_vec_make_levels.argtypes = [c_char_p, c_int, c_uint32, c_int, c_int, c_int, c_int]
    L.pgo_vec_make_config.restype = c_void_p
    L.pgo_vec_make_config.argtypes = [c_char_p, c_int, c_uint32, c_int, c_int, c_int, c_int, c_int]
    L.pgo_vec_make_flags.restype = c_void_p
    L.pgo_vec_make_flags.argtypes = [c_char_p, c_int, c_uint32, c_int, c_int, c_int, c_int, c_int, c_uint32]
    L.pgo_vec_make_threads.restype = c_void_p
    L.pgo_vec_make_threads.argtypes = [c_char_p, c_int, c_uint32, c_int, c_int, c_int, c_int, c_int, c_uint32, c_int]
    L.pgo_vec_reset_threads.argtypes = [c_void_p, c_int]
    L.pgo_make_mode.restype = c_void_p
    L.pgo_make_mode.argtypes = [c_char_p, c_uint32, c_int, c_int]
    L.pgo_resolve_mode.argtypes = [c_char_p, c_int]
    L.pgo_present.argtypes = [c_void_p]
    L.pgo_vec_reset.argtypes = [c_void_p, c_void_p, c_void_p]
    L.pgo_vec_close.argtypes = [c_void_p]
    L.pgo_vec_set_render.argtypes = [c_void_p, c_int]
    L.pgo_vec_step.argtypes = [c_void_p, c_void_p, c_uint32, c_int, c_int, c_void_p, c_void_p, c_void_p]
    L.pgo_vec_obs.argtypes = [c_void_p, c_void_p]
    L.pgo_vec_dump_state.argtypes = [c_void_p, c_int, POINTER(c_float), c_int]
    L.pgo_vec_dump_tiles.argtypes = [c_void_p, c_int, POINTER(c_uint8), c_int]
    L.pgo_vec_bench.restype = c_double
    L.pgo_vec_bench.argtypes = [c_void_p, c_int, c_uint32, c_int]
    _lib = L
    return L


def game_texture_names(game):
    """Relative asset paths (under assets/) a game loads — restated in tools/vendor_assets.py."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("vendor_assets", os.path.join(ROOT, "tools", "vendor_assets.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.GAMES[game]()


def register_textures(game):
    """Decode the game's PNGs with PIL and hand them to the oracle under the reference's asset paths."""
    from PIL import Image
    L = oracle()
    for rel in game_texture_names(game):
        if rel in _registered:
            continue
        im = Image.open(os.path.join(ASSETS, rel)).convert("RGBA")
        arr = np.ascontiguousarray(np.asarray(im, dtype=np.uint8))
        L.pgo_put_texture(("assets/" + rel).encode(), im.size[0], im.size[1], arr.ctypes.data_as(c_void_p))
        _registered.add(rel)


class OracleVec:
    """N oracle envs stepped in lock-step with the engine's auto-reset policy (oracle/pgo_api.cpp)."""

    def __init__(self, game, n, seed_base=1, env_offset=0, render=True, num_levels=0, start_level=0, mode=0, game_flags=0,
                 threads=1):
        if render:
            register_textures(game)
        self.L = oracle()
        self.n = n
        self.env_offset = env_offset
        self.threads = threads
        if threads > 1:  # the envs made (and, in reset(), given their first level) by several threads: full-size tests
            self.h = self.L.pgo_vec_make_threads(game.encode(), n, seed_base, env_offset, 1 if render else 0, num_levels,
                                                 start_level, mode, game_flags, threads)
        else:
            self.h = self.L.pgo_vec_make_flags(game.encode(), n, seed_base, env_offset, 1 if render else 0, num_levels,
                                               start_level, mode, game_flags)
        assert self.h, "oracle does not know game %r" % game
        self.obs = np.zeros((n, OBS_BYTES), np.uint8)
        self.reward = np.zeros(n, np.float32)
        self.done = np.zeros(n, np.uint8)

    def reset(self, mask=None, seeds=None):
        m = s = None
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
            m = mask.ctypes.data_as(c_void_p)
        if seeds is not None:
            seeds = np.ascontiguousarray(seeds, dtype=np.int32)
            s = seeds.ctypes.data_as(c_void_p)
        if m is None and s is None and self.threads > 1:
            self.L.pgo_vec_reset_threads(self.h, self.threads)
        else:
            self.L.pgo_vec_reset(self.h, m, s)
        return self.reset_obs()

    def reset_obs(self):
        self.L.pgo_vec_obs(self.h, self.obs.ctypes.data_as(c_void_p))
        return self.obs

    def step(self, actions=None, run_seed=0, threads=1):
        a = None
        if actions is not None:
            actions = np.ascontiguousarray(actions, dtype=np.int32)
            a = actions.ctypes.data_as(c_void_p)
        self.L.pgo_vec_step(self.h, a, run_seed, self.env_offset, threads, self.obs.ctypes.data_as(c_void_p),
                            self.reward.ctypes.data_as(c_void_p), self.done.ctypes.data_as(c_void_p))
        return self.obs, self.reward, self.done

    def set_render(self, on):
        """Drawing on / off from the next step on (logic is unaffected; obs is only meaningful for steps drawn)."""
        self.L.pgo_vec_set_render(self.h, 1 if on else 0)

    def state(self, env, cap=512):
        buf = (c_float * cap)()
        n = self.L.pgo_vec_dump_state(self.h, env, buf, cap)
        return np.array(buf[:min(n, cap)], np.float32)

    def tiles(self, env, cap=4096):
        buf = (c_uint8 * cap)()
        n = self.L.pgo_vec_dump_tiles(self.h, env, buf, cap)
        return np.array(buf[:n], np.uint8)

    def close(self):
        if self.h:
            self.L.pgo_vec_close(self.h)
            self.h = None
