"""ctypes binding of the engine's C ABI (include/procgen2_vec.h).

The shared library is the product; this module only declares prototypes.  It fails loudly when the
library has not been built — there is no Python or CPU fallback for the hot path.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int32, c_int64, c_uint8, c_uint32, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(HERE, "lib")
DEFAULT_LIB = os.path.join(LIB_DIR, "libprocgen2_hip.so")

OBS_BYTES = 12288
OBS_SHAPE = (64, 64, 3)
NUM_ACTIONS = 15

# include/procgen2_vec.h PGV_MODE_*
MODES = {"default": 0, "easy": 1, "hard": 2, "memory": 3, "extreme": 4}


class Config(ctypes.Structure):
    """include/procgen2_vec.h `pgv_config`."""
    _fields_ = [("struct_size", c_uint32), ("num_envs", c_int32), ("game", c_char_p), ("stream", c_void_p),
                ("device", c_int32), ("seed_base", c_uint32), ("env_offset", c_int32), ("num_levels", c_int32),
                ("start_level", c_int32), ("mode", c_int32), ("game_flags", c_uint32)]


def mode_id(mode):
    if mode is None:
        return 0
    if isinstance(mode, str):
        if mode not in MODES:
            raise ValueError("unknown distribution mode %r (one of %s)" % (mode, ", ".join(MODES)))
        return MODES[mode]
    return int(mode)


# include/procgen2_vec.h PGV_COINRUN_NO_*
COINRUN_NO_PIT, COINRUN_NO_CRATE, COINRUN_NO_DY, COINRUN_NO_MOBS = 1, 2, 4, 8
CHASER_FLOAT_ABS = JUMPER_FLOAT_ABS = 1  # include/procgen2_vec.h PGV_*_FLOAT_ABS


def make(lib, game, num_envs, device=0, seed_base=1, env_offset=0, stream=None, num_levels=0, start_level=0, mode=None,
         game_flags=0):
    """pgv_make_config → env handle (c_void_p)."""
    cfg = Config(ctypes.sizeof(Config), int(num_envs), game.encode(), stream, int(device), int(seed_base) & 0xFFFFFFFF,
                 int(env_offset), int(num_levels), int(start_level), mode_id(mode), int(game_flags))
    h = c_void_p()
    check(lib, lib.pgv_make_config(ctypes.byref(cfg), ctypes.byref(h)), "pgv_make")
    return h


_cached = {}


class EngineError(RuntimeError):
    pass


def load(path=None):
    """Load libprocgen2_hip.so (building is `python -m procgen2_amd.build` / __graft_entry__.build())."""
    path = os.path.abspath(path or os.environ.get("PROCGEN2_HIP_LIB") or DEFAULT_LIB)  # (the variable: experiment builds, tools/build_exp.py)
    if path in _cached:
        return _cached[path]
    if not os.path.exists(path):
        raise EngineError(
            "HIP engine library not found at %s — run `python -m procgen2_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback." % path)
    # One HIP runtime per process.  torch ships its own libamdhip64 under the same SONAME as /opt/rocm's, and the
    # dynamic loader hands whichever was loaded first to everyone who asks later.  The engine shares device pointers
    # and streams with torch (vec_env.py), so they must be the same runtime instance, and torch refuses to see a
    # device when it finds a foreign runtime already resident: load torch's first whenever torch is installed.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = ctypes.CDLL(path)
    P = c_void_p
    proto = {
        "pgv_game_name": (c_char_p, [c_int32]),
        "pgv_game_id": (c_int32, [c_char_p]),
        "pgv_make": (c_int32, [c_char_p, c_int32, c_int32, c_uint32, c_int32, P, POINTER(P)]),
        "pgv_make_levels": (c_int32, [c_char_p, c_int32, c_int32, c_uint32, c_int32, P, c_int32, c_int32, POINTER(P)]),
        "pgv_make_config": (c_int32, [POINTER(Config), POINTER(P)]),
        "pgv_game_modes": (c_uint32, [c_int32]),
        "pgv_mode": (c_int32, [P]),
        "pgv_close": (None, [P]),
        "pgv_reset": (c_int32, [P, P, P]),
        "pgv_step": (c_int32, [P, P]),
        "pgv_step_synthetic": (c_int32, [P, c_uint32]),
        "pgv_synthetic_action": (c_int32, [c_uint32, c_uint32, c_uint32]),
        "pgv_step_host": (c_int32, [P, P]),
        "pgv_reset_host": (c_int32, [P, P, P]),
        "pgv_decode_png": (c_int32, [c_char_p, POINTER(c_int32), POINTER(c_int32), P, ctypes.c_int64]),
        "pgv_sync": (c_int32, [P]),
        "pgv_generator_launches": (c_int64, [P]),
        "pgv_obs": (P, [P]),
        "pgv_reward": (P, [P]),
        "pgv_done": (P, [P]),
        "pgv_bind_outputs": (c_int32, [P, P, P, P]),
        "pgv_num_envs": (c_int32, [P]),
        "pgv_device": (c_int32, [P]),
        "pgv_stream": (P, [P]),
        "pgv_copy_out": (c_int32, [P, P, P, P]),
        "pgv_render_frame": (c_int32, [P, c_int32, c_int32, c_int32, P]),
        "pgv_snapshot_bytes": (c_int64, [P]),
        "pgv_save_state": (c_int32, [P, P, c_int64]),
        "pgv_load_state": (c_int32, [P, P, c_int64]),
        "pgv_step_synthetic_many": (c_int32, [P, c_int32, c_int32, c_uint32]),
        "pgv_timed_steps": (c_int32, [P, c_int32, c_uint32, POINTER(c_double), POINTER(c_double)]),
        "pgv_step_times": (c_int32, [P, c_int32, c_uint32, c_void_p, c_void_p]),
        "pgv_step_phases": (c_int32, [P, c_int32, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
        "pgv_step_phases_many": (c_int32, [P, c_int32, c_int32, c_uint32, c_void_p]),
        "pgv_set_debug": (c_int32, [P, c_int32]),
        "pgv_dump_state": (c_int32, [P, c_int32, POINTER(c_float), c_int32]),
        "pgv_dump_tiles": (c_int32, [P, c_int32, POINTER(c_uint8), c_int32]),
        "pgv_last_error": (c_char_p, []),
    }
    for name, (res, args) in proto.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _cached[path] = lib
    return lib


def check(lib, rc, what):
    if rc != 0:
        msg = lib.pgv_last_error()
        raise EngineError("%s failed: %s" % (what, msg.decode() if msg else "unknown error"))


EXPORTED_VEC_SYMBOLS = [
    "pgv_game_name", "pgv_game_id", "pgv_make", "pgv_make_levels", "pgv_make_config", "pgv_game_modes", "pgv_mode", "pgv_close", "pgv_reset", "pgv_step", "pgv_step_synthetic", "pgv_step_synthetic_many",
    "pgv_synthetic_action", "pgv_step_host", "pgv_reset_host", "pgv_decode_png", "pgv_sync", "pgv_generator_launches", "pgv_obs", "pgv_reward", "pgv_done", "pgv_bind_outputs", "pgv_num_envs",
    "pgv_device", "pgv_stream", "pgv_copy_out", "pgv_render_frame", "pgv_snapshot_bytes", "pgv_save_state", "pgv_load_state", "pgv_timed_steps", "pgv_step_times", "pgv_step_phases", "pgv_step_phases_many", "pgv_set_debug", "pgv_dump_state", "pgv_dump_tiles",
    "pgv_last_error",
]
EXPORTED_CENV_SYMBOLS = [
    "make_data", "reset_data", "step_data", "render_data", "cenv_get_env_version", "cenv_make", "cenv_reset",
    "cenv_step", "cenv_render", "cenv_close",
]
