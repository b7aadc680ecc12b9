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

_cached = {}


class EngineError(RuntimeError):
    pass


def load(path=None):
    """Load libprocgen2_hip.so (building is `python -m procgen2_amd.build` / __graft_entry__.build())."""
    path = os.path.abspath(path or DEFAULT_LIB)
    if path in _cached:
        return _cached[path]
    if not os.path.exists(path):
        raise EngineError(
            "HIP engine library not found at %s — run `python -m procgen2_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback." % path)
    lib = ctypes.CDLL(path)
    P = c_void_p
    proto = {
        "pgv_game_name": (c_char_p, [c_int32]),
        "pgv_game_id": (c_int32, [c_char_p]),
        "pgv_make": (c_int32, [c_char_p, c_int32, c_int32, c_uint32, c_int32, P, POINTER(P)]),
        "pgv_make_levels": (c_int32, [c_char_p, c_int32, c_int32, c_uint32, c_int32, P, c_int32, c_int32, POINTER(P)]),
        "pgv_close": (None, [P]),
        "pgv_reset": (c_int32, [P, P, P]),
        "pgv_step": (c_int32, [P, P]),
        "pgv_step_synthetic": (c_int32, [P, c_uint32]),
        "pgv_synthetic_action": (c_int32, [c_uint32, c_uint32, c_uint32]),
        "pgv_step_host": (c_int32, [P, P]),
        "pgv_reset_host": (c_int32, [P, P, P]),
        "pgv_decode_png": (c_int32, [c_char_p, POINTER(c_int32), POINTER(c_int32), P, ctypes.c_int64]),
        "pgv_sync": (c_int32, [P]),
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
        "pgv_timed_steps": (c_int32, [P, c_int32, c_uint32, POINTER(c_double), POINTER(c_double)]),
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
    "pgv_game_name", "pgv_game_id", "pgv_make", "pgv_make_levels", "pgv_close", "pgv_reset", "pgv_step", "pgv_step_synthetic",
    "pgv_synthetic_action", "pgv_step_host", "pgv_reset_host", "pgv_decode_png", "pgv_sync", "pgv_obs", "pgv_reward", "pgv_done", "pgv_bind_outputs", "pgv_num_envs",
    "pgv_device", "pgv_stream", "pgv_copy_out", "pgv_render_frame", "pgv_snapshot_bytes", "pgv_save_state", "pgv_load_state", "pgv_timed_steps", "pgv_set_debug", "pgv_dump_state", "pgv_dump_tiles",
    "pgv_last_error",
]
EXPORTED_CENV_SYMBOLS = [
    "make_data", "reset_data", "step_data", "render_data", "cenv_get_env_version", "cenv_make", "cenv_reset",
    "cenv_step", "cenv_render", "cenv_close",
]
