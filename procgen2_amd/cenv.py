"""CEnv — Python binding of the single-env `cenv` C ABI, mirroring the reference wrapper's surface.

Reference: cenv/cenv.py:152-380 (`class CEnv(gymnasium.Env)`): same constructor arguments, same
method names and return shapes, same error behaviour ("Non-zero error code!"), same marshalling rules
(python int → INT option, float → DOUBLE option, int action → {"action", INT, 1}, dict of ndarrays →
one key_value per entry).  Works with any library exporting the ABI of include/procgen2_cenv.h — the
engine's libCoinRun.so / libMaze.so / libprocgen2_hip.so, or the reference's own toy env.

gymnasium is optional here (it is not installed in the build image): when importable, CEnv derives
from gymnasium.Env and spaces are gymnasium spaces; otherwise light stand-ins with the same fields.
"""
import ctypes
from ctypes import POINTER, Structure, Union, c_bool, c_char_p, c_double, c_float, c_int32, c_ubyte, c_void_p

import numpy as np

try:  # pragma: no cover - depends on the environment
    import gymnasium as _gym
    _Base = _gym.Env
except Exception:  # gymnasium absent: minimal stand-ins
    _gym = None
    _Base = object

# cenv.h:29-39
VALUE_INT, VALUE_FLOAT, VALUE_DOUBLE, VALUE_BYTE, SPACE_BOX, SPACE_MULTI_DISCRETE = range(6)
_CTYPE = [c_int32, c_float, c_double, c_ubyte, c_float, c_int32]
_DTYPE = [np.int32, np.float32, np.float64, np.uint8, np.float32, np.int32]
_PY_TO_VALUE = {int: VALUE_INT, float: VALUE_DOUBLE}  # cenv.py:39-42
_NP_TO_VALUE = {np.dtype("int32"): VALUE_INT, np.dtype("float32"): VALUE_FLOAT, np.dtype("float64"): VALUE_DOUBLE,
                np.dtype("uint8"): VALUE_BYTE}


class Value(Union):
    _fields_ = [("i", c_int32), ("f", c_float), ("d", c_double), ("b", c_ubyte)]


class ValueBuffer(Union):
    _fields_ = [("i", POINTER(c_int32)), ("f", POINTER(c_float)), ("d", POINTER(c_double)), ("b", POINTER(c_ubyte))]


class KeyValue(Structure):
    _fields_ = [("key", c_char_p), ("value_type", c_int32), ("value_buffer_size", c_int32),
                ("value_buffer", ValueBuffer)]


class Option(Structure):
    _fields_ = [("name", c_char_p), ("value_type", c_int32), ("value", Value)]


class MakeData(Structure):
    _fields_ = [("observation_spaces_size", c_int32), ("observation_spaces", POINTER(KeyValue)),
                ("action_spaces_size", c_int32), ("action_spaces", POINTER(KeyValue))]


class ResetData(Structure):
    _fields_ = [("observations_size", c_int32), ("observations", POINTER(KeyValue)), ("infos_size", c_int32),
                ("infos", POINTER(KeyValue))]


class StepData(Structure):
    _fields_ = [("observations_size", c_int32), ("observations", POINTER(KeyValue)), ("reward", Value),
                ("terminated", c_bool), ("truncated", c_bool), ("infos_size", c_int32), ("infos", POINTER(KeyValue))]


class RenderData(Structure):
    _fields_ = [("value_type", c_int32), ("value_buffer_width", c_int32), ("value_buffer_height", c_int32),
                ("value_buffer_channels", c_int32), ("value_buffer", ValueBuffer)]


class Box:
    """Stand-in for gymnasium.spaces.Box when gymnasium is absent."""

    def __init__(self, low, high):
        self.low = np.asarray(low, np.float32)
        self.high = np.asarray(high, np.float32)
        self.shape = self.low.shape
        self.dtype = np.float32

    def __repr__(self):
        return "Box(%s, %s, %s)" % (self.low, self.high, self.shape)


class MultiDiscrete:
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, np.int64)
        self.shape = self.nvec.shape

    def __repr__(self):
        return "MultiDiscrete(%s)" % (self.nvec,)


def _copy_buffer(kv):
    """cenv.py:114-132 `_make_nd_array(..., own_data=True)`: a flat copy of `value_buffer_size` elements."""
    vt = int(kv.value_type)
    n = int(kv.value_buffer_size)
    dtype = np.dtype(_DTYPE[vt])
    addr = ctypes.cast(kv.value_buffer.b, c_void_p).value
    if n == 0 or not addr:
        return np.zeros(0, dtype)
    raw = (ctypes.c_char * (n * dtype.itemsize)).from_address(addr)
    return np.frombuffer(raw, dtype=dtype, count=n).copy()


def _space(kv):
    arr = _copy_buffer(kv)
    if int(kv.value_type) == SPACE_MULTI_DISCRETE:
        return _gym.spaces.MultiDiscrete(arr) if _gym else MultiDiscrete(arr)
    half = len(arr) // 2  # cenv.py:225 — BOX buffer = [low..., high...]
    return _gym.spaces.Box(arr[:half], arr[half:]) if _gym else Box(arr[:half], arr[half:])


def _options(options):
    if options is None:
        return None, 0
    arr = (Option * len(options))()
    for i, (k, v) in enumerate(options.items()):
        vt = _PY_TO_VALUE[type(v)]  # KeyError for unsupported types, as in the reference
        arr[i].name = k.encode("ascii")
        arr[i].value_type = vt
        if vt == VALUE_INT:
            arr[i].value.i = v
        else:
            arr[i].value.d = v
    return arr, len(options)


class CEnv(_Base):
    metadata = {"render_modes": ["human", "single_rgb_array"]}

    def __init__(self, lib_file_path, render_mode=None, options=None):
        self.lib = ctypes.CDLL(lib_file_path)
        L = self.lib
        L.cenv_get_env_version.restype = c_int32
        L.cenv_make.argtypes = [c_char_p, POINTER(Option), c_int32]
        L.cenv_make.restype = c_int32
        L.cenv_reset.argtypes = [POINTER(Option), c_int32]
        L.cenv_reset.restype = c_int32
        L.cenv_step.argtypes = [POINTER(KeyValue), c_int32]
        L.cenv_step.restype = c_int32
        L.cenv_render.restype = c_int32
        L.cenv_close.restype = None
        self.c_make_data = MakeData.in_dll(L, "make_data")
        self.c_reset_data = ResetData.in_dll(L, "reset_data")
        self.c_step_data = StepData.in_dll(L, "step_data")
        self.c_render_data = RenderData.in_dll(L, "render_data")

        opts, n = _options(options)
        if L.cenv_make(("" if render_mode is None else render_mode).encode("ascii"), opts, n) != 0:
            raise Exception("Non-zero error code!")
        md = self.c_make_data
        self.observation_space = {md.observation_spaces[i].key.decode(): _space(md.observation_spaces[i])
                                  for i in range(md.observation_spaces_size)}
        self.action_space = {md.action_spaces[i].key.decode(): _space(md.action_spaces[i])
                             for i in range(md.action_spaces_size)}

    def version(self):
        return int(self.lib.cenv_get_env_version())

    def step(self, action):
        keep = []
        if type(action) is int:
            c_action = c_int32(action)
            buf = ValueBuffer()
            buf.i = ctypes.pointer(c_action)
            c_actions = (KeyValue * 1)(KeyValue(b"action", VALUE_INT, 1, buf))
            num = 1
            keep.append(c_action)
        elif type(action) is dict:
            num = len(action)
            c_actions = (KeyValue * num)()
            for i, (k, v) in enumerate(action.items()):
                v = np.ascontiguousarray(v)
                keep.append(v)
                vt = _NP_TO_VALUE[v.dtype]
                c_actions[i].key = k.encode("ascii")
                c_actions[i].value_type = vt
                c_actions[i].value_buffer_size = len(v)
                c_actions[i].value_buffer.b = ctypes.cast(v.ctypes.data, POINTER(c_ubyte))
        else:
            raise Exception("Unrecognized action type! Supported are: int, np.array, Dict[np.array]")
        if self.lib.cenv_step(c_actions, num) != 0:
            raise Exception("Non-zero error code!")
        sd = self.c_step_data
        observation = {sd.observations[i].key.decode(): _copy_buffer(sd.observations[i])
                       for i in range(sd.observations_size)}
        info = {sd.infos[i].key.decode(): _copy_buffer(sd.infos[i]) for i in range(sd.infos_size)}
        return observation, float(sd.reward.f), bool(sd.terminated), bool(sd.truncated), info

    def reset(self, options=None):
        opts, n = _options(options)
        if self.lib.cenv_reset(opts, n) != 0:
            raise Exception("Non-zero error code!")
        rd = self.c_reset_data
        observation = {rd.observations[i].key.decode(): _copy_buffer(rd.observations[i])
                       for i in range(rd.observations_size)}
        info = {rd.infos[i].key.decode(): _copy_buffer(rd.infos[i]) for i in range(rd.infos_size)}
        return observation, info

    def render(self):
        self.lib.cenv_render()  # return value ignored, as in cenv.py:369
        r = self.c_render_data
        n = r.value_buffer_height * r.value_buffer_width * r.value_buffer_channels
        dtype = np.dtype(_DTYPE[int(r.value_type)])
        addr = ctypes.cast(r.value_buffer.b, c_void_p).value
        raw = (ctypes.c_char * (n * dtype.itemsize)).from_address(addr)
        arr = np.frombuffer(raw, dtype=dtype, count=n).copy()
        return arr.reshape(r.value_buffer_height, r.value_buffer_width, r.value_buffer_channels)

    def close(self):
        self.lib.cenv_close()
