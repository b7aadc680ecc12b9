"""Test-side driver of the HIP engine through its C ABI (ctypes, host-pointer entry points)."""
import ctypes
import os
import sys
from ctypes import c_float, c_uint8, c_void_p

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from procgen2_amd import lib as pglib  # noqa: E402

OBS_BYTES = pglib.OBS_BYTES


class EngineVec:
    def __init__(self, game, n, seed_base=1, env_offset=0, device=0, lib_path=None, num_levels=0, start_level=0, mode=None, game_flags=0):
        self.L = pglib.load(lib_path)
        self.n = n
        self.h = pglib.make(self.L, game, n, device=device, seed_base=seed_base, env_offset=env_offset,
                            num_levels=num_levels, start_level=start_level, mode=mode, game_flags=game_flags)
        self.obs = np.zeros((n, OBS_BYTES), np.uint8)
        self.reward = np.zeros(n, np.float32)
        self.done = np.zeros(n, np.uint8)

    def _fetch(self):
        pglib.check(self.L, self.L.pgv_copy_out(self.h, self.obs.ctypes.data_as(c_void_p),
                                                self.reward.ctypes.data_as(c_void_p),
                                                self.done.ctypes.data_as(c_void_p)), "pgv_copy_out")
        return self.obs, self.reward, self.done

    def reset(self, mask=None, seeds=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        s = None if seeds is None else np.ascontiguousarray(seeds, np.int32)
        pglib.check(self.L, self.L.pgv_reset_host(self.h, None if m is None else m.ctypes.data_as(c_void_p),
                                                  None if s is None else s.ctypes.data_as(c_void_p)), "pgv_reset_host")
        return self._fetch()[0]

    def step(self, actions=None, run_seed=0):
        if actions is None:
            pglib.check(self.L, self.L.pgv_step_synthetic(self.h, run_seed), "pgv_step_synthetic")
        else:
            a = np.ascontiguousarray(actions, np.int32)
            pglib.check(self.L, self.L.pgv_step_host(self.h, a.ctypes.data_as(c_void_p)), "pgv_step_host")
        return self._fetch()

    def step_quiet(self, run_seed=0):
        """One synthetic step, nothing copied out."""
        pglib.check(self.L, self.L.pgv_step_synthetic(self.h, run_seed), "pgv_step_synthetic")

    def fetch_scalars(self):
        """Rewards and dones only (the observation slab stays on the device)."""
        pglib.check(self.L, self.L.pgv_copy_out(self.h, None, self.reward.ctypes.data_as(c_void_p),
                                                self.done.ctypes.data_as(c_void_p)), "pgv_copy_out")
        return self.reward, self.done

    def frame(self, env, width, height):
        out = np.zeros((height, width, 3), np.uint8)
        pglib.check(self.L, self.L.pgv_render_frame(self.h, env, width, height, out.ctypes.data_as(c_void_p)),
                    "pgv_render_frame")
        return out

    def save_state(self):
        n = self.L.pgv_snapshot_bytes(self.h)
        buf = np.empty(n, np.uint8)
        pglib.check(self.L, self.L.pgv_save_state(self.h, buf.ctypes.data_as(c_void_p), n), "pgv_save_state")
        return buf

    def load_state(self, buf):
        pglib.check(self.L, self.L.pgv_load_state(self.h, buf.ctypes.data_as(c_void_p), buf.size), "pgv_load_state")

    def set_debug(self, flags):
        pglib.check(self.L, self.L.pgv_set_debug(self.h, flags), "pgv_set_debug")

    def state(self, env, cap=512):
        buf = (c_float * cap)()
        n = self.L.pgv_dump_state(self.h, env, buf, cap)
        return np.array(buf[:min(n, cap)], np.float32)

    def tiles(self, env, cap=4096):
        buf = (c_uint8 * cap)()
        n = self.L.pgv_dump_tiles(self.h, env, buf, cap)
        return np.array(buf[:n], np.uint8)

    def timed(self, steps, run_seed=0):
        total, render = ctypes.c_double(), ctypes.c_double()
        pglib.check(self.L, self.L.pgv_timed_steps(self.h, steps, run_seed, ctypes.byref(total), ctypes.byref(render)),
                    "pgv_timed_steps")
        return total.value, render.value

    def close(self):
        if self.h:
            self.L.pgv_close(self.h)
            self.h = None
