#!/usr/bin/env python3
"""Inner-loop check for kernel work on a GPU box: a short lock-step against the oracle (catches a broken picture in
seconds), then the render / whole-step time at 65 536 envs in steady state.

    python tools/perf_quick.py [--games coinrun,chaser] [--settle 400] [--steps 128] [--check 128x160]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from engine_util import EngineVec  # noqa: E402
from oracle_util import OracleVec  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--games", default="coinrun")
ap.add_argument("--settle", type=int, default=400)
ap.add_argument("--steps", type=int, default=128)
ap.add_argument("--check", default="128x160", help="ENVSxSTEPS of the lock-step, 0x0 to skip")
ap.add_argument("--envs", type=int, default=65536)
ap.add_argument("--debug", type=lambda v: int(v, 0), default=0, help="pgv_set_debug flags for the timed part (ablation bits: the -DPG_ABLATE library only)")
ap.add_argument("--lib", default=None, help="another build of the engine (e.g. lib/libprocgen2_hip_ablate.so)")
a = ap.parse_args()
cn, cs = (int(v) for v in a.check.split("x"))
for game in a.games.split(","):
    status = "unchecked"
    if cn and cs:
        eng, ora = EngineVec(game, cn, seed_base=3, lib_path=a.lib), OracleVec(game, cn, seed_base=3)
        ok = np.array_equal(eng.reset(), ora.reset_obs())
        idx = np.arange(cn)
        for s in range(cs):
            acts = np.array([ora.L.pgo_synthetic_action(1, s, e) for e in range(cn)], np.int32)
            acts = np.where((idx + s) % 7 < 2, (acts % 3) * 3 + 2, acts).astype(np.int32)  # some jumping / firing
            oe, re_, de = eng.step(acts)
            oo, ro, do = ora.step(acts, threads=8)
            if not (np.array_equal(oe, oo) and np.array_equal(re_, ro) and np.array_equal(de, do)):
                bad = np.nonzero((oe != oo).any(axis=1))[0]
                status = "MISMATCH at step %d (%d envs differ in obs, first %s)" % (s, bad.size, bad[:4])
                ok = False
                break
        if ok:
            status = "bit-exact %dx%d" % (cn, cs)
        eng.close()
        ora.close()
    e = EngineVec(game, a.envs, seed_base=1, lib_path=a.lib)
    e.reset()
    e.timed(a.settle)
    if a.debug:
        e.set_debug(a.debug)
    tot, ren = e.timed(a.steps)
    e.close()
    print("%-10s %-28s render %.4f ms  step %.4f ms  -> %.1f M env-steps/s" %
          (game, status, ren / a.steps, tot / a.steps, a.envs / (tot / a.steps) / 1e3), flush=True)
