#!/usr/bin/env python3
"""First-contact check on a GPU box: HIP engine vs CPU oracle, step by step, with diagnostics.

    python tools/gpu_check.py --game coinrun --envs 64 --steps 300

Prints the first mismatch (state vector, tiles, pixels) if any, then a short timing of the engine.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from engine_util import EngineVec  # noqa: E402
from oracle_util import OracleVec  # noqa: E402


def compare(game, n, steps, seed_base, run_seed, jumpy=False):
    eng = EngineVec(game, n, seed_base=seed_base)
    ora = OracleVec(game, n, seed_base=seed_base)
    bad = 0
    n_done = 0
    rew_sum = 0.0

    def diff(tag, step):
        nonlocal bad
        ok = True
        for e in range(n):
            se, so = eng.state(e), ora.state(e)
            if se.shape != so.shape or not np.array_equal(se.view(np.uint32), so.view(np.uint32)):
                print("[%s step %d env %d] STATE differs" % (tag, step, e))
                print("  engine:", se[:40])
                print("  oracle:", so[:40])
                ok = False
                break
            te, to = eng.tiles(e), ora.tiles(e)
            if not np.array_equal(te, to):
                idx = np.nonzero(te != to)[0]
                print("[%s step %d env %d] TILES differ at %d cells, first %s" % (tag, step, e, idx.size, idx[:8]))
                ok = False
                break
        return ok

    o_e = eng.reset()
    o_o = ora.reset_obs()
    if not np.array_equal(o_e, o_o):
        d = np.nonzero((o_e != o_o).any(axis=1))[0]
        print("RESET obs differ in %d/%d envs; first env %d: %d bytes differ" %
              (d.size, n, d[0], int((o_e[d[0]] != o_o[d[0]]).sum())))
        diff("reset", -1)
        bad += 1
    for s in range(steps):
        acts = np.array([ora.L.pgo_synthetic_action(run_seed, s, e) for e in range(n)], np.int32)
        if jumpy:  # bias towards the jump actions (2, 5, 8) so platformer agents climb and meet hazards
            idx = np.arange(n)
            acts = np.where((idx + s) % 5 < 3, (acts % 3) * 3 + 2, acts).astype(np.int32)
        oe, re_, de = eng.step(acts)
        oo, ro, do = ora.step(acts)
        n_done += int(do.sum())
        rew_sum += float(ro.sum())
        rew_ok = np.array_equal(re_.view(np.uint32), ro.view(np.uint32))
        done_ok = np.array_equal(de, do)
        obs_ok = np.array_equal(oe, oo)
        if not (rew_ok and done_ok and obs_ok):
            bad += 1
            print("step %d: reward %s done %s obs %s" % (s, rew_ok, done_ok, obs_ok))
            if not obs_ok:
                d = np.nonzero((oe != oo).any(axis=1))[0]
                e0 = d[0]
                px = np.nonzero(oe[e0] != oo[e0])[0]
                print("  obs differ in %d envs; env %d: %d bytes, first at byte %d (y=%d x=%d c=%d) eng=%d ora=%d" %
                      (d.size, e0, px.size, px[0], px[0] // 192, (px[0] % 192) // 3, px[0] % 3, oe[e0][px[0]],
                       oo[e0][px[0]]))
            diff("step", s)
            if bad >= 3:
                break
    print("%s: %d envs x %d steps: %s (%d episode ends, reward sum %.1f)" %
          (game, n, steps, "PARITY OK" if bad == 0 else "MISMATCH", n_done, rew_sum))
    eng.close()
    ora.close()
    return bad == 0


def timing(game, n, steps):
    eng = EngineVec(game, n, seed_base=1)
    eng.reset()
    eng.timed(16)
    total, render = eng.timed(steps)
    print("%s: %d envs, %d steps: %.3f ms/step (render kernel %.3f ms/step) -> %.2f M env-steps/s, obs write %.1f GB/s" %
          (game, n, steps, total / steps, render / steps, n * steps / total / 1e3, n * 12288 * steps / total / 1e6))
    eng.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--game", default="coinrun")
    ap.add_argument("--envs", type=int, default=64)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--seed-base", type=int, default=1)
    ap.add_argument("--run-seed", type=int, default=0)
    ap.add_argument("--jumpy", action="store_true")
    ap.add_argument("--time-envs", type=int, default=0)
    ap.add_argument("--time-steps", type=int, default=64)
    a = ap.parse_args()
    t0 = time.time()
    ok = compare(a.game, a.envs, a.steps, a.seed_base, a.run_seed, a.jumpy)
    print("compare took %.1fs" % (time.time() - t0))
    if a.time_envs:
        timing(a.game, a.time_envs, a.time_steps)
    sys.exit(0 if ok else 1)
