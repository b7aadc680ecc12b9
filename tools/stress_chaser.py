#!/usr/bin/env python3
"""Stress loop for chaser's concurrency: its levels are generated on a side stream BESIDE the step's logic
and render kernels, the envs being reset are drawn by a late pass behind a flag protocol — the one place a race was found
once (a once-in-fifty-runs trace failure in round 2).  Two kinds of run, repeated:

  * the reference's own 20 000-step reward/terminated traces (tests/golden/appendix_c.json, seeds 123 and 7, both
    readings of abs): ONE engine env each, caller-side reset after every terminal step, CRC-32 of the stream against the
    recording of the unmodified reference sources — no oracle in between;
  * a 4 096-env lock-step against the oracle, every observation byte, reward bit pattern and done flag of every step,
    with steering-heavy actions (episodes end every few steps somewhere in the batch, so every step has resets in it).

    python tools/stress_chaser.py [--traces 50] [--lockstep 20] [--envs 4096] [--steps 600]   > profiles/r04_stress_chaser.log
"""
import argparse
import json
import os
import struct
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from engine_util import EngineVec  # noqa: E402
from oracle_util import OracleVec  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--traces", type=int, default=50)
ap.add_argument("--lockstep", type=int, default=20)
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--steps", type=int, default=600)
a = ap.parse_args()

with open(os.path.join(ROOT, "tests", "golden", "appendix_c.json")) as f:
    golden = json.load(f)
traces = [t for t in golden["traces"] + golden["traces_float_abs"] if t["game"] == "chaser" and t["seed"] in (123, 7)]
failures = 0
t_start = time.time()
for rep in range(a.traces):
    for t in traces:
        eng = EngineVec("chaser", 1, seed_base=t["seed"], game_flags=t.get("flags", 0))
        eng.reset()
        s, crc, episodes = 1, 0, 0
        one = np.ones(1, np.uint8)
        for _ in range(golden["steps"]):
            s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
            _, r, d = eng.step(np.array([(s >> 16) % 15], np.int32))
            crc = zlib.crc32(struct.pack("<fB", float(r[0]), int(d[0])), crc)
            if d[0]:
                episodes += 1
                eng.reset(mask=one)
        eng.close()
        ok = "%08x" % crc == t["crc"] and episodes == t["episodes"]
        failures += 0 if ok else 1
        print("trace rep %2d seed %3d flags %d: crc %08x episodes %d %s" % (rep, t["seed"], t.get("flags", 0), crc, episodes,
                                                                           "ok" if ok else "FAIL (want %s, %d)" % (t["crc"], t["episodes"])), flush=True)
threads = bench.usable_cores()
for rep in range(a.lockstep):
    n = a.envs
    eng = EngineVec("chaser", n, seed_base=1000 + 17 * rep)
    ora = OracleVec("chaser", n, seed_base=1000 + 17 * rep, threads=threads)
    ok = np.array_equal(eng.reset(), ora.reset_obs())
    rng = np.random.default_rng(rep)
    hold = rng.choice([1, 3, 5, 7], n)
    ends, bad_step = 0, -1
    for s in range(a.steps):
        hold = np.where(rng.random(n) < 0.15, rng.choice([1, 3, 5, 7], n), hold)
        acts = np.where(rng.random(n) < 0.1, rng.integers(0, 15, n), hold).astype(np.int32)
        oe, re_, de = eng.step(acts)
        oo, ro, do = ora.step(acts, threads=threads)
        if not (np.array_equal(de, do) and np.array_equal(re_.view(np.uint32), ro.view(np.uint32)) and np.array_equal(oe, oo)):
            ok, bad_step = False, s
            break
        ends += int(do.sum())
    eng.close()
    ora.close()
    failures += 0 if ok else 1
    print("lock-step rep %2d: %d envs x %d steps, %d episodes ended %s" % (rep, n, a.steps, ends, "ok" if ok else "FAIL at step %d" % bad_step),
          flush=True)
print("stress_chaser: %d trace runs + %d lock-steps, %d failures, %.0f s" % (a.traces * len(traces), a.lockstep, failures, time.time() - t_start))
sys.exit(1 if failures else 0)
