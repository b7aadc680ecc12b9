#!/usr/bin/env python3
"""Where a wavefront of jumper's render kernel spends its life: a -DPG_TIMELINE build (tools/build_exp.py jumper tl
-DPG_TIMELINE) stamps s_memtime at the ends of its phases — after waiting for everything outstanding — and leaves the
stamps in the first bytes of the rows it stored.  Prints the mean length of each phase per wave (upper / lower rows), in
shader clocks and as a share of the wave's life.

    python tools/probe/jumper_timeline.py procgen2_amd/lib/libpg_exp_tl.so
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from engine_util import EngineVec  # noqa: E402

lib = sys.argv[1]
n = 65536
e = EngineVec("jumper", n, seed_base=1, lib_path=lib)
e.reset()
for _ in range(300):
    obs, _, _ = e.step(None, run_seed=3)
obs = np.asarray(obs).reshape(n, 64 * 64 * 3)
names = ["state loads", "spans + cell table", "row loop", "resolve", "sprites + bunny", "ring overlay", "needle + bar", "store"]
for half in (0, 1):
    raw = obs[:, half * 6144: half * 6144 + 72].copy().view(np.uint64).reshape(n, 9)
    ok = (raw[:, 8] > raw[:, 0]) & (raw[:, 8] - raw[:, 0] < 10_000_000)
    d = np.diff(raw[ok].astype(np.int64), axis=1)
    life = d.sum(axis=1)
    print("%s rows: %d waves, life %.0f clocks (median %.0f)" % ("upper" if half == 0 else "lower", ok.sum(), life.mean(), np.median(life)))
    for k, name in enumerate(names):
        print("   %-20s %8.0f  %5.1f %%" % (name, d[:, k].mean(), 100.0 * d[:, k].mean() / life.mean()))
