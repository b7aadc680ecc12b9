#!/usr/bin/env python3
"""Where a wavefront of a render kernel spends its life.  A -DPG_TIMELINE build of the game (tools/build_exp.py GAME tl
-DPG_TIMELINE; chaser.hip, jumper.hip, coinrun.hip, bossfight.hip and caveflyer.hip carry the stamps) reads s_memtime at the ends of its phases — after waiting for
everything outstanding — and leaves the stamps in the first bytes of the rows it stored.  Prints the mean length of each
phase per wave (upper / lower rows) in shader clocks and as a share of the wave's life.

    python tools/probe/wave_timeline.py chaser procgen2_amd/lib/libpg_exp_tl.so
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from engine_util import EngineVec  # noqa: E402

PHASES = {
    "jumper": ["pre-pass loads, cell table", "row loop", "compass resolve", "sprites + bunny", "ring overlay", "needle + bar", "store"],
    "chaser": ["list + state loads", "base layer copy", "point stamps", "first pass: fetch + keep", "draws", "store"],
    "coinrun": ["hand-off loads, cell table", "first sprite texels requested", "row loop", "sprite replay", "store"],
    "bossfight": ["hand-off loads", "background", "boss bullets", "second list (ship, shield, shots, agent)", "store"],
    "caveflyer": ["hand-off and state loads, cell table", "row loop", "draws resolved (one per lane)", "draw replay", "store"],
}
game, lib = sys.argv[1], sys.argv[2]
names = PHASES[game]
n = 65536
e = EngineVec(game, n, seed_base=1, lib_path=lib)
e.reset()
for _ in range(300):
    obs, _, _ = e.step(None, run_seed=3)
obs = np.asarray(obs).reshape(n, 64 * 64 * 3)
stamps = len(names) + 1
for half in (0, 1):
    raw = obs[:, half * 6144: half * 6144 + 8 * stamps].copy().view(np.uint64).reshape(n, stamps)
    ok = (raw[:, -1] > raw[:, 0]) & (raw[:, -1] - raw[:, 0] < 10_000_000)
    d = np.diff(raw[ok].astype(np.int64), axis=1)
    life = d.sum(axis=1)
    print("%s rows: %d waves, life %.0f clocks (median %.0f)" % ("upper" if half == 0 else "lower", ok.sum(), life.mean(), np.median(life)))
    for k, name in enumerate(names):
        print("   %-26s %8.0f  %5.1f %%" % (name, d[:, k].mean(), 100.0 * d[:, k].mean() / life.mean()))
