#!/usr/bin/env python3
"""Which frames the render pre-pass hands back to the complete path, and why: a -DPG_FAT_WHY build of the game
(tools/build_exp.py GAME fw -DPG_FAT_WHY) counts them per launch by reason and prints the counts of the previous launch.

    python tools/probe/fat_why.py coinrun procgen2_amd/lib/libpg_exp_fw.so [steps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from engine_util import EngineVec  # noqa: E402

game, lib = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
e = EngineVec(game, 65536, seed_base=1, lib_path=lib)
e.reset()
for _ in range(steps):
    e.step(None, run_seed=3)
e.close()
