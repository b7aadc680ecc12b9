#!/usr/bin/env python3
"""How often does a coinrun agent leave the reach its entity lanes pre-select hazards by (coinrun.hip hazard_near)?
Runs the -DPG_EXP_COUNT_FAR build (tools/build_exp.py coinrun far -DPG_EXP_COUNT_FAR), which marks such an env-step with
a reward of 12345, at 65 536 envs with the synthetic action stream."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from engine_util import EngineVec  # noqa: E402

n, steps = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 3000
eng = EngineVec("coinrun", n, seed_base=1, lib_path=os.path.join(ROOT, "procgen2_amd/lib/libpg_exp_far.so"))
eng.reset()
far = env_steps = 0
for s in range(steps):
    _, r, d = eng.step(None, run_seed=7)
    far += int((r == 12345.0).sum())
    env_steps += n
print("coinrun: %d of %d env-steps left the reach (%.3g)" % (far, env_steps, far / env_steps))
eng.close()
