#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_frames.json: per-step CRC-32 of the oracle's 12288-byte observation.

These vectors come from the repo's own oracle (raster spec of DESIGN.md), NOT from the reference: the
reference rasterises through an SDL3 pre-release that is absent here, so pixels are unpinned at that
boundary.  They guard against accidental changes of the spec / oracle.  Run in the build container:
    python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from test_oracle_golden import _frame_crcs  # noqa: E402

CASES = {"coinrun": [(123, 96), (7, 96), (4294967291, 48)], "maze": [(123, 96), (7, 64)],
         "bossfight": [(123, 120), (7, 120)],
         "climber": [(123, 120), (7, 120)],
         "caveflyer": [(123, 120), (7, 120)],
         "chaser": [(123, 120), (7, 120)],
         "jumper": [(123, 120), (7, 120)]}

# (game, PGV_MODE_*) of every non-default distribution mode: easy 1, hard 2, memory 3, extreme 4
MODE_CASES = [("coinrun", 1), ("climber", 1), ("bossfight", 1), ("chaser", 2), ("chaser", 4), ("maze", 1), ("maze", 3),
              ("caveflyer", 1), ("caveflyer", 3), ("jumper", 1), ("jumper", 3)]

if __name__ == "__main__":
    modes = {"%s:%d:%d:%d" % (g, m, 123, 80): _frame_crcs(g, 123, 80, m) for g, m in MODE_CASES}
    modes["maze:3:7:80"] = _frame_crcs("maze", 7, 80, 3)
    with open(os.path.join(HERE, "oracle_mode_frames.json"), "w") as f:
        json.dump(modes, f)
    print("wrote oracle_mode_frames.json")
    out = {g: {"%d:%d" % (seed, steps): _frame_crcs(g, seed, steps) for seed, steps in cases}
           for g, cases in CASES.items()}
    with open(os.path.join(HERE, "oracle_frames.json"), "w") as f:
        json.dump(out, f)
    print("wrote oracle_frames.json")
