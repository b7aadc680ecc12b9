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

if __name__ == "__main__":
    out = {g: {"%d:%d" % (seed, steps): _frame_crcs(g, seed, steps) for seed, steps in cases}
           for g, cases in CASES.items()}
    with open(os.path.join(HERE, "oracle_frames.json"), "w") as f:
        json.dump(out, f)
    print("wrote oracle_frames.json")
