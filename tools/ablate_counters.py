"""Per-section instruction counts of coinrun's render kernel: the -DPG_ABLATE build (python -m procgen2_amd.build
--ablate) run under `rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES …`, one group of launches per
debug-flag set; tools/ablate_counters.sh parses the per-dispatch counter CSV by launch order.

    python3 tools/ablate_counters.py run      (under rocprofv3; prints the plan as JSON on the last line)
"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from engine_util import EngineVec  # noqa: E402
LIB = os.path.join(ROOT, "procgen2_amd", "lib", "libprocgen2_hip_ablate.so")
GROUPS = [(0, "full"), (2, "no sprites"), (8, "no store"), (128, "no row loop"), (128 + 2, "no rows/sprites"),
          (128 + 2 + 8, "no rows/sprites/store"), (4 + 2 + 8, "preamble only"), (512, "rows never blend")]
PER = 6
GAME = os.environ.get("PG_GAME", "coinrun")
e = EngineVec(GAME, 65536, seed_base=1, lib_path=LIB)
e.reset()                      # 1 render launch
e.timed(int(os.environ.get("PG_SETTLE", "300")))   # settle
plan = []
for flags, name in GROUPS:
    e.set_debug(flags)
    tot, ren = e.timed(PER)
    plan.append({"flags": flags, "name": name, "launches": PER, "render_ms": ren / PER})
e.close()
print(json.dumps({"settle": int(os.environ.get("PG_SETTLE", "300")) + 1, "groups": plan}))
