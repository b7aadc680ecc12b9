"""Timing experiments on coinrun's render kernel: needs the -DPG_ABLATE build (python -m procgen2_amd.build --ablate).
The product library refuses these debug bits (include/procgen2_vec.h pgv_set_debug)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from engine_util import EngineVec  # noqa: E402
LIB = os.path.join(ROOT, "procgen2_amd", "lib", "libprocgen2_hip_ablate.so")
GAME = sys.argv[1] if len(sys.argv) > 1 else "coinrun"
for flags, name in ((0, "full"), (8192, "never general"), (4096, "always general"), (512, "rows never blend"),
                    (128, "no row loop"), (2, "no sprites"), (8, "no store"), (128 + 2, "no rows/sprites"),
                    (128 + 2 + 8, "no rows/spr/store"), (4 + 2 + 8, "nothing")):
    e = EngineVec(GAME, 65536, seed_base=1, lib_path=LIB)
    e.reset()
    e.timed(512)  # steady state: the agents spread over their levels
    e.set_debug(flags)
    tot, ren = e.timed(64)
    print("%-18s render %.3f ms  total %.3f ms" % (name, ren / 64, tot / 64))
    e.close()
