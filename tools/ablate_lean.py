"""Timing experiments on a game's LEAN render path (behind the pre-pass, pg_prepass.h): needs the -DPG_ABLATE build
(python -m procgen2_amd.build --ablate).  "render" = what sits between the engine's render events: the pre-pass kernel
+ the render kernel, or — with bit 22, which re-uses the previous frame's pre-pass — the render kernel alone."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from engine_util import EngineVec  # noqa: E402
LIB = os.path.join(ROOT, "procgen2_amd", "lib", "libprocgen2_hip_ablate.so")
GAME = sys.argv[1] if len(sys.argv) > 1 else "coinrun"
K = 1 << 22
for flags, name in ((0, "pre-pass + render"), (1 << 21, "complete path (no pre-pass)"), (K, "render alone"), (K | 2, "  no sprites"),
                    (K | 128, "  no row loop"), (K | 8, "  no store"), (K | 8192, "  never general"),
                    (K | 32768, "  every row's texels from LDS (mini-atlas bound)"), (K | 32768 | 65536, "  every second row's texels from LDS"), (K | 2 | 128, "  no rows/sprites"),
                    (K | 2 | 128 | 8, "  preamble only"), (K | 2 | 4 | 8, "  nothing")):
    e = EngineVec(GAME, 65536, seed_base=1, lib_path=LIB)
    e.reset()
    e.timed(512)  # steady state: the agents spread over their levels
    e.set_debug(flags)
    tot, ren = e.timed(64)
    print("%-30s render %.3f ms  total %.3f ms" % (name, ren / 64, tot / 64), flush=True)
    e.close()
