"""Mini workload for counter passes: GAME at 65 536 envs, settle, then a few steps (run under rocprofv3 --pmc …)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from engine_util import EngineVec  # noqa: E402
lib = os.environ.get("PG_LIB")
e = EngineVec(os.environ.get("PG_GAME", "coinrun"), 65536, seed_base=1, lib_path=os.path.join(ROOT, lib) if lib else None)
e.reset()
e.timed(int(os.environ.get("PG_SETTLE", "300")))
if os.environ.get("PG_DEBUG"):  # ablation bits: the -DPG_ABLATE library only (PG_LIB)
    e.set_debug(int(os.environ["PG_DEBUG"], 0))
e.timed(8)
e.close()
