"""Lock-step of the engine against the oracle with a mixed action stream; on the first mismatch prints the env, the step and
the differing entries of the two state dumps.

    python tools/lockstep_debug.py [GAME] [ENVS] [STEPS]        (default: chaser 256 3000)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from engine_util import EngineVec  # noqa: E402
from oracle_util import OracleVec  # noqa: E402

game = sys.argv[1] if len(sys.argv) > 1 else "chaser"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
eng, ora = EngineVec(game, n, seed_base=3), OracleVec(game, n, seed_base=3)
assert np.array_equal(eng.reset(), ora.reset_obs())
for s in range(steps):
    acts = np.array([ora.L.pgo_synthetic_action(1, s, e) for e in range(n)], np.int32)
    acts = np.where(acts % 2 == 1, acts, (acts*7+s) % 15).astype(np.int32)
    oe, re_, de = eng.step(acts)
    oo, ro, do = ora.step(acts, threads=8)
    if not (np.array_equal(re_, ro) and np.array_equal(de, do) and np.array_equal(oe, oo)):
        bad = np.nonzero((re_ != ro) | (de != do) | (oe != oo).any(axis=1))[0]
        e = bad[0]
        print("step", s, "envs", bad[:8], "reward", re_[e], ro[e], "done", de[e], do[e])
        a, b = eng.state(e, 400), ora.state(e, 400)
        d = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]
        print("state diff idx", d[:20], a[d[:20]], b[d[:20]])
        print(a[:16]); print(b[:16])
        break
else:
    print("no mismatch")
