import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from engine_util import EngineVec
from oracle_util import OracleVec
game, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
eng, ora = EngineVec(game, n, seed_base=3), OracleVec(game, n, seed_base=3)
print("reset equal", np.array_equal(eng.reset(), ora.reset_obs()))
idx = np.arange(n)
for s in range(steps):
    acts = np.array([ora.L.pgo_synthetic_action(1, s, e) for e in range(n)], np.int32)
    acts = np.where((idx + s) % 7 < 2, (acts % 3) * 3 + 2, acts).astype(np.int32)
    oe, _, _ = eng.step(acts)
    oo, _, _ = ora.step(acts, threads=8)
    if not np.array_equal(oe, oo):
        for e in np.nonzero((oe != oo).any(axis=1))[0][:3]:
            a, b = oe[e].reshape(64, 64, 3), oo[e].reshape(64, 64, 3)
            ys, xs = np.nonzero((a != b).any(axis=2))
            print("step", s, "env", e, "pixels differ", len(ys), "rows", ys.min(), ys.max(), "cols", xs.min(), xs.max())
            for y, x in list(zip(ys, xs))[:6]:
                print("   (y=%d,x=%d) engine %s oracle %s" % (y, x, a[y, x], b[y, x]))
            print("   state", ora.state(e)[:24])
        break
