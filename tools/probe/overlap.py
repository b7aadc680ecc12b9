"""Upper bound of overlapping a game's logic kernels with its render kernel: K shards of N/K envs on K streams,
stepping side by side with no join between steps.  usage: overlap.py GAME K [K…]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from procgen2_amd.vec_env import ProcgenVecEnv
game = sys.argv[1]
N = 65536
for K in [int(x) for x in sys.argv[2:]]:
    envs = [ProcgenVecEnv(game, N // K, seed_base=1, env_offset=k * (N // K)) for k in range(K)]
    for e in envs: e.reset()
    def run(steps):
        for _ in range(steps):
            for e in envs: e.step_synthetic(7, ordered=False)
        for e in envs: e.sync()
    run(300); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(200); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s K=%d  %.4f ms/step  %.1f M env-steps/s" % (game, K, dt / 200 * 1e3, N * 200 / dt / 1e6), flush=True)
    for e in envs: e.close()
