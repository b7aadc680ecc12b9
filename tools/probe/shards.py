#!/usr/bin/env python3
"""One game's batch as K engines of N/K envs side by side, each on its own stream (what --workload mixed does across
games): does the logic chain of one shard hide beside the render kernel of another?

    python tools/probe/shards.py coinrun 1 2 3 4
"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch  # noqa: E402
from procgen2_amd.vec_env import ProcgenVecEnv, step_many_synthetic  # noqa: E402

game, ks = sys.argv[1], [int(k) for k in sys.argv[2:]] or [1, 2]
N = 65536
for k in ks:
    per = N // k
    envs = [ProcgenVecEnv(game, per, device=0, seed_base=1, env_offset=i * per) for i in range(k)]
    for e in envs:
        e.reset()
    step_many_synthetic(envs, 600, 0)
    for e in envs:
        e.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step_many_synthetic(envs, 512, 0)
    for e in envs:
        e.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s  %d shard(s) of %d envs: %.4f ms per step of the whole batch -> %.1f M env-steps/s" % (game, k, per, dt / 512 * 1e3, per * k * 512 / dt / 1e6), flush=True)
    for e in envs:
        e.close()
