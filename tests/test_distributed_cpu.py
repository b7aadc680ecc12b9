"""The N>1 path without GPUs: env-index sharding and the optional rooted gather, exercised with
two gloo ranks.  Each rank's "engine output" is stood in by the CPU oracle over its shard — the oracle is only
the data source/checker here; what is under test is procgen2_amd.vec_env's sharding and gather plumbing."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from procgen2_amd.vec_env import gather_outputs, shard_range  # noqa: E402


def test_shard_range_partitions_exactly():
    for total, world in ((524288, 8), (65536, 8), (10, 3), (7, 8), (1, 1)):
        spans = [shard_range(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        for a, b in zip(spans, spans[1:]):
            assert a[1] == b[0]
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1
    assert shard_range(524288, 8, 3) == (196608, 262144)  # 65 536 per GPU (BASELINE.json configs[4])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, steps, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle_util import OracleVec
    lo, hi = shard_range(total, world, rank)
    ora = OracleVec("maze", hi - lo, seed_base=1, env_offset=lo, render=False)
    for s in range(steps):
        ora.step(None, run_seed=0)
    obs = torch.from_numpy(np.tile(ora.reward[:, None], (1, 4)).astype(np.float32))  # stand-in payload [n,4]
    rew = torch.from_numpy(ora.reward.copy())
    done = torch.from_numpy(ora.done.copy())
    g_obs, g_rew, g_done = gather_outputs(obs, rew, done, dst=0)
    if rank == 0:
        np.save(os.path.join(out_dir, "rew.npy"), g_rew.numpy())
        np.save(os.path.join(out_dir, "done.npy"), g_done.numpy())
        np.save(os.path.join(out_dir, "obs.npy"), g_obs.numpy())
    else:
        assert g_obs is None and g_rew is None and g_done is None
    ora.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_rollout_equals_single_process(tmp_path):
    total, steps, world = 21, 40, 2  # odd total: ranks hold different env counts
    mp.spawn(_worker, args=(world, _free_port(), total, steps, str(tmp_path)), nprocs=world, join=True)
    from oracle_util import OracleVec
    whole = OracleVec("maze", total, seed_base=1, env_offset=0, render=False)
    for s in range(steps):
        whole.step(None, run_seed=0)
    assert np.array_equal(np.load(tmp_path / "rew.npy"), whole.reward)
    assert np.array_equal(np.load(tmp_path / "done.npy"), whole.done)
    assert np.load(tmp_path / "obs.npy").shape == (total, 4)
    whole.close()
