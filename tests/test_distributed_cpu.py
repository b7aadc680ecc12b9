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
    from procgen2_amd.vec_env import RootGather
    lo, hi = shard_range(total, world, rank)
    ora = OracleVec("maze", hi - lo, seed_base=1, env_offset=lo, render=True)
    ora.reset()
    # the tensors a ProcgenVecEnv owns: obs u8 [n,64,64,3], reward f32 [n], done u8 [n]; the plan is built once
    obs = torch.zeros((hi - lo, 64, 64, 3), dtype=torch.uint8)
    rew = torch.zeros(hi - lo, dtype=torch.float32)
    done = torch.zeros(hi - lo, dtype=torch.uint8)
    plan = RootGather((obs, rew, done), dst=0)
    for s in range(steps):
        o, r, d = ora.step(None, run_seed=0)
        obs.copy_(torch.from_numpy(o.reshape(-1, 64, 64, 3)))
        rew.copy_(torch.from_numpy(r))
        done.copy_(torch.from_numpy(d))
        g_obs, g_rew, g_done = plan()  # every step, like bench.py --gather
        if rank == 0 and s in (0, steps - 1):
            np.save(os.path.join(out_dir, "obs%d.npy" % s), g_obs.numpy())
            np.save(os.path.join(out_dir, "rew%d.npy" % s), g_rew.numpy())
            np.save(os.path.join(out_dir, "done%d.npy" % s), g_done.numpy())
        if rank != 0:
            assert g_obs is None and g_rew is None and g_done is None
    if rank == 0:
        assert plan.slabs[0].shape == (total, 64, 64, 3) and plan.slabs[0].dtype == torch.uint8
    ora.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_rollout_equals_single_process(tmp_path):
    total, steps, world = 21, 40, 2  # odd total: ranks hold different env counts
    mp.spawn(_worker, args=(world, _free_port(), total, steps, str(tmp_path)), nprocs=world, join=True)
    from oracle_util import OracleVec
    whole = OracleVec("maze", total, seed_base=1, env_offset=0, render=True)
    whole.reset()
    for s in range(steps):
        o, r, d = whole.step(None, run_seed=0)
        if s in (0, steps - 1):  # the root's slab == the single-process batch, every observation byte
            assert np.array_equal(np.load(tmp_path / ("obs%d.npy" % s)).reshape(total, -1), o)
            assert np.array_equal(np.load(tmp_path / ("rew%d.npy" % s)), r)
            assert np.array_equal(np.load(tmp_path / ("done%d.npy" % s)), d)
    whole.close()


def _subgroup_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    group = dist.new_group([1, 2])  # group rank 0 = global rank 1: the two numberings differ
    if rank in (1, 2):
        n = 3 if rank == 1 else 5
        obs = torch.full((n, 64, 64, 3), rank, dtype=torch.uint8)
        rew = torch.full((n,), float(rank))
        done = torch.full((n,), rank, dtype=torch.uint8)
        g_obs, g_rew, g_done = gather_outputs(obs, rew, done, dst=0, group=group)  # dst is a GROUP rank
        if rank == 1:
            assert g_obs.shape == (8, 64, 64, 3)
            assert (g_obs[:3] == 1).all() and (g_obs[3:] == 2).all()
            assert g_rew.tolist() == [1.0] * 3 + [2.0] * 5 and g_done.tolist() == [1] * 3 + [2] * 5
            open(os.path.join(out_dir, "ok"), "w").write("1")
        else:
            assert g_obs is None
    dist.barrier()
    dist.destroy_process_group()


def test_gather_inside_a_subgroup_translates_ranks(tmp_path):
    mp.spawn(_subgroup_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    assert (tmp_path / "ok").exists()


def _mixed_worker(rank, world, port, per_rank, out_dir):
    """bench.py --workload mixed --gather, without GPUs: every rank owns ONE slab per output, seven "games" fill their
    blocks of it (here: a value that encodes rank, game and env), and a step is ONE RootGather of the three slabs — on
    the root straight into its slice of the gathered batch, which is never copied."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from procgen2_amd.vec_env import GAMES, RootGather
    base = per_rank // len(GAMES)
    counts = [base] * (len(GAMES) - 1) + [per_rank - base * (len(GAMES) - 1)]
    shapes = (((4, 4, 3), torch.uint8), ((), torch.float32), ((), torch.uint8))  # small frames: the plumbing is the same
    whole = None
    if rank == 0:
        whole = tuple(torch.zeros((world * per_rank,) + sh, dtype=dt) for sh, dt in shapes)
        local = tuple(t[:per_rank] for t in whole)
    else:
        local = tuple(torch.zeros((per_rank,) + sh, dtype=dt) for sh, dt in shapes)
    plan = RootGather(local, dst=0, slabs=whole)
    if rank == 0:
        assert all(a.data_ptr() == b.data_ptr() for a, b in zip(local, plan.slabs)), "the root's block is in place"
    for step in range(3):
        at = 0
        for g, count in enumerate(counts):  # what seven ProcgenVecEnv(out=slices) would do
            for t in local:
                t[at:at + count] = (rank * 16 + g * 2 + step) % 251
            at += count
        got = plan()
        if rank == 0:
            for t in got:
                for r in range(world):
                    at = r * per_rank
                    for g, count in enumerate(counts):
                        assert (t[at:at + count] == (r * 16 + g * 2 + step) % 251).all(), (step, r, g)
                        at += count
        else:
            assert all(t is None for t in got)
    if rank == 0:
        open(os.path.join(out_dir, "ok"), "w").write("1")
    dist.barrier()
    dist.destroy_process_group()


def test_mixed_workload_one_slab_one_gather_four_ranks(tmp_path):
    mp.spawn(_mixed_worker, args=(4, _free_port(), 23, str(tmp_path)), nprocs=4, join=True)  # 23 = 6·3 + 5: a remainder
    assert (tmp_path / "ok").exists()


class _SlabWriter:
    """What bench.py's gathered_steps() needs of a ProcgenVecEnv, on CPU tensors: step_synthetic(ordered=False) fills this
    engine's block of the rank's three slabs, publish() / consume() / sync() are recorded — the loop under test is
    bench.py's own, the collective a real RootGather over gloo."""

    def __init__(self, log, rank, game, blocks):
        self.log, self.rank, self.game, self.blocks, self.step = log, rank, game, blocks, 0

    def step_synthetic(self, run_seed, ordered=True):
        assert ordered is False
        for t in self.blocks:
            t.fill_((self.rank * 32 + self.game * 3 + self.step + run_seed) % 251)
        self.step += 1
        self.log.append(("step", self.game))

    def publish(self):
        self.log.append(("publish", self.game))

    def consume(self):
        self.log.append(("consume", self.game))

    def sync(self):
        self.log.append(("sync", self.game))


def _bench_loop_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from procgen2_amd.vec_env import GAMES, RootGather
    per_rank = 23 if rank == 0 else 31  # rank-dependent counts: the plan exchanges them once
    base = per_rank // len(GAMES)
    counts = [base] * (len(GAMES) - 1) + [per_rank - base * (len(GAMES) - 1)]
    shapes = (((4, 4, 3), torch.uint8), ((), torch.float32), ((), torch.uint8))
    local = tuple(torch.zeros((per_rank,) + sh, dtype=dt) for sh, dt in shapes)
    log, envs, at = [], [], 0
    for g, count in enumerate(counts):
        envs.append(_SlabWriter(log, rank, g, tuple(t[at:at + count] for t in local)))
        at += count
    plan = RootGather(local, dst=0)
    seen = []

    def gather():
        log.append(("gather", -1))
        got = plan()
        if rank == 0:
            seen.append(tuple(t.clone() for t in got))

    steps, run_seed = 4, 7
    bench.gathered_steps(envs, gather, steps, run_seed)
    # the order of one step: every engine steps, every engine publishes, ONE gather, every engine consumes
    n = len(GAMES)
    per_step = [("step", g) for g in range(n)] + [("publish", g) for g in range(n)] + [("gather", -1)] + [("consume", g) for g in range(n)]
    assert log == per_step * steps + [("sync", g) for g in range(n)]
    if rank == 0:
        sizes = [23, 31]
        assert plan.slabs[0].shape[0] == sum(sizes)
        for step, got in enumerate(seen):
            for t in got:
                at = 0
                for r, size in enumerate(sizes):
                    b = size // n
                    cs = [b] * (n - 1) + [size - b * (n - 1)]
                    for g, c in enumerate(cs):
                        assert (t[at:at + c] == (r * 32 + g * 3 + step + run_seed) % 251).all(), (step, r, g)
                        at += c
        open(os.path.join(out_dir, "ok"), "w").write("1")
    dist.barrier()
    dist.destroy_process_group()


def test_bench_gathered_step_loop_two_ranks_with_different_counts(tmp_path):
    """bench.py's own publish -> RootGather -> consume loop (bench.gathered_steps), two gloo ranks
    holding 23 and 31 envs."""
    mp.spawn(_bench_loop_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()
