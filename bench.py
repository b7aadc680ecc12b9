#!/usr/bin/env python3
"""Benchmark of the hot path: coinrun, 65 536 envs per GPU, synthetic random actions, observations in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs E] [--game coinrun]
    python bench.py --workload mixed [--gather]      # BASELINE.json configs[4]: all seven games, envs split 7 ways

One "step" = one pass of the hot path over the whole batch: the logic kernel (auto-reset or 4 physics
sub-steps per env) + the render kernel (64×64×3 observation per env into the contiguous slab).  Inputs
(actions) are generated on the device from a counter hash, so nothing crosses PCIe inside the timed region.
N > 1: one process per GPU, envs sharded by global index, no data-path collective (weak scaling: 65 536 envs per
GPU); time = max over ranks.  Started under torch.distributed.run (WORLD_SIZE set) this process is one rank; started
plainly with --gpus N > 1 it launches the N ranks itself — `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 …` as a CHILD, before this process has imported torch or touched HIP —
and relays rank 0's line.  --dry-launch prints that command instead of running it.

The timed region is preceded by --warmup steps AND --settle steps (default 512, untimed, reported): all envs are
made in the same step, and the first few hundred steps after that have almost no resets in them, so a short run would
time a transient that is ≈ 8 % faster than steady state.  The roofline leg is always measured on its own window of
≥ 256 steps after the timed region (`roofline.window`), whatever --steps says.

Prints ONE JSON line (rank 0) with `roofline` (render kernel, HIP-event timed on the engine's stream) and,
at N=1, `cpu_baseline` (the CPU oracle on a bounded sample of the same workload, all host cores).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Before torch or the engine touch HIP: the runtime reads this at its first call (see procgen2_amd/__init__.py — the
# mixed workload runs fourteen streams, and on the default four hardware queues they queue up behind each other).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

# 12288 obs write + 4 action read + 4 reward write + 1 done write.  (The synthetic path hashes its actions
# in the kernel instead of reading them: the 4 bytes of the action read are counted because the figure is the survey's,
# 0.03 % of it.)
ALGO_BYTES_PER_ENV_STEP = 12297
HBM_PEAK_GBPS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def csrc_fingerprint():
    """sha1 over the kernel sources (procgen2_amd/csrc, sorted by name): what a traffic profile was taken on."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "procgen2_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def measured_copy_bandwidth(torch, device, nbytes, repeats=5):
    """Device-to-device copy of a slab of the observation slab's size (hipMemcpyDtoD through torch), outside every
    timed region: bytes read + bytes written per second, the best of a few repeats.  What a pure streaming kernel
    reaches on THIS box, to be read beside the 8 TB/s spec peak."""
    src = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=device)
    src.zero_()
    dst.copy_(src)
    torch.cuda.synchronize()
    best = None
    for _ in range(repeats):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        dst.copy_(src)
        t1.record()
        t1.synchronize()
        ms = t0.elapsed_time(t1)
        best = ms if best is None or ms < best else best
    del src, dst
    return 2.0 * nbytes / (best * 1e-3) / 1e9


def measured_traffic(game):
    """HBM bytes per render launch from the committed PMC passes (tools/pmc_traffic.sh: FETCH_SIZE and WRITE_SIZE in
    separate rocprofv3 --pmc runs of this same command at 65 536 envs; KB → bytes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md §HBM prescribes for gfx950).  PMC counters cannot be read inside a timed run, so this is
    the latest profile on record, or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic_pmc.json")))
    if not files:
        return None, None, None
    with open(files[-1]) as f:
        data = json.load(f)
    entry = None
    for name, value in data.items():
        if name.startswith("_"):
            continue
        # "pg::variant0::<game>::render_kernel" (default distribution mode)
        # (chaser's is a template: "void pg::variant0::chaser::render_kernel<true>")
        if name.split("<")[0].endswith("::%s::render_kernel" % game) and ("variant" not in name or "variant0::" in name):
            entry = value
    if not entry:
        return None, None, None
    # Stale = taken on other kernel sources than the ones in the tree now (the profile carries their fingerprint; one
    # without a fingerprint predates the stamp and counts as stale).
    stale = data.get("_csrc_sha1") != csrc_fingerprint()
    return entry["bytes_corrected"], os.path.relpath(files[-1], ROOT), stale


def usable_cores():
    """Host cores this process may actually use: the scheduler affinity, capped by the cgroup CPU quota (a GPU box
    shows 256 logical CPUs with a 16-CPU quota; 256 threads there only queue behind each other)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:  # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline(game, run_seed, budget_s=20.0):
    """The CPU restatement (oracle/, kind "port") on a bounded sample of the same workload: same seeds
    (1 + env index), same action hash, same auto-reset policy, rendering on, one thread per host core."""
    import numpy as np
    from PIL import Image
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_util
    oracle_util.register_textures(game)
    L = oracle_util.oracle()
    cores = usable_cores()
    envs = 64 * cores
    h = L.pgo_vec_make(game.encode(), envs, 1, 0, 1)
    rate = L.pgo_vec_bench(h, 4, run_seed, cores)  # calibrate
    steps = max(8, int(budget_s * rate / envs))
    rate = L.pgo_vec_bench(h, steps, run_seed, cores)
    L.pgo_vec_close(h)
    return {"value": rate, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%s, first %d envs of the workload (seeds 1..%d), %d steps after a 4-step warm-up, "
                      "render on, %d threads" % (game, envs, envs, steps, cores)}


def cpu_baseline_mixed(games, counts, run_seed, budget_s=21.0):
    """The mixed workload on the CPU oracle: each game timed like cpu_baseline on its own bounded sample (budget split
    seven ways), combined as the workload combines them — a mixed step is counts[g] env-steps of every game g, so the
    rate is sum(counts) / sum(counts[g] / rate[g])."""
    per_game, cores, samples = {}, usable_cores(), []
    for game in games:
        b = cpu_baseline(game, run_seed, budget_s / len(games))
        per_game[game] = b["value"]
        samples.append(b["sample"])
    seconds_per_mixed_step = sum(c / per_game[g] for g, c in zip(games, counts))
    return {"value": sum(counts) / seconds_per_mixed_step, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "per_game": per_game,
            "sample": "each game on its own: " + "; ".join(samples) + " — combined in the workload's proportions %s" % (counts,)}


def launch_ranks(a):
    """--gpus N > 1 without a launcher: start the N ranks as children (one per GPU, RCCL rendezvous on 127.0.0.1).
    Nothing in this process has touched the GPU — torch is not even imported — so no initialised HIP runtime is ever
    replaced; the parent only waits and passes the children's output and exit code on."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    args = [x for x in sys.argv[1:] if x != "--dry-launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + args
    if a.dry_launch:
        print(json.dumps({"launch": cmd, "n_gpus": a.gpus}), flush=True)
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def first_contact_report(stage, rank, local_rank, world, error=None):
    """What a maintainer needs when the first multi-GPU run fails: which rank, on which device, at which stage, under
    which RCCL / HSA / rendezvous environment — as one JSON line on stderr instead of a bare traceback."""
    info = {"bench_error": "first contact failed at: " + stage, "rank": rank, "local_rank": local_rank, "world_size": world,
            "error": repr(error) if error is not None else "timed out",
            "env": {k: v for k, v in sorted(os.environ.items())
                    if k.startswith(("NCCL_", "RCCL_", "HSA_", "HIP_", "ROCR_", "GPU_", "MASTER_", "TORCHELASTIC_", "OMP_"))
                    or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK")}}
    try:
        import torch
        info["visible_devices"] = torch.cuda.device_count()
        if torch.cuda.is_available() and local_rank < torch.cuda.device_count():
            info["device"] = torch.cuda.get_device_name(local_rank)
    except Exception as exc:  # the report must come out whatever state torch is in
        info["device_query_error"] = repr(exc)
    print(json.dumps(info), file=sys.stderr, flush=True)


class FirstContact:
    """Guards the stages of a rank's first contact with the others (rendezvous, first barrier, first all_reduce, first
    gather): a stage that raises OR hangs past the timeout (BENCH_FIRST_CONTACT_TIMEOUT seconds, default 180; a collective
    over a broken link blocks rather than raises) ends the process non-zero behind a first_contact_report."""

    def __init__(self, rank, local_rank, world):
        self.rank, self.local_rank, self.world = rank, local_rank, world
        self.timeout = float(os.environ.get("BENCH_FIRST_CONTACT_TIMEOUT", "180"))

    def stage(self, name, fn):
        import threading

        def expired():
            first_contact_report(name, self.rank, self.local_rank, self.world)
            os._exit(3)  # (a blocked collective cannot be interrupted from Python)
        timer = threading.Timer(self.timeout, expired)
        timer.daemon = True
        timer.start()
        try:
            return fn()
        except BaseException as exc:
            first_contact_report(name, self.rank, self.local_rank, self.world, exc)
            raise SystemExit(3)
        finally:
            timer.cancel()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=512)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--game", default="coinrun")
    ap.add_argument("--mode", default=None, choices=["easy", "hard", "memory", "extreme"],
                    help="distribution mode of the game (default: the reference's compile-time one, which is what "
                         "BASELINE.json's metric is quoted on; other modes are reported without a CPU baseline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["single", "mixed"], default="single",
                    help="mixed = all seven games on every GPU, --envs split seven ways (the last game takes the "
                         "remainder), one HIP stream per game")
    ap.add_argument("--gather", action="store_true",
                    help="mixed workload only: rooted RCCL gather of obs/reward/done to rank 0 after every step")
    ap.add_argument("--settle", type=int, default=512,
                    help="untimed steps before the warm-up so that episode ends are spread out (steady state)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="with --gpus N > 1 and no WORLD_SIZE: print the torch.distributed.run command and exit")
    a = ap.parse_args()
    if a.steps < 1:
        ap.error("--steps must be >= 1")
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")

    if a.gpus > 1 and not (all(k in os.environ for k in ("WORLD_SIZE", "RANK", "MASTER_PORT")) or "TORCHELASTIC_RUN_ID" in os.environ):
        return launch_ranks(a)

    import torch
    from procgen2_amd.vec_env import ProcgenVecEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Under a launcher (WORLD_SIZE set) this is a rank of a process group even when it is the only one: the single-GPU
    # box then runs the same init / barrier / all_reduce / gather code the 8-GPU node will (tests/test_a_nccl_single_rank.py).
    # "Under a launcher" = its whole variable set is there: a scheduler or container image that merely exports WORLD_SIZE=1
    # leaves a plain single-GPU run a plain run (no rendezvous to fail at).
    distributed = all(k in os.environ for k in ("WORLD_SIZE", "RANK", "MASTER_PORT")) or "TORCHELASTIC_RUN_ID" in os.environ
    if not distributed:
        world, rank, local_rank = 1, 0, 0
    if distributed:
        import torch.distributed as dist
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        contact = FirstContact(rank, local_rank, world)
        contact.stage("torch.cuda.set_device(LOCAL_RANK)", lambda: torch.cuda.set_device(local_rank))
        contact.stage("init_process_group(nccl)", lambda: dist.init_process_group(
            "nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
            timeout=datetime.timedelta(seconds=contact.timeout)))

        def first_collectives():
            dist.barrier()
            probe = torch.ones(1, dtype=torch.float64, device="cuda")
            dist.all_reduce(probe, op=dist.ReduceOp.MAX)
            torch.cuda.synchronize()
        contact.stage("first barrier + all_reduce", first_collectives)
    n_gpus = world if distributed else 1
    if a.gpus != n_gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (a.gpus, n_gpus))

    run_seed = 0
    if a.workload == "mixed":
        return mixed(a, torch, ProcgenVecEnv, distributed, world, rank, local_rank, n_gpus, run_seed)
    env = ProcgenVecEnv(a.game, a.envs, device=local_rank, seed_base=1, env_offset=rank * a.envs,
                        distribution_mode=a.mode)
    env.reset()
    if a.settle > 0:
        env.timed_steps(a.settle, run_seed)      # untimed: spreads the episode ends out (see the module docstring)
    env.timed_steps(max(1, a.warmup), run_seed)  # untimed warm-up steps (same code path as the timed ones)

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    env.timed_steps(a.steps, run_seed, render_events=False)  # the steps and nothing else; returns after the stream has drained
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    fence()
    # Roofline leg and latency percentiles: the dominant kernel's launch durations and the per-step times over their own
    # steady-state window (HIP events on the engine's stream), independent of how short the timed region was asked to be.
    window = max(256, a.steps)
    ph = env.step_phases(window, run_seed)
    import numpy as np
    step_ms = ph["step"]
    render_avg_ms = float(ph["render"].mean())
    prepass_avg_ms, logic_avg_ms, late_avg_ms = float(ph["prepass"].mean()), float(ph["logic"].mean()), float(ph["late"].mean())
    p50_ms, mean_ms = float(np.median(step_ms)), float(step_ms.mean())
    per_rank_ms = [elapsed / a.steps * 1e3]
    if distributed:
        t = torch.tensor([elapsed, render_avg_ms, p50_ms, mean_ms, prepass_avg_ms, logic_avg_ms, late_avg_ms], dtype=torch.float64, device="cuda")
        mine = torch.tensor([elapsed / a.steps * 1e3], dtype=torch.float64, device="cuda")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)  # a straggler rank shows in the first SCALE record
        per_rank_ms = [float(x[0]) for x in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, render_avg_ms, p50_ms, mean_ms, prepass_avg_ms, logic_avg_ms, late_avg_ms = (float(x) for x in t)
    fence()
    copy_gbps = measured_copy_bandwidth(torch, env.device, a.envs * 12288) if rank == 0 else None

    done_frac = float(env.done.float().mean().item())
    env.close()

    if rank == 0:
        total_steps = float(n_gpus) * a.envs * a.steps
        value = total_steps / elapsed
        algo = ALGO_BYTES_PER_ENV_STEP * a.envs
        achieved = algo / (render_avg_ms * 1e-3) / 1e9
        # The path, not a launch: everything that turns the step's state into the frame (pre-pass + render kernel + late /
        # list kernels), and the whole step as `value` times it — a fraction that cannot rise by moving work across a
        # launch boundary.
        path_ms = prepass_avg_ms + render_avg_ms + late_avg_ms
        ms_per_step = elapsed / a.steps * 1e3
        traffic, traffic_src, traffic_stale = measured_traffic(a.game) if a.envs == 65536 and not a.mode else (None, None, None)
        line = {
            "metric": "env-steps/sec at 65536 envs, 64x64x3 obs",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": n_gpus,
            "steps": a.steps,
            "warmup": a.warmup,
            "settle_steps": a.settle,
            "ms_per_step": ms_per_step,
            "per_rank_ms": per_rank_ms,
            "p50_ms_per_step": p50_ms,
            "mean_ms_per_step_window": mean_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8/f32",
            "data": "synthetic",
            "config": {"workload": "%s, %d envs per GPU, uniform random actions 0..14 from a device counter hash, "
                                   "next-step auto-reset, seeds 1+global env index" % (a.game, a.envs),
                       "game": a.game, "distribution_mode": a.mode or "default", "envs_per_gpu": a.envs, "obs": "64x64x3 uint8", "parallelism": "env-shard x%d, no collective" % n_gpus},
            "obs_write_GBps": value * 12288 / 1e9,
            "roofline": {"bound": "hbm", "kernel": "%s::render_kernel" % a.game, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                         "whole_step_frac": algo / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "render_path": {"achieved": algo / (path_ms * 1e-3) / 1e9, "frac": algo / (path_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                         "avg_ms": path_ms, "what": "render pre-pass + render kernel + late / list kernels"},
                         "kernels": [{"name": "logic kernels (auto-reset install, agent / entities / resolve or the game's logic kernel)", "avg_us": logic_avg_ms * 1e3},
                                     {"name": "%s::setup_kernel (render pre-pass)" % a.game, "avg_us": prepass_avg_ms * 1e3},
                                     {"name": "%s::render_kernel" % a.game, "avg_us": render_avg_ms * 1e3},
                                     {"name": "late pass / list kernels behind the render launch", "avg_us": late_avg_ms * 1e3}],
                         "peak_measured": copy_gbps,
                         "peak_measured_how": "device-to-device copy of %d bytes (read + written bytes per second, best "
                                              "of 5), outside the timed region" % (a.envs * 12288),
                         "frac_of_measured": (achieved / copy_gbps) if copy_gbps else None,
                         "algorithmic_bytes_per_launch": algo,
                         "avg_launch_ms": render_avg_ms,
                         "window": "%d steps after the timed region (steps %d..%d since make), HIP events on the engine's "
                                   "stream between the phases of every step (pgv_step_phases): `frac` is the render kernel "
                                   "alone, `render_path` adds the pre-pass in front of it and what follows it inside the "
                                   "step, `whole_step_frac` is the same bytes over ms_per_step of the timed region"
                                   % (window, a.settle + max(1, a.warmup) + a.steps,
                                      a.settle + max(1, a.warmup) + a.steps + window - 1)},
            "done_fraction_last_step": done_frac,
        }
        if n_gpus == 1 and not a.no_cpu_baseline and not a.mode:
            line["cpu_baseline"] = cpu_baseline(a.game, run_seed)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


def gathered_steps(envs, gather, steps, run_seed):
    """The step loop of the gathered mixed workload: every engine steps on its own stream, then ONE rooted gather of the
    rank's three slabs — publish() orders the gather behind the engines' streams, consume() orders the engines' next step
    behind the gather, so a slab is neither sent before it is written nor overwritten while it is being sent.  No host
    synchronisation inside the loop.  (tests/test_distributed_cpu.py drives exactly this function over gloo.)"""
    for _ in range(steps):
        for e in envs:
            e.step_synthetic(run_seed, ordered=False)
        for e in envs:
            e.publish()   # torch's current stream waits for the engines' streams: no host synchronisation
        gather()          # one batch of point-to-point transfers: obs, reward, done slabs
        for e in envs:
            e.consume()   # … and the next step does not overwrite the slab while it is being sent
    for e in envs:
        e.sync()


def mixed(a, torch, ProcgenVecEnv, distributed, world, rank, local_rank, n_gpus, run_seed):
    """BASELINE.json configs[4]: the seven games side by side on every GPU.  Each game is its own vector env on its own
    HIP stream (they overlap on the device); the envs of a game are sharded over the ranks by global index like the
    single-game workload.  No collective unless --gather."""
    from procgen2_amd.vec_env import GAMES
    if distributed:
        import torch.distributed as dist
    from procgen2_amd.vec_env import RootGather
    base = a.envs // len(GAMES)
    counts = [base] * (len(GAMES) - 1) + [a.envs - base * (len(GAMES) - 1)]
    device = torch.device("cuda", local_rank)
    # ONE slab per GPU whatever the game: every game's env writes its block of it.  On the root of a
    # gathered run that slab is the root's slice of the gathered batch, so its own block is never copied.
    gathering = a.gather and distributed
    shapes = (((64, 64, 3), torch.uint8), ((), torch.float32), ((), torch.uint8))
    whole = None
    if gathering and rank == 0:
        whole = tuple(torch.zeros((world * a.envs,) + sh, dtype=dt, device=device) for sh, dt in shapes)
        local = tuple(t[rank * a.envs:(rank + 1) * a.envs] for t in whole)
    else:
        local = tuple(torch.zeros((a.envs,) + sh, dtype=dt, device=device) for sh, dt in shapes)
    envs, at = [], 0
    for game, count in zip(GAMES, counts):  # own stream each
        envs.append(ProcgenVecEnv(game, count, device=local_rank, seed_base=1, env_offset=rank * count,
                                  out=tuple(t[at:at + count] for t in local)))
        at += count
    for e in envs:
        e.reset()
    gather = None
    if gathering:  # 3 transfers per peer and step; the plan is one all_gather of the per-rank counts
        gather = FirstContact(rank, local_rank, world).stage("RootGather plan (all_gather of the counts)",
                                                             lambda: RootGather(local, dst=0, slabs=whole))
    if gather is not None:  # the first point-to-point transfers this node has ever made: guarded like the rendezvous

        def first_gather():
            for e in envs:
                e.publish()
            gather()
            for e in envs:
                e.consume()
            torch.cuda.synchronize()
        FirstContact(rank, local_rank, world).stage("first RootGather (batch_isend_irecv)", first_gather)

    from procgen2_amd.vec_env import step_many_synthetic

    def run(steps):
        if gather is None:  # no per-step work on the host: the whole launch loop in C
            step_many_synthetic(envs, steps, run_seed)
            for e in envs:
                e.sync()
            return
        gathered_steps(envs, gather, steps, run_seed)

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    if a.settle > 0:
        run(a.settle)
    run(max(1, a.warmup))
    fence()
    t0 = time.perf_counter()
    run(a.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    fence()
    # Per-game phases over a window of their own, all seven stepping side by side as in the timed region (events on each
    # game's stream; the streams overlap, so a game's numbers are what its stream saw, not a share of the wall clock).
    from procgen2_amd.vec_env import step_phases_many
    import numpy as np
    window = 128
    ph = step_phases_many(envs, window, run_seed).mean(axis=2)  # [game][step, logic, prepass, render, late]
    per_rank_ms = [elapsed / a.steps * 1e3]
    if distributed:
        mine = torch.tensor([elapsed / a.steps * 1e3], dtype=torch.float64, device="cuda")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [float(x[0]) for x in every]
    fence()
    for e in envs:
        e.close()
    if rank == 0:
        value = float(n_gpus) * a.envs * a.steps / elapsed
        algo = ALGO_BYTES_PER_ENV_STEP * a.envs
        achieved = algo / (elapsed / a.steps) / 1e9
        kernels, paths = [], {}
        for g, (game, count) in enumerate(zip(GAMES, counts)):
            for j, what in ((1, "logic kernels"), (2, "setup_kernel (render pre-pass)"), (3, "render_kernel"), (4, "late pass / list kernels")):
                kernels.append({"name": "%s: %s" % (game, what), "avg_us": float(ph[g][j]) * 1e3})
            path_ms = float(ph[g][2] + ph[g][3] + ph[g][4])
            paths[game] = {"envs": count, "avg_ms": path_ms, "achieved": ALGO_BYTES_PER_ENV_STEP * count / (path_ms * 1e-3) / 1e9,
                           "step_ms": float(ph[g][0])}
        line = {
            "metric": "env-steps/sec, all 7 games mixed, 64x64x3 obs",
            "value": value, "unit": "env-steps/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
            "settle_steps": a.settle,
            "ms_per_step": elapsed / a.steps * 1e3, "per_rank_ms": per_rank_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/f32", "data": "synthetic",
            "config": {"workload": "all seven games, %d envs per GPU split %s, one stream per game, uniform random "
                                   "actions from a device counter hash, next-step auto-reset" % (a.envs, counts),
                       "games": list(GAMES), "envs_per_gpu": a.envs, "obs": "64x64x3 uint8",
                       "parallelism": "env-shard x%d, %s" % (n_gpus, "rooted RCCL gather to rank 0 every step"
                                                             if a.gather and distributed else "no collective")},
            "obs_write_GBps": value * 12288 / 1e9,
            "roofline": {"bound": "hbm", "kernel": "whole step, all games (wall clock, not one kernel)",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "whole_step_frac": achieved / HBM_PEAK_GBPS,
                         "render_path": {"per_game": paths,
                                         "what": "per game, on its own stream while the other six run beside it: render pre-pass + "
                                                 "render kernel + late / list kernels of its share of the slab (the streams "
                                                 "overlap: these do not add up to the wall clock)"},
                         "kernels": kernels,
                         "window": "%d steps after the timed region, all seven games side by side, HIP events on each game's "
                                   "stream (pgv_step_phases_many)" % window,
                         "traffic": None},
        }
        if n_gpus == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_mixed(list(GAMES), counts, run_seed)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
