"""The RCCL code path on the one GPU a test box has.

The scaling curve at 1 / 2 / 4 / 8 GPUs is the driver's to measure on an 8-GPU node; what CAN be checked on one GPU is
that the code the ranks will run starts at all: `init_process_group("nccl")`, the barrier + all_reduce of the timing
fence, `RootGather`'s plan (an all_gather) around the engine's own slabs, `publish()` → gather → `consume()` every
step.  Each case starts `python -m torch.distributed.run --nproc-per-node 1 … bench.py` as a fresh CHILD process (never
an exec of this one), exactly as the driver launches the N > 1 runs, and reads rank 0's JSON line.

The module's name sorts first so that pytest collects and runs it before any test of this process has initialised HIP;
the children are independent of that either way.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _bench_under_the_launcher(args, timeout=600):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert out.returncode == 0, "rc %d\n--- stdout\n%s\n--- stderr\n%s" % (out.returncode, out.stdout[-2000:], out.stderr[-4000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_mixed_workload_with_the_rooted_gather_runs_over_nccl_on_one_rank():
    line = _bench_under_the_launcher(["--workload", "mixed", "--gather", "--steps", "8", "--warmup", "2", "--settle", "0",
                                      "--envs", "7168"])
    assert line["n_gpus"] == 1 and line["steps"] == 8 and line["value"] > 0
    assert "gather" in line["config"]["parallelism"], line["config"]["parallelism"]
    assert line["scaling"] == "weak"


def test_headline_workload_runs_as_a_rank_of_a_process_group():
    line = _bench_under_the_launcher(["--steps", "8", "--warmup", "2", "--settle", "0", "--no-cpu-baseline"])
    assert line["n_gpus"] == 1 and line["steps"] == 8 and line["value"] > 0
    assert line["config"]["parallelism"].startswith("env-shard x1")
    assert line["roofline"]["frac"] > 0 and "cpu_baseline" not in line
