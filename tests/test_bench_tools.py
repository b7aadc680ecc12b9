"""The bench line's regression gate (tools/check_bench.py) and bench.py's first-contact guard: host
logic only, no GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(value, frac, game="coinrun"):
    return {"metric": "env-steps/sec at 65536 envs, 64x64x3 obs", "value": value, "n_gpus": 1,
            "config": {"game": game, "envs_per_gpu": 65536, "distribution_mode": "default"},
            "roofline": {"whole_step_frac": frac}}


def _check(tmp_path, new, ref_dir):
    path = os.path.join(tmp_path, "new.json")
    with open(path, "w") as f:
        f.write("some log text\n" + json.dumps(new) + "\n")
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_bench.py"), path, "--against", ref_dir],
                          capture_output=True, text=True)


def test_check_bench_flags_a_drop_and_passes_a_gain(tmp_path):
    ref_dir = os.path.join(tmp_path, "ref")
    os.makedirs(ref_dir)
    with open(os.path.join(ref_dir, "bench_coinrun.json"), "w") as f:
        json.dump(_line(135.9e6, 0.209), f)
    with open(os.path.join(ref_dir, "bench_coinrun_driver_args.json"), "w") as f:  # (never the reference: another window)
        json.dump(_line(500e6, 0.9), f)
    ok = _check(tmp_path, _line(136.2e6, 0.2093), ref_dir)
    assert ok.returncode == 0, ok.stdout
    # round 4's regression: the headline down while a kernel's own fraction went up — whole_step_frac falls with it
    bad = _check(tmp_path, _line(122.2e6, 0.188), ref_dir)
    assert bad.returncode == 1 and "REGRESSION" in bad.stdout and "value fell" in bad.stdout
    within = _check(tmp_path, _line(133.0e6, 0.204), ref_dir)  # −2 %: inside the pool's box-to-box spread
    assert within.returncode == 0
    other = _check(tmp_path, _line(90e6, 0.14, game="chaser"), ref_dir)
    assert other.returncode == 2  # no reference line for that workload


def test_check_bench_accepts_the_committed_lines():
    """Every committed line of this round passes the gate against the round before (what tools/refresh_all.sh runs)."""
    new_dir, ref_dir = os.path.join(ROOT, "profiles", "bench_r06"), os.path.join(ROOT, "profiles", "bench_r05")
    if not os.path.isdir(new_dir):
        import pytest
        pytest.skip("no round-6 lines committed yet")
    for name in sorted(os.listdir(new_dir)):
        if "driver_args" in name or not name.endswith(".json"):
            continue
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_bench.py"), os.path.join(new_dir, name),
                            "--against", ref_dir], capture_output=True, text=True)
        assert r.returncode == 0, name + ": " + r.stdout


def test_first_contact_reports_a_raising_and_a_hanging_stage():
    """bench.FirstContact: a stage that raises, and one that never returns, both end the process with code 3 behind one
    JSON line on stderr that names the stage, the rank and the rendezvous environment."""
    code = (
        "import sys, time; sys.path.insert(0, %r); import bench\n"
        "c = bench.FirstContact(rank=5, local_rank=1, world=8)\n"
        "mode = sys.argv[1]\n"
        "c.stage('warm', lambda: 1)\n"
        "c.stage('first barrier', (lambda: 1 / 0) if mode == 'raise' else (lambda: time.sleep(60)))\n"
        "print('not reached')\n" % ROOT)
    env = dict(os.environ, BENCH_FIRST_CONTACT_TIMEOUT="2", MASTER_ADDR="127.0.0.1", NCCL_DEBUG="WARN")
    for mode in ("raise", "hang"):
        r = subprocess.run([sys.executable, "-c", code, mode], capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 3, (mode, r.returncode, r.stderr[-400:])
        assert "not reached" not in r.stdout
        report = json.loads([ln for ln in r.stderr.splitlines() if ln.startswith("{")][-1])
        assert report["rank"] == 5 and report["local_rank"] == 1 and report["world_size"] == 8
        assert "first barrier" in report["bench_error"]
        assert report["env"]["MASTER_ADDR"] == "127.0.0.1" and report["env"]["NCCL_DEBUG"] == "WARN"
        assert ("ZeroDivisionError" in report["error"]) == (mode == "raise")
