"""Host/device primitives of the engine (procgen2_amd/csrc/pg_rng.h, pg_order.h) compiled for the CPU and
checked against the real libstdc++ the reference depends on (SURVEY.md rows T1–T4 incl. their known answers)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_primitives_match_libstdcxx(tmp_path):
    exe = str(tmp_path / "test_primitives")
    subprocess.run(["g++", "-std=gnu++17", "-O2", "-mfma", "-ffp-contract=off", "-I" + os.path.join(ROOT, "procgen2_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpp", "test_primitives.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    for section in ("OK axis template", "OK rot box", "OK span nested", "OK sincos", "OK blend", "OK mt19937", "OK distributions", "OK hash_order", "OK sort", "ALL OK"):
        assert section in out.stdout
