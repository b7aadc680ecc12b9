"""The product build carries no experiment scaffolding: no -DPG_ABLATE / -DPG_TIMELINE / -DPG_MARKS
in the recipe of the libraries that ship, no environment variable steering the launch path of the engine."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "procgen2_amd", "csrc")


def test_product_build_flags_define_no_probe_macro(monkeypatch):
    monkeypatch.delenv("PG_MORE_FLAGS", raising=False)
    monkeypatch.delenv("PG_EXTRA_FLAGS", raising=False)
    import importlib
    import procgen2_amd.build as b
    b = importlib.reload(b)
    commands = []
    monkeypatch.setattr(b, "_run", lambda cmd, verbose: commands.append(cmd))
    monkeypatch.setattr(b, "_stale", lambda out, deps: True)
    monkeypatch.setattr(b, "_parallel", lambda jobs, verbose: commands.extend(jobs))
    b.build(force=True, verbose=False)
    assert len(commands) > 20  # every variant of every game, the engines, the links
    for cmd in commands:
        for word in cmd:
            assert not re.match(r"-DPG_(ABLATE|TIMELINE|MARKS|EXP_|FAT_WHY|CHASER_SKIP)", word), " ".join(cmd)
    compiles = [c for c in commands if "-c" in c]
    assert all("--offload-arch=gfx950" in c and "-ffp-contract=off" in c for c in compiles)


def test_probe_macros_are_defined_in_one_header_only():
    """PG_ABL / PG_MARK / PG_TL come from pg_probe.h and nowhere else, and expand to nothing unless an experiment build
    asks: a kernel source that rolled its own would dodge the flag test above."""
    for name in sorted(os.listdir(CSRC)):
        if name == "pg_probe.h" or not name.endswith((".hip", ".h", ".cpp")):
            continue
        text = open(os.path.join(CSRC, name)).read()
        assert not re.search(r"#\s*define\s+PG_(ABL|MARK|TL)\b", text), name
        assert "PG_TIMELINE" not in text and "PG_MARKS" not in text, name
        # (engine.hip's one `#ifndef PG_ABLATE` is the product REFUSING the experiment bits of the debug word)
        assert not re.search(r"#\s*if(def)?\s.*PG_ABLATE", text), name


def test_no_environment_variable_steers_the_launch_path():
    """getenv in the engine: the asset root and the hardware-queue count of the HIP runtime, nothing else."""
    seen = []
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h", ".cpp")):
            seen += re.findall(r'getenv\("([A-Z0-9_]+)"\)', open(os.path.join(CSRC, name)).read())
    assert sorted(seen) == ["GPU_MAX_HW_QUEUES", "PROCGEN2_ASSETS"], seen
