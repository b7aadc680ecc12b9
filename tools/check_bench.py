#!/usr/bin/env python3
"""Regression gate for bench.py lines.

    python tools/check_bench.py NEW.json [--against profiles/bench_r05] [--tolerance 0.03]
    python bench.py ... | python tools/check_bench.py -            # a line on stdin

NEW.json holds one bench.py JSON line (or a driver record with the line under "parsed").  The reference is the
committed line of the same workload under --against (matched by config.game / "mixed" and envs_per_gpu, the line with
the driver's arguments excluded).  Fails (exit 1) when `value` or `roofline.whole_step_frac` is more than --tolerance
below the reference: the r04 regression (headline down while the render kernel's own fraction went up) would have
tripped it.  Boxes of the pool differ by 1-3 %, hence the 3 % default; same-box pairs deserve a tighter one.
"""
import argparse
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_line(path):
    text = sys.stdin.read() if path == "-" else open(path).read()
    last = None
    for raw in text.splitlines():
        raw = raw.strip()
        if raw.startswith("{"):
            try:
                last = json.loads(raw)
            except ValueError:
                continue
    if last is None:
        last = json.loads(text)
    return last.get("parsed", last)


def key_of(line):
    cfg = line.get("config", {})
    game = cfg.get("game") or ("mixed" if "games" in cfg else None)
    return game, cfg.get("envs_per_gpu"), cfg.get("distribution_mode", "default"), line.get("n_gpus")


def reference_for(line, directory):
    best = None
    for path in sorted(glob.glob(os.path.join(directory, "*.json"))):
        if "driver_args" in os.path.basename(path):
            continue
        try:
            ref = load_line(path)
        except (ValueError, OSError):
            continue
        if "value" in ref and key_of(ref) == key_of(line):
            best = (path, ref)
    return best


def check(new, ref, tolerance):
    problems = []
    pairs = [("value", new.get("value"), ref.get("value")),
             ("roofline.whole_step_frac", new.get("roofline", {}).get("whole_step_frac"),
              ref.get("roofline", {}).get("whole_step_frac"))]
    for name, now, before in pairs:
        if now is None or before is None:
            continue
        if now < before * (1.0 - tolerance):
            problems.append("%s fell %.1f %%: %.6g -> %.6g" % (name, (1.0 - now / before) * 100.0, before, now))
    return problems


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("new")
    ap.add_argument("--against", default=os.path.join(ROOT, "profiles", "bench_r05"))
    ap.add_argument("--tolerance", type=float, default=0.03)
    a = ap.parse_args()
    new = load_line(a.new)
    found = reference_for(new, a.against)
    if not found:
        print("check_bench: no reference line for %s under %s" % (key_of(new), a.against))
        return 2
    path, ref = found
    problems = check(new, ref, a.tolerance)
    print("check_bench: %s against %s: value %.4g (was %.4g), whole_step_frac %s (was %s)" % (
        key_of(new)[0], os.path.relpath(path, ROOT), new["value"], ref["value"],
        new.get("roofline", {}).get("whole_step_frac"), ref.get("roofline", {}).get("whole_step_frac")))
    for p in problems:
        print("check_bench: REGRESSION: " + p)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
