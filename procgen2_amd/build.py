"""Build recipe for the HIP engine (gfx950 only) — explicit hipcc calls, outputs in-tree.

    python -m procgen2_amd.build            # build what is stale
    python -m procgen2_amd.build --force

Produces, under procgen2_amd/lib/:
    libprocgen2_hip.so   the engine: vector ABI (include/procgen2_vec.h) + cenv ABI (include/procgen2_cenv.h)
    libCoinRun.so        same objects, cenv_make defaults to coinrun  (reference name: games/coinrun/CMakeLists.txt)
    libMaze.so           same objects, cenv_make defaults to maze     (reference name: games/maze/CMakeLists.txt)

hipcc cross-compiles for gfx950 without a GPU present.  -ffp-contract=off is mandatory: the reference
x86-64 build forms no FMAs and every float result has to match it bit for bit.
"""
import argparse
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
OBJ = os.path.join(HERE, "build")
ARCH = "gfx950"

COMMON = os.environ.get("PG_MORE_FLAGS", "").split() + ["--offload-arch=" + ARCH, "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
          "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", "-I" + CSRC]

KERNEL_SOURCES = ["coinrun.hip", "maze.hip", "bossfight.hip", "climber.hip", "caveflyer.hip", "chaser.hip", "jumper.hip"]
# Distribution modes: a game's source is compiled once per variant (-DPG_VARIANT=k, csrc/pg_defs.h); variant 0 is the
# reference's compile-time default.  engine.hip's kVariants table maps (game, mode) to these.
VARIANTS = {"coinrun.hip": 1, "maze.hip": 3, "bossfight.hip": 2, "climber.hip": 2, "caveflyer.hip": 3, "chaser.hip": 3,
            "jumper.hip": 3}
HOST_SOURCES = ["png_decode.cpp"]
ENGINE = "engine.hip"
ALIASES = {"libprocgen2_hip.so": 0, "libCoinRun.so": 0, "libMaze.so": 1, "libBossFight.so": 2, "libClimber.so": 3,
           "libCaveFlyer.so": 4, "libChaser.so": 5,
           "libJumper.so": 6}


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    inc = os.path.join(HERE, "..", "include")
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
           [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def _parallel(jobs, verbose):
    """Independent compiler invocations, a few at a time (each hipcc is single-threaded and takes ~1 GiB)."""
    if not jobs:
        return
    from concurrent.futures import ThreadPoolExecutor
    workers = max(1, min(len(jobs), int(os.environ.get("PG_BUILD_JOBS", "0")) or min(6, os.cpu_count() or 1)))
    with ThreadPoolExecutor(workers) as pool:
        for f in [pool.submit(_run, cmd, verbose) for cmd in jobs]:
            f.result()


def build(force=False, verbose=True, ablate=False):
    """ablate=True: the timing-experiment build (-DPG_ABLATE, csrc/pg_render.h) of the engine alone, as
    lib/libprocgen2_hip_ablate.so with its objects in build_ablate/ — for tools/ablate_render.py, never the product."""
    common, aliases, obj_dir = COMMON, ALIASES, OBJ
    if ablate:
        common = COMMON + ["-DPG_ABLATE"] + os.environ.get("PG_EXTRA_FLAGS", "").split()  # (experiments only)
        aliases = {"libprocgen2_hip_ablate.so": 0}
        obj_dir = OBJ + "_ablate"
    return _build(force, verbose, common, aliases, obj_dir)


def _build(force, verbose, COMMON, ALIASES, OBJ):
    os.makedirs(LIB, exist_ok=True)
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    hdrs = _headers()
    objs, jobs = [], []
    for src in KERNEL_SOURCES + HOST_SOURCES:
        path = os.path.join(CSRC, src)
        for k in range(VARIANTS.get(src, 1)):
            obj = os.path.join(OBJ, os.path.splitext(src)[0] + ("_v%d.o" % k if src in VARIANTS else ".o"))
            if force or _stale(obj, [path] + hdrs):
                jobs.append([cc] + COMMON + ["-DPG_VARIANT=%d" % k, "-c", path, "-o", obj])
            objs.append(obj)
    engines = {}
    for lib, game in ALIASES.items():
        eng = os.path.join(OBJ, "engine_g%d.o" % game)
        if eng not in engines.values() and (force or _stale(eng, [os.path.join(CSRC, ENGINE)] + hdrs)):
            jobs.append([cc] + COMMON + ["-DPG_DEFAULT_GAME=%d" % game, "-c", os.path.join(CSRC, ENGINE), "-o", eng])
        engines[lib] = eng
    _parallel(jobs, verbose)
    outputs, links = [], []
    for lib, eng in engines.items():
        out = os.path.join(LIB, lib)
        if force or jobs or _stale(out, objs + [eng]):
            links.append([cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out] + objs + [eng, "-lz", "-ldl"])
        outputs.append(out)
    _parallel(links, verbose)
    if "libprocgen2_hip.so" in ALIASES:
        outputs.append(_build_selftest(force, verbose, cc, hdrs, COMMON))
    return outputs


def _build_selftest(force, verbose, cc, hdrs, common):
    """tests/hip/selftest.hip (+ selftest_rooms.hip once per world size) → lib/libpg_selftest.so: the device-side sweep of
    the bit-exact primitives (test infrastructure, loaded only by tests/test_primitives_gpu.py; in lib/ so that it travels
    to the GPU box)."""
    hip_dir = os.path.join(HERE, "..", "tests", "hip")
    src = os.path.join(hip_dir, "selftest.hip")
    rooms = os.path.join(hip_dir, "selftest_rooms.hip")
    out = os.path.join(LIB, "libpg_selftest.so")
    if os.path.exists(src) and os.path.exists(rooms) and (force or _stale(out, [src, rooms] + hdrs)):
        objs, jobs = [], []
        obj = os.path.join(OBJ, "selftest.o")
        jobs.append([cc] + common + ["-c", src, "-o", obj])
        objs.append(obj)
        for k in range(3):  # pg_rooms.h is a per-size header: variants 0 / 1 / 2 = 40 / 20 / 45 cells a side
            obj = os.path.join(OBJ, "selftest_rooms_v%d.o" % k)
            jobs.append([cc] + common + ["-DPG_VARIANT=%d" % k, "-c", rooms, "-o", obj])
            objs.append(obj)
        _parallel(jobs, verbose)
        _run([cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out] + objs, verbose)
    return out


def library_path(name="libprocgen2_hip.so"):
    return os.path.join(LIB, name)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--ablate", action="store_true", help="build lib/libprocgen2_hip_ablate.so (-DPG_ABLATE) instead")
    a = ap.parse_args()
    for o in build(force=a.force, verbose=not a.quiet, ablate=a.ablate):
        print("built", os.path.relpath(o))
    sys.exit(0)
