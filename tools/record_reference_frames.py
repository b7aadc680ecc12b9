#!/usr/bin/env python3
"""Record observation frames from the REAL reference (needs a box with the reference built against SDL3 + SDL3_image —
not this image: see tools/record_reference_frames.md).  Writes tests/golden/sdl_frames.npz, which
tests/test_sdl_frames.py then holds the oracle (CPU) and the HIP engine (GPU) to, byte for byte.

    python tools/record_reference_frames.py --reference /path/to/Procgen2 --build /path/to/Procgen2/build \
        [--out tests/golden/sdl_frames.npz] [--steps 96] [--seeds 1 7 123]

It binds the reference's own shared libraries through the reference's own cenv/cenv.py (class CEnv: make with a seed
option, reset, step) — nothing of this repo is in the loop.  Actions are the LCG of SURVEY.md Appendix C (the same stream
the reward traces use), resets are the caller-side `if terminated: reset()` of game_test.py:36-40.

File format (numpy .npz, one entry per array):
    games    [R]        '<U16'   game name of record r
    seeds    [R]        int64    the seed handed to cenv_make
    actions  [R, S]     int32    the action of step s (0..14)
    frames   [R, S+1, 64, 64, 3] uint8   frames[r, 0] = the observation reset() returned, frames[r, s+1] = after step s
                                         (after the caller-side reset where step s terminated)
    resets   [R, S]     uint8    1 where step s terminated and the caller reset
    sdl      []         '<U64'   SDL_GetRevision() / version string of the SDL3 the reference was linked against
"""
import argparse
import importlib.util
import os
import sys

import numpy as np

LIBS = {"coinrun": "CoinRun", "maze": "Maze", "caveflyer": "CaveFlyer", "bossfight": "BossFight", "chaser": "Chaser",
        "jumper": "Jumper", "climber": "Climber"}


def lcg_actions(seed, steps):
    s, out = (seed * 2654435761) & 0xffffffff, []
    for _ in range(steps):
        s = (s * 1664525 + 1013904223) & 0xffffffff
        out.append((s >> 16) % 15)
    return np.array(out, np.int32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="checkout of Farama-Foundation/Procgen2")
    ap.add_argument("--build", required=True, help="its build directory (holds lib<Game>.so)")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sdl_frames.npz"))
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--seeds", type=int, nargs="+", default=[1, 7, 123])
    ap.add_argument("--sdl", default="unknown", help="version / revision string of the SDL3 linked in")
    a = ap.parse_args()
    spec = importlib.util.spec_from_file_location("ref_cenv", os.path.join(a.reference, "cenv", "cenv.py"))
    ref_cenv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_cenv)
    os.chdir(a.reference)  # the games open "assets/..." relative to the working directory
    games, seeds, actions, frames, resets = [], [], [], [], []
    for game, lib in LIBS.items():
        path = None
        for root, _, files in os.walk(a.build):
            for f in files:
                if f in ("lib%s.so" % lib, "%s.dll" % lib, "lib%s.dylib" % lib):
                    path = os.path.join(root, f)
        if path is None:
            print("skipping %s: no lib%s under %s" % (game, lib, a.build), file=sys.stderr)
            continue
        for seed in a.seeds:
            env = ref_cenv.CEnv(path, options={"seed": seed})
            obs, _ = env.reset()
            acts = lcg_actions(seed, a.steps)
            fr, rs = [np.asarray(obs["screen"], np.uint8).reshape(64, 64, 3).copy()], []
            for act in acts:
                obs, _, term, _, _ = env.step(int(act))
                if term:
                    obs, _ = env.reset()
                rs.append(1 if term else 0)
                fr.append(np.asarray(obs["screen"], np.uint8).reshape(64, 64, 3).copy())
            env.close()
            games.append(game)
            seeds.append(seed)
            actions.append(acts)
            frames.append(np.stack(fr))
            resets.append(np.array(rs, np.uint8))
    np.savez_compressed(a.out, games=np.array(games), seeds=np.array(seeds, np.int64), actions=np.stack(actions),
                        frames=np.stack(frames), resets=np.stack(resets), sdl=np.array(a.sdl))
    print("wrote %s: %d records of %d steps" % (a.out, len(games), a.steps))


if __name__ == "__main__":
    main()
