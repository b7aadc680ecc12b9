#!/usr/bin/env python3
"""Records outputs of THE REFERENCE's own compiled sources (oracle/_ref, built by `make -C oracle` from /root/reference
— see oracle/ref_driver.cpp) as fixtures that travel to the GPU box, where the reference tree does not exist:

  maze_levels  games/maze: what a fresh cenv_make(seed = L) builds first (side draw, Kruskal, goal), worlds 25/15/31
  setmaze      games/chaser + jumper: generate_maze / generate_maze_no_dead_ends on a fresh engine
  rooms        games/caveflyer + jumper: find_best_room iteration order, find_path, expand_room order on random caves

Run in the build container:  python tests/golden/make_ref_fixtures.py   (writes tests/golden/ref_fixtures.json)
Inputs and expected outputs only — no reference source text."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ref_util  # noqa: E402


def bits(a):
    return "".join(str(int(v)) for v in a)


def main():
    assert ref_util.have_reference_build(), "needs /root/reference (oracle/_ref)"
    ref = ref_util.Side("ref")
    out = {"generated_by": "tests/golden/make_ref_fixtures.py from oracle/_ref (reference sources compiled as-is)",
           "maze_levels": [], "setmaze": [], "rooms": []}
    for world, seeds in ((25, range(0, 64)), (15, range(100, 116)), (31, range(200, 216))):
        for seed in seeds:
            dim, grid, nxt = ref.maze_level(seed, world)
            out["maze_levels"].append({"seed": seed, "world": world, "dim": dim, "grid": bits(grid), "next": nxt})
    for dim, nde, seeds in ((11, 0, range(0, 64)), (13, 0, range(0, 16)), (19, 0, range(0, 16)), (13, 1, range(0, 16)),
                            (6, 1, range(0, 8)), (15, 1, range(0, 8))):
        for seed in seeds:
            grid, nxt = ref.setmaze_generate(seed, dim, nde)
            out["setmaze"].append({"seed": seed, "dim": dim, "no_dead_ends": nde, "grid": bits(grid), "next": nxt})
    rng = np.random.default_rng(5)
    for gw, gh, count in ((40, 40, 6), (20, 20, 6), (45, 45, 2)):
        for _ in range(count):
            cave = ref.rooms_update(gw, gh, ref_util.random_cave(rng, gw, gh), 2)
            a, b, e = int(rng.integers(0, 1 << 30)), int(rng.integers(0, 1 << 30)), 4
            best, path, wide = ref.rooms_analyse(gw, gh, cave, a, b, e)
            out["rooms"].append({"gw": gw, "gh": gh, "cave": bits(cave), "src_sel": a, "dst_sel": b, "expand": e,
                                 "best_order": best.tolist(), "path": path.tolist(), "wide_order": wide.tolist()})
    with open(os.path.join(HERE, "ref_fixtures.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", os.path.join(HERE, "ref_fixtures.json"))


if __name__ == "__main__":
    main()
