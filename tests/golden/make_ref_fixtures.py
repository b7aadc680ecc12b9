#!/usr/bin/env python3
"""Records outputs of THE REFERENCE's own compiled sources (oracle/_ref, built by `make -C oracle` from /root/reference
— see oracle/ref_driver.cpp) as fixtures that travel to the GPU box, where the reference tree does not exist:

  maze_levels  games/maze: what a fresh cenv_make(seed = L) builds first (side draw, Kruskal, goal), worlds 25/15/31
  setmaze      games/chaser + jumper: generate_maze / generate_maze_no_dead_ends on a fresh engine
  rooms        games/caveflyer + jumper: two automaton updates, find_best_room iteration order, find_path, expand_room
               order on random caves
  (ref_aabb_ecs.npz)  games/*/helpers.cpp check_collision / get_collision_overlap on 12 000 rectangle pairs; games/*/ecs.cpp:
               24 create / destroy / remove-component / clear scripts against the reference's Coordinator, run one after
               the other in a fresh process (the sets keep their bucket arrays across clear()), with the entity ids it
               handed out and the iteration order of its three systems' entity sets after every operation

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
            raw = ref_util.random_cave(rng, gw, gh)
            cave = ref.rooms_update(gw, gh, raw, 2)
            a, b, e = int(rng.integers(0, 1 << 30)), int(rng.integers(0, 1 << 30)), 4
            best, path, wide = ref.rooms_analyse(gw, gh, cave, a, b, e)
            out["rooms"].append({"gw": gw, "gh": gh, "raw": bits(raw), "cave": bits(cave), "src_sel": a, "dst_sel": b, "expand": e,
                                 "best_order": best.tolist(), "path": path.tolist(), "wide_order": wide.tolist()})
    with open(os.path.join(HERE, "ref_fixtures.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", os.path.join(HERE, "ref_fixtures.json"))
    aabb_and_ecs(ref)


def aabb_and_ecs(ref):
    """helpers.cpp and ecs.cpp outputs as arrays (tests/golden/ref_aabb_ecs.npz).  The ECS part must come from a process
    in which ref_ecs_script has not run before: the Coordinator is the reference's global `c`, and its sets' bucket
    arrays survive clear_entities() — which is exactly what the device twin has to reproduce."""
    rng = np.random.default_rng(11)
    n = 12000
    a, b = ref_util.rect_pairs(rng, n)  # the games' hit boxes on a 1/64 lattice (exact touching is common) + raw random ones
    hit, overlap = ref.collisions(a, b)
    scripts = []
    for i in range(24):
        ops, args = ref_util.ecs_random_script(rng, 120 + 40 * (i % 5))
        ids, orders = ref.ecs_script(ops, args)
        assert ids.max() < 2000, "entity ids must stay below the device twin's table"
        scripts.append((ops, args, ids, orders))
    np.savez_compressed(os.path.join(HERE, "ref_aabb_ecs.npz"), a=a, b=b, hit=hit, overlap=overlap,
                        n_scripts=np.int32(len(scripts)),
                        **{"ops%d" % i: v[0] for i, v in enumerate(scripts)}, **{"args%d" % i: v[1] for i, v in enumerate(scripts)},
                        **{"ids%d" % i: v[2] for i, v in enumerate(scripts)}, **{"orders%d" % i: v[3] for i, v in enumerate(scripts)})
    print("wrote", os.path.join(HERE, "ref_aabb_ecs.npz"), "(%d hits of %d pairs)" % (int(hit.sum()), n))


if __name__ == "__main__":
    main()
