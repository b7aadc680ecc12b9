"""The oracle against THE REFERENCE ITSELF, where the reference compiles as-is (VERDICT r01 "missing #1").

oracle/_ref holds the reference's SDL-free hot-path sources built where they lie with plain g++ — maze_generator.cpp
(maze; chaser == jumper), room_generator.cpp (caveflyer == jumper), helpers.cpp, ecs.cpp — behind the small driver
oracle/ref_driver.cpp.  Each test runs the reference and the oracle's restatement (oracle/pgo_hooks.cpp → the same
Carver / carve_merged / Rooms / IdPool / IdSet the oracle's games use) on the same seeds and inputs and demands equal
outputs: grids, free-cell lists, the mt19937 position afterwards, `best_room` ITERATION ORDER, paths, widened sets in
iteration order, AABB results bit for bit, entity ids and per-system set orders.  SURVEY.md rows pinned this way:
T1/T2 (through the generators' draws), T3, T5, H1, H2, G2 generator, G5/G6 Kruskal, G3g/G6 room pipeline.

Where /root/reference is absent the prebuilt oracle/_ref is used if present; otherwise these tests skip and the
committed fixtures (tests/golden/ref_fixtures.json, test_oracle_matches_reference_fixtures) still hold the oracle to
recorded reference outputs."""
import json
import os

import numpy as np
import pytest

import ref_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURES = os.path.join(ROOT, "tests", "golden", "ref_fixtures.json")


@pytest.fixture(scope="module")
def sides():
    if not ref_util.have_reference_build():
        pytest.skip("oracle/_ref not built and /root/reference absent")
    return ref_util.Side("ref"), ref_util.Side("pgo")


def test_maze_kruskal_matches_reference(sides):
    """games/maze/maze_generator.cpp:55-139,183-195 — every odd side the game can draw (3..25 hard, ..31 memory), 1 000+
    seeds in all, incl. seeds that go through `unsigned long` wrap-around (negative int32)."""
    ref, ora = sides
    n = 0
    for dim in range(3, 33, 2):
        for seed in list(range(dim * 1000, dim * 1000 + 70)) + [0xFFFFFFFB, 0x80000000 + dim]:
            for objects in (1, 3):
                if objects > 1 and dim < 5:
                    continue
                g0, f0, x0 = ref.maze_generate(seed, dim, dim, objects)
                g1, f1, x1 = ora.maze_generate(seed, dim, dim, objects)
                assert np.array_equal(g0, g1) and np.array_equal(f0, f1) and x0 == x1, (dim, seed, objects)
                n += 1
    assert n >= 1000
    for w, h in ((5, 9), (11, 7), (3, 15)):  # non-square: the generator takes both
        for seed in range(40):
            g0, f0, x0 = ref.maze_generate(seed, w, h, 1)
            g1, f1, x1 = ora.maze_generate(seed, w, h, 1)
            assert np.array_equal(g0, g1) and np.array_equal(f0, f1) and x0 == x1


def test_maze_level_head_matches_reference(sides):
    """maze/tilemap.cpp:65-70 on a fresh engine: side draw + Kruskal + goal placement, all three world sizes."""
    ref, ora = sides
    for world in (25, 15, 31):
        for seed in range(1, 401):
            d0, g0, x0 = ref.maze_level(seed, world)
            d1, g1, x1 = ora.maze_level(seed, world)
            assert d0 == d1 and np.array_equal(g0, g1) and x0 == x1, (world, seed)


def test_set_merge_kruskal_matches_reference(sides):
    """games/{chaser,jumper}/maze_generator.cpp:47-173: chaser's sides 11/13/19, jumper's 13/6/15 with dead-end removal."""
    ref, ora = sides
    n = 0
    for dim, nde in ((11, 0), (13, 0), (19, 0), (13, 1), (6, 1), (15, 1), (5, 1), (7, 0)):
        for seed in range(7000 + dim, 7000 + dim + 180):
            g0, x0 = ref.setmaze_generate(seed, dim, nde)
            g1, x1 = ora.setmaze_generate(seed, dim, nde)
            assert np.array_equal(g0, g1) and x0 == x1, (dim, nde, seed)
            n += 1
    assert n >= 1000


def test_room_pipeline_matches_reference(sides):
    """games/{caveflyer,jumper}/room_generator.cpp:4-202 on 1 000+ random caves of the games' sizes: CA update,
    find_best_room's iteration order (T3: reaches the level), find_path, expand_room's set order."""
    ref, ora = sides
    rng = np.random.default_rng(20261002)
    n = 0
    for gw, gh, count in ((40, 40, 500), (20, 20, 300), (45, 45, 200), (12, 30, 60)):
        for _ in range(count):
            cave = ref_util.random_cave(rng, gw, gh, rng.choice([0.4, 0.5, 0.55]))
            iters = int(rng.integers(0, 4))
            u0 = ref.rooms_update(gw, gh, cave, iters)
            u1 = ora.rooms_update(gw, gh, cave, iters)
            assert np.array_equal(u0, u1)
            a, b, e = int(rng.integers(0, 1 << 30)), int(rng.integers(0, 1 << 30)), int(rng.integers(0, 5))
            r0 = ref.rooms_analyse(gw, gh, u0, a, b, e)
            r1 = ora.rooms_analyse(gw, gh, u0, a, b, e)
            for x, y in zip(r0, r1):
                assert np.array_equal(x, y), (gw, gh, iters, a, b, e)
            n += 1
    assert n >= 1000


def test_aabb_helpers_match_reference_on_a_million_pairs(sides):
    """helpers.cpp:40-46 check_collision, :48-108 get_collision_overlap — 10⁶ pairs, results compared as bit patterns."""
    ref, ora = sides
    rng = np.random.default_rng(7)
    a, b = ref_util.rect_pairs(rng, 1_000_000)
    h0, o0 = ref.collisions(a, b)
    h1, o1 = ora.collisions(a, b)
    assert 0.05 < h0.mean() < 0.95  # the sample exercises both outcomes
    assert np.array_equal(h0, h1)
    assert np.array_equal(o0.view(np.uint32), o1.view(np.uint32))


def test_entity_ids_and_system_set_order_match_reference(sides):
    """ecs.cpp:3-83 against IdPool + IdSet under create / destroy / remove-component / clear sequences; bucket counts
    survive clear() on both sides, so the scripts are run in the same order on both (each side keeps its history)."""
    ref, ora = sides
    rng = np.random.default_rng(11)
    for n_ops in (50, 400, 120, 1500, 30, 700, 200, 3000):
        ops, args = ref_util.ecs_random_script(rng, n_ops)
        i0, o0 = ref.ecs_script(ops, args)
        i1, o1 = ora.ecs_script(ops, args)
        assert np.array_equal(i0, i1)
        assert np.array_equal(o0, o1)
    # SURVEY.md T3 known answers, through the reference itself: fresh ids 0..29 with component A
    # (history above left big buckets; the known-answer orders need a fresh process → covered by the fixture test)


def test_oracle_matches_reference_fixtures():
    """Runs everywhere (no reference tree needed): recorded outputs of oracle/_ref (tests/golden/make_ref_fixtures.py)."""
    fx = json.load(open(FIXTURES))
    ora = ref_util.Side("pgo")
    for rec in fx["maze_levels"]:
        dim, grid, nxt = ora.maze_level(rec["seed"], rec["world"])
        assert dim == rec["dim"] and nxt == rec["next"]
        assert "".join(str(int(v)) for v in grid) == rec["grid"], rec["seed"]
    for rec in fx["setmaze"]:
        grid, nxt = ora.setmaze_generate(rec["seed"], rec["dim"], rec["no_dead_ends"])
        assert nxt == rec["next"] and "".join(str(int(v)) for v in grid) == rec["grid"], rec
    for rec in fx["rooms"]:
        cave = np.array([int(ch) for ch in rec["cave"]], np.int32)
        best, path, wide = ora.rooms_analyse(rec["gw"], rec["gh"], cave, rec["src_sel"], rec["dst_sel"], rec["expand"])
        assert best.tolist() == rec["best_order"] and path.tolist() == rec["path"] and wide.tolist() == rec["wide_order"]


# ------------------------------------------------------------------------------------------------------------------
# Whole envs against the recorded reference generator outputs, through level-seed mode: level number L is by
# definition what a fresh cenv_make(seed = L) builds first, and that is what the fixtures hold for L = the fixture seed.
# ------------------------------------------------------------------------------------------------------------------
def _mix32(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def _level_number(num_levels, start_level, chain_seed, k):
    return (start_level + _mix32(_mix32(chain_seed) + k) % num_levels) & 0xFFFFFFFF


def _maze_expectation(rec):
    """maze/tilemap.cpp:72-87: the maze centred in a world of walls, tile (x, y) at y + x·H; the goal's world position."""
    world, dim = rec["world"], rec["dim"]
    grid = np.array([int(ch) for ch in rec["grid"]], np.int32).reshape(dim + 2, dim + 2)  # [x][y], padded
    margin = (world - dim) // 2
    tiles = np.ones((world, world), np.uint8)  # [x][y]
    inner = grid[1:-1, 1:-1]
    tiles[margin:margin + dim, margin:margin + dim] = (inner == 1).astype(np.uint8)
    gx, gy = [int(v[0]) for v in np.nonzero(inner == 2)]
    goal = (gx + margin + 0.5, world - 1 - (gy + margin) + 0.5)
    return tiles.reshape(-1), goal


def _chaser_expectation(rec):
    """chaser/tilemap.cpp:127-141: tile (x, y) is a wall where the padded maze grid holds 1."""
    dim = rec["dim"]
    grid = np.array([int(ch) for ch in rec["grid"]], np.int32).reshape(dim + 2, dim + 2)
    return (grid[1:-1, 1:-1] == 1).astype(np.uint8).reshape(-1)


LEVEL_CASES = [  # game, mode, fixture key, world / dim, first fixture seed, fixture count
    ("maze", None, "maze_levels", 25, 0, 64), ("maze", "easy", "maze_levels", 15, 100, 16),
    ("maze", "memory", "maze_levels", 31, 200, 16),
    ("chaser", None, "setmaze", 11, 0, 64), ("chaser", "hard", "setmaze", 13, 0, 16), ("chaser", "extreme", "setmaze", 19, 0, 16),
]


def _check_levels_against_fixtures(make_vec, n_envs):
    fx = json.load(open(FIXTURES))
    for game, mode, key, size, first, count in LEVEL_CASES:
        if key == "maze_levels":
            recs = {r["seed"]: r for r in fx[key] if r["world"] == size}
        else:
            recs = {r["seed"]: r for r in fx[key] if r["dim"] == size and not r["no_dead_ends"]}
        vec = make_vec(game, n_envs, 900, count, first, mode)  # made AND reset once: every env is on its level k = 1
        seen = set()
        for i in range(n_envs):
            number = _level_number(count, first, 900 + i, 1)  # k = 0 was make's never-observed level (D1)
            seen.add(number)
            rec = recs[number]
            tiles = vec.tiles(i, cap=size * size)
            if game == "maze":
                want, goal = _maze_expectation(rec)
                assert np.array_equal(tiles, want), (game, mode, i, number)
                st = vec.state(i)
                assert (float(st[3]), float(st[4])) == goal, (game, mode, i, number)
            else:
                assert np.array_equal(tiles, _chaser_expectation(rec)), (game, mode, i, number)
        assert len(seen) >= min(count, n_envs) // 2
        vec.close()


def test_oracle_levels_equal_reference_generator_fixtures():
    from oracle_util import OracleVec

    def make(game, n, seed_base, num_levels, start_level, mode):
        m = 0 if mode is None else {"easy": 1, "hard": 2, "memory": 3, "extreme": 4}[mode]
        # (OracleVec's make already includes the caller's first reset())
        return OracleVec(game, n, seed_base=seed_base, render=False, num_levels=num_levels, start_level=start_level, mode=m)
    _check_levels_against_fixtures(make, 96)


@pytest.mark.gpu
def test_engine_levels_equal_reference_generator_fixtures():
    """The HIP engine's level kernels against outputs of the reference's own maze_generator.cpp (maze: union-find
    Kruskal + goal; chaser: set-merge Kruskal), no oracle in between."""
    from engine_util import EngineVec

    def make(game, n, seed_base, num_levels, start_level, mode):
        eng = EngineVec(game, n, seed_base=seed_base, num_levels=num_levels, start_level=start_level, mode=mode)
        eng.reset()
        return eng
    _check_levels_against_fixtures(make, 192)
