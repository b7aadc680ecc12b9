"""Distribution modes: compile-time `Config` constants in the reference, int options here.

CPU: the oracle's modes against what the reference's sources say they change; the engine's (game, mode) table.
GPU: the HIP engine in every non-default mode against the oracle, bit for bit.
"""
import ctypes

import numpy as np
import pytest

from oracle_util import OracleVec, oracle, register_textures

from procgen2_amd import lib as pglib

EASY, HARD, MEMORY, EXTREME = 1, 2, 3, 4
# game → (default, every mode it offers)
TABLE = {
    "coinrun": (HARD, {EASY, HARD}),
    "maze": (HARD, {EASY, HARD, MEMORY}),
    "bossfight": (HARD, {EASY, HARD}),
    "climber": (HARD, {EASY, HARD}),
    "caveflyer": (HARD, {EASY, HARD, MEMORY}),
    "chaser": (EASY, {EASY, HARD, EXTREME}),
    "jumper": (HARD, {EASY, HARD, MEMORY}),
}
NON_DEFAULT = sorted((g, m) for g, (d, ms) in TABLE.items() for m in ms if m != d)


def test_engine_mode_table_matches_the_oracle():
    L = pglib.load()
    O = oracle()
    for gid in range(7):
        game = L.pgv_game_name(gid).decode()
        default, modes = TABLE[game]
        assert L.pgv_game_modes(gid) == sum(1 << m for m in modes), game
        assert O.pgo_resolve_mode(game.encode(), 0) == default, game
        for m in (EASY, HARD, MEMORY, EXTREME):
            built = m in modes
            # the oracle may know modes the engine does not offer yet, never the other way round
            assert (O.pgo_resolve_mode(game.encode(), m) == m) or not built, (game, m)
    assert L.pgv_game_modes(99) == 0


def _rollout(game, mode, n=8, steps=120, seed_base=21):
    v = OracleVec(game, n, seed_base=seed_base, render=False, mode=mode)
    out = []
    for s in range(steps):
        _, r, d = v.step(None, run_seed=5)
        out.append((r.copy(), d.copy()))
    state = [v.state(e) for e in range(n)]
    tiles = [v.tiles(e) for e in range(n)]
    v.close()
    return out, state, tiles


def test_oracle_default_mode_is_the_named_default():
    for game, (default, _) in TABLE.items():
        a, sa, ta = _rollout(game, 0, n=3, steps=40)
        b, sb, tb = _rollout(game, default, n=3, steps=40)
        assert all(np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) for x, y in zip(a, b)), game
        assert all(np.array_equal(x, y) for x, y in zip(ta, tb)), game


def test_oracle_coinrun_easy_is_hard():
    # `allow_monsters = !cfg.easy_mode` is never read (coinrun/tilemap.cpp:148): the two modes are the same game
    a, sa, ta = _rollout("coinrun", EASY)
    b, sb, tb = _rollout("coinrun", HARD)
    assert all(np.array_equal(x, y) for x, y in zip(ta, tb))
    assert all(np.array_equal(x, y) for x, y in zip(sa, sb))


def test_oracle_climber_easy_spawns_fewer_enemies():
    # enemy_prob .2 instead of .5 (climber/tilemap.cpp:118): same platforms up to the first enemy draw, fewer enemies
    def enemies(mode):
        v = OracleVec("climber", 48, seed_base=100, render=False, mode=mode)
        total = sum(int(v.state(e).size) for e in range(48))  # the state dump lists every entity
        v.close()
        return total
    assert enemies(EASY) < enemies(HARD)


def test_oracle_bossfight_easy_bullets_are_slower():
    a, _, _ = _rollout("bossfight", EASY, n=16, steps=200)
    b, _, _ = _rollout("bossfight", HARD, n=16, steps=200)
    assert any(not np.array_equal(x[1], y[1]) for x, y in zip(a, b))  # different episodes …
    assert sum(int(x[1].sum()) for x in a) <= sum(int(y[1].sum()) for y in b)  # … and the agent survives longer


@pytest.mark.gpu
@pytest.mark.parametrize("game,mode", NON_DEFAULT)
def test_engine_mode_matches_oracle(game, mode):
    from engine_util import EngineVec
    register_textures(game)
    n, steps = 128, 260
    eng = EngineVec(game, n, seed_base=17, mode=mode)
    ora = OracleVec(game, n, seed_base=17, mode=mode)
    assert eng.L.pgv_mode(eng.h) == mode
    L = ora.L
    assert np.array_equal(eng.reset(), ora.reset_obs()), "reset frame"
    for s in range(steps):
        a = np.array([L.pgo_synthetic_action(9, s, e) for e in range(n)], np.int32)
        oe, re_, de = eng.step(a)
        oo, ro, do = ora.step(a)
        assert np.array_equal(de, do), "done, step %d" % s
        assert np.array_equal(re_.view(np.uint32), ro.view(np.uint32)), "reward bits, step %d" % s
        if not np.array_equal(oe, oo):
            bad = np.nonzero((oe != oo).any(axis=1))[0]
            raise AssertionError("obs differ at step %d in %d envs (first env %d)" % (s, bad.size, bad[0]))
    for e in range(0, n, 16):
        assert np.array_equal(eng.state(e).view(np.uint32), ora.state(e).view(np.uint32)), "state env %d" % e
        assert np.array_equal(eng.tiles(e), ora.tiles(e)), "tiles env %d" % e
    eng.close()
    ora.close()


@pytest.mark.gpu
def test_engine_rejects_modes_a_game_does_not_have():
    from engine_util import EngineVec
    with pytest.raises(pglib.EngineError, match="no distribution mode"):
        EngineVec("coinrun", 4, mode=EXTREME)


# ---------------------------------------------------------------------------------------------------------------------
# coinrun's allow_pit / allow_crate / allow_dy / allow_mobs (pgv_config.game_flags, PGV_COINRUN_NO_*)
# ---------------------------------------------------------------------------------------------------------------------
NO_PIT, NO_CRATE, NO_DY, NO_MOBS = 1, 2, 4, 8


def _coinrun_tiles(flags, n=24):
    v = OracleVec("coinrun", n, seed_base=300, render=False, game_flags=flags)
    tiles = [v.tiles(e).reshape(64, 64) for e in range(n)]  # [x][y]
    sizes = [v.state(e).size for e in range(n)]
    v.close()
    return tiles, sizes


def test_oracle_coinrun_switches_remove_what_they_name():
    base, base_sizes = _coinrun_tiles(0)
    ids = sorted({int(t) for lvl in base for t in np.unique(lvl & 15)})
    # no crates: the crate tile id disappears from every level (find it as the id that only the default set has)
    nocrate, _ = _coinrun_tiles(NO_CRATE)
    gone = {int(t) for lvl in base for t in np.unique(lvl & 15)} - {int(t) for lvl in nocrate for t in np.unique(lvl & 15)}
    assert len(gone) == 1, (ids, gone)
    # no dy (and no pits / crates to stack on it): the floor stays one tile high from the start column to the coin wall
    flat, _ = _coinrun_tiles(NO_DY | NO_PIT | NO_CRATE)
    for lvl in flat:
        heights = [(int(((lvl[x, :62] & 15) != 0).sum())) for x in range(1, 63)]
        run = 0
        while run < len(heights) and heights[run] == 1:
            run += 1
        assert run >= 8 and all(h > 30 for h in heights[run:]), heights  # then the solid block right of the coin
    assert any(len({int(((lvl[x, :62] & 15) != 0).sum()) for x in range(1, 20)}) > 1 for lvl in base)  # default: steps
    # no mobs, no pits: fewer entities than the default set (the state dump lists every entity)
    calm, calm_sizes = _coinrun_tiles(NO_MOBS | NO_PIT)
    assert sum(calm_sizes) < sum(base_sizes)
    # and 0 is the reference
    again, again_sizes = _coinrun_tiles(0)
    assert all(np.array_equal(a, b) for a, b in zip(base, again)) and base_sizes == again_sizes


def test_oracle_rejects_flags_of_other_games():
    O = oracle()
    O.pgo_make_config.restype = ctypes.c_void_p
    O.pgo_make_config.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_uint32]
    assert not O.pgo_make_config(b"maze", 1, 0, 0, 1)
    assert not O.pgo_make_config(b"coinrun", 1, 0, 0, 16)
    h = O.pgo_make_config(b"coinrun", 1, 0, 0, 15)
    assert h
    O.pgo_close(h)


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [NO_PIT, NO_CRATE, NO_DY, NO_MOBS, NO_PIT | NO_CRATE | NO_DY | NO_MOBS])
def test_engine_coinrun_switches_match_oracle(flags):
    from engine_util import EngineVec
    register_textures("coinrun")
    n, steps = 128, 160
    eng = EngineVec("coinrun", n, seed_base=41, game_flags=flags)
    ora = OracleVec("coinrun", n, seed_base=41, game_flags=flags)
    L = ora.L
    assert np.array_equal(eng.reset(), ora.reset_obs()), "reset frame"
    for s in range(steps):
        a = np.array([L.pgo_synthetic_action(13, s, e) for e in range(n)], np.int32)
        oe, re_, de = eng.step(a)
        oo, ro, do = ora.step(a)
        assert np.array_equal(de, do) and np.array_equal(re_.view(np.uint32), ro.view(np.uint32)), s
        assert np.array_equal(oe, oo), s
    for e in range(0, n, 16):
        assert np.array_equal(eng.tiles(e), ora.tiles(e)), e
    eng.close()
    ora.close()


@pytest.mark.gpu
def test_engine_rejects_flags_of_other_games():
    from engine_util import EngineVec
    with pytest.raises(pglib.EngineError, match="game_flags"):
        EngineVec("maze", 4, game_flags=1)
    with pytest.raises(pglib.EngineError, match="game_flags"):
        EngineVec("coinrun", 4, game_flags=16)
