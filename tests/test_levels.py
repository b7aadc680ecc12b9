"""Level-seed mode (include/procgen2_vec.h pgv_make_levels).

The reference has no such option; the semantics are pinned to something the reference does have: level number L is
what a fresh `cenv_make(seed = L)` builds as its level 0.  The CPU tests check the oracle's restatement of that
against directly made oracle envs; the GPU tests check the HIP engine against the oracle, bit for bit.
"""
import ctypes

import numpy as np
import pytest

from oracle_util import OBS_BYTES, OracleVec, oracle, register_textures

GAMES = ["coinrun", "maze", "bossfight", "climber", "caveflyer", "chaser", "jumper"]


def mix32(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def level_number(num_levels, start_level, chain_seed, k):
    return (start_level + mix32(mix32(chain_seed) + k) % num_levels) & 0xFFFFFFFF


def _fresh_make_obs(game, number):
    """First frame and state of level 0 of a fresh make(seed=number), straight from a single oracle env."""
    L = oracle()
    L.pgo_present.argtypes = [ctypes.c_void_p]
    h = L.pgo_make(game.encode(), number, 1)
    L.pgo_present(h)
    obs = np.ctypeslib.as_array(L.pgo_obs(h), shape=(OBS_BYTES,)).copy()
    L.pgo_close(h)
    return obs


@pytest.mark.parametrize("game", GAMES)
def test_oracle_level_numbers_are_fresh_makes(game):
    register_textures(game)
    n, num_levels, start = 6, 5, 1000
    v = OracleVec(game, n, seed_base=40, num_levels=num_levels, start_level=start)
    first = v.reset_obs().copy()
    for i in range(n):
        number = level_number(num_levels, start, 40 + i, 1)  # k = 0 is make's hidden level
        assert start <= number < start + num_levels
        assert np.array_equal(first[i], _fresh_make_obs(game, number)), i
    # later levels: follow every env through a few episode ends
    drawn = [2] * n
    pending = [False] * n
    seen = 0
    for s in range(600):
        obs, _, done = v.step(None, run_seed=3)
        for i in range(n):
            if pending[i]:
                number = level_number(num_levels, start, 40 + i, drawn[i])
                drawn[i] += 1
                assert np.array_equal(obs[i], _fresh_make_obs(game, number)), (i, s)
                seen += 1
            pending[i] = bool(done[i])
        if seen >= 8:
            break
    assert seen > 0 or game not in ("maze", "bossfight", "chaser")  # these three end episodes within 600 steps
    v.close()


@pytest.mark.parametrize("game", ["maze", "bossfight", "chaser"])
def test_oracle_one_level_repeats_forever_and_reseed_restarts_the_sequence(game):
    register_textures(game)
    n = 4
    v = OracleVec(game, n, seed_base=7, num_levels=1, start_level=31337)
    first = v.reset_obs().copy()
    assert all(np.array_equal(first[0], first[i]) for i in range(n))  # one level: every env plays it
    assert np.array_equal(first[0], _fresh_make_obs(game, 31337))
    v.close()
    # many levels: a reseeding reset restarts the env's sequence at k = 0
    v = OracleVec(game, n, seed_base=7, num_levels=1000, start_level=0)
    seeds = np.array([5, 5, -9, 123456], np.int32)
    obs = v.reset(seeds=seeds).copy()
    for i in range(n):
        assert np.array_equal(obs[i], _fresh_make_obs(game, level_number(1000, 0, int(seeds[i]) & 0xFFFFFFFF, 0))), i
    assert np.array_equal(obs[0], obs[1])
    again = v.reset().copy()  # no seeds: k = 1
    for i in range(n):
        assert np.array_equal(again[i], _fresh_make_obs(game, level_number(1000, 0, int(seeds[i]) & 0xFFFFFFFF, 1))), i
    v.close()


def test_level_mode_off_is_the_reference_behaviour():
    register_textures("maze")
    a = OracleVec("maze", 3, seed_base=5)
    b = OracleVec("maze", 3, seed_base=5, num_levels=0, start_level=99)
    assert np.array_equal(a.reset_obs(), b.reset_obs())
    for s in range(30):
        oa, ra, da = a.step(None, run_seed=1)
        ob, rb, db = b.step(None, run_seed=1)
        assert np.array_equal(oa, ob) and np.array_equal(ra, rb) and np.array_equal(da, db)
    a.close()
    b.close()


# ---------------------------------------------------------------------------------------------------------------------
# HIP engine vs oracle
# ---------------------------------------------------------------------------------------------------------------------
def _jumpy(L, run_seed, step, n):
    """Synthetic actions biased so that platformer agents move and die: episode ends are what is being tested."""
    a = np.array([L.pgo_synthetic_action(run_seed, step, e) for e in range(n)], np.int32)
    return a


@pytest.mark.gpu
@pytest.mark.parametrize("game,steps", [("coinrun", 300), ("maze", 520), ("bossfight", 250), ("climber", 250),
                                        ("caveflyer", 250), ("chaser", 200), ("jumper", 250)])
@pytest.mark.parametrize("prefetch", [True, False])
def test_engine_level_mode_matches_oracle(game, steps, prefetch):
    from engine_util import EngineVec
    if not prefetch and game in ("bossfight", "chaser"):
        pytest.skip("no prefetch in this game")
    n, num_levels, start = 96, 7, 50
    eng = EngineVec(game, n, seed_base=3, num_levels=num_levels, start_level=start)
    if not prefetch:
        eng.set_debug(256)
    ora = OracleVec(game, n, seed_base=3, num_levels=num_levels, start_level=start)
    L = ora.L
    assert np.array_equal(eng.reset(), ora.reset_obs()), "reset frame"  # the oracle vec has done its first reset
    ends = 0
    for s in range(steps):
        a = _jumpy(L, 11, s, n)
        oe, re_, de = eng.step(a)
        oo, ro, do = ora.step(a)
        assert np.array_equal(de, do), "done, step %d" % s
        assert np.array_equal(re_.view(np.uint32), ro.view(np.uint32)), "reward bits, step %d" % s
        if not np.array_equal(oe, oo):
            bad = np.nonzero((oe != oo).any(axis=1))[0]
            raise AssertionError("obs differ at step %d in %d envs (first env %d)" % (s, bad.size, bad[0]))
        ends += int(do.sum())
        if s == steps // 2:  # masked reseeding reset in the middle: restarts those envs' sequences
            mask = (np.arange(n) % 3 == 0).astype(np.uint8)
            seeds = (np.arange(n, dtype=np.int32) % 4) - 1
            assert np.array_equal(eng.reset(mask=mask, seeds=seeds), ora.reset(mask=mask, seeds=seeds)), "masked reset"
    for e in range(0, n, 12):
        assert np.array_equal(eng.state(e).view(np.uint32), ora.state(e).view(np.uint32)), "state env %d" % e
        assert np.array_equal(eng.tiles(e), ora.tiles(e)), "tiles env %d" % e
    assert ends > 0 or game in ("climber", "jumper", "coinrun", "caveflyer"), ends
    eng.close()
    ora.close()


@pytest.mark.gpu
def test_engine_one_level_means_one_first_frame_at_scale():
    """4096 maze envs on one level: every env's every episode starts with the same frame (mazes time out at 500)."""
    from engine_util import EngineVec
    n = 4096
    register_textures("maze")  # (_fresh_make_obs goes to the oracle directly: not only when an earlier test has done this)
    eng = EngineVec("maze", n, seed_base=1, num_levels=1, start_level=77)
    first = eng.reset().copy()
    assert (first == first[0]).all()
    assert np.array_equal(first[0], _fresh_make_obs("maze", 77))
    pending = np.zeros(n, bool)
    checked = 0
    for s in range(505):
        obs, _, done = eng.step(None, run_seed=2)
        if pending.any():
            assert (obs[pending] == first[0]).all(), s
            checked += int(pending.sum())
        pending = done.astype(bool)
    assert checked >= n
    eng.close()


@pytest.mark.gpu
def test_engine_snapshot_carries_the_level_sequence():
    from engine_util import EngineVec
    n = 256
    eng = EngineVec("bossfight", n, seed_base=9, num_levels=3, start_level=0)
    eng.reset()
    for s in range(60):
        eng.step(None, run_seed=4)
    snap = eng.save_state()
    tail = [tuple(x.copy() for x in eng.step(None, run_seed=4)) for _ in range(200)]
    eng.load_state(snap)
    for k, (o, r, d) in enumerate(tail):
        o2, r2, d2 = eng.step(None, run_seed=4)
        assert np.array_equal(d, d2) and np.array_equal(r, r2) and np.array_equal(o, o2), k
    other = EngineVec("bossfight", n, seed_base=9, num_levels=4, start_level=0)
    with pytest.raises(Exception):
        other.load_state(snap)
    other.close()
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("game,every", [("coinrun", 4), ("maze", 2)])
def test_generator_cadence_does_not_depend_on_how_the_caller_synchronises(game, every):
    """the level generator is launched every Game::pregen_every()-th step — a count of steps, not
    the host's view of the side stream.  A caller that drains the stream after every step (any policy loop) and one that
    enqueues all its steps ahead (bench.py) get the same number of generator launches, and the same frames."""
    from engine_util import EngineVec
    from procgen2_amd import lib as pglib
    steps, n = 64, 256
    counts, frames = [], []
    for drain in (True, False):
        e = EngineVec(game, n, seed_base=5)
        e.reset()
        before = e.L.pgv_generator_launches(e.h)
        for s in range(steps):
            e.step_quiet(run_seed=9)
            if drain:
                pglib.check(e.L, e.L.pgv_sync(e.h), "pgv_sync")
        counts.append(e.L.pgv_generator_launches(e.h) - before)
        frames.append(e._fetch()[0].copy())
        e.close()
    assert counts[0] == counts[1] == steps // every, counts
    assert np.array_equal(frames[0], frames[1])
