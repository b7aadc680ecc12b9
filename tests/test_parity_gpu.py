"""Parity tests proper (need an MI355X): the HIP engine, called through its C ABI, against the CPU oracle on
the same seeds and actions.  Bar: bit-exact observations (u8), rewards (f32 bit patterns) and dones.

Small/medium sizes compare every byte of every step; BASELINE.json's full sizes (65 536 envs) use
size-independent properties: shard invariance (an env's trajectory depends only on its global index),
determinism, and byte-compare of a strided sample of envs against the oracle.
"""
import ctypes
import os

import numpy as np
import pytest

from engine_util import EngineVec
from oracle_util import OracleVec, oracle

pytestmark = pytest.mark.gpu

from procgen2_amd import cenv as pgcenv  # noqa: E402
from procgen2_amd import lib as pglib  # noqa: E402


def _actions(L, run_seed, step, n, offset=0):
    return np.array([L.pgo_synthetic_action(run_seed, step, offset + e) for e in range(n)], np.int32)


def _lockstep(game, n, steps, seed_base=1, run_seed=0, check_state_every=0, game_flags=0, threads=1, debug=0):
    eng = EngineVec(game, n, seed_base=seed_base, game_flags=game_flags)
    if debug:
        eng.set_debug(debug)
    ora = OracleVec(game, n, seed_base=seed_base, game_flags=game_flags)
    L = ora.L
    assert np.array_equal(eng.reset(), ora.reset_obs()), "reset frame"
    resets = 0
    for s in range(steps):
        a = _actions(L, run_seed, s, n)
        oe, re_, de = eng.step(a)
        oo, ro, do = ora.step(a, threads=threads)
        assert np.array_equal(de, do), "done, step %d" % s
        assert np.array_equal(re_.view(np.uint32), ro.view(np.uint32)), "reward bits, step %d" % s
        if not np.array_equal(oe, oo):
            bad = np.nonzero((oe != oo).any(axis=1))[0]
            raise AssertionError("obs differ at step %d in %d envs (first env %d, %d bytes)" %
                                 (s, bad.size, bad[0], int((oe[bad[0]] != oo[bad[0]]).sum())))
        resets += int(do.sum())
        if check_state_every and s % check_state_every == 0:
            for e in range(0, n, max(1, n // 8)):
                assert np.array_equal(eng.state(e).view(np.uint32), ora.state(e).view(np.uint32)), "state env %d" % e
                assert np.array_equal(eng.tiles(e), ora.tiles(e)), "tiles env %d" % e
    eng.close()
    ora.close()
    return resets


def test_coinrun_lockstep_256_envs():
    # configs[1] scaled to what the scalar oracle renders in seconds; 400 steps cross several episode ends
    resets = _lockstep("coinrun", 256, 400, check_state_every=50)
    assert resets > 0, "no episode ended: the auto-reset path was not exercised"


def test_coinrun_hazards_the_long_way():
    """coinrun's entity lanes hand resolve_kernel only the hazards that come near where the agent STARTED the step, and the
    agent's lane checks that it stayed within that reach (coinrun.hip hazard_near); where it did not — never, in a normal
    run — resolve_kernel works every hazard's boxes out again.  pgv_set_debug bit 24 sets the reach to nothing, so that
    every agent that moves fails the check: the same rewards, dones and frames as the oracle through episode ends (deaths
    by saw and by mob are most of them)."""
    resets = _lockstep("coinrun", 192, 300, seed_base=5, run_seed=3, check_state_every=60, debug=1 << 24)
    assert resets > 0


@pytest.mark.parametrize("game_flags", [0, 1])
def test_chaser_enemies_the_long_way(game_flags):
    """chaser's enemies take their turns side by side, one per lane, on outputs peeked from the env's stream where each is
    thought to start (chaser.hip advance); where the peek does not reach — the last few of the stream's 624 words, a
    rejected range draw — they go one after the other on the stream itself.  pgv_set_debug bit 25 sends every sub-step
    that way: the same rewards, dones, frames and enemy states as the oracle through episode ends, for both `abs`
    readings (D21).  (The default path is what every other chaser test runs.)"""
    resets = _lockstep("chaser", 256, 400, seed_base=11, run_seed=5, check_state_every=50, debug=1 << 25, game_flags=game_flags)
    assert resets > 0


def test_bossfight_long_lockstep_over_many_changes_of_buffers():
    """bossfight draws about six numbers a step, so an env's stream runs out every hundred steps or so and its gang
    changes buffers (pg_gang.h GangRng::refill) — in both directions, with the next block made ahead by setup_kernel each
    time, through in-step resets that draw from whichever buffer is current: 700 steps of 256 envs against the oracle."""
    resets = _lockstep("bossfight", 256, 700, seed_base=31, run_seed=6, check_state_every=100)
    assert resets > 0


@pytest.mark.parametrize("game", ["bossfight", "chaser"])
def test_plain_masked_reset_after_the_streams_changed_buffers(game):
    """bossfight's and chaser's random streams live in two buffers (pg_gang.h GangRng: a gang that runs out of numbers
    changes buffers, the next block having been made ahead of time).  A reset WITHOUT a seed continues the env's stream:
    whoever generates the level must read it from where the gang left it (chaser generate, bossfight begin_level).  Sixty
    steps in, most streams have changed buffers at least once; a third of the envs — some of them due for an auto-reset —
    are then reset by hand, without seeds, and the batch goes on: frames, rewards and dones as the oracle's throughout."""
    n = 192
    eng, ora = EngineVec(game, n, seed_base=21), OracleVec(game, n, seed_base=21)
    L = ora.L
    assert np.array_equal(eng.reset(), ora.reset_obs())
    for s in range(160):
        a = _actions(L, 9, s, n)
        oe, re_, de = eng.step(a)
        oo, ro, do = ora.step(a)
        assert np.array_equal(de, do) and np.array_equal(re_.view(np.uint32), ro.view(np.uint32)), "step %d" % s
        assert np.array_equal(oe, oo), "obs, step %d" % s
        if s in (60, 61, 110):
            mask = ((np.arange(n) + s) % 3 == 0).astype(np.uint8)
            assert np.array_equal(eng.reset(mask=mask), ora.reset(mask=mask)), "masked reset at step %d" % s
    eng.close()
    ora.close()


@pytest.mark.parametrize("game", ["coinrun", "maze", "bossfight", "climber", "caveflyer", "chaser", "jumper"])
def test_row_composer_equals_draw_list_replay(game):
    """The fused background+tile row composer (pg_render.h compose_rows) against the one-blit-at-a-time replay of
    the reference's draw list, the same engine in both modes, every byte of 512 envs over 150 steps."""
    n = 512
    fast, slow = EngineVec(game, n, seed_base=77), EngineVec(game, n, seed_base=77)
    slow.set_debug(1)
    assert np.array_equal(fast.reset(), slow.reset())
    for s in range(150):
        of, rf, df = fast.step(None, run_seed=4)
        os_, rs, ds = slow.step(None, run_seed=4)
        assert np.array_equal(of, os_), "step %d" % s
        assert np.array_equal(rf, rs) and np.array_equal(df, ds)
    fast.close()
    slow.close()


@pytest.mark.parametrize("game", ["coinrun", "climber", "caveflyer", "bossfight", "jumper", "chaser", "maze"])
def test_render_pre_pass_equals_the_complete_path(game):
    """The lean frames — composed from what the render pre-pass (pg_prepass.h setup_kernel) left in device memory — against
    the same engine with the pre-pass switched off (pgv_set_debug bit 21: every frame's workgroup does its own set-up, the
    path the pre-pass's `fat` frames take anyway; bossfight has no such frames — its complete kernel is a second kernel, chosen
    by the host): every byte of 1 024 envs over 300 steps, through episode ends (coinrun:
    steps that end early and owe their entities a redo), explicit masked resets and the auto-resets in between."""
    n = 1024
    lean, full = EngineVec(game, n, seed_base=901), EngineVec(game, n, seed_base=901)
    full.set_debug(1 << 21)
    assert np.array_equal(lean.reset(), full.reset())
    ends = 0
    for s in range(300):
        ol, rl, dl = lean.step(None, run_seed=6)
        of, rf, df = full.step(None, run_seed=6)
        assert np.array_equal(ol, of), "step %d" % s
        assert np.array_equal(rl.view(np.uint32), rf.view(np.uint32)) and np.array_equal(dl, df)
        ends += int(dl.sum())
        if s % 97 == 50:  # an explicit reset of some envs (the pre-pass runs behind it with the same mask)
            mask = (np.arange(n) % 5 == s % 5).astype(np.uint8)
            assert np.array_equal(lean.reset(mask=mask), full.reset(mask=mask)), "masked reset at step %d" % s
    assert ends > 0 or game in ("caveflyer", "maze")
    lean.close()
    full.close()


@pytest.mark.parametrize("game", ["coinrun", "climber", "caveflyer", "jumper"])
def test_frames_the_pre_pass_hands_back_equal_the_prepared_ones(game):
    """The pre-pass leaves a frame its tables do not hold (more visible draws than a wave has lanes, a window beyond the
    cell table, …) to the complete path — inside the render kernel (coinrun, climber, caveflyer) or by a kernel of its own
    walking a list (jumper).  Climber and jumper never have such a frame in a normal run, so pgv_set_debug bit 23 makes the
    pre-pass hand back every third env's: those frames, their neighbours' and everything after them must be what the
    engine without the switch produces, under masked resets too (the list is rebuilt by every pre-pass launch)."""
    n = 768
    plain, thirds = EngineVec(game, n, seed_base=77), EngineVec(game, n, seed_base=77)
    thirds.set_debug(1 << 23)
    assert np.array_equal(plain.reset(), thirds.reset())
    for s in range(120):
        op, rp, dp = plain.step(None, run_seed=8)
        ot, rt, dt = thirds.step(None, run_seed=8)
        assert np.array_equal(op, ot), "step %d" % s
        assert np.array_equal(rp.view(np.uint32), rt.view(np.uint32)) and np.array_equal(dp, dt)
        if s % 40 == 20:
            mask = (np.arange(n) % 4 == 1).astype(np.uint8)  # (env 0, which the list's counter is reset by, sits this one out)
            assert np.array_equal(plain.reset(mask=mask), thirds.reset(mask=mask)), "masked reset at step %d" % s
    plain.close()
    thirds.close()


def test_coinrun_other_seeds_and_action_stream():
    _lockstep("coinrun", 64, 300, seed_base=4294967000, run_seed=9)  # seeds wrap through 2^32 like `unsigned long`→u32


def test_maze_lockstep_with_timeouts():
    resets = _lockstep("maze", 96, 620, check_state_every=100)  # > 500: every env hits the step cap (D5)
    assert resets >= 96


def test_bossfight_lockstep_many_episodes():
    # BASELINE.json configs[3] game: ~87-step episodes under random actions, so 500 steps cross several resets per env;
    # exercises the in-step RNG draws, the glibc-exact sin/cos twin and the rotated bullet blits (raster rule S6).
    resets = _lockstep("bossfight", 192, 500, check_state_every=60)
    assert resets >= 192


def test_bossfight_fire_heavy_actions():
    # action 9 = fire: agent bullets, shield bounces (RNG), boss hit points and the phase machine
    n = 64
    eng, ora = EngineVec("bossfight", n, seed_base=21), OracleVec("bossfight", n, seed_base=21)
    assert np.array_equal(eng.reset(), ora.reset_obs())
    rng = np.random.default_rng(5)
    for s in range(700):
        a = np.where(rng.random(n) < 0.6, 9, rng.integers(0, 15, n)).astype(np.int32)
        oe, re_, de = eng.step(a)
        oo, ro, do = ora.step(a)
        assert np.array_equal(de, do) and np.array_equal(re_.view(np.uint32), ro.view(np.uint32)), s
        assert np.array_equal(oe, oo), s
    for e in range(0, n, 8):
        assert np.array_equal(eng.state(e, 400).view(np.uint32), ora.state(e, 400).view(np.uint32)), e
    eng.close()
    ora.close()


def test_caveflyer_lockstep_with_fire():
    # BASELINE.json configs[2] game.  40% fire / thrust-heavy actions: bullets, target destruction (+3, entity sets
    # shrink and the draw list is rebuilt), exhaust particles (rotated + alpha blits), deaths -> the level generator
    # (hashtable-ordered largest room, BFS path, widening) runs for the auto-resets.
    n = 160
    eng, ora = EngineVec("caveflyer", n, seed_base=41), OracleVec("caveflyer", n, seed_base=41)
    assert np.array_equal(eng.reset(), ora.reset_obs())
    rng = np.random.default_rng(8)
    ends, rew = 0, 0.0
    for s in range(700):
        u = rng.random(n)
        a = np.where(u < 0.25, 9, np.where(u < 0.55, rng.choice([2, 5, 8], n), rng.integers(0, 15, n))).astype(np.int32)
        oe, re_, de = eng.step(a)
        oo, ro, do = ora.step(a)
        assert np.array_equal(de, do) and np.array_equal(re_.view(np.uint32), ro.view(np.uint32)), s
        assert np.array_equal(oe, oo), s
        ends += int(do.sum())
        rew += float(ro.sum())
        if s % 100 == 0:
            for e in range(0, n, 20):
                assert np.array_equal(eng.state(e).view(np.uint32), ora.state(e).view(np.uint32)), (s, e)
                assert np.array_equal(eng.tiles(e), ora.tiles(e)), (s, e)
    assert ends > 10 and rew > 0.0, (ends, rew)
    eng.close()
    ora.close()


@pytest.mark.parametrize("game,steps", [("caveflyer", 200), ("maze", 560), ("coinrun", 260), ("climber", 200),
                                        ("jumper", 200)])
def test_level_prefetch_equals_synchronous_generation(game, steps):
    """pg_prefetch.h: levels generated ahead of time on the side stream and installed at reset against the same engine
    with prefetch off (every reset generates inside the step) — every byte of 2048 envs, over enough steps that both
    the ready-slot path and the not-ready-yet fallbacks are taken (maze: all envs time out together at step 500)."""
    n = 2048
    ahead, sync = EngineVec(game, n, seed_base=900), EngineVec(game, n, seed_base=900)
    sync.set_debug(256)
    assert np.array_equal(ahead.reset(), sync.reset())
    ends = 0
    for s in range(steps):
        oa, ra, da = ahead.step(None, run_seed=6)
        os_, rs, ds = sync.step(None, run_seed=6)
        assert np.array_equal(da, ds) and np.array_equal(ra, rs), s
        assert np.array_equal(oa, os_), s
        ends += int(da.sum())
    assert ends > 100 or game in ("climber", "jumper"), ends
    ahead.close()
    sync.close()


def test_chaser_lockstep_steering_actions():
    # Only actions 1, 3, 5, 7 steer in chaser; with mostly those the agent roams, eats points (entity sets shrink, draw
    # list rebuilt through the introsort twin on up to 70 equal keys), takes orbs (enemies flee, get eaten, respawn as
    # eggs: in-step RNG) and is caught (auto-reset: Kruskal maze + unordered_set-ordered spawn cells).
    n = 192
    eng, ora = EngineVec("chaser", n, seed_base=51), OracleVec("chaser", n, seed_base=51)
    assert np.array_equal(eng.reset(), ora.reset_obs())
    rng = np.random.default_rng(11)
    ends, rew = 0, 0.0
    hold = rng.choice([1, 3, 5, 7], n)
    for s in range(700):
        turn = rng.random(n) < 0.15
        hold = np.where(turn, rng.choice([1, 3, 5, 7], n), hold)
        a = np.where(rng.random(n) < 0.1, rng.integers(0, 15, n), hold).astype(np.int32)
        oe, re_, de = eng.step(a)
        oo, ro, do = ora.step(a)
        assert np.array_equal(de, do) and np.array_equal(re_.view(np.uint32), ro.view(np.uint32)), s
        assert np.array_equal(oe, oo), s
        ends += int(do.sum())
        rew += float(ro.sum())
        if s % 100 == 0:
            for e in range(0, n, 24):
                assert np.array_equal(eng.state(e, 1024).view(np.uint32), ora.state(e, 1024).view(np.uint32)), (s, e)
                assert np.array_equal(eng.tiles(e), ora.tiles(e)), (s, e)
    assert ends > 50 and rew > 50.0, (ends, rew)
    eng.close()
    ora.close()


def test_jumper_lockstep_jump_heavy_actions():
    # Jump-biased actions: double jumps with cooldown, spikes (deaths -> maze/cave generator on the auto-reset),
    # carrots (+10), dust particles (alpha blits) and the compass HUD (screen-space blits, glibc-exact atan2f).
    n = 160
    eng, ora = EngineVec("jumper", n, seed_base=61), OracleVec("jumper", n, seed_base=61)
    assert np.array_equal(eng.reset(), ora.reset_obs())
    idx = np.arange(n)
    ends, rew = 0, 0.0
    for s in range(800):
        a = _actions(ora.L, 5, s, n)
        a = np.where((idx + s) % 5 < 3, (a % 3) * 3 + 2, a).astype(np.int32)
        oe, re_, de = eng.step(a)
        oo, ro, do = ora.step(a)
        assert np.array_equal(de, do) and np.array_equal(re_.view(np.uint32), ro.view(np.uint32)), s
        assert np.array_equal(oe, oo), s
        ends += int(do.sum())
        rew += float(ro.sum())
        if s % 160 == 0:
            for e in range(0, n, 20):
                assert np.array_equal(eng.state(e).view(np.uint32), ora.state(e).view(np.uint32)), (s, e)
                assert np.array_equal(eng.tiles(e), ora.tiles(e)), (s, e)
    assert ends > 40, (ends, rew)
    eng.close()
    ora.close()


@pytest.mark.parametrize("game", ["coinrun", "caveflyer", "bossfight", "chaser"])
def test_snapshot_restore_replays_the_same_rollout(game):
    """pgv_save_state / pgv_load_state: after a restore the batch continues exactly as it did the first time —
    observations, rewards, dones — across auto-resets, prefetched levels (coinrun, caveflyer) and in-step RNG draws
    (bossfight, chaser); a snapshot is refused by an env of another size."""
    n = 192
    eng = EngineVec(game, n, seed_base=5)
    eng.reset()
    for s in range(40):
        eng.step(None, run_seed=8)
    snap = eng.save_state()
    first = []
    for s in range(120):
        o, r, d = eng.step(None, run_seed=8)
        first.append((o.copy(), r.copy(), d.copy()))
    assert sum(int(d.sum()) for _, _, d in first) > 0
    eng.load_state(snap)
    for s in range(120):
        o, r, d = eng.step(None, run_seed=8)
        assert np.array_equal(o, first[s][0]) and np.array_equal(r, first[s][1]) and np.array_equal(d, first[s][2]), s
    other = EngineVec(game, n + 1, seed_base=5)
    with pytest.raises(pglib.EngineError):
        other.load_state(snap)
    other.close()
    eng.close()


FRAME_GAMES = [("coinrun", 0), ("maze", 0), ("bossfight", 0), ("climber", 0), ("caveflyer", 0), ("chaser", 0),
               ("jumper", 0),
               # non-default distribution modes with their own camera / world size (include/procgen2_vec.h PGV_MODE_*)
               ("maze", 3), ("chaser", 4), ("jumper", 3), ("caveflyer", 1)]


@pytest.mark.parametrize("game,mode", FRAME_GAMES)
def test_human_frame_matches_the_oracle(game, mode):
    """cenv_render's W×H frame (render_game(false)): the GPU frame kernel (pg_frame.h) against the
    oracle's paint() at the same size — the default 512×512 window, a small and a non-square one — after a reset and
    along a rollout, for several envs."""
    from oracle_util import register_textures
    n = 6
    eng = EngineVec(game, n, seed_base=71, mode=mode)
    L = oracle()
    register_textures(game)
    hs = [L.pgo_make_mode(game.encode(), 71 + i, 1, mode) for i in range(n)]
    for h in hs:
        L.pgo_reset(h, 0, 0)
    eng.reset()

    def check(tag):
        for env, (w, h) in ((0, (512, 512)), (1, (160, 160)), (2, (200, 120)), (5, (64, 64))):
            want = np.zeros((h, w, 3), np.uint8)
            L.pgo_render_frame(hs[env], w, h, want.ctypes.data_as(ctypes.c_void_p))
            got = eng.frame(env, w, h)
            if not np.array_equal(got, want):
                bad = np.argwhere((got != want).any(axis=2))
                raise AssertionError("%s: env %d %dx%d: %d pixels differ, first at (y=%d, x=%d)" %
                                     (tag, env, w, h, len(bad), bad[0][0], bad[0][1]))

    check("reset")
    pending = [False] * n
    for s in range(90):
        a = _actions(L, 4, s, n)
        eng.step(a)
        for i, h in enumerate(hs):
            if pending[i]:
                L.pgo_reset(h, 0, 0)
                pending[i] = False
            else:
                L.pgo_step(h, int(a[i]))
                pending[i] = bool(L.pgo_terminated(h))
        if s % 30 == 29:
            check("step %d" % s)
    for h in hs:
        L.pgo_close(h)
    eng.close()


def test_climber_lockstep_jump_heavy_actions():
    # Uniform random actions rarely leave the floor; biasing towards the jump actions (2, 5, 8) makes agents climb,
    # collect crystals (entity destruction -> draw-list rebuild) and die on mobs (auto-reset, new level).
    n = 192
    eng, ora = EngineVec("climber", n, seed_base=31), OracleVec("climber", n, seed_base=31)
    assert np.array_equal(eng.reset(), ora.reset_obs())
    idx = np.arange(n)
    ends, rew = 0, 0.0
    for s in range(900):
        a = _actions(ora.L, 2, s, n)
        a = np.where((idx + s) % 5 < 3, (a % 3) * 3 + 2, a).astype(np.int32)
        oe, re_, de = eng.step(a)
        oo, ro, do = ora.step(a)
        assert np.array_equal(de, do) and np.array_equal(re_.view(np.uint32), ro.view(np.uint32)), s
        assert np.array_equal(oe, oo), s
        ends += int(do.sum())
        rew += float(ro.sum())
        if s % 150 == 0:
            for e in range(0, n, 24):
                assert np.array_equal(eng.state(e).view(np.uint32), ora.state(e).view(np.uint32)), (s, e)
                assert np.array_equal(eng.tiles(e), ora.tiles(e)), (s, e)
    assert ends > 50 and rew > 10.0, (ends, rew)
    eng.close()
    ora.close()


def test_maze_out_of_range_actions_follow_reference_quirk():
    # D6/D20: actions 9..15 teleport 2–3 cells in maze; others ignore them.
    n = 32
    eng, ora = EngineVec("maze", n, seed_base=3), OracleVec("maze", n, seed_base=3)
    assert np.array_equal(eng.reset(), ora.reset_obs())
    rng = np.random.default_rng(1)
    for s in range(120):
        a = rng.integers(0, 16, n).astype(np.int32)
        oe, re_, de = eng.step(a)
        oo, ro, do = ora.step(a)
        assert np.array_equal(oe, oo) and np.array_equal(de, do) and np.array_equal(re_, ro), s
    eng.close()
    ora.close()


@pytest.mark.parametrize("game", ["coinrun", "maze", "bossfight", "climber", "caveflyer", "chaser", "jumper"])
def test_reset_with_seed_option_and_mask(game):
    """cenv_reset's "seed" option (coinrun.cpp:313-317) per env, and masked resets leaving other envs untouched."""
    n = 16
    eng = EngineVec(game, n, seed_base=11)
    L = oracle()
    from oracle_util import register_textures
    register_textures(game)
    hs = [L.pgo_make(game.encode(), 11 + i, 1) for i in range(n)]
    for h in hs:
        L.pgo_reset(h, 0, 0)
    o = eng.reset()
    for i, h in enumerate(hs):
        assert np.array_equal(o[i], np.ctypeslib.as_array(L.pgo_obs(h), shape=(12288,)))
    pending = [False] * n
    for s in range(20):
        a = _actions(L, 2, s, n)
        eng.step(a)
        for i, h in enumerate(hs):
            if pending[i]:  # reference loop: `if term: env.reset()` replaces the next step
                L.pgo_reset(h, 0, 0)
                pending[i] = False
            else:
                L.pgo_step(h, int(a[i]))
                pending[i] = bool(L.pgo_terminated(h))
    before = eng.obs.copy()
    mask = np.zeros(n, np.uint8)
    mask[::3] = 1
    seeds = np.arange(n, dtype=np.int32) * 7 - 5  # includes negative seeds: int → u32 wrap
    o = eng.reset(mask=mask, seeds=seeds).copy()
    for i, h in enumerate(hs):
        if mask[i]:
            L.pgo_reset(h, 1, int(seeds[i]))
            assert np.array_equal(o[i], np.ctypeslib.as_array(L.pgo_obs(h), shape=(12288,))), i
        else:
            assert np.array_equal(o[i], before[i]), i
    pending = [p and not mask[i] for i, p in enumerate(pending)]  # an explicit reset clears a pending one
    for s in range(30):
        a = _actions(L, 3, s, n)
        oe, _, _ = eng.step(a)
        for i, h in enumerate(hs):
            if pending[i]:  # reference loop: `if term: env.reset()` replaces the next step
                L.pgo_reset(h, 0, 0)
                pending[i] = False
            else:
                L.pgo_step(h, int(a[i]))
                pending[i] = bool(L.pgo_terminated(h))
            if not np.array_equal(oe[i], np.ctypeslib.as_array(L.pgo_obs(h), shape=(12288,))):
                raise AssertionError("env %d step %d" % (i, s))
    for h in hs:
        L.pgo_close(h)
    eng.close()


def test_cenv_abi_single_env_matches_reference_loop():
    """The drop-in path: CEnv("libCoinRun.so", options={"seed": s}) → reset → step(int) … with the caller doing
    `if term: reset()` — exactly game_test.py:36-40 — against the oracle driven the same way."""
    for libname, game in (("libCoinRun.so", "coinrun"), ("libMaze.so", "maze"), ("libBossFight.so", "bossfight"),
                          ("libClimber.so", "climber"), ("libCaveFlyer.so", "caveflyer"),
                          ("libChaser.so", "chaser"), ("libJumper.so", "jumper")):
        env = pgcenv.CEnv(os.path.join(pglib.LIB_DIR, libname), options={"seed": 123})
        assert list(env.observation_space) == ["screen"] and list(env.action_space) == ["action"]
        assert list(env.action_space["action"].nvec) == [15]
        from oracle_util import register_textures
        register_textures(game)
        L = oracle()
        h = L.pgo_make(game.encode(), 123, 1)
        L.pgo_reset(h, 0, 0)
        obs, info = env.reset()
        assert obs["screen"].shape == (12288,) and obs["screen"].dtype == np.uint8 and info == {}
        assert np.array_equal(obs["screen"], np.ctypeslib.as_array(L.pgo_obs(h), shape=(12288,)))
        s = 1
        for i in range(700 if game == "maze" else 300):
            s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
            a = (s >> 16) % 15
            obs, rew, term, trunc, info = env.step(int(a))
            L.pgo_step(h, a)
            assert np.array_equal(obs["screen"], np.ctypeslib.as_array(L.pgo_obs(h), shape=(12288,))), i
            assert rew == L.pgo_reward(h) and term == bool(L.pgo_terminated(h)) and trunc is False
            if term:
                obs, _ = env.reset()
                L.pgo_reset(h, 0, 0)
                assert np.array_equal(obs["screen"], np.ctypeslib.as_array(L.pgo_obs(h), shape=(12288,)))
        frame = env.render()
        assert frame.shape == (512, 512, 3)
        env.close()
        L.pgo_close(h)


def test_cenv_abi_batched_through_unmodified_wrapper_shapes():
    """N envs through the reference wrapper's call shapes — action {"action": int32[N]},
    observations "screen" BYTE[N*12288] + "reward" FLOAT[N] + "terminated" BYTE[N]."""
    n = 8
    env = pgcenv.CEnv(os.path.join(pglib.LIB_DIR, "libprocgen2_hip.so"), options={"seed": 1, "num_envs": n, "game": 0})
    ora = OracleVec("coinrun", n, seed_base=1)
    obs, _ = env.reset()
    assert np.array_equal(obs["screen"].reshape(n, 12288), ora.reset_obs())
    for s in range(40):
        a = _actions(ora.L, 0, s, n)
        obs, rew, term, trunc, _ = env.step({"action": a})
        oo, ro, do = ora.step(a)
        assert np.array_equal(obs["screen"].reshape(n, 12288), oo)
        assert np.array_equal(obs["reward"], ro) and np.array_equal(obs["terminated"], do)
        assert abs(rew - float(ro.mean())) < 1e-6
    env.close()
    ora.close()


@pytest.mark.parametrize("game,n,steps", [("coinrun", 65536, 24), ("bossfight", 65536, 100), ("caveflyer", 32768, 40)])
def test_shard_invariance_and_determinism_at_full_size(game, n, steps):
    """BASELINE.json's full sizes (configs: coinrun 65 536, bossfight 65 536, caveflyer 32 768 envs): a shard created
    with env_offset reproduces the same global envs byte for byte (so results do not depend on the GPU count), two
    identical runs agree, and a strided sample of the envs matches the oracle.  (bossfight runs long enough for a large
    share of its envs to have ended an episode and drawn their next level inside the step.)"""
    big = EngineVec(game, n, seed_base=1)
    big.reset()
    ends = 0
    for s in range(steps):
        _, _, d = big.step(None, run_seed=0)
        ends += int(d.sum())
    obs_big, rew_big, done_big = (x.copy() for x in big._fetch())
    big.close()
    if game == "bossfight":
        assert ends > n // 4, ends

    lo = (n * 5) // 8
    shard = EngineVec(game, 512, seed_base=1, env_offset=lo)
    shard.reset()
    for s in range(steps):
        shard.step(None, run_seed=0)
    o, r, d = shard._fetch()
    assert np.array_equal(o, obs_big[lo:lo + 512]) and np.array_equal(r, rew_big[lo:lo + 512])
    assert np.array_equal(d, done_big[lo:lo + 512])
    shard.close()

    again = EngineVec(game, n, seed_base=1)
    again.reset()
    for s in range(steps):
        again.step(None, run_seed=0)
    o2, r2, d2 = again._fetch()
    assert np.array_equal(o2, obs_big) and np.array_equal(r2, rew_big) and np.array_equal(d2, done_big)
    again.close()

    L = oracle()
    from oracle_util import register_textures
    register_textures(game)
    for g in range(0, n, n // 16 + 3):  # 16 envs spread over the whole range
        h = L.pgo_make(game.encode(), 1 + g, 1)
        L.pgo_reset(h, 0, 0)
        pending = False
        for s in range(steps):
            if pending:
                L.pgo_reset(h, 0, 0)
                pending = False
            else:
                L.pgo_step(h, L.pgo_synthetic_action(0, s, g))
                pending = bool(L.pgo_terminated(h))
        assert np.array_equal(obs_big[g], np.ctypeslib.as_array(L.pgo_obs(h), shape=(12288,))), g
        L.pgo_close(h)


def _every_env_against_the_oracle(game, n, steps, env_offset=0, chunk=16384, eng=None, return_dones=False):
    """Every env of an engine batch against the oracle: rewards and dones of every step, every observation byte of the
    reset frame, of the middle step and of the last one.  The engine runs first (device-generated actions); the oracle
    then replays the envs in chunks of `chunk` (seeds and the action hash depend on the global index only) with all
    host threads, so its memory stays bounded whatever n is.  The oracle draws only the frames that are compared
    (OracleVec.set_render: drawing does not feed back into the game), which is what makes hundreds of steps of 65 536
    envs affordable on the host."""
    own = eng is None
    if own:
        eng = EngineVec(game, n, seed_base=1, env_offset=env_offset)
    checkpoints = {-1: eng.reset().copy()}
    rewards, dones = np.zeros((steps, n), np.float32), np.zeros((steps, n), np.uint8)
    for s in range(steps):
        eng.step_quiet(run_seed=0)
        if s in (steps // 2, steps - 1):
            checkpoints[s] = eng._fetch()[0].copy()
            rewards[s], dones[s] = eng.reward, eng.done
        else:
            rewards[s], dones[s] = eng.fetch_scalars()
    if own:
        eng.close()
    threads = _host_threads()
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        ora = OracleVec(game, hi - lo, seed_base=1, env_offset=env_offset + lo, threads=threads)
        assert np.array_equal(ora.reset_obs(), checkpoints[-1][lo:hi]), (game, "reset frame", lo)  # (made = reset once)
        for s in range(steps):
            ora.set_render(s in checkpoints)
            oo, ro, do = ora.step(None, run_seed=0, threads=threads)
            assert np.array_equal(do, dones[s, lo:hi]), (game, "done", s, lo)
            assert np.array_equal(ro.view(np.uint32), rewards[s, lo:hi].view(np.uint32)), (game, "reward bits", s, lo)
            if s in checkpoints and not np.array_equal(oo, checkpoints[s][lo:hi]):
                bad = np.nonzero((oo != checkpoints[s][lo:hi]).any(axis=1))[0]
                raise AssertionError("%s: obs differ at step %d in %d envs (first: global env %d)" %
                                     (game, s, bad.size, env_offset + lo + bad[0]))
        ora.close()
    return dones if return_dones else int(dones.sum())


@pytest.mark.parametrize("game,n,steps", [("coinrun", 65536, 48), ("bossfight", 65536, 100), ("caveflyer", 32768, 48),
                                          ("chaser", 65536, 400), ("jumper", 65536, 96), ("climber", 65536, 96),
                                          ("maze", 65536, 520)])
def test_every_env_at_full_size_matches_the_oracle(game, n, steps):
    """BASELINE.json configs[1..3] at their full sizes, EVERY env (a strided sample cannot see a bug that needs a
    particular env index mod something, LDS slot or XCD to show): 65 536 coinrun, 65 536 bossfight (long enough for
    thousands of episodes to end and draw their next level inside the step), 32 768 caveflyer envs — and the other four
    games at 65 536: chaser long enough that tens of thousands of episodes end inside the window (its levels are
    generated on a side stream beside the step's logic and render kernels, the reset envs drawn by a late pass behind a
    flag protocol: the one place a race was found once), maze through its 500-step cap in every env."""
    dones = _every_env_against_the_oracle(game, n, steps, return_dones=True)
    ends = int(dones.sum())
    if game == "bossfight":
        assert ends > 1000, ends
    if game == "chaser":
        assert ends >= 10000, ends  # the side-stream generator and the late render pass were exercised
    if game == "maze":
        assert bool(dones.any(axis=0).all()), "every maze env ends by step 500 (maze.cpp:308-310)"


def test_vec_env_torch_zero_copy_matches_c_abi():
    import torch
    from procgen2_amd.vec_env import ProcgenVecEnv
    n = 128
    env = ProcgenVecEnv("coinrun", n, seed_base=1)
    ref = EngineVec("coinrun", n, seed_base=1)
    o = env.reset()
    env.sync()
    assert o.shape == (n, 64, 64, 3) and o.dtype == torch.uint8 and o.is_cuda
    assert np.array_equal(o.cpu().numpy().reshape(n, -1), ref.reset())
    L = oracle()
    for s in range(30):
        a = _actions(L, 0, s, n)
        o, r, d = env.step(torch.from_numpy(a).cuda())
        oe, re_, de = ref.step(a)
        env.sync()
        assert np.array_equal(o.cpu().numpy().reshape(n, -1), oe)
        assert np.array_equal(r.cpu().numpy(), re_) and np.array_equal(d.cpu().numpy(), de)
    env.close()
    ref.close()


def _appendix_c():
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "appendix_c.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("trace", _appendix_c()["traces"] + _appendix_c()["traces_float_abs"],
                         ids=lambda t: "%s-%d%s" % (t["game"], t["seed"], "-float-abs" if t.get("flags") else ""))
def test_engine_reproduces_the_reference_reward_done_traces(trace):
    """The HIP engine against the reference itself, without the oracle in between: SURVEY.md Appendix C recorded the
    CRC-32 of the (reward f32, terminated u8) stream of the UNMODIFIED reference sources for make(seed) → reset →
    20 000 LCG actions with `if terminated: reset()`.  One engine env with that seed, the same actions, an explicit
    reset after every terminal step (which is what the caller's loop does; it replaces the engine's auto-reset)."""
    import struct
    import zlib
    steps = _appendix_c()["steps"]
    eng = EngineVec(trace["game"], 1, seed_base=trace["seed"], game_flags=trace.get("flags", 0))  # flags: appendix_c.json
    eng.reset()
    s, crc, episodes, total, length, lengths = 1, 0, 0, 0.0, 0, []
    one = np.ones(1, np.uint8)
    for _ in range(steps):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        _, r, d = eng.step(np.array([(s >> 16) % 15], np.int32))
        crc = zlib.crc32(struct.pack("<fB", float(r[0]), int(d[0])), crc)
        total += float(r[0])
        length += 1
        if d[0]:
            episodes += 1
            lengths.append(length)
            length = 0
            eng.reset(mask=one)
    eng.close()
    assert "%08x" % crc == trace["crc"]
    assert episodes == trace["episodes"]
    assert abs(total - trace["reward_sum"]) < 1e-3
    want = trace["first_lengths"]
    assert lengths[:len(want)] == want


@pytest.mark.parametrize("game,steps", [("coinrun", 3000), ("chaser", 1200), ("climber", 2000)])
def test_long_rollout_at_full_size_stays_on_the_oracle(game, steps):
    """Soak: 65 536 envs for thousands of steps — every env goes through many episodes, prefetched levels, synchronous
    fallbacks and (chaser) in-step generation — then a strided sample of envs must still be where the oracle is:
    last observation, reward, done."""
    from oracle_util import register_textures
    n = 65536
    eng = EngineVec(game, n, seed_base=1)
    eng.reset()
    for s in range(steps):
        pglib.check(eng.L, eng.L.pgv_step_synthetic(eng.h, 5), "pgv_step_synthetic")
    obs, rew, done = (x.copy() for x in eng._fetch())
    eng.close()
    L = oracle()
    register_textures(game)
    for g in range(0, n, n // 12 + 7):
        h = L.pgo_make(game.encode(), 1 + g, 1)
        L.pgo_reset(h, 0, 0)
        pending, r, d = False, 0.0, 0
        for s in range(steps):
            if pending:
                L.pgo_reset(h, 0, 0)
                pending, r, d = False, 0.0, 0
            else:
                L.pgo_step(h, L.pgo_synthetic_action(5, s, g))
                r, d = L.pgo_reward(h), int(L.pgo_terminated(h))
                pending = bool(d)
        assert np.array_equal(obs[g], np.ctypeslib.as_array(L.pgo_obs(h), shape=(12288,))), g
        assert np.float32(r) == rew[g] and d == done[g], g
        L.pgo_close(h)


def _host_threads():
    import bench
    return bench.usable_cores()


def test_coinrun_4096_envs_256_steps_every_byte():
    """BASELINE.json configs[1] at its stated size: coinrun, 4 096 envs, every observation byte, reward bit pattern and
    done flag of every env for 256 steps against the oracle (all host threads)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    resets = _lockstep("coinrun", 4096, 256, check_state_every=64, threads=_host_threads())
    assert resets > 0


def test_chaser_float_abs_variant_lockstep():
    """PGV_CHASER_FLOAT_ABS (include/procgen2_vec.h): the other reading of chaser's abs(<float>) calls — every byte
    against the oracle's same variant, steering-heavy actions so that cell-centre alignment matters."""
    n = 192
    eng = EngineVec("chaser", n, seed_base=51, game_flags=pglib.CHASER_FLOAT_ABS)
    ora = OracleVec("chaser", n, seed_base=51, game_flags=pglib.CHASER_FLOAT_ABS)
    plain = OracleVec("chaser", n, seed_base=51, render=False)
    assert np.array_equal(eng.reset(), ora.reset_obs())
    rng = np.random.default_rng(12)
    ends, differs = 0, False
    hold = rng.choice([1, 3, 5, 7], n)
    for s in range(700):
        hold = np.where(rng.random(n) < 0.15, rng.choice([1, 3, 5, 7], n), hold)
        a = np.where(rng.random(n) < 0.1, rng.integers(0, 15, n), hold).astype(np.int32)
        oe, re_, de = eng.step(a)
        oo, ro, do = ora.step(a)
        _, rp, dp = plain.step(a)
        assert np.array_equal(de, do) and np.array_equal(re_.view(np.uint32), ro.view(np.uint32)), s
        assert np.array_equal(oe, oo), s
        differs = differs or not (np.array_equal(do, dp) and np.array_equal(ro, rp))
        ends += int(do.sum())
        if s % 100 == 0:
            for e in range(0, n, 24):
                assert np.array_equal(eng.state(e, 1024).view(np.uint32), ora.state(e, 1024).view(np.uint32)), (s, e)
    assert ends > 20 and differs, "the variant must actually change the game"
    for v in (eng, ora, plain):
        v.close()


def test_jumper_float_abs_variant_lockstep():
    """PGV_JUMPER_FLOAT_ABS: jumper/common_systems.cpp:198 decides when dust particles are emitted — pixels only."""
    resets = _lockstep("jumper", 128, 400, seed_base=61, run_seed=5, check_state_every=80, game_flags=pglib.JUMPER_FLOAT_ABS)
    assert resets >= 0


def test_engine_refuses_unknown_game_flags():
    for game, flags in (("maze", 1), ("chaser", 2), ("jumper", 4), ("coinrun", 16)):
        with pytest.raises(pglib.EngineError):
            EngineVec(game, 4, game_flags=flags)


def test_mixed_seven_game_slice_of_configs4():
    """BASELINE.json configs[4], one GPU's share: seven vector envs (65 536 envs split seven ways, the last game takes
    the remainder) on seven HIP streams of one device, all writing their blocks of ONE [65 536, 64, 64, 3] slab
    (bench.py --workload mixed lays them out like this), stepped side by side with device-generated
    actions and no ordering between them.  After 48 steps EVERY env of every game must be where the oracle is (rewards
    and dones of every step, every observation byte of the last); after 200, a strided sample still is.  The envs sit
    at the global indices rank 3 of 8 would own (env_offset)."""
    import torch
    from oracle_util import register_textures
    from procgen2_amd.vec_env import GAMES, ProcgenVecEnv
    per_gpu, rank, first, steps = 65536, 3, 48, 200
    base = per_gpu // len(GAMES)
    counts = [base] * (len(GAMES) - 1) + [per_gpu - base * (len(GAMES) - 1)]
    assert counts == [9362] * 6 + [9364]
    slab = (torch.zeros((per_gpu, 64, 64, 3), dtype=torch.uint8, device="cuda"),
            torch.zeros(per_gpu, dtype=torch.float32, device="cuda"), torch.zeros(per_gpu, dtype=torch.uint8, device="cuda"))
    envs, at = [], 0
    for g, c in zip(GAMES, counts):
        envs.append(ProcgenVecEnv(g, c, seed_base=1, env_offset=rank * c, out=tuple(t[at:at + c] for t in slab)))
        assert envs[-1].obs.data_ptr() == slab[0][at:at + c].data_ptr()
        at += c
    for e in envs:
        e.reset()
    rewards, dones = np.zeros((first, per_gpu), np.float32), np.zeros((first, per_gpu), np.uint8)
    for s in range(first):
        for e in envs:
            e.step_synthetic(0, ordered=False)  # each on its own stream, nothing waits for anything
        for e in envs:
            e.sync()
        rewards[s], dones[s] = slab[1].cpu().numpy(), slab[2].cpu().numpy()
    obs_first = slab[0].cpu().numpy().reshape(per_gpu, -1)
    threads, at = _host_threads(), 0
    for game, count in zip(GAMES, counts):
        ora = OracleVec(game, count, seed_base=1, env_offset=rank * count, threads=threads)  # (made = reset once)
        for s in range(first):
            oo, ro, do = ora.step(None, run_seed=0, threads=threads)
            assert np.array_equal(do, dones[s, at:at + count]), (game, "done", s)
            assert np.array_equal(ro.view(np.uint32), rewards[s, at:at + count].view(np.uint32)), (game, "reward bits", s)
        if not np.array_equal(oo, obs_first[at:at + count]):
            bad = np.nonzero((oo != obs_first[at:at + count]).any(axis=1))[0]
            raise AssertionError("%s: obs differ after %d steps in %d envs (first: env %d of the game)" % (game, first, bad.size, bad[0]))
        ora.close()
        at += count
    del obs_first
    for s in range(first, steps):
        for e in envs:
            e.step_synthetic(0, ordered=False)
    for e in envs:
        e.sync()
    L = oracle()
    for e, game, count in zip(envs, GAMES, counts):
        obs = e.obs.cpu().numpy().reshape(count, -1)
        rew, done = e.reward.cpu().numpy(), e.done.cpu().numpy()
        register_textures(game)
        for i in list(range(0, count, count // 6 + 1)) + [count - 1]:
            g = rank * count + i  # global env index: seed 1 + g, action hash over g
            h = L.pgo_make(game.encode(), (1 + g) & 0xFFFFFFFF, 1)
            L.pgo_reset(h, 0, 0)
            pending, r, d = False, 0.0, 0
            for s in range(steps):
                if pending:
                    L.pgo_reset(h, 0, 0)
                    pending, r, d = False, 0.0, 0
                else:
                    L.pgo_step(h, L.pgo_synthetic_action(0, s, g))
                    r, d = L.pgo_reward(h), int(L.pgo_terminated(h))
                    pending = bool(d)
            assert np.array_equal(obs[i], np.ctypeslib.as_array(L.pgo_obs(h), shape=(12288,))), (game, i)
            assert np.float32(r) == rew[i] and d == done[i], (game, i)
            L.pgo_close(h)
        e.close()
    torch.cuda.synchronize()


def test_cenv_abi_batched_maze_64_envs_through_the_timeout():
    """BASELINE.json configs[0]'s shape on the engine: libMaze.so with num_envs = 64 through the reference wrapper's
    call shapes for 520 steps — every env runs into maze's 500-step cap (D5) and is auto-reset."""
    n = 64
    env = pgcenv.CEnv(os.path.join(pglib.LIB_DIR, "libMaze.so"), options={"seed": 1, "num_envs": n})
    ora = OracleVec("maze", n, seed_base=1)
    obs, _ = env.reset()
    assert np.array_equal(obs["screen"].reshape(n, 12288), ora.reset_obs())
    ends = 0
    for s in range(520):
        a = _actions(ora.L, 0, s, n)
        obs, rew, term, trunc, _ = env.step({"action": a})
        oo, ro, do = ora.step(a)
        assert np.array_equal(obs["screen"].reshape(n, 12288), oo), s
        assert np.array_equal(obs["reward"], ro) and np.array_equal(obs["terminated"], do), s
        assert trunc is False and term == bool(do.all())
        ends += int(do.sum())
    assert ends >= n
    env.close()
    ora.close()


def test_bossfight_step_and_reset_after_a_non_square_human_frame():
    """D15 (bossfight.cpp:434,458, common_systems.cpp:227-228,513-514): reset() and both update()s read the camera size
    and scale the LAST render left in the global renderer.  After a human-size cenv_render of a non-square window the
    next step clamps against another screen rectangle and the next reset spawns the agent and the barriers on another
    row; the observation render that follows puts 64×64 back.  Vector path (pgv_render_frame of single envs) and the
    drop-in cenv path, against the oracle doing the same calls."""
    from oracle_util import register_textures
    register_textures("bossfight")
    L = oracle()
    n, W, H = 8, 200, 120
    eng = EngineVec("bossfight", n, seed_base=9)
    hs = [L.pgo_make(b"bossfight", 9 + i, 1) for i in range(n)]
    for h in hs:
        L.pgo_reset(h, 0, 0)
    assert np.array_equal(eng.reset(), np.stack([np.ctypeslib.as_array(L.pgo_obs(h), shape=(12288,)) for h in hs]))
    want = np.zeros((H, W, 3), np.uint8)
    pending = [False] * n
    for s in range(140):
        if s % 9 == 4:  # a human frame of two envs between steps; env 3 is also reset right after its frame
            for env in (1, 3):
                L.pgo_render_frame(hs[env], W, H, want.ctypes.data_as(ctypes.c_void_p))
                assert np.array_equal(eng.frame(env, W, H), want), (s, env)
            if s % 18 == 4:
                mask = np.zeros(n, np.uint8)
                mask[3] = 1
                o = eng.reset(mask=mask)
                L.pgo_reset(hs[3], 0, 0)
                pending[3] = False
                assert np.array_equal(o[3], np.ctypeslib.as_array(L.pgo_obs(hs[3]), shape=(12288,))), s
                assert np.array_equal(eng.state(3, 400).view(np.uint32), _oracle_state(L, hs[3], 400).view(np.uint32)), s
        a = np.where(np.arange(n) % 2 == 0, 9, _actions(L, 3, s, n)).astype(np.int32)
        oe, re_, de = eng.step(a)
        for i, h in enumerate(hs):
            if pending[i]:
                L.pgo_reset(h, 0, 0)
                pending[i] = False
                assert re_[i] == 0.0 and de[i] == 0
            else:
                L.pgo_step(h, int(a[i]))
                pending[i] = bool(L.pgo_terminated(h))
                assert re_[i] == np.float32(L.pgo_reward(h)) and de[i] == int(L.pgo_terminated(h)), (s, i)
            assert np.array_equal(oe[i], np.ctypeslib.as_array(L.pgo_obs(h), shape=(12288,))), (s, i)
    for h in hs:
        L.pgo_close(h)
    eng.close()
    # the drop-in path: CEnv with a 200×120 window, render() then reset()
    env = pgcenv.CEnv(os.path.join(pglib.LIB_DIR, "libBossFight.so"), options={"seed": 5, "width": W, "height": H})
    h = L.pgo_make(b"bossfight", 5, 1)
    L.pgo_reset(h, 0, 0)
    obs, _ = env.reset()
    assert np.array_equal(obs["screen"], np.ctypeslib.as_array(L.pgo_obs(h), shape=(12288,)))
    for k in range(12):
        obs, rew, term, _, _ = env.step(9 if k % 2 else 7)
        L.pgo_step(h, 9 if k % 2 else 7)
        if k % 4 == 1:
            frame = env.render()
            L.pgo_render_frame(h, W, H, want.ctypes.data_as(ctypes.c_void_p))
            assert frame.shape == (H, W, 3) and np.array_equal(frame, want)
        if k == 5:  # render() was the last thing that touched the camera: this reset sees the window's
            obs, _ = env.reset()
            L.pgo_reset(h, 0, 0)
        assert np.array_equal(obs["screen"], np.ctypeslib.as_array(L.pgo_obs(h), shape=(12288,))), k
    env.close()
    L.pgo_close(h)


def _oracle_state(L, h, cap):
    buf = (ctypes.c_float * cap)()
    m = L.pgo_dump_state(h, buf, cap)
    return np.array(buf[:min(m, cap)], np.float32)


def _seven_game_slab(per_game, rank=0):
    """Seven vector envs on their own streams writing their blocks of ONE slab triple, as bench.py --workload mixed lays
    them out (and test_mixed_seven_game_slice_of_configs4 checks against the oracle)."""
    import torch
    from procgen2_amd.vec_env import GAMES, ProcgenVecEnv
    total = per_game * len(GAMES)
    slab = (torch.zeros((total, 64, 64, 3), dtype=torch.uint8, device="cuda"),
            torch.zeros(total, dtype=torch.float32, device="cuda"), torch.zeros(total, dtype=torch.uint8, device="cuda"))
    envs = [ProcgenVecEnv(g, per_game, seed_base=1, env_offset=rank * per_game,
                          out=tuple(t[k * per_game:(k + 1) * per_game] for t in slab)) for k, g in enumerate(GAMES)]
    for e in envs:
        e.reset()
    for e in envs:
        e.sync()
    return envs, slab


def test_publish_and_consume_keep_unordered_steps_and_their_readers_apart():
    """The stream-ordering contract of ProcgenVecEnv (vec_env.py publish / consume), on one GPU:
    seven games step side by side with step_synthetic(ordered=False) — each on its own stream, nothing waits for
    anything — and after every step a READER on torch's current stream (here: a deliberately late device-to-device copy
    of the three slabs into a history ring, the stand-in for bench.py's rooted RCCL gather) must see that step's
    outputs, whole: publish() orders the reader behind the step, consume() orders the NEXT step behind the reader.
    160 steps with no host synchronisation inside; afterwards every copied step equals a synchronous run of the same
    envs, and a handful of envs per game are where the oracle is.

    The test has teeth: the same loop without consume() — the engines then run ahead of the late reader and overwrite
    the slab under it — is run too and MUST come out torn (observed: every one of its copied steps differs)."""
    import torch
    from oracle_util import register_textures
    from procgen2_amd.vec_env import GAMES
    per_game, steps = 128, 160
    total = per_game * len(GAMES)

    def rollout(consume):
        envs, slab = _seven_game_slab(per_game)
        hist = tuple(torch.zeros((steps,) + tuple(t.shape), dtype=t.dtype, device="cuda") for t in slab)
        torch.cuda.synchronize()
        for s in range(steps):
            for e in envs:
                e.step_synthetic(0, ordered=False)
            for e in envs:
                e.publish()
            torch.cuda._sleep(2_000_000)  # the reader is late: about a millisecond, several steps' worth
            for t, h in zip(slab, hist):
                h[s].copy_(t, non_blocking=True)
            if consume:
                for e in envs:
                    e.consume()
        for e in envs:
            e.sync()
        torch.cuda.synchronize()
        out = tuple(h.cpu().numpy() for h in hist)
        for e in envs:
            e.close()
        return out

    # the reference rollout: the same envs, the host waiting for every step
    envs, slab = _seven_game_slab(per_game)
    want = (np.zeros((steps, total, 64, 64, 3), np.uint8), np.zeros((steps, total), np.float32), np.zeros((steps, total), np.uint8))
    for s in range(steps):
        for e in envs:
            e.step_synthetic(0, ordered=False)
        for e in envs:
            e.sync()
        for w, t in zip(want, slab):
            w[s] = t.cpu().numpy()
    for e in envs:
        e.close()

    got = rollout(consume=True)
    for s in range(steps):
        assert np.array_equal(got[2][s], want[2][s]), ("done", s)
        assert np.array_equal(got[1][s].view(np.uint32), want[1][s].view(np.uint32)), ("reward", s)
        assert np.array_equal(got[0][s], want[0][s]), ("obs", s)
    torn = rollout(consume=False)
    differing = sum(not np.array_equal(torn[0][s], want[0][s]) for s in range(steps))
    assert differing > steps // 2, "without consume() the late reader must see later steps' frames (%d of %d differ)" % (differing, steps)

    # … and the rollout itself is the oracle's: a few envs of every game after the last step
    L = oracle()
    last = want[0][steps - 1].reshape(total, -1)
    for k, game in enumerate(GAMES):
        register_textures(game)
        for i in (0, per_game // 3, per_game - 1):
            h = L.pgo_make(game.encode(), 1 + i, 1)
            L.pgo_reset(h, 0, 0)
            pending, r, d = False, 0.0, 0
            for s in range(steps):
                if pending:
                    L.pgo_reset(h, 0, 0)
                    pending, r, d = False, 0.0, 0
                else:
                    L.pgo_step(h, L.pgo_synthetic_action(0, s, i))
                    r, d = L.pgo_reward(h), int(L.pgo_terminated(h))
                    pending = bool(d)
            assert np.array_equal(last[k * per_game + i], np.ctypeslib.as_array(L.pgo_obs(h), shape=(12288,))), (game, i)
            assert np.float32(r) == want[1][steps - 1][k * per_game + i] and d == want[2][steps - 1][k * per_game + i], (game, i)
            L.pgo_close(h)


def test_step_many_synthetic_equals_stepping_the_envs_one_by_one():
    """pgv_step_synthetic_many (the launch loop of bench.py --workload mixed, in C) against a per-env
    pgv_step_synthetic loop from the same snapshot: identical observations, rewards and dones for every env of every
    game, and the per-step timing entry point pgv_step_times is sane."""
    import torch
    from procgen2_amd.vec_env import GAMES, step_many_synthetic
    per_game, steps = 192, 40
    envs, slab = _seven_game_slab(per_game)
    snaps = [e.save_state() for e in envs]
    step_many_synthetic(envs, steps, 3)
    for e in envs:
        e.sync()
    many = tuple(t.cpu().numpy().copy() for t in slab)
    for e, snap in zip(envs, snaps):
        e.load_state(snap)
    for s in range(steps):
        for e in envs:
            e.step_synthetic(3, ordered=False)
    for e in envs:
        e.sync()
    one = tuple(t.cpu().numpy() for t in slab)
    assert np.array_equal(many[2], one[2]) and np.array_equal(many[1].view(np.uint32), one[1].view(np.uint32))
    assert np.array_equal(many[0], one[0])
    # pgv_step_times: one entry per step, all positive, render within the step, and the sum close to a timed run's total
    e = envs[0]
    step_ms, render_ms = e.step_times(32, 3)
    assert step_ms.shape == (32,) and render_ms.shape == (32,)
    assert (step_ms > 0).all() and (render_ms > 0).all() and (render_ms <= step_ms * 1.05).all()
    total_ms, _ = e.timed_steps(32, 3)
    assert 0.3 * total_ms < float(step_ms.sum()) < 3.0 * total_ms, (float(step_ms.sum()), total_ms)
    # pgv_step_phases: the same step cut in four by events on the stream — the parts are non-negative and add up to the step
    ph = e.step_phases(16, 3)
    assert set(ph) == {"step", "logic", "prepass", "render", "late"} and all(v.shape == (16,) for v in ph.values())
    parts = ph["logic"] + ph["prepass"] + ph["render"] + ph["late"]
    assert (ph["logic"] > 0).all() and (ph["render"] > 0).all() and (ph["prepass"] >= 0).all() and (ph["late"] >= 0).all()
    assert np.allclose(parts, ph["step"], rtol=0.02, atol=0.005), (parts, ph["step"])
    # … and of all seven games side by side (the mixed bench line's window); the rollout it makes is the one
    # step_many_synthetic makes
    from procgen2_amd.vec_env import step_phases_many
    snaps = [x.save_state() for x in envs]
    many_ms = step_phases_many(envs, 8, 3)
    for x in envs:
        x.sync()
    assert many_ms.shape == (len(envs), 5, 8) and (many_ms[:, 0] > 0).all() and (many_ms[:, 3] > 0).all()
    assert np.allclose(many_ms[:, 1:].sum(axis=1), many_ms[:, 0], rtol=0.02, atol=0.005)
    after_phases = tuple(t.cpu().numpy().copy() for t in slab)
    for x, snap in zip(envs, snaps):
        x.load_state(snap)
    step_many_synthetic(envs, 8, 3)
    for x in envs:
        x.sync()
    assert all(np.array_equal(a, t.cpu().numpy()) for a, t in zip(after_phases, slab))
    for e in envs:
        e.close()
    torch.cuda.synchronize()
