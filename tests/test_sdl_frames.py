"""Frames recorded from the REAL reference (SDL3 software renderer) against the oracle and the engine.

tests/golden/sdl_frames.npz does not exist in this repo: the reference cannot be built in its image (SDL3 pre-release +
SDL3_image are absent; no stand-ins are written).  tools/record_reference_frames.md says how to make it on a box that has
them; until then these tests skip and the observation pixels stay pinned to the in-repo raster spec only
("parity unpinned at the SDL boundary")."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FRAMES = os.path.join(HERE, "golden", "sdl_frames.npz")

needs_frames = pytest.mark.skipif(not os.path.exists(FRAMES), reason="tests/golden/sdl_frames.npz not recorded "
                                  "(needs the reference built against SDL3: tools/record_reference_frames.md)")


def load_records(path=FRAMES):
    """The file format of tools/record_reference_frames.py, validated."""
    z = np.load(path)
    games, seeds, actions, frames, resets = z["games"], z["seeds"], z["actions"], z["frames"], z["resets"]
    r, s = actions.shape
    assert games.shape == (r,) and seeds.shape == (r,) and resets.shape == (r, s)
    assert frames.shape == (r, s + 1, 64, 64, 3) and frames.dtype == np.uint8
    assert actions.min() >= 0 and actions.max() <= 14
    return [dict(game=str(games[k]), seed=int(seeds[k]), actions=actions[k], frames=frames[k], resets=resets[k]) for k in range(r)]


def describe(expected, got, what):
    diff = np.abs(expected.astype(np.int16) - got.astype(np.int16))
    return "%s: %d pixels differ, largest channel difference %d" % (what, int((diff.max(axis=-1) > 0).sum()), int(diff.max()))


def replay_oracle(rec):
    from oracle_util import oracle, register_textures
    L = oracle()
    register_textures(rec["game"])
    h = L.pgo_make(rec["game"].encode(), rec["seed"], 1)
    L.pgo_reset(h, 0, 0)
    out = [np.ctypeslib.as_array(L.pgo_obs(h), shape=(12288,)).reshape(64, 64, 3).copy()]
    for s, act in enumerate(rec["actions"]):
        L.pgo_step(h, int(act))
        term = bool(L.pgo_terminated(h))
        assert term == bool(rec["resets"][s]), "%s seed %d: terminated differs at step %d" % (rec["game"], rec["seed"], s)
        if term:
            L.pgo_reset(h, 0, 0)
        out.append(np.ctypeslib.as_array(L.pgo_obs(h), shape=(12288,)).reshape(64, 64, 3).copy())
    L.pgo_close(h)
    return np.stack(out)


@needs_frames
def test_oracle_frames_equal_the_reference_sdl_frames():
    for rec in load_records():
        got = replay_oracle(rec)
        for s in range(got.shape[0]):
            assert np.array_equal(got[s], rec["frames"][s]), describe(
                rec["frames"][s], got[s], "%s seed %d, frame %d (0 = reset)" % (rec["game"], rec["seed"], s))


@needs_frames
@pytest.mark.gpu
def test_engine_frames_equal_the_reference_sdl_frames():
    """One HIP env per record through the cenv-shaped single-env calls — no oracle in between."""
    from engine_util import EngineVec
    for rec in load_records():
        eng = EngineVec(rec["game"], 1, seed_base=rec["seed"], device=0)
        obs = eng.reset().reshape(64, 64, 3)
        assert np.array_equal(obs, rec["frames"][0]), describe(rec["frames"][0], obs, "%s seed %d, reset" % (rec["game"], rec["seed"]))
        for s, act in enumerate(rec["actions"]):
            o, _, d = eng.step(np.array([int(act)], np.int32))
            if d[0]:  # the engine resets on its NEXT step; the recording reset at once
                o, _, _ = eng.step(np.array([0], np.int32))
            o = o.reshape(64, 64, 3)
            assert np.array_equal(o, rec["frames"][s + 1]), describe(rec["frames"][s + 1], o, "%s seed %d, step %d" % (rec["game"], rec["seed"], s))
        eng.close()


def test_the_frame_file_format_round_trips(tmp_path):
    """The format the recorder writes and the loader reads is one format (checked on oracle frames standing in for
    recorded ones — this proves nothing about SDL, only that a recorded file would be consumed as documented)."""
    from oracle_util import oracle
    oracle()
    steps, seed, game = 12, 7, "maze"
    s, acts = (seed * 2654435761) & 0xffffffff, []
    for _ in range(steps):
        s = (s * 1664525 + 1013904223) & 0xffffffff
        acts.append((s >> 16) % 15)
    acts = np.array(acts, np.int32)
    rec = dict(game=game, seed=seed, actions=acts, resets=np.zeros(steps, np.uint8), frames=None)
    rec["resets"] = np.zeros(steps, np.uint8)
    frames = None
    try:
        frames = replay_oracle(rec)
    except AssertionError:  # an episode ended inside the sample: record where, as the recorder would
        from oracle_util import register_textures
        L = oracle()
        register_textures(game)
        h = L.pgo_make(game.encode(), seed, 1)
        L.pgo_reset(h, 0, 0)
        for k, act in enumerate(acts):
            L.pgo_step(h, int(act))
            if L.pgo_terminated(h):
                rec["resets"][k] = 1
                L.pgo_reset(h, 0, 0)
        L.pgo_close(h)
        frames = replay_oracle(rec)
    path = os.path.join(tmp_path, "sdl_frames.npz")
    np.savez_compressed(path, games=np.array([game]), seeds=np.array([seed], np.int64), actions=acts[None], frames=frames[None],
                        resets=rec["resets"][None], sdl=np.array("none: oracle frames, format check only"))
    (back,) = load_records(path)
    assert back["game"] == game and back["seed"] == seed and np.array_equal(back["frames"], frames)
    assert np.array_equal(replay_oracle(back), frames)
