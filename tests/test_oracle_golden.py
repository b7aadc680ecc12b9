"""Pins the CPU oracle: (1) against the reward/terminated CRC traces that SURVEY.md Appendix C recorded from
the unmodified reference game sources; (2) frame checksums of the oracle itself as a regression guard
(raster-spec pixels are not pinned by the reference: no SDL in the image — DESIGN.md §oracle)."""
import ctypes
import json
import os
import zlib

import numpy as np
import pytest

import oracle_util

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
IMPLEMENTED = ("coinrun", "maze", "bossfight", "climber", "caveflyer", "chaser", "jumper")

with open(os.path.join(GOLDEN, "appendix_c.json")) as f:
    APPENDIX_C = json.load(f)


@pytest.mark.parametrize("trace", [t for t in APPENDIX_C["traces"] if t["game"] in IMPLEMENTED],
                         ids=lambda t: "%s-%d" % (t["game"], t["seed"]))
def test_reference_reward_done_trace(oracle_lib, trace):
    crc, episodes, total = ctypes.c_uint32(), ctypes.c_int(), ctypes.c_double()
    lens = (ctypes.c_int * 6)()
    rc = oracle_lib.pgo_trace(trace["game"].encode(), trace["seed"], APPENDIX_C["steps"], crc, episodes, total, lens, 6)
    assert rc == 0
    assert "%08x" % crc.value == trace["crc"]
    assert episodes.value == trace["episodes"]
    assert abs(total.value - trace["reward_sum"]) < 1e-3
    want = trace["first_lengths"]
    assert list(lens)[:len(want)] == want


@pytest.mark.parametrize("trace", APPENDIX_C["traces_float_abs"], ids=lambda t: "%s-%d-float-abs" % (t["game"], t["seed"]))
def test_reference_reward_done_trace_float_abs(oracle_lib, trace):
    """The other reading of the same unmodified sources (appendix_c.json `_more_source`): game_flags bit 0."""
    crc, episodes, total = ctypes.c_uint32(), ctypes.c_int(), ctypes.c_double()
    lens = (ctypes.c_int * 6)()
    rc = oracle_lib.pgo_trace_flags(trace["game"].encode(), trace["seed"], APPENDIX_C["steps"], trace["flags"], crc, episodes,
                                    total, lens, 6)
    assert rc == 0
    assert "%08x" % crc.value == trace["crc"]
    assert episodes.value == trace["episodes"]
    assert abs(total.value - trace["reward_sum"]) < 1e-3
    want = trace["first_lengths"]
    assert list(lens)[:len(want)] == want


def _frame_crcs(game, seed, steps, mode=0):
    oracle_util.register_textures(game)
    L = oracle_util.oracle()
    h = L.pgo_make_mode(game.encode(), seed, 1, mode)
    L.pgo_reset(h, 0, 0)
    out = []
    s = 1
    for i in range(steps):
        frame = np.ctypeslib.as_array(L.pgo_obs(h), shape=(12288,))
        out.append(zlib.crc32(frame.tobytes()))
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        L.pgo_step(h, (s >> 16) % 15)
        if L.pgo_terminated(h):
            L.pgo_reset(h, 0, 0)
    L.pgo_close(h)
    return out


@pytest.mark.parametrize("game", IMPLEMENTED)
def test_oracle_frame_checksums(game):
    with open(os.path.join(GOLDEN, "oracle_frames.json")) as f:
        gold = json.load(f)
    for key, want in gold[game].items():
        seed, steps = (int(v) for v in key.split(":"))
        assert _frame_crcs(game, seed, steps) == want, "%s seed %d" % (game, seed)


def test_oracle_frame_checksums_in_every_distribution_mode():
    """The non-default distribution modes (tests/test_modes.py) have no reference trace; these checksums of the
    oracle's own frames at least keep its reading of them from drifting unnoticed."""
    with open(os.path.join(GOLDEN, "oracle_mode_frames.json")) as f:
        gold = json.load(f)
    assert len(gold) == 12
    for key, want in gold.items():
        game, mode, seed, steps = key.split(":")
        assert _frame_crcs(game, int(seed), int(steps), int(mode)) == want, key


def test_oracle_reset_frame_has_no_sprites_and_reseed_repeats():
    """D2 (reset frame drawn before the first sprite update) and the reset "seed" option (coinrun.cpp:313-317)."""
    oracle_util.register_textures("coinrun")
    L = oracle_util.oracle()
    h = L.pgo_make(b"coinrun", 5, 1)
    L.pgo_reset(h, 1, 77)
    a = np.ctypeslib.as_array(L.pgo_obs(h), shape=(12288,)).copy()
    for _ in range(10):
        L.pgo_step(h, 7)
    L.pgo_reset(h, 1, 77)
    b = np.ctypeslib.as_array(L.pgo_obs(h), shape=(12288,)).copy()
    L.pgo_close(h)
    # same seed → same level and backdrop; only the camera differs (D3: reset frame uses the last camera)
    assert a.shape == b.shape and a.any() and b.any()
