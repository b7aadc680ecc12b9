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


def test_oracle_reseeded_reset_repeats_the_level_and_keeps_the_old_camera():
    """The reset "seed" option (coinrun.cpp:313-317) rebuilds the same level, and D3: the reset frame is drawn with the
    camera the previous episode's last step left (coinrun/common_systems.cpp:238-239 is the only writer), so the two
    reset frames of the same level differ exactly by that camera."""
    oracle_util.register_textures("coinrun")
    L = oracle_util.oracle()

    def state(h):
        buf = (ctypes.c_float * 512)()
        n = L.pgo_dump_state(h, buf, 512)
        return np.array(buf[:min(n, 512)], np.float32)

    def tiles(h):
        buf = (ctypes.c_uint8 * 4096)()
        n = L.pgo_dump_tiles(h, buf, 4096)
        return np.array(buf[:n], np.uint8)

    h = L.pgo_make(b"coinrun", 5, 1)
    L.pgo_reset(h, 1, 77)
    a, sa, ta = np.ctypeslib.as_array(L.pgo_obs(h), shape=(12288,)).copy(), state(h), tiles(h)
    assert sa[7] == 0.0 and sa[8] == 0.0  # camera {0, 0}: nothing has stepped since make (renderer.h:17)
    for _ in range(10):
        L.pgo_step(h, 7)  # walk right: the camera follows the agent
    cam = state(h)[7:9].copy()
    assert cam[0] > 0.0
    L.pgo_reset(h, 1, 77)
    b, sb, tb = np.ctypeslib.as_array(L.pgo_obs(h), shape=(12288,)).copy(), state(h), tiles(h)
    L.pgo_close(h)
    assert np.array_equal(ta, tb)                      # same seed → same level
    assert np.array_equal(sa[9:], sb[9:])              # … same themes, backdrop, entities
    assert np.array_equal(sb[7:9], cam)                # the camera is the one the last step left (D3)
    assert not np.array_equal(a, b)                    # so the two reset frames of one level differ
    assert np.array_equal(sa[:7], sb[:7])              # agent spawn state identical


def test_oracle_aabb_and_ecs_match_the_recorded_reference_outputs():
    """tests/golden/ref_aabb_ecs.npz (helpers.cpp / ecs.cpp of the reference, recorded by make_ref_fixtures.py) against
    the oracle's own pieces — in a fresh process, because both sides' entity sets keep their bucket arrays across
    clear(): the scripts only reproduce when run from the beginning, in order.  Travels to boxes without /root/reference."""
    import subprocess
    import sys
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
import ref_util
z = np.load(os.path.join(%r, "golden", "ref_aabb_ecs.npz"))
ora = ref_util.Side("pgo")
hit, ov = ora.collisions(z["a"], z["b"])
assert np.array_equal(hit, z["hit"]) and np.array_equal(ov.view(np.uint32), z["overlap"].view(np.uint32))
for k in range(int(z["n_scripts"])):
    ids, orders = ora.ecs_script(z["ops%%d" %% k], z["args%%d" %% k])
    assert np.array_equal(ids, z["ids%%d" %% k]), k
    assert np.array_equal(orders, z["orders%%d" %% k]), k
print("ok")
''' % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]
