"""procgen2_amd.gym_vector: the Gymnasium VectorEnv-shaped wrapper, exercised on the CPU over an
oracle-backed stand-in engine (the wrapper only needs reset/step/close/num_envs), and on the GPU over the real one."""
import ctypes

import numpy as np
import pytest

import oracle_util
from procgen2_amd.gym_vector import GymVectorAdapter, NUM_ACTIONS, make_spaces


class OracleEngine:
    """The engine's contract on the CPU: N single-env oracle objects, next-step auto-reset, reseeding masked resets."""

    def __init__(self, game, n, seed_base=1):
        oracle_util.register_textures(game)
        self.L = oracle_util.oracle()
        self.game = game
        self.num_envs = n
        self.h = [self.L.pgo_make(game.encode(), seed_base + i, 1) for i in range(n)]
        self.pending = [False] * n
        self.obs = np.zeros((n, 64, 64, 3), np.uint8)
        self.reward = np.zeros(n, np.float32)
        self.done = np.zeros(n, np.uint8)

    def _grab(self, i):
        self.obs[i] = np.ctypeslib.as_array(self.L.pgo_obs(self.h[i]), shape=(12288,)).reshape(64, 64, 3)

    def reset(self, mask=None, seeds=None):
        for i, h in enumerate(self.h):
            if mask is not None and not mask[i]:
                continue
            self.L.pgo_reset(h, 0 if seeds is None else 1, 0 if seeds is None else int(seeds[i]))
            self.pending[i] = False
            self.reward[i] = 0.0
            self.done[i] = 0
            self._grab(i)
        return self.obs

    def step(self, actions):
        actions = np.asarray(actions)
        for i, h in enumerate(self.h):
            if self.pending[i]:
                self.L.pgo_reset(h, 0, 0)
                self.pending[i] = False
                self.reward[i], self.done[i] = 0.0, 0
            else:
                self.L.pgo_step(h, int(actions[i]))
                self.reward[i] = self.L.pgo_reward(h)
                self.done[i] = 1 if self.L.pgo_terminated(h) else 0
                self.pending[i] = bool(self.done[i])
            self._grab(i)
        return self.obs, self.reward, self.done

    def render_frame(self, index=0, width=512, height=512):
        out = np.zeros((height, width, 3), np.uint8)
        self.L.pgo_render_frame(self.h[index], width, height, out.ctypes.data_as(ctypes.c_void_p))
        return out

    def close(self):
        for h in self.h:
            self.L.pgo_close(h)
        self.h = []


def test_spaces_have_the_vector_env_shapes():
    single_obs, single_act, obs, act = make_spaces(5)
    assert single_obs.shape == (64, 64, 3) and single_obs.dtype == np.uint8
    assert int(single_obs.low.min()) == 0 and int(single_obs.high.max()) == 255
    assert single_act.n == NUM_ACTIONS
    assert obs.shape == (5, 64, 64, 3)
    assert act.shape == (5,) and int(np.asarray(act.nvec)[0]) == NUM_ACTIONS
    assert single_act.contains(single_act.sample())
    assert act.contains(np.asarray(act.sample()))


def test_wrapper_over_the_oracle_engine_reset_step_autoreset():
    n = 6
    env = GymVectorAdapter(OracleEngine("maze", n), output="numpy")
    assert env.num_envs == n and env.observation_space.shape == (n, 64, 64, 3)
    obs, info = env.reset(seed=40)
    assert obs.shape == (n, 64, 64, 3) and obs.dtype == np.uint8 and info == {}
    first = obs.copy()
    rng = np.random.default_rng(0)
    ended_at = {}
    for s in range(520):  # maze episodes end by step 500 at the latest (timeout counts as terminated, D5)
        obs, reward, terminated, truncated, info = env.step(rng.integers(0, NUM_ACTIONS, n))
        assert reward.shape == (n,) and terminated.dtype == bool and not truncated.any()
        for i in np.nonzero(terminated)[0]:
            ended_at.setdefault(int(i), s)
        for i, t in ended_at.items():
            if s == t + 1:  # NEXT_STEP autoreset: reward 0, not terminated, a fresh level
                assert reward[i] == 0.0 and not terminated[i]
    assert len(ended_at) == n
    # same seeds → same first frames; scalar seed = seed + i, explicit list accepted
    again, _ = env.reset(seed=40)
    assert np.array_equal(again, first)
    listed, _ = env.reset(seed=[40 + i for i in range(n)])
    assert np.array_equal(listed, first)
    # masked reset leaves the other envs' observations alone
    obs, *_ = env.step(np.zeros(n, np.int64))
    before = obs.copy()
    mask = np.array([1, 0, 0, 1, 0, 0], bool)
    after, _ = env.reset(seed=7, options={"reset_mask": mask})
    assert np.array_equal(after[~mask], before[~mask])
    with pytest.raises(ValueError):
        env.step(np.zeros(n + 1, np.int64))
    with pytest.raises(ValueError):
        env.reset(options={"nonsense": 1})
    env.close()
    env.close()  # idempotent


def test_same_step_autoreset_walks_the_same_episodes_as_next_step():
    """SAME_STEP (SURVEY.md §8f-1 `final_observation`): the terminal step returns the next episode's first frame, the
    terminal frame travels in info["final_obs"]; the frames an env shows are NEXT_STEP's with the reset step taken out."""
    n, steps = 5, 520
    rng = np.random.default_rng(4)
    actions = rng.integers(0, NUM_ACTIONS, (steps, n))
    same = GymVectorAdapter(OracleEngine("maze", n), output="numpy", autoreset_mode="same_step")
    nxt = GymVectorAdapter(OracleEngine("maze", n), output="numpy")
    o_s, _ = same.reset(seed=11)
    o_n, _ = nxt.reset(seed=11)
    assert np.array_equal(o_s, o_n)
    ends = 0
    for s in range(200):
        obs, rew, term, trunc, info = same.step(actions[s])
        if term.any():
            assert set(info) == {"final_obs", "_final_obs", "final_obs_compact", "final_obs_env"}
            assert np.array_equal(info["final_obs_env"], np.nonzero(term)[0]) and np.array_equal(info["_final_obs"], term)
            assert info["final_obs_compact"].shape == (int(term.sum()), 64, 64, 3)
            # Gymnasium's layout: indexed by ENV — an object array with None for the envs that go on
            assert info["final_obs"].shape == (n,) and info["final_obs"].dtype == object
            for e in range(n):
                if info["_final_obs"][e]:
                    k = int(np.searchsorted(info["final_obs_env"], e))
                    assert info["final_obs"][e].shape == (64, 64, 3)
                    assert np.array_equal(info["final_obs"][e], info["final_obs_compact"][k])
                else:
                    assert info["final_obs"][e] is None
            ends += int(term.sum())
        else:
            assert info == {}
    assert ends > 0
    same.close()
    nxt.close()
    # one env, exact comparison: SAME_STEP's stream == NEXT_STEP's with the reset steps (which take any action) removed
    a = GymVectorAdapter(OracleEngine("maze", 1, seed_base=3), output="numpy", autoreset_mode="same_step")
    b = GymVectorAdapter(OracleEngine("maze", 1, seed_base=3), output="numpy")
    a.reset()
    b.reset()
    for s in range(steps):
        oa, ra, ta, _, ia = a.step(actions[s, :1])
        ob, rb, tb, _, _ = b.step(actions[s, :1])
        assert ra[0] == rb[0] and ta[0] == tb[0]
        if ta[0]:
            assert np.array_equal(ia["final_obs"][0], ob[0])  # the terminal frame
            ob, rb, tb, _, _ = b.step(np.zeros(1, np.int64))  # NEXT_STEP spends a step on the reset
            assert rb[0] == 0.0 and not tb[0]
        assert np.array_equal(oa, ob), s
    a.close()
    b.close()
    with pytest.raises(ValueError):
        GymVectorAdapter(OracleEngine("maze", 1), autoreset_mode="sometimes")


@pytest.mark.gpu
def test_same_step_autoreset_on_the_hip_engine_matches_the_oracle_engine():
    import torch
    from procgen2_amd.gym_vector import ProcgenGymVectorEnv
    n = 16
    gpu = ProcgenGymVectorEnv("bossfight", n, seed=1, autoreset_mode="same_step")  # short episodes, in-step random draws
    cpu = GymVectorAdapter(OracleEngine("bossfight", n), output="numpy", autoreset_mode="same_step")
    assert np.array_equal(gpu.reset(seed=5)[0].cpu().numpy(), cpu.reset(seed=5)[0])
    rng = np.random.default_rng(9)
    ends = 0
    kept = []  # (the info's tensor as handed out, a copy of what it held then): ownership passes to the caller
    for s in range(300):
        a = rng.integers(0, NUM_ACTIONS, n)
        og, rg, tg, _, ig = gpu.step(torch.as_tensor(a, dtype=torch.int32, device="cuda"))
        oc, rc, tc, _, ic = cpu.step(a)
        assert np.array_equal(og.cpu().numpy(), oc), s
        assert np.array_equal(rg.cpu().numpy(), rc) and np.array_equal(tg.cpu().numpy(), tc)
        assert set(ig) == set(ic)
        if ic:
            kept.append((ig["final_obs"], ig["final_obs"].cpu().numpy().copy(), ic["_final_obs"].copy()))
            # rows of envs that did not end this step are zero, not an older episode's frame
            assert not ig["final_obs"][torch.as_tensor(~ic["_final_obs"], device="cuda")].any()
            assert np.array_equal(ig["final_obs_compact"].cpu().numpy(), ic["final_obs_compact"])
            assert np.array_equal(ig["final_obs_env"].cpu().numpy(), ic["final_obs_env"])
            # indexed by env id, as code written against Gymnasium reads it: infos["final_obs"][i] where infos["_final_obs"][i]
            assert tuple(ig["final_obs"].shape) == (n, 64, 64, 3) and ig["final_obs"].is_cuda
            for e in np.nonzero(ic["_final_obs"])[0]:
                assert bool(ig["_final_obs"][e])
                assert np.array_equal(ig["final_obs"][e].cpu().numpy(), ic["final_obs"][e]), (s, e)
            ends += len(ic["final_obs_env"])
    assert ends > 10
    # an info kept from step t still holds step t's terminal frames after all the later steps (r04 advisor finding: one
    # persistent buffer was handed out again and again)
    assert len(kept) > 5
    for tensor, then, _ in kept:
        assert np.array_equal(tensor.cpu().numpy(), then)
    gpu.close()
    cpu.close()


@pytest.mark.gpu
def test_wrapper_over_the_hip_engine_matches_the_oracle_engine():
    import torch
    from procgen2_amd.gym_vector import ProcgenGymVectorEnv
    n = 8
    gpu = ProcgenGymVectorEnv("coinrun", n, seed=1)
    cpu = GymVectorAdapter(OracleEngine("coinrun", n), output="numpy")
    o_g, _ = gpu.reset(seed=900)
    o_c, _ = cpu.reset(seed=900)
    assert isinstance(o_g, torch.Tensor) and o_g.is_cuda and tuple(o_g.shape) == (n, 64, 64, 3)
    assert np.array_equal(o_g.cpu().numpy(), o_c)
    rng = np.random.default_rng(3)
    for s in range(120):
        a = rng.integers(0, NUM_ACTIONS, n)
        og, rg, tg, ug, _ = gpu.step(torch.as_tensor(a, dtype=torch.int32, device="cuda"))
        oc, rc, tc, uc, _ = cpu.step(a)
        assert np.array_equal(og.cpu().numpy(), oc), s
        assert np.array_equal(rg.cpu().numpy(), rc) and np.array_equal(tg.cpu().numpy(), tc)
        assert not bool(ug.any())
    gpu.close()
    cpu.close()


def test_render_returns_the_human_frame_only_in_rgb_array_mode():
    env = GymVectorAdapter(OracleEngine("maze", 2), output="numpy")
    env.reset()
    assert env.render() is None
    env.close()
    env = GymVectorAdapter(OracleEngine("maze", 2), output="numpy", render_mode="rgb_array", render_size=(96, 80))
    obs, _ = env.reset()
    frame = env.render(index=1)
    assert frame.shape == (80, 96, 3) and frame.dtype == np.uint8 and frame.any()
    small = GymVectorAdapter(OracleEngine("maze", 2), output="numpy", render_mode="rgb_array", render_size=(64, 64))
    obs2, _ = small.reset()
    assert np.array_equal(small.render(index=0), obs2[0])  # at 64×64 the human frame is the observation
    with pytest.raises(ValueError):
        GymVectorAdapter(OracleEngine("maze", 1), render_mode="human")
    env.close()
    small.close()
