"""Gymnasium `VectorEnv`-shaped front end of the engine.

The reference's wrapper (`cenv/cenv.py:152-380`) is a single `gym.Env` whose `observation_space` is a dict of 1-element
Boxes; what RL code imports is a vector env with real spaces.  This class gives the engine that surface:

    env = ProcgenGymVectorEnv("coinrun", num_envs=4096)          # needs a HIP device
    obs, info = env.reset(seed=123)                               # uint8 [N, 64, 64, 3]
    obs, reward, terminated, truncated, info = env.step(actions)  # actions: int array/tensor [N]

* `single_observation_space = Box(0, 255, (64, 64, 3), uint8)`, `single_action_space = Discrete(15)` and their batched
  forms — real `gymnasium.spaces` objects when gymnasium is importable, otherwise light stand-ins with the same
  attributes (`shape`, `dtype`, `low`, `high`, `n`, `contains`, `sample`); nothing else of gymnasium is needed.
* Autoreset is Gymnasium 1.x's NEXT_STEP mode by default, which is exactly the engine's policy: the step
  after a terminal one ignores the action, returns the first observation of the new episode with reward 0, terminated
  False.  `autoreset_mode="same_step"` gives Gymnasium's SAME_STEP mode instead: the terminal step itself returns the
  first observation of the next episode, and the terminal observation travels in `info["final_obs"]` (see step()).  It
  is the reference caller's own loop — `if terminated: obs, info = env.reset()` (`game_test.py:36-40`) — done for the
  envs that ended, with their random streams continuing, so both modes walk through the same levels.
  `truncated` is always False (the reference never truncates: `coinrun.cpp:367`).
* `reset(seed=s)` reseeds env i with `s + i` (a list/array gives one seed per env); `reset(options={"reset_mask": m})`
  resets only the envs where `m` is true, as Gymnasium's vector API allows.
* Outputs are views of the engine's buffers: torch tensors on the device by default (zero copy), numpy arrays with
  `output="numpy"` (one device→host copy per step).

The class only talks to an *engine* object with `reset(mask, seeds) -> obs`, `step(actions) -> (obs, reward, done)`,
`close()`, `num_envs`; `ProcgenVecEnv` is the real one.  The CPU tests drive the same code with an oracle-backed
stand-in, so the wrapper logic is covered without a GPU.
"""
import numpy as np

NUM_ACTIONS = 15
OBS_SHAPE = (64, 64, 3)

try:  # optional
    import gymnasium as _gym
    from gymnasium import spaces as _spaces
    _VectorBase = _gym.vector.VectorEnv
except Exception:  # gymnasium absent: same attribute surface, no dependency
    _gym = None
    _spaces = None
    _VectorBase = object


class _Box:
    def __init__(self, low, high, shape, dtype):
        self.low = np.full(shape, low, dtype=dtype)
        self.high = np.full(shape, high, dtype=dtype)
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self._rng = np.random.default_rng()

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and x.dtype == self.dtype

    def sample(self):
        return self._rng.integers(0, 256, self.shape, dtype=np.int64).astype(self.dtype)

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def __repr__(self):
        return "Box(%s, %s, %s, %s)" % (self.low.flat[0], self.high.flat[0], self.shape, self.dtype)


class _Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype(np.int64)
        self._rng = np.random.default_rng()

    def contains(self, x):
        return 0 <= int(x) < self.n

    def sample(self):
        return int(self._rng.integers(0, self.n))

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def __repr__(self):
        return "Discrete(%d)" % self.n


class _MultiDiscrete:
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape = self.nvec.shape
        self.dtype = np.dtype(np.int64)
        self._rng = np.random.default_rng()

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(((x >= 0) & (x < self.nvec)).all())

    def sample(self):
        return self._rng.integers(0, self.nvec)

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def __repr__(self):
        return "MultiDiscrete(%d x %d)" % (self.nvec.size, int(self.nvec.flat[0]) if self.nvec.size else 0)


def make_spaces(num_envs):
    """(single_observation, single_action, observation, action) spaces."""
    if _spaces is not None:
        single_obs = _spaces.Box(0, 255, OBS_SHAPE, np.uint8)
        single_act = _spaces.Discrete(NUM_ACTIONS)
        obs = _spaces.Box(0, 255, (num_envs,) + OBS_SHAPE, np.uint8)
        act = _spaces.MultiDiscrete(np.full(num_envs, NUM_ACTIONS, dtype=np.int64))
    else:
        single_obs = _Box(0, 255, OBS_SHAPE, np.uint8)
        single_act = _Discrete(NUM_ACTIONS)
        obs = _Box(0, 255, (num_envs,) + OBS_SHAPE, np.uint8)
        act = _MultiDiscrete(np.full(num_envs, NUM_ACTIONS, dtype=np.int64))
    return single_obs, single_act, obs, act


class GymVectorAdapter(_VectorBase):
    """The wrapper logic over any engine object (see module docstring)."""

    metadata = {"render_modes": ["rgb_array"], "autoreset_mode": "next_step"}

    def __init__(self, engine, output="torch", render_mode=None, render_size=(512, 512), autoreset_mode="next_step"):
        if output not in ("torch", "numpy"):
            raise ValueError("output must be 'torch' or 'numpy'")
        if autoreset_mode not in ("next_step", "same_step"):
            raise ValueError("autoreset_mode must be 'next_step' or 'same_step'")
        self.autoreset_mode = autoreset_mode
        if render_mode not in (None, "rgb_array"):
            raise ValueError("render_mode must be None or 'rgb_array'")
        self.engine = engine
        self.num_envs = int(engine.num_envs)
        self.output = output
        self.render_size = (int(render_size[0]), int(render_size[1]))
        (self.single_observation_space, self.single_action_space, self.observation_space,
         self.action_space) = make_spaces(self.num_envs)
        self.render_mode = render_mode
        self.closed = False
        self.metadata = dict(self.metadata, autoreset_mode=autoreset_mode)
        if _gym is not None:
            try:
                modes = _gym.vector.AutoresetMode
                self.metadata["autoreset_mode"] = modes.SAME_STEP if autoreset_mode == "same_step" else modes.NEXT_STEP
            except Exception:
                pass

    # -- helpers ---------------------------------------------------------------------------------
    def _out(self, x, dtype=None):
        if self.output == "numpy":
            x = x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)
            return x.astype(dtype, copy=False) if dtype is not None else x
        return x

    def _seeds(self, seed):
        if seed is None:
            return None
        if np.isscalar(seed):
            base = int(seed)
            seeds = (np.arange(self.num_envs, dtype=np.int64) + base)
        else:
            seeds = np.asarray(list(seed), dtype=np.int64)
            if seeds.shape != (self.num_envs,):
                raise ValueError("expected %d seeds, got shape %s" % (self.num_envs, seeds.shape))
        # the engine takes int32 and seeds mt19937 with the value mod 2^32, as `rng.seed(options[i].value.i)` does
        return (seeds & 0xFFFFFFFF).astype(np.uint32).view(np.int32)

    # -- Gymnasium vector API ----------------------------------------------------------------------
    def reset(self, *, seed=None, options=None):
        mask = None
        if options:
            unknown = set(options) - {"reset_mask"}
            if unknown:
                raise ValueError("unknown reset option(s): %s" % ", ".join(sorted(unknown)))
            if options.get("reset_mask") is not None:
                mask = np.asarray(options["reset_mask"]).astype(np.uint8)
                if mask.shape != (self.num_envs,):
                    raise ValueError("reset_mask must have shape (%d,)" % self.num_envs)
        obs = self.engine.reset(mask=mask, seeds=self._seeds(seed))
        return self._out(obs), {}

    def step(self, actions):
        n = actions.numel() if hasattr(actions, "numel") else np.asarray(actions).size
        if n != self.num_envs:
            raise ValueError("expected %d actions, got %d" % (self.num_envs, n))
        obs, reward, done = self.engine.step(actions)
        terminated = done != 0
        truncated = np.zeros(self.num_envs, dtype=bool) if self.output == "numpy" else (done != done)
        info = {}
        if self.autoreset_mode == "same_step" and bool(terminated.any()):
            # SAME_STEP: keep the terminal frames, then reset exactly the envs that ended (streams continue), which
            # also takes them off the engine's own next-step reset.  reward / terminated stay the terminal step's.
            # Gymnasium's layout — both keys are indexed by ENV:
            #   info["_final_obs"]  bool [num_envs]: which envs ended;
            #   info["final_obs"]   numpy output: object array [num_envs], the env's terminal frame or None;
            #                       torch output: uint8 [num_envs, 64, 64, 3] on the device, valid where _final_obs is
            #                       true, zero elsewhere — a FRESH tensor every step (Gymnasium hands out fresh data: an
            #                       info kept from step t is not touched by step t + 1, so a replay buffer may keep it).
            # Beside them, for callers that want the batch of terminal frames without the holes:
            #   info["final_obs_compact"] [k, 64, 64, 3] in env order, info["final_obs_env"] [k] their indices.
            if hasattr(terminated, "nonzero") and not isinstance(terminated, np.ndarray):  # torch
                mask, rew = terminated.clone(), reward.clone()
                where = mask.nonzero().flatten()
                final = obs[where].clone()
            else:
                mask, rew = np.array(terminated, dtype=bool), np.array(reward)
                where = np.nonzero(mask)[0]
                final = np.array(obs[where])
            if self.output == "numpy":
                final_np, where_np = self._out(final), self._out(where)
                by_env = np.full(self.num_envs, None, dtype=object)
                for k, e in enumerate(where_np):
                    by_env[int(e)] = final_np[k]
            else:
                by_env = obs.new_zeros(obs.shape) if hasattr(obs, "new_zeros") else np.zeros_like(obs)
                by_env[where] = final
            obs = self.engine.reset(mask=mask, seeds=None)  # (clears reward and done of those envs in the engine)
            reward, terminated = rew, mask
            info = {"final_obs": by_env, "_final_obs": self._out(mask, bool), "final_obs_compact": self._out(final),
                    "final_obs_env": self._out(where)}
        return self._out(obs), self._out(reward), self._out(terminated, bool), truncated, info

    def render(self, index=0):
        """The human-size frame of one env (the reference's `cenv_render`, coinrun.cpp:393-411; default 512×512 as its
        window): uint8 [H, W, 3].  None unless the env was made with render_mode="rgb_array"."""
        if self.render_mode != "rgb_array":
            return None
        w, h = self.render_size
        return self.engine.render_frame(int(index), w, h)

    def close(self, **kwargs):
        if not self.closed:
            self.engine.close()
            self.closed = True

    def close_extras(self, **kwargs):  # gymnasium.vector.VectorEnv.close() calls this
        if not self.closed:
            self.engine.close()
            self.closed = True

    @property
    def unwrapped(self):
        return self

    def __repr__(self):
        return "%s(%s, num_envs=%d)" % (type(self).__name__, getattr(self.engine, "game", "?"), self.num_envs)


class ProcgenGymVectorEnv(GymVectorAdapter):
    """`GymVectorAdapter` over the HIP engine.  Raises if there is no HIP device (no CPU fallback)."""

    def __init__(self, game, num_envs, device=0, seed=1, env_offset=0, output="torch", num_levels=0, start_level=0,
                 distribution_mode=None, render_mode=None, render_size=(512, 512), autoreset_mode="next_step"):
        from .vec_env import ProcgenVecEnv
        super().__init__(ProcgenVecEnv(game, num_envs, device=device, seed_base=seed, env_offset=env_offset,
                                       num_levels=num_levels, start_level=start_level,
                                       distribution_mode=distribution_mode), output=output, render_mode=render_mode,
                         render_size=render_size, autoreset_mode=autoreset_mode)
        self.game = game
