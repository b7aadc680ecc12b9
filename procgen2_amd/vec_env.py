"""ProcgenVecEnv — N concurrent envs on one MI355X, tensors stay in HBM.

Counterpart of the reference's `CEnv` (cenv/cenv.py:152-380) for the vector extension
(include/procgen2_vec.h): same life cycle (make → reset → step … close), but observations,
rewards and dones are zero-copy torch tensors over the engine's output slab instead of per-call
numpy copies (SURVEY.md §8b "Consequence for a vector engine").

Multi-GPU: one process per GPU; rank r owns global envs [r*N, (r+1)*N) (seed = seed_base + global
index, so results do not depend on the GPU count).  There is no collective in reset/step/render;
`gather()` is the optional rooted gather of obs/reward/done to rank 0 over RCCL.
"""
import ctypes
import os
from ctypes import c_void_p

import torch

from . import lib as pglib

GAMES = ("coinrun", "maze", "bossfight", "climber", "caveflyer", "chaser", "jumper")


def shard_range(total_envs, world_size, rank):
    """Contiguous env-index block of `rank`: [lo, hi).  Earlier ranks take the remainder."""
    base, rem = divmod(total_envs, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ProcgenVecEnv:
    def __init__(self, game, num_envs, device=0, seed_base=1, env_offset=0, lib_path=None, num_levels=0,
                 start_level=0, distribution_mode=None, game_flags=0, out=None):
        """out = (obs, reward, done): caller-owned result tensors on `device` — uint8 [N,64,64,3], float32 [N],
        uint8 [N], contiguous — e.g. slices of one slab that several envs (the games of a mixed workload) fill side by
        side ("one contiguous [N_local,64,64,3] slab regardless of game").  Default: own tensors."""
        if not torch.cuda.is_available():
            raise pglib.EngineError("ProcgenVecEnv needs a HIP device (torch.cuda.is_available() is False); "
                                    "there is no CPU fallback")
        self.L = pglib.load(lib_path)
        self.game = game
        self.num_envs = int(num_envs)
        self.env_offset = int(env_offset)
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        # The engine gets a stream of its own (a torch stream, so torch can order against it): torch's default stream
        # has handle 0, which the C ABI reads as "create one", and work on an unrelated stream would race with the
        # caller's.  Every call below makes the engine's stream wait for the caller's current stream (actions, masks)
        # and the caller's current stream wait for the engine's (obs, reward, done).
        # (PG_STREAM_PRIORITY_<GAME>=-1: a high-priority stream for that game's engine — an A/B switch for several engines on
        # one GPU, bench.py --workload mixed; default 0)
        self._stream = torch.cuda.Stream(device=self.device, priority=int(os.environ.get("PG_STREAM_PRIORITY_" + game.upper(), "0")))
        # num_levels > 0: a finite level set (include/procgen2_vec.h pgv_make_levels); 0 = every level is new.
        # distribution_mode: None / "default" = the reference's compile-time config, or "easy" | "hard" | "memory" |
        # "extreme" where the game has it (pgv_game_modes).
        h = pglib.make(self.L, game, self.num_envs, device=device, seed_base=seed_base, env_offset=self.env_offset,
                       stream=c_void_p(self._stream.cuda_stream), num_levels=num_levels, start_level=start_level,
                       mode=distribution_mode, game_flags=game_flags)
        self.num_levels, self.start_level = int(num_levels), int(start_level)
        self.distribution_mode = {v: k for k, v in pglib.MODES.items()}[self.L.pgv_mode(h)]
        self._h = h
        # torch owns the result buffers; the engine writes straight into them.
        if out is not None:
            self.obs, self.reward, self.done = out
            want = (((self.num_envs, 64, 64, 3), torch.uint8), ((self.num_envs,), torch.float32),
                    ((self.num_envs,), torch.uint8))
            for t, (shape, dtype) in zip(out, want):
                if tuple(t.shape) != shape or t.dtype != dtype or t.device != self.device or not t.is_contiguous():
                    self.L.pgv_close(h)
                    self._h = None
                    raise ValueError("out: expected a contiguous %s tensor of shape %s on %s" % (dtype, shape, self.device))
        else:
            self.obs = torch.zeros((self.num_envs, 64, 64, 3), dtype=torch.uint8, device=self.device)
            self.reward = torch.zeros(self.num_envs, dtype=torch.float32, device=self.device)
            self.done = torch.zeros(self.num_envs, dtype=torch.uint8, device=self.device)
        pglib.check(self.L, self.L.pgv_bind_outputs(self._h, c_void_p(self.obs.data_ptr()),
                                                    c_void_p(self.reward.data_ptr()), c_void_p(self.done.data_ptr())),
                    "pgv_bind_outputs")
        self.single_observation_shape = (64, 64, 3)
        self.num_actions = pglib.NUM_ACTIONS

    # -- life cycle ------------------------------------------------------------------------------
    def reset(self, mask=None, seeds=None):
        """cenv_reset on every env (or those with mask != 0); seeds (int32[N]) reseed the streams."""
        m = None
        s = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        if seeds is not None:
            s = torch.as_tensor(seeds, device=self.device).to(torch.int32).contiguous()
        self._before()
        pglib.check(self.L, self.L.pgv_reset(self._h, c_void_p(m.data_ptr()) if m is not None else None,
                                             c_void_p(s.data_ptr()) if s is not None else None), "pgv_reset")
        self._after()
        self._keep = (m, s)
        return self.obs

    def step(self, actions):
        """actions: int32[N] tensor on this device (or anything torch.as_tensor accepts).
        Returns (obs u8[N,64,64,3], reward f32[N], done u8[N]) — views of the engine's buffers, valid
        until the next step; work is enqueued on torch's current stream."""
        a = torch.as_tensor(actions, device=self.device).to(torch.int32).contiguous()
        if a.numel() != self.num_envs:
            raise ValueError("expected %d actions, got %d" % (self.num_envs, a.numel()))
        self._before()
        pglib.check(self.L, self.L.pgv_step(self._h, c_void_p(a.data_ptr())), "pgv_step")
        self._after()
        self._keep = (a,)
        return self.obs, self.reward, self.done

    def step_synthetic(self, run_seed=0, ordered=True):
        """One step with device-generated actions.  ordered=False skips the stream hand-shake with the caller's current
        stream (several envs stepping side by side on their own streams; call sync() before reading the outputs)."""
        if ordered:
            self._before()
        pglib.check(self.L, self.L.pgv_step_synthetic(self._h, run_seed), "pgv_step_synthetic")
        if ordered:
            self._after()
        return self.obs, self.reward, self.done

    def _before(self):
        self._stream.wait_stream(torch.cuda.current_stream(self.device))

    def _after(self):
        torch.cuda.current_stream(self.device).wait_stream(self._stream)

    def timed_steps(self, steps, run_seed=0, render_events=True):
        """(total_ms, render_kernel_ms_sum) from HIP events on the engine's stream.  render_events=False: the region holds
        the steps and nothing else (no event pair per render launch); the second value is then None."""
        total, render = ctypes.c_double(), ctypes.c_double()
        pglib.check(self.L, self.L.pgv_timed_steps(self._h, steps, run_seed, ctypes.byref(total),
                                                   ctypes.byref(render) if render_events else None), "pgv_timed_steps")
        return total.value, (render.value if render_events else None)

    def step_times(self, steps, run_seed=0):
        """Per-step detail of `steps` synthetic steps (HIP events on the engine's stream): (step_ms[steps],
        render_ms[steps]) as numpy float32 arrays — for latency percentiles and the roofline window."""
        import numpy as np
        step_ms, render_ms = np.zeros(steps, np.float32), np.zeros(steps, np.float32)
        pglib.check(self.L, self.L.pgv_step_times(self._h, steps, run_seed, c_void_p(step_ms.ctypes.data),
                                                  c_void_p(render_ms.ctypes.data)), "pgv_step_times")
        return step_ms, render_ms

    def step_phases(self, steps, run_seed=0):
        """`steps` synthetic steps cut into their phases by HIP events on the engine's stream (include/procgen2_vec.h
        pgv_step_phases): a dict of numpy float32 arrays step / logic / prepass / render / late, milliseconds per step;
        logic + prepass + render + late = step."""
        import numpy as np
        names = ("step", "logic", "prepass", "render", "late")
        out = {k: np.zeros(steps, np.float32) for k in names}
        pglib.check(self.L, self.L.pgv_step_phases(self._h, steps, run_seed, *(c_void_p(out[k].ctypes.data) for k in names)),
                    "pgv_step_phases")
        return out

    def render_frame(self, index=0, width=512, height=512):
        """The human-size frame of env `index` (cenv_render, render_game(false)): uint8 [height, width, 3] on the host."""
        import numpy as np
        out = np.zeros((height, width, 3), np.uint8)
        pglib.check(self.L, self.L.pgv_render_frame(self._h, index, width, height, c_void_p(out.ctypes.data)),
                    "pgv_render_frame")
        return out

    def save_state(self):
        """Snapshot of the whole batch (state, RNG streams, prefetched levels, outputs) as a numpy byte array."""
        import numpy as np
        n = self.L.pgv_snapshot_bytes(self._h)
        buf = np.empty(n, np.uint8)
        pglib.check(self.L, self.L.pgv_save_state(self._h, c_void_p(buf.ctypes.data), n), "pgv_save_state")
        return buf

    def load_state(self, buf):
        import numpy as np
        buf = np.ascontiguousarray(buf, np.uint8)
        pglib.check(self.L, self.L.pgv_load_state(self._h, c_void_p(buf.ctypes.data), buf.size), "pgv_load_state")

    def sync(self):
        pglib.check(self.L, self.L.pgv_sync(self._h), "pgv_sync")

    def publish(self):
        """Order torch's current stream behind the engine's stream (after step_synthetic(ordered=False)), without
        blocking the host: whatever is enqueued on the current stream next sees the step's outputs.  This covers
        read-after-write only: the NEXT unordered step would overwrite obs / reward / done while a reader enqueued here
        (a gather, a copy) is still at them — call consume() after enqueueing the readers."""
        self._after()

    def consume(self):
        """The counterpart of publish(): order the engine's stream behind torch's current stream, so the next
        step_synthetic(ordered=False) does not overwrite the outputs before what was enqueued on the current stream
        (and on streams it has been made to wait for, like a collective's) has read them."""
        self._before()

    def close(self):
        if self._h:
            self.L.pgv_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- multi-GPU -------------------------------------------------------------------------------
    def gather(self, dst=0, group=None):
        """Rooted gather of (obs, reward, done) to rank `dst` (a rank of `group`); other ranks get None.  Optional: the
        hot path itself never communicates.  The plan (per-rank env counts, the root's slabs) is made on the first call
        and reused: a call is then one batch of point-to-point transfers and no host synchronisation."""
        key = (dst, id(group))
        plan = self._gathers.get(key) if hasattr(self, "_gathers") else None
        if plan is None:
            if not hasattr(self, "_gathers"):
                self._gathers = {}
            plan = self._gathers[key] = RootGather((self.obs, self.reward, self.done), dst=dst, group=group)
        return plan()


def step_many_synthetic(envs, steps, run_seed=0):
    """`steps` synthetic steps of several ProcgenVecEnv on one device side by side, each on its own stream, with no
    ordering against the caller's stream and no host work between the launches (include/procgen2_vec.h
    pgv_step_synthetic_many).  Call sync() on the envs before reading their outputs."""
    handles = (c_void_p * len(envs))(*[e._h for e in envs])
    pglib.check(envs[0].L, envs[0].L.pgv_step_synthetic_many(handles, len(envs), int(steps), int(run_seed)),
                "pgv_step_synthetic_many")


def step_phases_many(envs, steps, run_seed=0):
    """`steps` synthetic steps of several ProcgenVecEnv side by side (as step_many_synthetic), every step of every env cut
    into its phases by HIP events on that env's stream (pgv_step_phases_many): numpy float32 [len(envs), 5, steps] —
    step, logic, prepass, render, late, milliseconds."""
    import numpy as np
    out = np.zeros((len(envs), 5, int(steps)), np.float32)
    handles = (c_void_p * len(envs))(*[e._h for e in envs])
    pglib.check(envs[0].L, envs[0].L.pgv_step_phases_many(handles, len(envs), int(steps), int(run_seed), c_void_p(out.ctypes.data)),
                "pgv_step_phases_many")
    return out


class RootGather:
    """Rooted gather of a tuple of per-rank tensors `[n_r, ...]` into `[sum n_r, ...]` slabs on rank `dst` of `group`
    (SURVEY.md §8e).  Ranks may hold different env counts.

    Built once: the counts are exchanged once (one all_gather of an int per rank) and the root allocates one slab per
    tensor.  Every call is one `batch_isend_irecv`: the root receives each peer's block STRAIGHT into its slice of the
    slab (no temporaries, no concatenation) while copying its own block, every peer sends its tensors as they are.  On
    GPUs this is RCCL grouped send/recv, so the root's inbound xGMI links (one per peer, point to point) all run at
    once; the CPU tests run the same code over gloo.  The returned slabs are reused by the next call."""

    def __init__(self, tensors, dst=0, group=None, slabs=None):
        """slabs: the root's preallocated `[sum n_r, ...]` tensors (one per input).  When the root's own input IS its
        slice of the slab (the engine writes straight into it), its block is not copied."""
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.tensors = tuple(tensors)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)  # rank inside `group`, like dst
        self.dst = dst
        # P2P ops address peers by GLOBAL rank.
        self.global_rank = [dist.get_global_rank(group, r) if group is not None else r for r in range(self.world)]
        n = self.tensors[0].shape[0]
        counts = [None] * self.world
        dist.all_gather_object(counts, int(n), group=group)  # once, on the host: no device sync per call later
        self.counts = [int(c) for c in counts]
        self.offsets = [0]
        for c in self.counts:
            self.offsets.append(self.offsets[-1] + c)
        self.slabs = None
        if self.rank == dst:
            total = self.offsets[-1]
            if slabs is not None:
                self.slabs = tuple(slabs)
                for t, slab in zip(self.tensors, self.slabs):
                    if (tuple(slab.shape) != (total,) + tuple(t.shape[1:]) or slab.dtype != t.dtype or slab.device != t.device
                            or not slab.is_contiguous()):
                        raise ValueError("RootGather: a slab does not match [%d, ...] of its tensor (shape, dtype, device, "
                                         "contiguity)" % total)
            else:
                self.slabs = tuple(torch.empty((total,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
                                   for t in self.tensors)

    def __call__(self):
        dist = self.dist
        ops = []
        if self.rank == self.dst:
            for t, slab in zip(self.tensors, self.slabs):
                for r in range(self.world):
                    part = slab[self.offsets[r]:self.offsets[r + 1]]  # contiguous: a block of leading indices
                    if r == self.rank:
                        if part.data_ptr() != t.data_ptr():  # (already in place when the input is the slab's slice)
                            part.copy_(t)
                    elif self.counts[r]:
                        ops.append(dist.P2POp(dist.irecv, part, self.global_rank[r], group=self.group))
        elif self.counts[self.rank]:
            for t in self.tensors:
                ops.append(dist.P2POp(dist.isend, t.contiguous(), self.global_rank[self.dst], group=self.group))
        if ops:
            for q in dist.batch_isend_irecv(ops):
                q.wait()
        return self.slabs if self.rank == self.dst else tuple(None for _ in self.tensors)


def gather_outputs(obs, reward, done, dst=0, group=None):
    """One-shot form of RootGather (builds the plan, runs it once)."""
    return RootGather((obs, reward, done), dst=dst, group=group)()
