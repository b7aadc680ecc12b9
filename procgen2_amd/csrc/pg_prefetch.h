// Level prefetch: taking procedural level generation off the step's critical path.
//
// In the games without in-episode RNG draws (coinrun, maze, caveflyer, climber, jumper) the next level of an env is
// a function of its generator state alone (the mt19937 stream and the hashtable bucket counts that survive
// clear()), so it can be produced any time after the current level — not only at the moment the episode ends.
// Generation is latency-bound serial work (libstdc++ hashtable replay, BFS, Kruskal: milliseconds for ONE env, but
// ~1 µs per env in bulk), which a step that must wait for it cannot hide, while an episode lasts hundreds to
// thousands of steps.  So every env owns a shadow slot holding its next level:
//
//   * a generator kernel on a side stream fills the shadow slots of envs whose slot is kQueued;
//   * the step's install kernel (main stream) copies a kReady slot into the live state when the env resets and
//     queues the slot again — a few KB of coalesced copies instead of the generation;
//   * if the slot is not ready yet (an episode shorter than the generator's latency) the install kernel either
//     takes the job itself (kQueued → kSync, synchronous generation as before) or, when a side-stream wave is
//     already on it (kBusy), waits for that wave.  Results are identical on every path: one generator function,
//     one per-env chain state, and the slot word guarantees at most one generator per env at a time.
//
// The slot word is the only cross-stream synchronisation: agent-scope acquire/release atomics (the two kernels may
// run concurrently on different XCDs, whose L2s are not coherent with each other without them).
#pragma once

#include "pg_defs.h"
#include "pg_engine.h"

namespace pg {

enum SlotState : int32_t {
    kSlotIdle = 0,    // nothing requested (prefetch off, or before the first level)
    kSlotQueued = 1,  // next level wanted, nobody working on it
    kSlotBusy = 2,    // a side-stream wave is generating it
    kSlotReady = 3,   // shadow slot holds the next level
    kSlotSync = 4,    // the main stream generates synchronously; side-stream generators keep out
};

#if defined(__HIPCC__)
PG_D int32_t slot_load(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }
PG_D void slot_store(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
PG_D bool slot_cas(int32_t* p, int32_t expect, int32_t want) {
    return __hip_atomic_compare_exchange_strong(p, &expect, want, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE,
                                                __HIP_MEMORY_SCOPE_AGENT);
}

// Called by lane 0 of a main-stream wave that needs env's next level now.  Returns kSlotReady (copy the shadow
// slot) or kSlotSync (generate it yourself; the slot is yours).  A kBusy generator is a resident wave that needs
// nothing from the caller, so waiting for it cannot deadlock; the spin is bounded anyway and falls back to a trap.
PG_D int32_t slot_acquire_for_install(int32_t* p) {
    for (long spin = 0; spin < (1L << 22); spin++) {  // ≈ seconds
        const int32_t st = slot_load(p);
        if (st == kSlotReady) return kSlotReady;
        if (st == kSlotQueued || st == kSlotIdle) {
            if (slot_cas(p, st, kSlotSync)) return kSlotSync;
            continue;
        }
        __builtin_amdgcn_s_sleep(32);  // kSlotBusy
    }
    __builtin_trap();
    return kSlotSync;
}

// The level kernel shared by the prefetching games.  G supplies
//     State  (members: int n; Level* shadow; int32_t* slot),  Level (POD, size a multiple of 4),  GenLds (scratch),
//     generate(s, env, L, lv, reseed, seed, lane)  — one wavefront advances env's generator chain, level into lv (LDS)
//     install(s, env, lv, lane)                     — lv becomes env's live state
//     fresh_chain(s, env), fresh_live(s, env)       — what cenv_make leaves in the generator chain besides the RNG
//                                                     (hashtable bucket counts) / in the live state (camera, pools):
//                                                     level-seed mode rebuilds every level from that (LevelPlan).
// A block serves the envs [blockIdx·span, +span) that need a level, one after the other:
//   mode 0  cenv_make: seed = seed_base + env index, level 0 generated synchronously;
//   mode 1  explicit reset (mask, optional seeds; a seed restarts the env's generator chain);
//   mode 2  auto-reset of the envs whose previous step terminated (StepIO::pending 1 → 2 tells the logic kernel);
//   mode 3  side stream: fill the shadow slots that are kSlotQueued.
// `prefetch` = whether a served env queues its next level.
// Optional hook: G::served(s, env) — called by lane 0 after an auto-reset (mode 2) has given env its level.
template <class G, class = void>
struct HasServed {
    static constexpr bool value = false;
};
template <class G>
struct HasServed<G, decltype(static_cast<void>(&G::served))> {
    static constexpr bool value = true;
};

// `due_mark` / `served_mark`: what mode 2 looks for in StepIO::pending and what it leaves there for an env it has served —
// 1 and 2, or reset_due_mark(step) / reset_served_mark(step) for the games whose logic kernel runs the prefetched installs
// in its own grid (install_prefetched below).
// The body is a device function (one wavefront of 64 lanes serving the envs [base, base + span)) so that a game can also
// run mode 2 as a row of blocks of one of its own kernels (coinrun: resolve_kernel's blockIdx.y == 1).
template <class G>
PG_D void level_serve(const typename G::State& s, int mode, int span, int prefetch, uint32_t seed_base, int env_offset,
                      const uint8_t* mask, const int32_t* seeds, StepIO io, LevelPlan plan, int served_mark, int due_mark,
                      int base, int lane) {
    using Level = typename G::Level;
    bool want = false;
    if (lane < span && base + lane < s.n) {
        const int e = base + lane;
        if (mode == 0)
            want = true;
        else if (mode == 1)
            want = !mask || mask[e];
        else if (mode == 2)
            want = io.pending[e] == due_mark;
        else
            want = __hip_atomic_load(&s.slot[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kSlotQueued;
    }
    unsigned long long todo = __ballot(want);
    if (!todo) return;
    __shared__ typename G::GenLds L;
    __shared__ Level lv;
    __shared__ int32_t verdict;
    __shared__ uint32_t numbered;  // level-seed mode: the level number this env builds next
    const bool levels = plan.num_levels > 0;
    // lane 0: (re)start the env's level sequence if asked, and in level-seed mode draw its next level number
    auto next_level = [&](int env, bool restart, uint32_t chain_seed) {
        if (lane == 0) {
            if (restart) {
                plan.chain_seed[env] = chain_seed;
                plan.drawn[env] = 0;
            }
            if (levels) {
                const uint32_t k = plan.drawn[env];
                plan.drawn[env] = k + 1;
                numbered = level_number(plan.num_levels, plan.start_level, plan.chain_seed[env], k);
                G::fresh_chain(s, env);
            }
        }
        __syncthreads();
    };
    constexpr int kWords = static_cast<int>(sizeof(Level) / 4);
    static_assert(sizeof(Level) % 4 == 0, "Level is copied as 32-bit words");
    while (todo) {
        const int env = base + __builtin_ctzll(todo);
        todo &= todo - 1;
        uint32_t* shadow = reinterpret_cast<uint32_t*>(&s.shadow[env]);
        uint32_t* local = reinterpret_cast<uint32_t*>(&lv);
        if (mode == 3) {
            if (lane == 0) verdict = slot_cas(&s.slot[env], kSlotQueued, kSlotBusy) ? 1 : 0;
            __syncthreads();
            const bool mine = verdict != 0;
            __syncthreads();
            if (!mine) continue;
            next_level(env, false, 0u);
            G::generate(s, env, L, lv, levels, numbered, lane);
            __syncthreads();
            for (int k = lane; k < kWords; k += 64) shadow[k] = local[k];
            __threadfence();
            __syncthreads();
            if (lane == 0) slot_store(&s.slot[env], kSlotReady);
            continue;
        }
        const bool reseed = mode == 0 || (mode == 1 && seeds != nullptr);
        if (lane == 0) {
            if (mode == 0) {
                verdict = kSlotSync;
            } else {
                int32_t got = slot_acquire_for_install(&s.slot[env]);
                if (got == kSlotReady && reseed) {  // the prepared level belongs to the abandoned chain
                    slot_store(&s.slot[env], kSlotSync);
                    got = kSlotSync;
                }
                verdict = got;
            }
        }
        __syncthreads();
        const int32_t how = verdict;
        __threadfence();
        if (how == kSlotReady) {
            for (int k = lane; k < kWords; k += 64) local[k] = shadow[k];
        } else {
            const uint32_t seed = mode == 0 ? seed_base + static_cast<uint32_t>(env_offset + env)
                                            : (seeds ? static_cast<uint32_t>(seeds[env]) : 0u);
            next_level(env, reseed, seed);
            G::generate(s, env, L, lv, reseed || levels, levels ? numbered : seed, lane);
        }
        __syncthreads();
        if (levels && lane == 0) G::fresh_live(s, env);
        __syncthreads();
        G::install(s, env, lv, lane);
        __threadfence();
        __syncthreads();
        if (lane == 0) {
            slot_store(&s.slot[env], prefetch ? kSlotQueued : kSlotIdle);
            if (mode != 0) {
                io.reward[env] = 0.0f;
                io.done[env] = 0;
                io.pending[env] = mode == 2 ? static_cast<uint8_t>(served_mark) : 0;
                if constexpr (HasServed<G>::value) {
                    if (mode == 2) G::served(s, env);
                }
            }
        }
        __syncthreads();
    }
}

template <class G>
__global__ void __launch_bounds__(64) level_kernel(typename G::State s, int mode, int span, int prefetch,
                                                   uint32_t seed_base, int env_offset, const uint8_t* mask,
                                                   const int32_t* seeds, StepIO io, LevelPlan plan, int served_mark, int due_mark) {
    // A level is one long dependent chain per wavefront; whatever shares its SIMD (a logic kernel running beside an
    // in-step reset, a render kernel beside the prefetcher) is many short ones: let the chain issue first.
    __builtin_amdgcn_s_setprio(3);
    level_serve<G>(s, mode, span, prefetch, seed_base, env_offset, mask, seeds, io, plan, served_mark, due_mark,
                   static_cast<int>(blockIdx.x) * span, static_cast<int>(threadIdx.x));
}

// The copy-only part of an auto-reset (mode 2 above, slot kSlotReady) as a device function, for a game that runs it
// INSIDE the grid of its first logic kernel instead of as a launch in front of it (round 5, coinrun: blockIdx.y == 1 of
// agent_kernel).  An env that resets does not step, so its install and the other envs' agents have nothing to wait for
// in each other; as a launch of its own the install was 26 µs of latency chain — pending byte, slot word, shadow level,
// flags, the stores, a fence — in front of every step, for the sixty-odd envs of 65 536 that reset in it.  What is left
// to the level kernel, launched behind (mode 2, unchanged): the envs whose slot is NOT ready (an episode shorter than the
// generator's latency) — they keep pending == 1 and are generated synchronously there, as before; in steady state that
// launch finds nothing (one load per lane).  The generator's 256 registers stay out of the logic kernel this way
// (the fused kernel WITH the generator ran at one wave per SIMD and took the sum of the two chains).
// One wavefront of 64 lanes (the copies stride by 64); `lv` is a Level in LDS.  Served envs are left at pending == served_mark, as mode 2 leaves them.
//
// The pending byte when install and logic share a launch — a game whose ONE logic kernel both reads the byte (is this
// step the env's reset?) and writes it (the env terminated: reset it next step).  Two races to keep out:
//   * a logic lane may look at its env's byte before, while or after the install row serves it: "due in step t" and
//     "served in step t" must both mean "this step is the env's reset";
//   * the install row scans the bytes while logic lanes of the same launch write them: an env that terminates NOW must not
//     look due NOW (it was, with a plain 1: four of 128 mazes were reset in the step that ended them).
// So the marks carry the parity of the step they are about: reset_due_mark(t) = 4 | t mod 2, written by the logic lane of
// step t − 1; reset_served_mark(t) = 2 | t mod 2, written by whoever serves the env in step t.  In step t, due(t) and
// served(t) mean reset; anything else — 0, served(t − 1) — means step, and the logic lane overwrites it with its verdict.
// Nothing has to clear a mark.  (coinrun writes the byte in resolve_kernel, whose second row of blocks generates the
// levels that were not ready — level_serve, mode 2 — beside the lanes that write it: the same two races, the same marks.)
PG_HD int reset_due_mark(uint32_t step_index) { return 4 | static_cast<int>(step_index & 1u); }
PG_HD int reset_served_mark(uint32_t step_index) { return 2 | static_cast<int>(step_index & 1u); }
PG_HD bool resets_in_step(int pending, uint32_t step_index) {
    return pending == reset_due_mark(step_index) || pending == reset_served_mark(step_index);
}

template <class G>
PG_D void install_prefetched(const typename G::State& s, int base, int span, int prefetch, StepIO io, LevelPlan plan,
                             typename G::Level& lv, int lane, int served_mark = 2, int due_mark = 1) {
    using Level = typename G::Level;
    constexpr int kWords = static_cast<int>(sizeof(Level) / 4);
    const bool want = lane < span && base + lane < s.n && io.pending[base + lane] == due_mark;
    unsigned long long todo = __ballot(want);
    const bool levels = plan.num_levels > 0;
    while (todo) {  // wave-uniform
        const int env = base + __builtin_ctzll(todo);
        todo &= todo - 1;
        const int32_t st = __builtin_amdgcn_readfirstlane(slot_load(&s.slot[env]));  // acquire, agent scope
        if (st != kSlotReady) continue;  // queued, busy or idle: the level kernel behind this launch takes it
        __threadfence();
        const uint32_t* shadow = reinterpret_cast<const uint32_t*>(&s.shadow[env]);
        uint32_t* local = reinterpret_cast<uint32_t*>(&lv);
        for (int k = lane; k < kWords; k += 64) local[k] = shadow[k];
        __syncthreads();
        if (levels && lane == 0) G::fresh_live(s, env);
        __syncthreads();
        G::install(s, env, lv, lane);
        __threadfence();
        __syncthreads();
        if (lane == 0) {
            slot_store(&s.slot[env], prefetch ? kSlotQueued : kSlotIdle);
            io.reward[env] = 0.0f;
            io.done[env] = 0;
            io.pending[env] = static_cast<uint8_t>(served_mark);
            if constexpr (HasServed<G>::value) G::served(s, env);
        }
        __syncthreads();
    }
}

// Host side of the same: the four launches a prefetching game needs.
template <class G>
struct LevelLaunch {
    static void make(hipStream_t st, const typename G::State& s, int prefetch, uint32_t seed_base, int env_offset,
                     LevelPlan plan) {
        hipLaunchKernelGGL(level_kernel<G>, dim3(s.n), dim3(64), 0, st, s, 0, 1, prefetch, seed_base, env_offset,
                           nullptr, nullptr, StepIO{}, plan, 2, 1);
    }
    static void reset(hipStream_t st, const typename G::State& s, int prefetch, const uint8_t* mask,
                      const int32_t* seeds, StepIO io, LevelPlan plan) {
        hipLaunchKernelGGL(level_kernel<G>, dim3(s.n), dim3(64), 0, st, s, 1, 1, prefetch, 0u, 0, mask, seeds, io,
                           plan, 2, 1);
    }
    // span = envs per wavefront, served one after the other.  With prefetch a served env costs a copy, so 64 per wave
    // is right; a game that generates inside the step (chaser) wants few, or the step waits for the unluckiest wave:
    // (number of its envs that reset in this step) × (one generation).
#ifndef PG_RESET_SPAN
#define PG_RESET_SPAN 64
#endif
    static void auto_reset(hipStream_t st, const typename G::State& s, int prefetch, StepIO io, LevelPlan plan,
                           int span = PG_RESET_SPAN, int served_mark = 2, int due_mark = 1) {
        hipLaunchKernelGGL(level_kernel<G>, dim3((s.n + span - 1) / span), dim3(64), 0, st, s, 2, span, prefetch, 0u, 0,
                           nullptr, nullptr, io, plan, served_mark, due_mark);
    }
    // bulk: most slots are queued (after make / a full reset) → a wavefront per env; otherwise few envs per wave,
    // because the queued envs of one wave are served one after the other.
    static void pregen(hipStream_t side, const typename G::State& s, bool bulk, LevelPlan plan) {
        const int span = bulk ? 1 : 8;
        hipLaunchKernelGGL(level_kernel<G>, dim3((s.n + span - 1) / span), dim3(64), 0, side, s, 3, span, 1, 0u, 0,
                           nullptr, nullptr, StepIO{}, plan, 2, 1);
    }
};
#endif

}  // namespace pg
