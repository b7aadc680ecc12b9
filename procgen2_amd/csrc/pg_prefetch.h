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
    for (long spin = 0; spin < (1L << 34); spin++) {
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
#endif

}  // namespace pg
