// Gangs: G adjacent lanes of a wavefront step ONE env together (logic kernels of the games with entity lists).
//
// A lane-per-env logic kernel is 1 024 wavefronts at 65 536 envs — one per SIMD, nothing to hide a dependent chain
// behind — and each walks max-over-its-64-envs many bullets / mobs one after the other.  A gang turns the entity
// loops sideways: the G lanes take G consecutive entries of a list per trip, the env's scalars are held (uniformly)
// by all of them, and the order-dependent parts of the reference's loops — a count that shrinks while the loop runs,
// a `break` at the first hit, random draws in visiting order — become prefix counts over a G-bit ballot.  8 192
// wavefronts instead of 1 024 (G = 8), each with a chain an order of magnitude shorter.
//
// Memory rule: whatever a gang keeps in global memory across its loops is only ever read and written by the lane that
// OWNS it (ring slot k belongs to lane k mod G), so no value travels between lanes through memory; what does travel
// goes through ballots.  The one exception is the mt19937 state, see GangRng.
#pragma once

#include "pg_defs.h"
#include "pg_rng.h"

namespace pg {

#if defined(__HIPCC__)

PG_D void gang_fence() {  // stores of any lane above are visible to loads of any lane of the wavefront below
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int G>
struct Gang {
    static_assert(G == 4 || G == 8 || G == 16 || G == 32, "gang width");
    static constexpr uint32_t kAll = G == 32 ? 0xffffffffu : (1u << G) - 1u;
    int g;      // my lane within the gang
    int shift;  // the gang's first lane within the wavefront

    PG_D static Gang at(int thread) {
        Gang q;
        q.g = thread & (G - 1);
        q.shift = (thread & 63) & ~(G - 1);
        return q;
    }
    // One bit per lane of my gang (bit g = lane g).  Every branch around a call must be gang-uniform.
    PG_D uint32_t ballot(bool p) const { return static_cast<uint32_t>(__ballot(p) >> shift) & kAll; }
    PG_D bool any(bool p) const { return ballot(p) != 0; }

    // A trip over a ring of K slots filled at `next`, newest first, W lanes wide (W ≤ G and W | K; lanes g ≥ W sit the
    // trip out — the caller masks them): the trip covers list positions [i0, i0 + W); my slot is the one of them I own
    // (slot mod W = g), `rank` is its place in the trip.
    struct Trip {
        int rank, i, slot, turn;
    };
    template <int K, int W = G>
    PG_D Trip trip(int next, int i0) const {
        static_assert(K % W == 0 && W <= G, "ring size must be a multiple of the trip width");
        Trip t;
        const int c = (next - 1 - i0) & (W - 1);
        t.rank = (c - g) & (W - 1);
        t.i = i0 + t.rank;
        t.slot = (next - 1 - t.i) & (K - 1);
        t.turn = (W - 1 - c) & (W - 1);
        return t;
    }
    // The ballot of p over the trip's W lanes with bit r = the lane whose rank in the trip is r.
    template <int W = G>
    PG_D uint32_t ranked(const Trip& t, bool p) const {
        constexpr uint32_t kLanes = W == 32 ? 0xffffffffu : (1u << W) - 1u;
        const uint32_t rev = __brev(ballot(p) & kLanes) >> (32 - W);  // rev[q] = lane W-1-q
        return ((rev >> t.turn) | (rev << ((W - t.turn) & 31))) & kLanes;
    }
    PG_D static int before(uint32_t ranked_mask, int rank) { return __popc(ranked_mask & ((1u << rank) - 1u)); }
};

// The env's mt19937 stream as a gang uses it: the index lives in a register (uniform over the gang), words are read
// straight from the env's 2 500 bytes in global memory, and the regeneration of the 624 words is shared by the gang's
// lanes (same words as mt_twist).  This is the one place where lanes read what other lanes wrote: gang_fence()
// separates the chunks.
template <int G>
struct GangRng {
    uint32_t* x;
    int idx;
    Gang<G> q;
    // The next G words of the stream, one per lane (lane g: word win0 + g), fetched together: a step's handful of
    // uniform draws are lane-to-lane reads instead of a dependent round trip to memory each.
    uint32_t win;
    int win0;
    // Optional (round 6): a second buffer of 624 words per env and a selector byte — bit 1: the stream's CURRENT words
    // are in the second buffer (`other`) instead of the env's own (`home`); bit 0: whichever buffer is not current holds
    // the NEXT 624 words already, worked out ahead of time by somebody else (pg_rng.h mt_next_block_wave, from a kernel
    // that runs when no gang does).  A gang that runs out of numbers then just changes buffers (refill) instead of
    // regenerating the block in six dependent memory round trips — chaser's logic kernel spent 80 of its 180 µs in
    // wavefronts waiting for one of their eight gangs to do that; copying a block made ahead still cost 50.  The index
    // stays in home[kMtN].  Anybody else who reads the stream looks at the selector first (GangRng users' level
    // generators, snapshots).  No selector: regenerate in place, as before.
    uint32_t* home = nullptr;
    uint32_t* other = nullptr;
    uint8_t* sel_at = nullptr;
    int sel = 0;

    PG_D void fetch_window() {
        win0 = idx;
        const int at = idx + q.g;
        win = x[at < kMtN ? at : kMtN - 1];
    }
    PG_D static GangRng open(uint32_t* words, Gang<G> gang, uint32_t* second = nullptr, uint8_t* selector = nullptr) {
        GangRng r{words, static_cast<int>(words[kMtN]), gang, 0u, 0, words, second, selector, 0};
        if (selector != nullptr) {
            r.sel = *selector;
            if (r.sel & 2) r.x = second;
        }
        r.fetch_window();
        return r;
    }
    PG_D void close() const {
        if (q.g == 0) home[kMtN] = static_cast<uint32_t>(idx);
    }

    template <int kPer>
    PG_D void chunk(int i0, int count, int from) const {  // x[i] ← x[i+from] ^ mix(x[i], x[i+1]) for i in [i0, i0+count)
        uint32_t v[kPer];
#pragma unroll
        for (int t = 0; t < kPer; t++) {
            const int i = i0 + q.g + G * t;
            if (i < i0 + count) v[t] = x[i + from] ^ mt_mix(x[i], x[i + 1]);
        }
        gang_fence();
#pragma unroll
        for (int t = 0; t < kPer; t++) {
            const int i = i0 + q.g + G * t;
            if (i < i0 + count) x[i] = v[t];
        }
        gang_fence();
    }
    __device__ __noinline__ void twist() const {
        constexpr int kPer = 128 / G < 8 ? 128 / G : 8, kChunk = G * kPer, kFirst = kMtN - kMtM;  // kChunk ≤ 227: a chunk never reads what it writes
        for (int i = 0; i < kFirst; i += kChunk) chunk<kPer>(i, kFirst - i < kChunk ? kFirst - i : kChunk, kMtM);
        for (int i = kFirst; i < kMtN - 1; i += kChunk)
            chunk<kPer>(i, kMtN - 1 - i < kChunk ? kMtN - 1 - i : kChunk, kMtM - kMtN);
        if (q.g == 0) x[kMtN - 1] = x[kMtM - 1] ^ mt_mix(x[kMtN - 1], x[0]);
        gang_fence();
    }

    // The stream has run out: the next 624 words — in the other buffer if somebody has made them already, else regenerated
    // in place (in whichever buffer is current).
    PG_D void refill() {
        if (sel & 1) {  // (gang-uniform)
            x = (sel & 2) ? home : other;
            sel = (sel ^ 2) & 2;
            if (q.g == 0) *sel_at = static_cast<uint8_t>(sel);
            return;
        }
        twist();
    }

    // One engine output, the same for every lane of the gang (all of them call).
    PG_D uint32_t next() {
        if (idx >= kMtN) {
            refill();
            idx = 0;
            fetch_window();
        }
        if (idx - win0 >= G) fetch_window();
        const uint32_t w = static_cast<uint32_t>(__shfl(static_cast<int>(win), idx - win0, G));
        idx++;
        return mt_temper(w);
    }
    PG_D float canonical() { return canonical_of(next()); }
    PG_D float real(float a, float b) { return canonical() * (b - a) + a; }  // = rng_real
    PG_D int integer(int lo, int hi) {                                        // = rng_int
        const uint32_t range = static_cast<uint32_t>(hi) - static_cast<uint32_t>(lo) + 1u;
        uint64_t product = static_cast<uint64_t>(next()) * range;
        uint32_t low = static_cast<uint32_t>(product);
        if (low < range) {
            const uint32_t threshold = (0u - range) % range;
            while (low < threshold) {
                product = static_cast<uint64_t>(next()) * range;
                low = static_cast<uint32_t>(product);
            }
        }
        return lo + static_cast<int>(product >> 32);
    }
    // `total` outputs drawn one after the other by the lanes that want one: a wanting lane passes how many wanting
    // lanes come before it (`ahead`) and gets its output; the others get 0.  All lanes call, `total` is gang-uniform.
    PG_D uint32_t next_in_order(bool want, int ahead, int total) {
        const int at = idx + ahead;
        uint32_t w = (want && at < kMtN) ? x[at] : 0u;
        if (idx + total > kMtN) {  // the stream runs out in the middle: the rest comes from the next 624 words
            gang_fence();
            refill();
            if (want && at >= kMtN) w = x[at - kMtN];
            idx += total - kMtN;
            fetch_window();
        } else {
            idx += total;
        }
        return want ? mt_temper(w) : 0u;
    }
};

#endif

}  // namespace pg
