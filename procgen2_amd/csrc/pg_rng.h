// Per-env random stream: MT19937 plus the two libstdc++ 11 distributions the games use.
//
// Reference: every game owns one `std::mt19937 rng` for its whole life (games/coinrun/coinrun.cpp:34,235,316)
// and draws through std::uniform_int_distribution<int> / std::uniform_real_distribution<float>
// (SURVEY.md rows T1, T2).  Results are bit-exact against libstdc++ 11:
//   * uniform_int(a,b): Lemire's nearly-divisionless method on 32-bit draws
//     (/usr/include/c++/11/bits/uniform_int_dist.h:246-270,311-317);
//   * uniform_real<float>(a,b): generate_canonical<float,24> = one draw, u32→float (RNE) / 2^32,
//     clamped below 1, then c*(b-a)+a with separate roundings (bits/random.tcc:3348-3380, random.h:1865-1871).
// State layout: 624 × u32 words + one index word, wherever the caller keeps them (LDS in the kernels).
#pragma once

#include "pg_defs.h"

namespace pg {

constexpr int kMtN = 624;
constexpr int kMtM = 397;
constexpr int kMtWords = kMtN + 1;  // state + index, the per-env footprint in HBM (2500 B)

// mt19937::seed(value): x[0]=value, x[i]=1812433253*(x[i-1]^(x[i-1]>>30))+i; index = 624.
PG_HD void mt_seed(uint32_t* x, uint32_t seed) {
    x[0] = seed;
    for (int i = 1; i < kMtN; i++) x[i] = 1812433253u * (x[i - 1] ^ (x[i - 1] >> 30)) + static_cast<uint32_t>(i);
    x[kMtN] = kMtN;
}

PG_HD uint32_t mt_mix(uint32_t hi, uint32_t lo) {
    uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
    return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// In-place regeneration of all 624 words (one lane; amortised over 624 draws).
#if defined(__HIP_DEVICE_COMPILE__)
// On the device the state lives in global memory and the lane that runs out holds its whole wavefront up — in a logic
// kernel that is one wavefront per 64 envs, so nearly every step of bossfight or chaser has some wavefront here, and
// the step waits for it.  Written as a plain loop this is ~90 dependent memory round trips (the compiler forms groups
// of six or eight words); taking kChunk words at a time — every load of a chunk issued before any of its stores: a
// chunk reads only words no earlier store of the SAME chunk writes, `from` lies 227 or 397 words away — makes it 20.
// Same words as the loop below.
template <int kCount>
PG_D void mt_twist_chunk(uint32_t* x, int i0, int from) {
    uint32_t own[kCount + 1], far[kCount];
#pragma unroll
    for (int k = 0; k <= kCount; k++) own[k] = x[i0 + k];
#pragma unroll
    for (int k = 0; k < kCount; k++) far[k] = x[i0 + from + k];
#pragma unroll
    for (int k = 0; k < kCount; k++) x[i0 + k] = far[k] ^ mt_mix(own[k], own[k + 1]);
}
PG_D void mt_twist(uint32_t* x) {
    constexpr int kChunk = 32, kFirst = kMtN - kMtM;  // 227 = 7·32 + 3;  623 − 227 = 396 = 12·32 + 12
    for (int i = 0; i + kChunk <= kFirst; i += kChunk) mt_twist_chunk<kChunk>(x, i, kMtM);
    mt_twist_chunk<kFirst % kChunk>(x, kFirst - kFirst % kChunk, kMtM);
    constexpr int kSecond = kMtN - 1 - kFirst;
    for (int i = kFirst; i + kChunk <= kMtN - 1; i += kChunk) mt_twist_chunk<kChunk>(x, i, kMtM - kMtN);
    mt_twist_chunk<kSecond % kChunk>(x, kMtN - 1 - kSecond % kChunk, kMtM - kMtN);
    x[kMtN - 1] = x[kMtM - 1] ^ mt_mix(x[kMtN - 1], x[0]);
}
#else
PG_HD void mt_twist(uint32_t* x) {
    for (int i = 0; i < kMtN - kMtM; i++) x[i] = x[i + kMtM] ^ mt_mix(x[i], x[i + 1]);
    for (int i = kMtN - kMtM; i < kMtN - 1; i++) x[i] = x[i + kMtM - kMtN] ^ mt_mix(x[i], x[i + 1]);
    x[kMtN - 1] = x[kMtM - 1] ^ mt_mix(x[kMtN - 1], x[0]);
}
#endif

PG_HD uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

PG_HD uint32_t mt_next(uint32_t* x) {
    uint32_t idx = x[kMtN];
    if (idx >= static_cast<uint32_t>(kMtN)) {
        mt_twist(x);
        idx = 0;
    }
    const uint32_t y = x[idx];
    x[kMtN] = idx + 1;
    return mt_temper(y);
}

// std::uniform_int_distribution<int>(lo, hi)(rng), lo <= hi, hi - lo < 2^32 - 1.
PG_HD int rng_int(uint32_t* x, int lo, int hi) {
    const uint32_t range = static_cast<uint32_t>(hi) - static_cast<uint32_t>(lo) + 1u;
    uint64_t product = static_cast<uint64_t>(mt_next(x)) * range;
    uint32_t low = static_cast<uint32_t>(product);
    if (low < range) {
        const uint32_t threshold = (0u - range) % range;
        while (low < threshold) {
            product = static_cast<uint64_t>(mt_next(x)) * range;
            low = static_cast<uint32_t>(product);
        }
    }
    return lo + static_cast<int>(product >> 32);
}

// std::generate_canonical<float, 24>(rng) for one engine output
PG_HD float canonical_of(uint32_t word) {
    float c = static_cast<float>(word) / 4294967296.0f;
    if (c >= 1.0f) c = 0.99999994f;  // nextafter(1.0f, 0.0f) = 0x3f7fffff
    return c;
}
PG_HD float rng_canonical(uint32_t* x) { return canonical_of(mt_next(x)); }

// std::uniform_real_distribution<float>(a, b)(rng)
PG_HD float rng_real(uint32_t* x, float a, float b) { return rng_canonical(x) * (b - a) + a; }

#if defined(__HIPCC__)
// mt_twist by one wavefront (x in LDS): the recurrence x[i] ← x[i+397 mod 624] ^ mix(x[i], x[i+1]) reads, within
// each of the index ranges [0,227) [227,454) [454,623), only words that range does not write — except its own
// x[i] / x[i+1] neighbours, hence read-all, barrier, write-all per range.  Same words as mt_twist.
__device__ inline void mt_twist_wave(uint32_t* x, int lane) {
    const int lo[3] = {0, kMtN - kMtM, 2 * (kMtN - kMtM)};
    const int hi[3] = {kMtN - kMtM, 2 * (kMtN - kMtM), kMtN - 1};
    for (int ph = 0; ph < 3; ph++) {
        uint32_t v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = lo[ph] + lane + 64 * r;
            if (i < hi[ph]) v[r] = x[ph == 0 ? i + kMtM : i + kMtM - kMtN] ^ mt_mix(x[i], x[i + 1]);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = lo[ph] + lane + 64 * r;
            if (i < hi[ph]) x[i] = v[r];
        }
        __syncthreads();
    }
    if (lane == 0) x[kMtN - 1] = x[kMtM - 1] ^ mt_mix(x[kMtN - 1], x[0]);
    __syncthreads();
}

// The stream's NEXT 624 words from its current ones, OUT of place, by one wavefront (round 6).  What mt_twist makes of
// x in place is a function of x alone, so it can be worked out into a second buffer at any time after x was (re)generated
// — long before the stream runs out — by somebody who is not in a hurry (pg_gang.h GangRng::refill: the gang that runs out
// then copies 624 words, one memory round trip, instead of regenerating them in six dependent ones).  Word i of the new
// block, in three ranges: [0, 227): cur[i + 397] ^ mix(cur[i], cur[i + 1]); [227, 454) and [454, 623): NEW[i − 227] ^
// mix(cur[i], cur[i + 1]) — lane l takes words l, l + 64, … of every range, so the new word it needs is one it has made
// itself; 623: NEW[396] ^ mix(cur[623], NEW[0]).  Every load (of `cur` only) leaves before the first is waited for.
// cur and next are in device memory; every lane of the wave calls.
__device__ inline void mt_next_block_wave(const uint32_t* cur, uint32_t* next, int lane) {
    constexpr int kA = kMtN - kMtM;  // 227
    uint32_t own[3][4], up[3][4], far[4];
#pragma unroll
    for (int ph = 0; ph < 3; ph++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int j = lane + 64 * r, i = ph * kA + j;
            const bool in = j < kA && i < kMtN - 1;
            own[ph][r] = in ? cur[i] : 0u;
            up[ph][r] = in ? cur[i + 1] : 0u;
            if (ph == 0) far[r] = in ? cur[i + kMtM] : 0u;
        }
    const uint32_t last = cur[kMtN - 1];
    uint32_t made[3][4];
#pragma unroll
    for (int ph = 0; ph < 3; ph++)
#pragma unroll
        for (int r = 0; r < 4; r++) made[ph][r] = (ph == 0 ? far[r] : made[ph - 1][r]) ^ mt_mix(own[ph][r], up[ph][r]);
#pragma unroll
    for (int ph = 0; ph < 3; ph++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int j = lane + 64 * r, i = ph * kA + j;
            if (j < kA && i < kMtN - 1) next[i] = made[ph][r];
        }
    // word 623: NEW[396] = range 1, j = 169 = 41 + 64·2 (lane 41, r = 2); NEW[0] = range 0, lane 0, r = 0
    const uint32_t new396 = static_cast<uint32_t>(__shfl(static_cast<int>(made[1][2]), 41));
    const uint32_t new0 = static_cast<uint32_t>(__shfl(static_cast<int>(made[0][0]), 0));
    static_assert(kMtM - 1 - kA == 169 && 169 == 41 + 64 * 2, "where NEW[396] is made");
    if (lane == 0) next[kMtN - 1] = new396 ^ mt_mix(last, new0);
}

// Wave-uniform draws from a stream held in LDS: every lane calls them together and gets the same value; the
// regeneration of the 624 words, when due, is done by all lanes (mt_twist_wave) instead of by one.
__device__ inline uint32_t wave_mt_next(uint32_t* x, int lane) {
    int idx = static_cast<int>(x[kMtN]);
    if (idx >= kMtN) {
        __syncthreads();
        mt_twist_wave(x, lane);
        idx = 0;
    }
    const uint32_t y = x[idx];
    __syncthreads();  // everyone has read the index
    if (lane == 0) x[kMtN] = static_cast<uint32_t>(idx + 1);
    __syncthreads();
    return mt_temper(y);
}
__device__ inline int wave_rng_int(uint32_t* x, int lo, int hi, int lane) {  // = rng_int
    const uint32_t range = static_cast<uint32_t>(hi) - static_cast<uint32_t>(lo) + 1u;
    uint64_t product = static_cast<uint64_t>(wave_mt_next(x, lane)) * range;
    uint32_t low = static_cast<uint32_t>(product);
    if (low < range) {
        const uint32_t threshold = (0u - range) % range;
        while (low < threshold) {
            product = static_cast<uint64_t>(wave_mt_next(x, lane)) * range;
            low = static_cast<uint32_t>(product);
        }
    }
    return lo + static_cast<int>(product >> 32);
}
__device__ inline float wave_rng_real(uint32_t* x, float a, float b, int lane) {  // = rng_real
    return canonical_of(wave_mt_next(x, lane)) * (b - a) + a;
}

// `count` consecutive draws of uniform_real_distribution<float>(0,1)(rng) by one wavefront: f(k, value) is called
// once for every k in [0, count) by whichever lane owns that draw.  Leaves the stream where `count` calls of
// rng_real(x, 0, 1) would.  The caller needs a barrier before other lanes read what f stored.
template <class F>
__device__ inline void wave_draws(uint32_t* x, int count, int lane, F f) {
    int done = 0;
    while (done < count) {
        int idx = static_cast<int>(x[kMtN]);
        __syncthreads();
        if (idx >= kMtN) {
            mt_twist_wave(x, lane);
            idx = 0;
        }
        const int take = (kMtN - idx) < (count - done) ? (kMtN - idx) : (count - done);
        for (int k = lane; k < take; k += 64) f(done + k, canonical_of(mt_temper(x[idx + k])) * (1.0f - 0.0f) + 0.0f);
        if (lane == 0) x[kMtN] = static_cast<uint32_t>(idx + take);
        __syncthreads();
        done += take;
    }
}

// out[k] = 1 when the k-th of `count` draws is below one half.
__device__ inline void wave_coin_flips(uint32_t* x, uint8_t* out, int count, int lane) {
    wave_draws(x, count, lane, [&](int k, float v) { out[k] = v < 0.5f ? 1 : 0; });
}
#endif

}  // namespace pg
