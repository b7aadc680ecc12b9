// Iteration order of a std::unordered_set<int> (libstdc++ 11) after inserting n distinct keys one by one, computed by
// one wavefront without replaying the node list (the serial twin is pg_order.h's HashOrder, which the unit tests pin
// to the real container).
//
// With B buckets fixed, inserting a sequence into an empty table leaves the node list as: buckets in order of their
// FIRST insertion, latest first; inside a bucket, latest insertion first — call that R(B, sequence).  The rank of
// element i under R is
//     (elements in buckets first touched after mine) + (elements of my bucket inserted after me),
// a suffix sum over first-touch positions plus a walk of my bucket's short chain: all lanes busy, a handful of LDS
// operations per element, instead of ~2n dependent pointer-chasing steps on one lane.
//   * A rehash re-inserts the nodes in list order under the new B, so after it (and the inserts that follow before
//     the next one) the list is R(B', current list ++ later keys).
//   * Inserts that follow WITHOUT a rehash (a set that kept a large bucket array across clear(), or the policy's
//     "next_resize = buckets" branch) leave the existing nodes in place: R(B, reverse(current list) ++ later keys),
//     because R applied to the reverse of an R-ordered list reproduces it.
// The growth policy itself (_Prime_rehash_policy, one insert at a time) is replayed in scalar code; it only decides
// where the rounds end.
#pragma once

#include "pg_defs.h"
#include "pg_order.h"

namespace pg {

struct SetOrderScratch {  // LDS
    int32_t* touch;     // [≥ final bucket count]  first position per bucket
    int32_t* chain;     // [≥ final bucket count]  per-bucket chain head
    int16_t* link;      // [≥ n]                   chain links
    int32_t* tail_sum;  // [≥ n + 1]
    int16_t* tmp;       // [≥ n]
};

// keys[0..n): in = the keys in insertion order, out = the set's iteration order.  buckets / next_resize: the set's
// bucket count and _M_next_resize before (a fresh set: 1 / 0; a clear()ed set keeps both) and after.
// All 64 lanes call; contains barriers.
__device__ inline void wave_set_order(int16_t* keys, int n, int32_t& buckets, int32_t& next_resize,
                                      const SetOrderScratch& S, int lane) {
    int have = 0;
    int B = buckets, NR = next_resize;
    while (have < n) {
        // the policy check of the insert that finds `have` elements (pg_order.h hash_insert)
        bool rehash = false;
        if (have + 1 > NR) {
            int floor_min = have + 1;
            if (NR == 0 && floor_min < 11) floor_min = 11;
            if (floor_min >= B) {
                int want = floor_min + 1;
                if (want < B * 2) want = B * 2;
                B = hash_next_bkt(want);
                NR = B;
                rehash = true;
            } else {
                NR = B;
            }
        }
        const int m = n < NR ? n : NR;  // inserts up to the next policy check that can change anything
        // the round's sequence: existing nodes (forward after a rehash, reversed otherwise), then the new keys
        if (!rehash && have > 1) {
            for (int i = lane; i < have / 2; i += 64) {
                const int16_t a = keys[i], b = keys[have - 1 - i];
                keys[i] = b;
                keys[have - 1 - i] = a;
            }
        }
        for (int b = lane; b < B; b += 64) {
            S.touch[b] = 0x7fffffff;
            S.chain[b] = -1;
        }
        __syncthreads();
        for (int i = lane; i < m; i += 64) {
            const int b = hash_mod(keys[i], B);
            atomicMin(&S.touch[b], i);
            S.link[i] = static_cast<int16_t>(atomicExch(&S.chain[b], i));
        }
        __syncthreads();
        // per element: its bucket's population and how many of it came later; first-touch positions carry the
        // population into the suffix sum
        for (int i = lane; i < m; i += 64) {
            const int b = hash_mod(keys[i], B);
            int all = 0;
            for (int q = S.chain[b]; q >= 0; q = S.link[q]) all++;
            S.tail_sum[i] = S.touch[b] == i ? all : 0;
        }
        if (lane == 0) S.tail_sum[m] = 0;
        __syncthreads();
        {   // tail_sum[p] ← Σ_{q ≥ p} tail_sum[q]: each lane owns a contiguous strip, strips combined by a wave scan
            const int strip = (m + 63) / 64;
            const int lo = lane * strip < m ? lane * strip : m, hi = (lo + strip) < m ? (lo + strip) : m;
            int mine = 0;
            for (int p = lo; p < hi; p++) mine += S.tail_sum[p];
            int above = mine;  // inclusive suffix over lanes
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_down(above, off);
                if (lane + off < 64) above += t;
            }
            int run = above - mine;  // everything in higher strips
            for (int p = hi - 1; p >= lo; p--) {
                run += S.tail_sum[p];
                S.tail_sum[p] = run;
            }
        }
        __syncthreads();
        for (int i = lane; i < m; i += 64) {
            const int b = hash_mod(keys[i], B);
            int after = 0;
            for (int q = S.chain[b]; q >= 0; q = S.link[q]) after += q > i ? 1 : 0;
            S.tmp[S.tail_sum[S.touch[b] + 1] + after] = keys[i];
        }
        __syncthreads();
        for (int i = lane; i < m; i += 64) keys[i] = S.tmp[i];
        __syncthreads();
        have = m;
    }
    buckets = B;
    next_resize = NR;
}

}  // namespace pg
