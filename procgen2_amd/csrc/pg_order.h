// Emulation of the two libstdc++ behaviours that decide iteration / draw order in the reference
// (SURVEY.md §0 fact 5, rows T3 and T4):
//
//  * HashOrder  — std::unordered_set<int> as used for every System's entity set (games/*/ecs.h:212-215)
//                 and by some level generators.  hash(int) = identity, bucket = key % B, nodes in one
//                 singly linked list; insert into an empty bucket goes to the global list head, into a
//                 non-empty bucket right behind that bucket's "before" node; single-insert growth
//                 B: 1→13→29→59→127→257→541→1109→2357 when size+1 > B; clear() keeps B.
//                 (libstdc++ 11 hashtable.h: _M_insert_unique_node / _M_insert_bucket_begin /
//                 _M_rehash_aux / _M_erase, hashtable_policy.h: _Prime_rehash_policy.)
//  * sort_by_key — std::sort (introsort: median-of-3 quick partitions down to 16, then insertion sort)
//                 with the games' comparator `a.first < b.first` on (z, entity) pairs
//                 (games/coinrun/common_systems.cpp:36-38).  Equal keys are NOT kept in order.
//
// Both are plain integer code; tests/test_primitives.py drives them against the real containers.
#pragma once

#include "pg_defs.h"

namespace pg {

// ---------------------------------------------------------------------------------------------
// HashOrder: storage is caller-provided so it can sit in LDS, scratch or host memory.
//   next[key]   : successor key in the node list, kNil at the tail            (size ≥ max key + 1)
//   before[b]   : the node *preceding* bucket b's first node: kNil (empty bucket), kHead
//                 (the list's before-begin sentinel) or a key                     (size ≥ max B)
// ---------------------------------------------------------------------------------------------
constexpr int16_t kNil = -1;
constexpr int16_t kHead = -2;

struct HashOrder {
    int16_t* next;
    int16_t* before;
    int32_t head;         // first key or kNil
    int32_t buckets;      // B
    int32_t count;
    int32_t next_resize;  // _Prime_rehash_policy::_M_next_resize
};

// The slice of libstdc++'s prime table reachable by single inserts below 2357 elements.
PG_HD int32_t hash_next_bkt(int32_t want) {
    // __fast_bkt for tiny sizes (only 13 is reachable: first insert asks for max(12, 2)).
    if (want <= 13) return 13;
    const int32_t primes[] = {29, 59, 127, 257, 541, 1109, 2357};
    for (int i = 0; i < 7; i++)
        if (primes[i] >= want) return primes[i];
    return 2357;
}

// key % b for 0 ≤ key < 2^15, 1 ≤ b < 2^12.  The device has no integer divider: a 32-bit `%` is a ~40-instruction
// dependent chain, and the replay does several per node; one reciprocal multiply plus a fix-up is exact here.
PG_HD int32_t hash_mod(int32_t key, int32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int32_t q = static_cast<int32_t>(static_cast<float>(key) * __builtin_amdgcn_rcpf(static_cast<float>(b)));
    int32_t r = key - q * b;
    if (r < 0) r += b;
    if (r >= b) r -= b;
    return r;
#else
    return key % b;
#endif
}

PG_HD void hash_init(HashOrder& h, int16_t* next, int16_t* before) {
    h.next = next;
    h.before = before;
    h.head = kNil;
    h.buckets = 1;
    h.count = 0;
    h.next_resize = 0;
    before[0] = kNil;
}

PG_HD int16_t hash_get_next(const HashOrder& h, int16_t node) { return node == kHead ? (int16_t)h.head : h.next[node]; }
PG_HD void hash_set_next(HashOrder& h, int16_t node, int16_t v) {
    if (node == kHead)
        h.head = v;
    else
        h.next[node] = v;
}

// unordered_set::clear(): nodes freed, bucket array zeroed, B and the rehash policy untouched.
PG_HD void hash_clear(HashOrder& h) {
    for (int b = 0; b < h.buckets; b++) h.before[b] = kNil;
    h.head = kNil;
    h.count = 0;
}

PG_HD void hash_rehash(HashOrder& h, int32_t nb) {  // _M_rehash_aux(n, true_type)
    for (int b = 0; b < nb; b++) h.before[b] = kNil;
    int16_t p = static_cast<int16_t>(h.head);
    h.head = kNil;
    int32_t bbegin_bkt = 0;
    while (p != kNil) {
        int16_t nxt = h.next[p];
        int32_t b = hash_mod(p, nb);
        if (h.before[b] == kNil) {
            h.next[p] = static_cast<int16_t>(h.head);
            h.head = p;
            h.before[b] = kHead;
            if (h.next[p] != kNil) h.before[bbegin_bkt] = p;
            bbegin_bkt = b;
        } else {
            int16_t prev = h.before[b];
            h.next[p] = hash_get_next(h, prev);
            hash_set_next(h, prev, p);
        }
        p = nxt;
    }
    h.buckets = nb;
}

// Returns the node before `key` in its bucket, or kNil when absent (_M_find_before_node).
PG_HD int16_t hash_find_before(const HashOrder& h, int32_t key) {
    int32_t b = hash_mod(key, h.buckets);
    int16_t prev = h.before[b];
    if (prev == kNil) return kNil;
    int16_t p = hash_get_next(h, prev);
    for (;;) {
        if (p == key) return prev;
        int16_t nx = h.next[p];
        if (nx == kNil || hash_mod(nx, h.buckets) != b) return kNil;
        prev = p;
        p = nx;
    }
}

PG_HD bool hash_contains(const HashOrder& h, int32_t key) { return h.count > 0 && hash_find_before(h, key) != kNil; }

// unordered_set::insert(key); `known_absent` skips the lookup when the caller has its own membership test.
PG_HD void hash_insert(HashOrder& h, int32_t key, bool known_absent = false) {
    if (!known_absent && h.count > 0 && hash_find_before(h, key) != kNil) return;
    // _Prime_rehash_policy::_M_need_rehash(B, count, 1), max_load_factor = 1
    if (h.count + 1 > h.next_resize) {
        int32_t floor_min = h.count + 1;
        if (h.next_resize == 0 && floor_min < 11) floor_min = 11;
        if (floor_min >= h.buckets) {
            int32_t want = floor_min + 1;
            if (want < h.buckets * 2) want = h.buckets * 2;
            int32_t nb = hash_next_bkt(want);
            h.next_resize = nb;
            hash_rehash(h, nb);
        } else {
            h.next_resize = h.buckets;
        }
    }
    // _M_insert_bucket_begin
    int32_t b = hash_mod(key, h.buckets);
    int16_t k = static_cast<int16_t>(key);
    if (h.before[b] != kNil) {
        int16_t prev = h.before[b];
        h.next[k] = hash_get_next(h, prev);
        hash_set_next(h, prev, k);
    } else {
        h.next[k] = static_cast<int16_t>(h.head);
        h.head = k;
        if (h.next[k] != kNil) h.before[hash_mod(h.next[k], h.buckets)] = k;
        h.before[b] = kHead;
    }
    h.count++;
}

// unordered_set::erase(key)
PG_HD void hash_erase(HashOrder& h, int32_t key) {
    if (h.count == 0) return;
    int16_t prev = hash_find_before(h, key);
    if (prev == kNil) return;
    int32_t b = hash_mod(key, h.buckets);
    int16_t n = static_cast<int16_t>(key);
    int16_t nx = h.next[n];
    if (prev == h.before[b]) {
        // _M_remove_bucket_begin
        int32_t nb = (nx != kNil) ? hash_mod(nx, h.buckets) : 0;
        if (nx == kNil || nb != b) {
            if (nx != kNil) h.before[nb] = h.before[b];
            if (h.before[b] == kHead) h.head = nx;
            h.before[b] = kNil;
        }
    } else if (nx != kNil) {
        int32_t nb = hash_mod(nx, h.buckets);
        if (nb != b) h.before[nb] = prev;
    }
    hash_set_next(h, prev, nx);
    h.count--;
}

// ---------------------------------------------------------------------------------------------
// std::sort on (key, id) pairs with comparator key_a < key_b   (bits/stl_algo.h: __sort,
// __introsort_loop, __unguarded_partition_pivot, __move_median_to_first, __final_insertion_sort;
// _S_threshold = 16).  The heap-sort fallback (depth limit 2*floor(log2 n)) is kept for completeness.
// ---------------------------------------------------------------------------------------------
struct ZItem {
    float z;
    int32_t id;
};

PG_HD void zswap(ZItem& a, ZItem& b) {
    ZItem t = a;
    a = b;
    b = t;
}

PG_HD void sort_adjust_heap(ZItem* first, int hole, int len, ZItem value) {  // __adjust_heap + __push_heap
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (first[child].z < first[child - 1].z) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && first[parent].z < value.z) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

PG_HD void sort_heapsort(ZItem* first, int n) {  // __partial_sort(first, last, last)
    if (n < 2) return;
    for (int parent = (n - 2) / 2;; parent--) {  // __make_heap
        ZItem v = first[parent];
        sort_adjust_heap(first, parent, n, v);
        if (parent == 0) break;
    }
    for (int last = n - 1; last > 0; last--) {  // __sort_heap → __pop_heap
        ZItem v = first[last];
        first[last] = first[0];
        sort_adjust_heap(first, 0, last, v);
    }
}

PG_HD void sort_insertion(ZItem* a, int first, int last) {  // __insertion_sort on [first,last)
    if (first == last) return;
    for (int i = first + 1; i < last; i++) {
        ZItem v = a[i];
        if (v.z < a[first].z) {
            for (int k = i; k > first; k--) a[k] = a[k - 1];
            a[first] = v;
        } else {  // __unguarded_linear_insert
            int k = i;
            while (v.z < a[k - 1].z) {
                a[k] = a[k - 1];
                k--;
            }
            a[k] = v;
        }
    }
}

PG_HD void sort_by_key(ZItem* a, int n) {
    if (n < 2) return;
    // __introsort_loop with an explicit stack of pending [first,last) ranges (the library recurses on
    // the right part and loops on the left; the partitions are disjoint so order of visiting is free).
    int stack_first[32], stack_last[32], stack_depth[32];
    int sp = 0;
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) lg++;
    stack_first[sp] = 0;
    stack_last[sp] = n;
    stack_depth[sp] = 2 * lg;
    sp++;
    while (sp > 0) {
        sp--;
        int first = stack_first[sp], last = stack_last[sp], depth = stack_depth[sp];
        while (last - first > 16) {
            if (depth == 0) {
                sort_heapsort(a + first, last - first);
                break;
            }
            depth--;
            // __unguarded_partition_pivot
            int mid = first + (last - first) / 2;
            {  // __move_median_to_first(first, first+1, mid, last-1)
                int ia = first + 1, ib = mid, ic = last - 1;
                if (a[ia].z < a[ib].z) {
                    if (a[ib].z < a[ic].z)
                        zswap(a[first], a[ib]);
                    else if (a[ia].z < a[ic].z)
                        zswap(a[first], a[ic]);
                    else
                        zswap(a[first], a[ia]);
                } else if (a[ia].z < a[ic].z)
                    zswap(a[first], a[ia]);
                else if (a[ib].z < a[ic].z)
                    zswap(a[first], a[ic]);
                else
                    zswap(a[first], a[ib]);
            }
            int lo = first + 1, hi = last;
            for (;;) {  // __unguarded_partition(first+1, last, pivot=first)
                while (a[lo].z < a[first].z) lo++;
                hi--;
                while (a[first].z < a[hi].z) hi--;
                if (!(lo < hi)) break;
                zswap(a[lo], a[hi]);
                lo++;
            }
            int cut = lo;
            stack_first[sp] = cut;
            stack_last[sp] = last;
            stack_depth[sp] = depth;
            sp++;
            last = cut;
        }
    }
    // __final_insertion_sort
    if (n > 16) {
        sort_insertion(a, 0, 16);
        for (int i = 16; i < n; i++) {  // __unguarded_insertion_sort
            ZItem v = a[i];
            int k = i;
            while (v.z < a[k - 1].z) {
                a[k] = a[k - 1];
                k--;
            }
            a[k] = v;
        }
    } else {
        sort_insertion(a, 0, n);
    }
}

// ---------------------------------------------------------------------------------------------
// Every game gives all its sprites the same z, so std::sort never sees `a < b` hold and its
// result is a fixed permutation of its input that depends on the element count alone.  equal_key_ranks[n][r] = the
// position sort_by_key gives to the r-th of n equal-key elements; with it a draw list is rebuilt by scattering the
// surviving sprites (in set order) instead of re-running the introsort twin on a private array in every lane whose
// sprite set changed.  Table: rank_offset(n) + r, n ≤ kRankMax; built on the host from sort_by_key itself.
// ---------------------------------------------------------------------------------------------
constexpr int kRankMax = 208;  // chaser extreme_mode: 198 sprites
PG_HD int rank_offset(int n) { return n * (n - 1) / 2; }
constexpr int kRankTableBytes = (kRankMax + 1) * kRankMax / 2;

inline void build_equal_key_ranks(uint8_t* out) {
    for (int n = 1; n <= kRankMax; n++) {
        ZItem items[kRankMax];
        for (int k = 0; k < n; k++) items[k] = {1.0f, k};
        sort_by_key(items, n);
        for (int k = 0; k < n; k++) out[rank_offset(n) + items[k].id] = static_cast<uint8_t>(k);
    }
}

}  // namespace pg
