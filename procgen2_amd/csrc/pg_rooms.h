// Room_Generator on the device, shared by caveflyer and jumper (games/caveflyer/room_generator.{h,cpp}; jumper's copy
// is identical): a W×W (40×40 by default) cellular-automaton cave, its largest 4-connected room in the iteration order of the
// reference's std::unordered_set<int>, the BFS path between two cells.  One wavefront per level, all state in LDS.
//
// What is serial in the reference and what is done about it:
//   * build_room / find_path walk a FIFO cell by cell → flood(): the FIFO is consumed in chunks of up to 64 entries by
//     the whole wave and the discoveries of a chunk are ordered exactly as the serial loop would order them;
//   * the rooms are std::unordered_set<int>, and `for (int i : best_room)` hands libstdc++'s node order to the level
//     → set_order(): that order in closed form (rank by first touch of the bucket, then by insertion position,
//     re-applied at every rehash) instead of replaying up to 1600 dependent pointer-chasing insertions.
#pragma once

#include "pg_defs.h"
#include "pg_order.h"
#include "pg_rng.h"
#include "pg_setorder.h"

// World side: the including game defines PG_ROOMS_DIM per distribution mode (pg_defs.h PG_VARIANT) before this header.
#ifndef PG_ROOMS_DIM
#define PG_ROOMS_DIM 40
#endif

namespace pg {
namespace PG_VARIANT_NS {
namespace rooms {

constexpr int W = PG_ROOMS_DIM, H = PG_ROOMS_DIM, kCells = W * H;

struct RoomsLds {
    uint32_t mt[kMtWords];
    uint8_t grid[kCells];    // Room_Generator::grid after the two automaton iterations
    uint8_t aux[kCells];     // automaton double buffer → room membership → path `covered` → widening layer + 1 (1 = path)
    union {
        int32_t claim[kCells];  // flood(): lowest (parent slot, direction) reaching a cell within one chunk
        int32_t tail_sum[kCells + 8];  // set_order(): elements listed before the bucket first touched at [p]
        struct {
            int16_t next[kCells];  // small hashtable replays (entity sets); the picked object indices
            int16_t before[128];
        };
    };
    int32_t touch[2368];  // set_order(): first insertion index per bucket (≤ 2357 buckets)
    int32_t chain[2368];  // set_order(): per-bucket chain of insertion indices (head; links in L.parent)
    int16_t queue[kCells + 4];   // BFS queue / `expanded` / widening layer A
    int16_t parent[kCells + 4];  // `parents` / widening layer B
    int16_t cells[kCells];   // free_cells / the goal path
    int32_t path_len, goal_cell, agent_cell;
};

PG_D int cell_of(int x, int y) { return y + H * x; }

// Room_Generator::update (room_generator.cpp:20-35): ≥5 walls among the 9 Moore cells (out of bounds = wall).
PG_D void automaton(const uint8_t* src, uint8_t* dst, int lane) {
    for (int c = lane; c < kCells; c += 64) {
        const int x = c / H, y = c % H;
        int n = 0;
        for (int a = -1; a <= 1; a++)
            for (int b = -1; b <= 1; b++) {
                const int nx = x + a, ny = y + b;
                n += (nx < 0 || ny < 0 || nx >= W || ny >= H) ? 1 : src[cell_of(nx, ny)];
            }
        dst[c] = n >= 5 ? 1 : 0;
    }
}

// The order of a System's entity set for this episode: `keys` inserted in creation order into a set that kept its
// bucket array across clear() (packed = buckets | next_resize << 16).
PG_D void episode_order(int32_t& packed, const uint8_t* keys, int n, uint8_t* out, int16_t* next, int16_t* before) {
    HashOrder h;
    h.next = next;
    h.before = before;
    h.head = kNil;
    h.buckets = packed & 0xffff;
    h.next_resize = packed >> 16;
    h.count = 0;
    for (int b = 0; b < h.buckets; b++) before[b] = kNil;
    for (int k = 0; k < n; k++) hash_insert(h, keys[k]);
    int16_t p = static_cast<int16_t>(h.head);
    for (int k = 0; k < n; k++) {
        out[k] = static_cast<uint8_t>(p);
        p = next[p];
    }
    packed = h.buckets | (h.next_resize << 16);
}

// Breadth-first flood from `start` over open cells by the whole wavefront, reproducing the reference's queue exactly
// (room_generator.cpp:37-75 build_room and :77-136 find_path share it): the FIFO is consumed in chunks of up to 64
// entries, lane j expanding entry j; a cell reached by several (entry, direction) pairs of one chunk goes to the
// lowest pair — the one the serial loop would have reached first — and the winners are appended in (entry,
// direction) order by a wave prefix sum.  L.queue receives the FIFO (entry 0 = start; a start with open neighbours
// appears a second time, as in the reference, because it is not marked when pushed), L.parent the queue slot each
// entry was discovered from when `parents` is set.  L.aux marks discovered cells; L.claim must be all-ones.
// Stops early once `stop_cell` has been discovered.  Returns the FIFO length.
PG_D int flood(RoomsLds& L, int start, bool parents, int stop_cell, int lane) {
    int qh = 0, qt = 1;
    if (lane == 0) {
        L.queue[0] = static_cast<int16_t>(start);
        L.parent[0] = -1;
    }
    __syncthreads();
    while (qh < qt) {
        const int take = (qt - qh) < 64 ? (qt - qh) : 64;
        int reach[4] = {-1, -1, -1, -1};
        if (lane < take) {
            const int cur = L.queue[qh + lane];
            const int x = cur / H, y = cur % H;
#pragma unroll
            for (int d = 0; d < 4; d++) {  // (i, j) = (-1,0) (0,-1) (0,1) (1,0)
                const int nx = x + (d == 0 ? -1 : d == 3 ? 1 : 0), ny = y + (d == 1 ? -1 : d == 2 ? 1 : 0);
                if (nx < 0 || ny < 0 || nx >= W || ny >= H) continue;
                const int ni = cell_of(nx, ny);
                if (!L.aux[ni] && L.grid[ni] == 0) {
                    reach[d] = ni;
                    atomicMin(&L.claim[ni], lane * 4 + d);
                }
            }
        }
        __syncthreads();
        bool win[4];
        int mine = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            win[d] = reach[d] >= 0 && L.claim[reach[d]] == lane * 4 + d;
            mine += win[d] ? 1 : 0;
        }
        int upto = mine;  // inclusive prefix sum over lanes
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(upto, off);
            if (lane >= off) upto += t;
        }
        const int total = __shfl(upto, 63);
        __syncthreads();
        int w = qt + upto - mine;
        bool found = false;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            if (reach[d] >= 0) L.claim[reach[d]] = 0x7fffffff;
            if (win[d]) {
                L.queue[w] = static_cast<int16_t>(reach[d]);
                if (parents) L.parent[w] = static_cast<int16_t>(qh + lane);
                L.aux[reach[d]] = 1;
                found = found || reach[d] == stop_cell;
                w++;
            }
        }
        qh += take;
        qt += total;
        __syncthreads();
        if (__ballot(found)) break;
    }
    return qt;
}

// Iteration order of a fresh std::unordered_set<int> after inserting the n distinct keys L.cells[0..n) one by one:
// pg_setorder.h (closed-form ranks per rehash round instead of replaying up to 1600 dependent insertions).
PG_D void set_order(RoomsLds& L, int n, int lane) {
    int32_t buckets = 1, next_resize = 0;
    wave_set_order(L.cells, n, buckets, next_resize, SetOrderScratch{L.touch, L.chain, L.parent, L.tail_sum, L.queue},
                   lane);
}

// Room_Generator::find_best_room (room_generator.cpp:138-160): the largest 4-connected room (the first one on
// ties), its cells in the iteration order of the reference's std::unordered_set (`best_room = next_room` keeps
// it): the FIFO's discovery order is the insertion order.  Result in L.cells; returns the size.  A lone open cell
// forms a room of size zero (it is never inserted).
PG_D int best_room(RoomsLds& L, int lane) {
    for (int c = lane; c < kCells; c += 64) {
        L.aux[c] = 0;
        L.claim[c] = 0x7fffffff;
    }
    __syncthreads();
    int best_n = -1, from = 0;
    for (;;) {
        // next start: the lowest open cell ≥ from that no room holds yet
        int first = kCells;
        for (int c = from + lane; c < kCells && first == kCells; c += 64)
            if (L.grid[c] == 0 && !L.aux[c]) first = c;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int t = __shfl_xor(first, off);
            first = t < first ? t : first;
        }
        if (first >= kCells) break;
        const int n = flood(L, first, false, -1, lane) - 1;
        if (n > best_n) {
            best_n = n;
            for (int k = lane; k < n; k += 64) L.cells[k] = L.queue[1 + k];
        }
        from = first + 1;
        __syncthreads();
    }
    __syncthreads();
    set_order(L, best_n, lane);  // overwrites L.claim: the floods are done
    return best_n;
}

// Room_Generator::find_path (:77-136): the BFS tree's path src → dst into L.cells; length in L.path_len.
PG_D void goal_path(RoomsLds& L, int src, int dst, int lane) {
    for (int c = lane; c < kCells; c += 64) {
        L.aux[c] = 0;  // `covered`; the source is not in it, so it is reached (and expanded) a second time
        L.claim[c] = 0x7fffffff;
    }
    __syncthreads();
    const int n = flood(L, src, true, dst, lane);
    if (lane == 0) {
        int at = n - 1;
        while (L.queue[at] != dst) at--;  // dst ≠ src has exactly one entry
        int len = 0;
        for (int k = at; k >= 0; k = L.parent[k]) len++;
        int w = len;
        for (int k = at; k >= 0; k = L.parent[k]) L.cells[--w] = L.queue[k];
        L.path_len = len;
    }
    __syncthreads();
}

// expand_room(wide_path, 4) (room_generator.cpp:162-202) applied to the path in L.cells[0..L.path_len): four layers
// of 8-neighbour growth through open cells.  Result: L.aux = 0 outside the widened set, 1 on the path, layer + 1
// elsewhere.
PG_D void widen(RoomsLds& L, int lane) {
    for (int c = lane; c < kCells; c += 64) L.aux[c] = 0;
    __syncthreads();
    // expand_room(wide_path, 4) (room_generator.cpp:162-202): four layers of 8-neighbour growth through open cells.
    // Only membership matters downstream, so each layer is one data-parallel pass: a cell joins layer k+1 when a
    // neighbour sits in layer k (a concurrent write can only turn a 0 into k+1, which no lane of this pass matches).
    for (int k = lane; k < L.path_len; k += 64) L.aux[L.cells[k]] = 1;
    __syncthreads();
    for (int layer = 1; layer <= 4; layer++) {
        for (int c = lane; c < kCells; c += 64) {
            if (L.aux[c] || L.grid[c]) continue;
            const int x = c / H, y = c % H;
            bool hit = false;
            for (int a = -1; a <= 1; a++)
                for (int b = -1; b <= 1; b++) {
                    const int nx = x + a, ny = y + b;
                    if ((a | b) == 0 || nx < 0 || ny < 0 || nx >= W || ny >= H) continue;
                    hit = hit || L.aux[cell_of(nx, ny)] == layer;
                }
            if (hit) L.aux[c] = static_cast<uint8_t>(layer + 1);
        }
        __syncthreads();
    }
}

}  // namespace rooms
}  // namespace PG_VARIANT_NS
}  // namespace pg
