// Randomised Kruskal maze carving shared by maze, chaser and jumper (device only, one wavefront per maze, LDS).
//
// Reference: games/maze/maze_generator.cpp:55-139 (union by rank + path halving) and games/chaser|jumper/
// maze_generator.cpp:47-130 (per-cell std::unordered_set merging).  Both remove a drawn wall exactly when its two
// cells are in different components, so any disjoint-set structure yields the same maze; the draws are what has to
// match: one uniform_int over the walls still present, in construction order.
#pragma once

#include "pg_defs.h"
#include "pg_rng.h"

namespace pg {

// Randomised Kruskal with union by rank + path halving over a 1-cell padded grid (maze_generator.cpp:55-139), on
// LDS.  The reference erases the drawn wall from a std::vector (`walls.erase(walls.begin() + n)`), i.e. the draw
// selects the n-th wall still present in construction order: kept here as a 320-bit presence mask and a
// select-the-n-th-set-bit, instead of moving the tail of the array 312 times.  MAXDIM = the largest (odd) maze side.
template <int MAXDIM>
struct KruskalLdsT {
    static constexpr int kMaxDim = MAXDIM, kPadDim = kMaxDim + 2;
    static constexpr int kMaxSegs = 2 * ((kMaxDim - 1) / 2) * ((kMaxDim + 1) / 2);  // 312 walls between cells at 25×25
    static constexpr int kPresentWords = (kMaxSegs + 63) / 64;
    uint64_t present[kPresentWords];
    uint8_t grid[kPadDim * kPadDim + 3];
    uint8_t rank[kMaxDim * kMaxDim + 3];
    int16_t parent[kMaxDim * kMaxDim + 1];
    int16_t open_cells[kPadDim * kPadDim + 1];
    uint8_t seen[kMaxDim * kMaxDim + 3];
    uint8_t segs[kMaxSegs][4];
    int32_t n_open, drop_cell;
};
using KruskalLds = KruskalLdsT<25>;

template <class LDS>
struct CarverT {
    LDS& L;
    int mw, mh, aw, ah;

    PG_D int idx(int x, int y) const { return y + ah * x; }
    PG_D int get(int x, int y) const {
        if (x < 0 || y < 0 || x >= aw || y >= ah) return 1;
        return L.grid[idx(x, y)];
    }
    PG_D int root(int c) {
        int cur = c;
        while (L.parent[cur] != cur) {
            L.parent[cur] = L.parent[L.parent[cur]];
            cur = L.parent[cur];
        }
        return cur;
    }
    PG_D void open(int x, int y) {  // maze_generator.cpp:34-45
        L.grid[idx(x + 1, y + 1)] = 0;
        const int cell = y + mh * x;
        if (!L.seen[cell]) {
            L.open_cells[L.n_open++] = static_cast<int16_t>(cell);
            L.seen[cell] = 1;
        }
    }
    PG_D int nth_present(int n) const {
        int w = 0;
        for (;; w++) {
            const int c = __popcll(L.present[w]);
            if (n < c) break;
            n -= c;
        }
        uint64_t v = L.present[w];
        for (int k = 0; k < n; k++) v &= v - 1;
        return w * 64 + __builtin_ctzll(v);
    }
    // All lanes call; the draws are wave-uniform, the union-find runs on lane 0.
    PG_D void carve(int dim, uint32_t* mt, int lane) {
        mw = mh = dim;
        aw = ah = dim + 2;
        for (int k = lane; k < aw * ah; k += 64) {
            L.grid[k] = 1;
            L.open_cells[k] = 0;
        }
        for (int k = lane; k < mw * mh; k += 64) {
            L.parent[k] = static_cast<int16_t>(k);
            L.rank[k] = 0;
            L.seen[k] = 0;
        }
        __syncthreads();
        int n_segs = 0;
        if (lane == 0) {
            L.grid[idx(1, 1)] = 0;
            L.n_open = 0;
            for (int a = 1; a < mw; a += 2)
                for (int b = 0; b < mh; b += 2)
                    if (a > 0 && a < mw - 1) {
                        L.segs[n_segs][0] = static_cast<uint8_t>(a - 1);
                        L.segs[n_segs][1] = static_cast<uint8_t>(b);
                        L.segs[n_segs][2] = static_cast<uint8_t>(a + 1);
                        L.segs[n_segs][3] = static_cast<uint8_t>(b);
                        n_segs++;
                    }
            for (int a = 0; a < mw; a += 2)
                for (int b = 1; b < mh; b += 2)
                    if (b > 0 && b < mh - 1) {
                        L.segs[n_segs][0] = static_cast<uint8_t>(a);
                        L.segs[n_segs][1] = static_cast<uint8_t>(b - 1);
                        L.segs[n_segs][2] = static_cast<uint8_t>(a);
                        L.segs[n_segs][3] = static_cast<uint8_t>(b + 1);
                        n_segs++;
                    }
            for (int w = 0; w < LDS::kPresentWords; w++) {
                const int left = n_segs - 64 * w;
                L.present[w] = left >= 64 ? ~0ull : (left > 0 ? ((1ull << left) - 1ull) : 0ull);
            }
        }
        n_segs = ((dim - 1) / 2) * ((dim + 1) / 2) * 2;  // both loops: odd a in [1, dim-2] × even b in [0, dim-1]
        __syncthreads();
        for (; n_segs > 0; n_segs--) {
            const int pick = wave_rng_int(mt, 0, n_segs - 1, lane);
            if (lane == 0) {
                const int at = nth_present(pick);
                L.present[at >> 6] &= ~(1ull << (at & 63));
                const int x1 = L.segs[at][0], y1 = L.segs[at][1], x2 = L.segs[at][2], y2 = L.segs[at][3];
                const int r0 = root(y1 + mh * x1);
                const int r1 = root(y2 + mh * x2);
                const int mx = (x1 + x2) / 2, my = (y1 + y2) / 2;
                const int centre = my + mh * mx;
                if (get(mx + 1, my + 1) == 1 && r0 != r1) {
                    open(x1, y1);
                    open(mx, my);
                    open(x2, y2);
                    if (L.rank[r0] > L.rank[r1]) {
                        L.parent[r1] = static_cast<int16_t>(r0);
                        L.parent[centre] = static_cast<int16_t>(r0);
                    } else {
                        L.parent[r0] = static_cast<int16_t>(r1);
                        L.parent[centre] = static_cast<int16_t>(r1);
                        if (L.rank[r0] == L.rank[r1]) L.rank[r1]++;
                    }
                }
            }
        }
        __syncthreads();
    }
    // maze_generator.cpp:183-195; START_CELL = 10 is compared with the cell index (D7).
    PG_D void drop(int kind, uint32_t* mt, int lane) {
        const int n_open = L.n_open;
        int k = wave_rng_int(mt, 0, n_open - 1, lane);
        while (L.open_cells[k] == -1 || L.open_cells[k] == 10) k = wave_rng_int(mt, 0, n_open - 1, lane);
        __syncthreads();
        if (lane == 0) {
            const int cell = L.open_cells[k];
            L.open_cells[k] = -1;
            L.grid[idx(cell / mh + 1, cell % mh + 1)] = static_cast<uint8_t>(kind);
        }
        __syncthreads();
    }
};
using Carver = CarverT<KruskalLds>;

}  // namespace pg
