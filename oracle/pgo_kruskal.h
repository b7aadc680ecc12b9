// ORACLE — TEST INFRASTRUCTURE ONLY (see pgo_common.h header).
//
// The two Kruskal maze generators of the reference, as free-standing pieces so that tests can run them next to the
// reference's own compiled sources (oracle/_ref, tests/test_reference_pin.py):
//   * Carver        — games/maze/maze_generator.{h,cpp}: union by rank with path halving, free-cell list, place_object;
//   * carve_merged  — games/{chaser,jumper}/maze_generator.cpp:47-130 (identical files): per-cell sets merged by
//                     relabelling; only the labels reach the result;
//   * open_dead_ends — the second half of generate_maze_no_dead_ends (:132-173), used by jumper.
#pragma once

#include <unordered_set>
#include <vector>

#include "pgo_common.h"

namespace pgo {

// Randomised Kruskal over a padded grid (maze/maze_generator.h, maze_generator.cpp).
struct Carver {
    static constexpr int kPad = 1;
    static constexpr int kInvalid = -1, kSpace = 0, kBrick = 1, kStartCellQuirk = 10;  // maze_generator.h:15-18
    int mw = 0, mh = 0, aw = 0, ah = 0;
    std::vector<int> grid, rank, parent, open_cells;
    std::unordered_set<int> open_set;
    int n_open = 0;

    int idx(int x, int y) const { return y + ah * x; }
    int get(int x, int y) const {
        if (x < 0 || y < 0 || x >= aw || y >= ah) return 1;
        return grid[idx(x, y)];
    }
    int root(int c) {  // maze_generator.cpp:47-53, path halving
        int cur = c;
        while (parent[cur] != cur) cur = parent[cur] = parent[parent[cur]];
        return cur;
    }
    void open(int x, int y) {  // maze_generator.cpp:34-45
        grid[idx(x + kPad, y + kPad)] = kSpace;
        int cell = y + mh * x;
        if (open_set.find(cell) == open_set.end()) {
            open_cells[n_open] = cell;
            open_set.insert(cell);
            n_open++;
        }
    }
    void carve(int w, int h, Rng& rng) {  // maze_generator.cpp:55-139
        mw = w;
        mh = h;
        aw = w + 2 * kPad;
        ah = h + 2 * kPad;
        rank.assign(aw * ah, 0);
        parent.assign(aw * ah, 0);
        open_cells.assign(aw * ah, 0);
        grid.assign(aw * ah, kBrick);
        grid[idx(kPad, kPad)] = kSpace;
        n_open = 0;
        open_set.clear();
        for (int i = 0; i < mw * mh; i++) parent[i] = i;

        struct Seg {
            int x1, y1, x2, y2;
        };
        std::vector<Seg> segs;
        for (int i = 1; i < mw; i += 2)
            for (int j = 0; j < mh; j += 2)
                if (i > 0 && i < mw - 1) segs.push_back({i - 1, j, i + 1, j});
        for (int i = 0; i < mw; i += 2)
            for (int j = 1; j < mh; j += 2)
                if (j > 0 && j < mh - 1) segs.push_back({i, j - 1, i, j + 1});

        while (!segs.empty()) {
            int n = rng.irange(0, static_cast<int>(segs.size()) - 1);
            Seg s = segs[n];
            int r0 = root(s.y1 + mh * s.x1);
            int r1 = root(s.y2 + mh * s.x2);
            int mx = (s.x1 + s.x2) / 2, my = (s.y1 + s.y2) / 2;
            int centre = my + mh * mx;
            if (get(mx + kPad, my + kPad) == kBrick && r0 != r1) {
                open(s.x1, s.y1);
                open(mx, my);
                open(s.x2, s.y2);
                if (rank[r0] > rank[r1]) {
                    parent[r1] = r0;
                    parent[centre] = r0;
                } else {
                    parent[r0] = r1;
                    parent[centre] = r1;
                    if (rank[r0] == rank[r1]) rank[r1]++;
                }
            }
            segs.erase(segs.begin() + n);
        }
    }
    void drop(int kind, Rng& rng) {  // maze_generator.cpp:183-195 (D7: compares the cell index with 10)
        int k = rng.irange(0, n_open - 1);
        while (open_cells[k] == kInvalid || open_cells[k] == kStartCellQuirk) k = rng.irange(0, n_open - 1);
        int cell = open_cells[k];
        open_cells[k] = kInvalid;
        grid[idx(cell / mh + kPad, cell % mh + kPad)] = kind;
    }
};

// chaser/jumper maze_generator.cpp:47-130: Kruskal over per-cell sets; only the set labels matter for the result.
// grid is (dim+2)², padded with walls, indexed y + (dim+2)·x.
inline void carve_merged(int dim, std::vector<int>& grid, Rng& rng) {
    const int ah = dim + 2;
    grid.assign(ah * ah, 1);
    grid[1 + ah * 1] = 0;
    std::vector<int> label(dim * dim);
    for (int i = 0; i < dim * dim; i++) label[i] = i;
    struct Seg {
        int x1, y1, x2, y2;
    };
    std::vector<Seg> walls;
    for (int i = 1; i < dim; i += 2)
        for (int j = 0; j < dim; j += 2)
            if (i > 0 && i < dim - 1) walls.push_back({i - 1, j, i + 1, j});
    for (int i = 0; i < dim; i += 2)
        for (int j = 1; j < dim; j += 2)
            if (j > 0 && j < dim - 1) walls.push_back({i, j - 1, i, j + 1});
    while (!walls.empty()) {
        const int n = rng.irange(0, static_cast<int>(walls.size()) - 1);
        const Seg w = walls[n];
        const int s0 = label[w.y1 + dim * w.x1], s1 = label[w.y2 + dim * w.x2];
        const int x0 = (w.x1 + w.x2) / 2, y0 = (w.y1 + w.y2) / 2;
        const int centre = y0 + dim * x0;
        if (grid[(y0 + 1) + ah * (x0 + 1)] == 1 && s0 != s1) {
            grid[(w.y1 + 1) + ah * (w.x1 + 1)] = 0;
            grid[(y0 + 1) + ah * (x0 + 1)] = 0;
            grid[(w.y2 + 1) + ah * (w.x2 + 1)] = 0;
            for (int& l : label)
                if (l == s0) l = s1;
            label[centre] = s1;
        }
        walls.erase(walls.begin() + n);
    }
}

// maze_generator.cpp:132-173, the pass generate_maze_no_dead_ends runs after generate_maze.
inline void open_dead_ends(int dim, std::vector<int>& grid, Rng& rng) {
    const int ah = dim + 2;
    for (int i = 0; i < ah * ah; i++) {
        if (grid[i] != 0) continue;
        const int x = i / ah, y = i % ah;
        const int nb[4] = {y + ah * (x - 1), y + ah * (x + 1), (y - 1) + ah * x, (y + 1) + ah * x};
        int spaces = 0, wallsn = 0;
        for (int n = 0; n < 4; n++) {
            if (grid[nb[n]] == 0)
                spaces++;
            else if (grid[nb[n]] == 1)
                wallsn++;
        }
        if (spaces == 1 && wallsn > 0) {
            const int pick = rng.irange(0, wallsn - 1);
            for (int n = 0; n < 4; n++) {
                const int cell = nb[(pick + n) % wallsn];  // indexes the neighbour list, not the walls (kept)
                const int cx = cell / ah, cy = cell % ah;
                if (cx >= 1 && cy >= 1 && cx < ah - 1 && cy < ah - 1 && grid[cell] == 1) {
                    grid[cell] = 0;
                    break;
                }
            }
        }
    }
}

}  // namespace pgo
