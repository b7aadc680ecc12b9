// Tile-map helpers shared by the tile-based games' logic kernels (device only).
//
//  * TileWinT — a 4×4 window of tile ids in collide() coordinates (x, y ↦ tile (x, H-1-y)), 3 bits per cell in one
//    64-bit word: fetched with 16 independent byte loads (one memory round trip), after which every lookup of a
//    collide() is a shift and a mask.  Cells outside the window fall back to a direct load, so where the window
//    is placed affects speed only.
//  * collide_plain — System_Tilemap::get_collision "variant B" (maze, caveflyer, chaser, climber, jumper share one
//    body, e.g. games/climber/tilemap.cpp:200-258): pass 1 resolves y where the overlap is
//    wider than tall, pass 2 resolves x otherwise, over tiles floor(x)..ceil(x+w) × floor(y)..ceil(y+h), the
//    rectangle mutating between tiles.  coinrun's variant A (one-way crates) lives in coinrun.hip.
#pragma once

#include "pg_geom.h"

namespace pg {

template <int W, int H, int OOB>
struct TileWinT {
    const uint8_t* tiles;  // column-major y + x*H, low 3 bits = tile id
    int ax, ay;
    uint64_t bits;

    PG_D static int direct(const uint8_t* tiles, int x, int y) {
        const int ty = H - 1 - y;
        if (x < 0 || ty < 0 || x >= W || ty >= H) return OOB;
        return tiles[ty + x * H] & 7;
    }
    PG_D static TileWinT fetch(const uint8_t* tiles, int ax, int ay) {
        TileWinT w{tiles, ax, ay, 0};
        // A column of the window is four consecutive bytes of the map (index ty + x·H, ty = H − 1 − y): four loads of a
        // word — at any byte address, which global memory allows — instead of sixteen of a byte.  UNCONDITIONAL loads, so
        // that they are all in flight together (a load behind the bounds test is a load in a branch of its own, and the
        // compiler waits for each before it enters the next): a column beyond the map reads column 0, a word that would
        // stick out above or below is read from the nearest place inside and shifted, and a cell outside the map is
        // replaced afterwards.
        static_assert(H >= 4, "a column holds a word");
        const int ty_lo = H - 4 - ay;  // the window's row ay + 3; row ay + j sits in byte 3 − j of the column's word
        const int ty_at = ty_lo < 0 ? 0 : (ty_lo > H - 4 ? H - 4 : ty_lo);
        const int delta = ty_at - ty_lo;  // ∈ [−3, 3] where any cell is inside
        const uint32_t sh = 8u * static_cast<uint32_t>(delta < 0 ? -delta : delta);
        uint32_t col[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int x = ax + c;
            __builtin_memcpy(&col[c], tiles + ty_at + ((x < 0 || x >= W) ? 0 : x) * H, 4);
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int x = ax + c;
            const uint32_t word = sh >= 32u ? 0u : (delta >= 0 ? col[c] << sh : col[c] >> sh);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int ty = H - 1 - (ay + j);
                const bool inside = !(x < 0 || ty < 0 || x >= W || ty >= H);
                const int tile = inside ? static_cast<int>((word >> (8 * (3 - j))) & 7u) : OOB;
                w.bits |= static_cast<uint64_t>(tile) << (3 * (c + 4 * j));
            }
        }
        return w;
    }
    PG_D int cell(int dx, int dy) const { return static_cast<int>((bits >> (3 * (dx + 4 * dy))) & 7u); }  // inside the window
    PG_D int at(int x, int y) const {
        const unsigned dx = static_cast<unsigned>(x - ax), dy = static_cast<unsigned>(y - ay);
        if (dx < 4u && dy < 4u) return cell(static_cast<int>(dx), static_cast<int>(dy));
        return direct(tiles, x, y);
    }
    // Does the window hold every tile collide_plain() scans for box r (floor(x)..ceil(x+w) × floor(y)..ceil(y+h))?
    PG_D bool holds(const Box& r) const {
        const int x0 = static_cast<int>(floorf(r.x)), y0 = static_cast<int>(floorf(r.y));
        const int x1 = static_cast<int>(ceilf(r.x + r.w)), y1 = static_cast<int>(ceilf(r.y + r.h));
        return x0 >= ax && y0 >= ay && x1 < ax + 4 && y1 < ay + 4;
    }
    // The window for box r with its spare column / row on the side the box is heading (dir: velocity signs).
    PG_D static TileWinT around(const uint8_t* tiles, const Box& r, float dir_x, float dir_y) {
        const int x0 = static_cast<int>(floorf(r.x)), y0 = static_cast<int>(floorf(r.y));
        return fetch(tiles, x0 - (dir_x < 0.0f ? 1 : 0), y0 - (dir_y < 0.0f ? 1 : 0));
    }
};

// The same window for a map that only has walls and empty cells, from one 64-bit word per map column: bit y (collide()
// coordinates) = wall at (x, y), bits from H up set (above the map is wall), and out of the map sideways or below is wall
// too (OOB).  Four words instead of sixteen bytes, and a logic kernel can hold the whole map of an env in 8·W bytes of LDS.
template <int W, int H, int WALL, int EMPTY>
struct BitWinT {
    static_assert(H + 8 <= 60, "column words keep their top bits for the rows above the map");
    const uint64_t* cols;  // [W]
    int ax, ay;
    uint32_t bits;  // bit 4·dx + dy

    PG_D static uint32_t rows4(uint64_t word, int y) {  // the wall bits of rows y..y+3 of a column
        const int at = y < -4 ? -4 : (y > 56 ? 56 : y);
        return static_cast<uint32_t>((((word << 4) | 0xfull) >> (at + 4)) & 15ull);
    }
    PG_D static BitWinT fetch(const uint64_t* cols, int ax, int ay) {
        BitWinT w{cols, ax, ay, 0u};
        uint64_t word[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = ax + k;
            word[k] = cols[(x < 0 || x >= W) ? 0 : x];
            if (x < 0 || x >= W) word[k] = ~0ull;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) w.bits |= rows4(word[k], ay) << (4 * k);
        return w;
    }
    PG_D int cell(int dx, int dy) const { return ((bits >> (4 * dx + dy)) & 1u) ? WALL : EMPTY; }
    PG_D int at(int x, int y) const {
        const unsigned dx = static_cast<unsigned>(x - ax), dy = static_cast<unsigned>(y - ay);
        if (dx < 4u && dy < 4u) return cell(static_cast<int>(dx), static_cast<int>(dy));
        if (x < 0 || x >= W) return WALL;
        return (rows4(cols[x], y) & 1u) ? WALL : EMPTY;
    }
};

struct TileHit {
    float x, y;
    bool any;
};

// get_collision_overlap (pg_geom.h box_overlap) without its early return: the same values through selects.
PG_D Box box_overlap_flat(const Box& a, const Box& b) {
    const bool hit = (a.x < b.x + b.w) & (a.x + a.w > b.x) & (a.y < b.y + b.h) & (a.y + a.h > b.y);
    const float ddx = fabsf(a.x - b.x), ddy = fabsf(a.y - b.y);
    const bool left = a.x <= b.x, top = a.y <= b.y;
    float w = (left ? a.w : b.w) - ddx, h = (top ? a.h : b.h) - ddy;
    const float wcap = a.w > b.w ? b.w : a.w, hcap = a.h > b.h ? b.h : a.h;
    w = w >= wcap ? wcap : w;
    h = h >= hcap ? hcap : h;
    return Box{hit ? (left ? b.x : a.x) : 0.0f, hit ? (top ? b.y : a.y) : 0.0f, hit ? w : 0.0f, hit ? h : 0.0f};
}

// collide_plain(win, r, solid).any alone, for callers that only ask whether the box touches something (a bullet, an
// enemy that turns round): `any` is set by the first solid cell whose overlap with the box is not 0×0, and until then
// the box has not moved — so it is "some solid cell of the scan has a non-empty overlap with r", and get_collision_overlap
// is separable: hit = hit_x ∧ hit_y, its width a function of the x axis and its height of the y axis alone.  Three
// columns and three rows are evaluated with box_overlap's own arithmetic, the nine cells are bit tests.  The box must
// be at most a cell wide and tall (scan of at most 3×3) and the window placed at (floor(r.x), floor(r.y)).
struct AxisTouch {
    uint32_t hit, zero;  // bit k: the box overlaps cell k of this axis; the overlap's extent is 0
};
PG_D AxisTouch axis_touch(float a, float aw, int c0, int c1) {  // cells c0..c1 (≤ c0+2), each [c, c+1)
    AxisTouch t{0u, 0u};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float b = static_cast<float>(c0 + k), bw = 1.0f;
        const bool in = c0 + k <= c1;
        const bool hit = (a < b + bw) & (a + aw > b);
        const float dd = fabsf(a - b);
        float w = (a <= b ? aw : bw) - dd;
        const float cap = aw > bw ? bw : aw;
        w = w >= cap ? cap : w;
        t.hit |= (in & hit) ? 1u << k : 0u;
        t.zero |= (w == 0.0f) ? 1u << k : 0u;
    }
    return t;
}
template <class Win, class Pred>
PG_D bool collide_any(const Win& win, const Box& r, Pred solid) {
    const int x0 = static_cast<int>(floorf(r.x)), y0 = static_cast<int>(floorf(r.y));
    const int x1 = static_cast<int>(ceilf(r.x + r.w)), y1 = static_cast<int>(ceilf(r.y + r.h));
    const AxisTouch tx = axis_touch(r.x, r.w, x0, x1), ty = axis_touch(r.y, r.h, y0, y1);
    const int wx = x0 - win.ax, wy = y0 - win.ay;
    bool any = false;
#pragma unroll
    for (int dy = 0; dy < 3; dy++)
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            const bool touch = ((tx.hit >> dx) & (ty.hit >> dy) & 1u) != 0;           // box_hit (and the cell is scanned)
            const bool empty = ((tx.zero >> dx) & (ty.zero >> dy) & 1u) != 0;         // o.w == 0 && o.h == 0
            any = any | (touch & !empty & solid(win.cell((wx + dx) & 3, (wy + dy) & 3)));
        }
    return any;
}

// PG_WALK_SKIP: a step of the flat walk is skipped — wave-uniformly — when no lane's box meets a solid cell there: whoever
// takes a cell meets it (strictly: get_collision_overlap is all zeros otherwise), so the walk's results are the same, and
// of the eighteen steps of a walk most find nobody: the row under a mob's probe is floor for every mob and met by none.
// climber's logic kernel 96.2 -> 87.7 µs, coinrun's 43.8 -> 42.3 (its own variant A walk has the same line), jumper's and
// caveflyer's unchanged (a lane per env / the gang walk).  0: every step, as before.
#ifndef PG_WALK_SKIP
#define PG_WALK_SKIP 1
#endif
// kFlat: see below — pays where boxes are a tile in size (climber −2 %), not where they are bullets (caveflyer +2 %).
template <bool kFlat = false, class Win, class Pred>
PG_D TileHit collide_plain(const Win& win, Box r, Pred solid) {
    bool any = false;
    const int x0 = static_cast<int>(floorf(r.x)), y0 = static_cast<int>(floorf(r.y));
    const int x1 = static_cast<int>(ceilf(r.x + r.w)), y1 = static_cast<int>(ceilf(r.y + r.h));
    const float mid_x = r.x + r.w * 0.5f, mid_y = r.y + r.h * 0.5f;
    Box cell{0.0f, 0.0f, 1.0f, 1.0f};
    if (kFlat && ((x1 - x0 <= 2) & (y1 - y0 <= 2) & (x0 >= win.ax) & (y0 >= win.ay) & (x1 < win.ax + 4) & (y1 < win.ay + 4))) {
        // At most three by three cells, all inside the window: nine fixed steps with every decision a select (bitwise,
        // so that nothing short-circuits into a branch) — same cells, same order, same arithmetic as the loops below.
        // In a logic kernel (one wavefront per 64 envs, its SIMD to itself) the loops' taken branches cost more than
        // the arithmetic of the cells they skip.
        const int wx = x0 - win.ax, wy = y0 - win.ay;
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                const bool in = (x0 + dx <= x1) & (y0 + dy <= y1);
                const int ox_ = in ? dx : 0, oy_ = in ? dy : 0;  // (a cell past the box: look at the first one, ignore it)
                const bool is_solid = solid(win.cell(wx + ox_, wy + oy_));
                cell.x = static_cast<float>(x0 + ox_);
                cell.y = static_cast<float>(y0 + oy_);
#if PG_WALK_SKIP
                if (__ballot(in & is_solid & box_hit(r, cell)) == 0) continue;
#endif
                const Box o = box_overlap_flat(r, cell);
                const bool take = in & is_solid & !((o.w == 0.0f) & (o.h == 0.0f)) & (o.w > o.h);
                r.y = take ? (o.y + o.h * 0.5f > mid_y ? cell.y - r.h : cell.y + cell.h) : r.y;
                any = any | take;
            }
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                const bool in = (x0 + dx <= x1) & (y0 + dy <= y1);
                const int ox_ = in ? dx : 0, oy_ = in ? dy : 0;
                const bool is_solid = solid(win.cell(wx + ox_, wy + oy_));
                cell.x = static_cast<float>(x0 + ox_);
                cell.y = static_cast<float>(y0 + oy_);
#if PG_WALK_SKIP
                if (__ballot(in & is_solid & box_hit(r, cell)) == 0) continue;
#endif
                const Box o = box_overlap_flat(r, cell);
                const bool take = in & is_solid & !((o.w == 0.0f) & (o.h == 0.0f)) & (o.w <= o.h);
                r.x = take ? (o.x + o.w * 0.5f > mid_x ? cell.x - r.w : cell.x + cell.w) : r.x;
                any = any | take;
            }
        return {r.x, r.y, any};
    }
    for (int y = y0; y <= y1; y++)
        for (int x = x0; x <= x1; x++) {
            if (!solid(win.at(x, y))) continue;
            cell.x = static_cast<float>(x);
            cell.y = static_cast<float>(y);
            const Box o = box_overlap(r, cell);
            if (o.w == 0.0f && o.h == 0.0f) continue;
            if (o.w > o.h) {
                r.y = (o.y + o.h * 0.5f > mid_y ? cell.y - r.h : cell.y + cell.h);
                any = true;
            }
        }
    for (int y = y0; y <= y1; y++)
        for (int x = x0; x <= x1; x++) {
            if (!solid(win.at(x, y))) continue;
            cell.x = static_cast<float>(x);
            cell.y = static_cast<float>(y);
            const Box o = box_overlap(r, cell);
            if (o.w == 0.0f && o.h == 0.0f) continue;
            if (o.w <= o.h) {
                r.x = (o.x + o.w * 0.5f > mid_x ? cell.x - r.w : cell.x + cell.w);
                any = true;
            }
        }
    return {r.x, r.y, any};
}

// collide_plain<true> for a box that every lane of a gang of eight holds alike (the env's own body, pg_gang.h): the nine
// cells of a pass are looked at SIDE BY SIDE instead of one after the other.  The walk is sequential only through the
// box — a cell that is taken moves it, and the cells behind are held against the moved box — so: lane k judges cell k
// against the box as it stands; the first cell that takes (a ballot) is exactly the one the walk would take next, since
// none before it did and the box has not moved since they were judged; it moves the box, and the cells behind it are
// judged again.  A pass ends when nobody takes: one round more than it has takes — one or two for a body that rests on
// a floor, against nine steps.  The ninth cell, the last of the walk, is judged by every lane once the eight are settled.
// Same cells, same order, same arithmetic; `q`: Gang<8>.  (caveflyer's ship, round 6: the walk was 31 of its logic
// kernel's 102 µs — every lane of the gang doing all of it.)
template <class Q, class Win, class Pred>
PG_D TileHit collide_plain_gang8(const Q& q, const Win& win, Box r, Pred solid) {
    const int x0 = static_cast<int>(floorf(r.x)), y0 = static_cast<int>(floorf(r.y));
    const int x1 = static_cast<int>(ceilf(r.x + r.w)), y1 = static_cast<int>(ceilf(r.y + r.h));
    if (!((x1 - x0 <= 2) & (y1 - y0 <= 2) & (x0 >= win.ax) & (y0 >= win.ay) & (x1 < win.ax + 4) & (y1 < win.ay + 4)))
        return collide_plain<true>(win, r, solid);  // (gang-uniform: the box is)
    const float mid_x = r.x + r.w * 0.5f, mid_y = r.y + r.h * 0.5f;
    const int wx = x0 - win.ax, wy = y0 - win.ay;
    bool any = false;
    // cell k of the walk against box `at`: does it take, and where does it put the box (y in the first pass, x in the second)
    auto judge = [&](int k, const Box& at, bool second, float& put) {
        const int dy = k / 3, dx = k - 3 * dy;
        const bool in = (x0 + dx <= x1) & (y0 + dy <= y1);
        const int ox_ = in ? dx : 0, oy_ = in ? dy : 0;  // (a cell past the box: look at the first one, ignore it)
        const bool is_solid = solid(win.cell(wx + ox_, wy + oy_));
        const Box cell{static_cast<float>(x0 + ox_), static_cast<float>(y0 + oy_), 1.0f, 1.0f};
        const Box o = box_overlap_flat(at, cell);
        const bool some = in & is_solid & !((o.w == 0.0f) & (o.h == 0.0f));
        put = second ? (o.x + o.w * 0.5f > mid_x ? cell.x - at.w : cell.x + cell.w)
                     : (o.y + o.h * 0.5f > mid_y ? cell.y - at.h : cell.y + cell.h);
        return some & (second ? (o.w <= o.h) : (o.w > o.h));
    };
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        const bool second = pass == 1;
        int settled = 0;  // cells below this one have had their say
        for (;;) {        // (gang-uniform)
            float put;
            const bool take = (q.g >= settled) & judge(q.g, r, second, put);
            const uint32_t takers = q.ballot(take);
            if (takers == 0u) break;
            const int first = __ffs(takers) - 1;
            const float moved = __shfl(put, first, 8);
            if (second)
                r.x = moved;
            else
                r.y = moved;
            any = true;
            settled = first + 1;
        }
        float put;
        if (judge(8, r, second, put)) {
            if (second)
                r.x = put;
            else
                r.y = put;
            any = true;
        }
    }
    return {r.x, r.y, any};
}

}  // namespace pg
