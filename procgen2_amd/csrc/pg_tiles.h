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
        // Sixteen UNCONDITIONAL loads — a cell outside the map reads cell 0 and is replaced afterwards — so that they are
        // all in flight together.  (A load behind the bounds test is a load in a branch of its own, and the compiler
        // waits for each before it enters the next: sixteen memory round trips instead of one.)
        int t[16];
        bool inside[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int x = ax + (k & 3), ty = H - 1 - (ay + (k >> 2));
            inside[k] = !(x < 0 || ty < 0 || x >= W || ty >= H);
            t[k] = tiles[inside[k] ? ty + x * H : 0];
        }
#pragma unroll
        for (int k = 0; k < 16; k++) w.bits |= static_cast<uint64_t>(inside[k] ? (t[k] & 7) : OOB) << (3 * k);
        return w;
    }
    PG_D int at(int x, int y) const {
        const unsigned dx = static_cast<unsigned>(x - ax), dy = static_cast<unsigned>(y - ay);
        if (dx < 4u && dy < 4u) return static_cast<int>((bits >> (3 * (dx + 4 * dy))) & 7u);
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
                const bool is_solid = solid(static_cast<int>((win.bits >> (3 * ((wx + ox_) + 4 * (wy + oy_)))) & 7u));
                cell.x = static_cast<float>(x0 + ox_);
                cell.y = static_cast<float>(y0 + oy_);
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
                const bool is_solid = solid(static_cast<int>((win.bits >> (3 * ((wx + ox_) + 4 * (wy + oy_)))) & 7u));
                cell.x = static_cast<float>(x0 + ox_);
                cell.y = static_cast<float>(y0 + oy_);
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

}  // namespace pg
