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

template <class Win, class Pred>
PG_D TileHit collide_plain(const Win& win, Box r, Pred solid) {
    bool any = false;
    const int x0 = static_cast<int>(floorf(r.x)), y0 = static_cast<int>(floorf(r.y));
    const int x1 = static_cast<int>(ceilf(r.x + r.w)), y1 = static_cast<int>(ceilf(r.y + r.h));
    const float mid_x = r.x + r.w * 0.5f, mid_y = r.y + r.h * 0.5f;
    Box cell{0.0f, 0.0f, 1.0f, 1.0f};
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
