// The render pre-pass: everything a frame needs that depends on the env's state but not on the pixel, worked out DENSELY —
// lane = (env, grid column | grid row), (env, pixel column | pixel row), (env, cell), (env, draw) — by a kernel of its
// own between the logic kernels and the render kernel, and handed to the env's render workgroup through device memory.
//
// Why: the render kernels are bound by vector instructions, and more than half of coinrun's (and jumper's) were this
// set-up — Renderer::render_texture's exact float divisions for sixteen tile columns, sixteen tile rows and a few dozen
// sprites (games/*/renderer.cpp:5-82), then the tables that turn spans into per-pixel candidates — executed by a
// workgroup whose 128 lanes stand for 128 pixels, at a quarter of the lanes or less (pg_render.h compose_spans,
// compose_hand_build, resolve_draw in the sprite pass).  Here the same arithmetic runs with every lane busy, once per env
// instead of once per wave, and the render wave starts from what it used to end its set-up with (pg_render.h
// ComposeRegs + the cell table + its draws, resolved and culled).  The price is ≈ 1 KB written and read again per env
// (the observation it precedes is 12 KB), kept small by packing: a pixel column or row is one word.
//
// A frame the scheme does not cover (a span wider than MAXSPAN, a third covering tile, more than 64 visible draws, a
// game-specific rare case) is marked `fat`: its render workgroup then takes the game's complete path, as before.
#pragma once

#include "pg_render.h"

namespace pg {

constexpr int kPrepMetaWords = 32;  // one 128-byte line per env
constexpr int kPrepKinds = 16;      // tile kinds a game may have (cell byte → byte offset of the kind's texture)
constexpr int kPrepDraws = 64;      // visible draws a lean frame may have (one per lane of the render wave)
enum PrepMetaWord {
    PM_SECOND = 0,  // 64-bit row masks (lo, hi): two grid rows cover the pixel row
    PM_SOFT = 2,    //   … a covering grid row (or the background) shows texels that are not opaque
    PM_HARD = 4,    //   … so many that the one-texel attempt is not made
    PM_FLAGS = 6,   // bit 0: fat (take the complete path); bits 8-15: number of draws
    PM_BGX = 7,     // background, x axis: d0 | dn << 16 (signed halves), then s0 | sn << 16
    PM_BGY = 9,     // background, y axis
    PM_BGTEX = 11,  // background: first texel in the atlas
    PM_WIDTHS = 12, // background texture width | tile texture width << 16
    PM_KINDS = 16,  // kPrepKinds words: byte offset in the atlas of each tile kind's texture
};

// Where the pre-pass leaves its results (device memory, per env; not part of the state: nothing survives a frame).
struct PrepOut {
    uint32_t* axes;   // [n][128]        pixel columns 0-63, pixel rows 0-63: texel a | texel b << 8 | grid index a << 16 |
                      //                 has a << 24 | has b << 25   (b = a + 1 by construction)
    uint32_t* axes2;  // [n][64]         (games with a second tile texture) pixel rows in it: texel a | texel b << 8 | has bits
    uint8_t* cells;   // [n][GRID*GRID]  tile kind of every grid cell of the window, 0xff = none
    uint32_t* meta;   // [n][kPrepMetaWords]
    uint32_t* draws;  // [n][kPrepDraws][kBlitWords]  (+ 2 words for games with rotated draws: see the game)
};
inline size_t prep_bytes(int n, int grid, int draw_words, bool second) {
    auto up = [](size_t b) { return (b + 255) & ~size_t(255); };
    return up(size_t(n) * 128 * 4) + up(second ? size_t(n) * 64 * 4 : 0) + up(size_t(n) * grid * grid) +
           up(size_t(n) * kPrepMetaWords * 4) + up(size_t(n) * kPrepDraws * draw_words * 4);
}
inline PrepOut prep_bind(void* base, int n, int grid, int draw_words, bool second) {
    auto up = [](size_t b) { return (b + 255) & ~size_t(255); };
    uint8_t* p = static_cast<uint8_t*>(base);
    PrepOut o{};
    o.axes = reinterpret_cast<uint32_t*>(p);
    p += up(size_t(n) * 128 * 4);
    o.axes2 = second ? reinterpret_cast<uint32_t*>(p) : nullptr;
    p += up(second ? size_t(n) * 64 * 4 : 0);
    o.cells = p;
    p += up(size_t(n) * grid * grid);
    o.meta = reinterpret_cast<uint32_t*>(p);
    p += up(size_t(n) * kPrepMetaWords * 4);
    o.draws = reinterpret_cast<uint32_t*>(p);
    return o;
}

#if defined(__HIPCC__)

// What the game's pre-pass kernel knows about one env's frame before the generic phases run (in LDS, one per env of the
// workgroup): camera, tile window, the tile layer's texture size and scale, the background's draw.
struct PrepView {
    Camera cam;
    int32_t x0, y0, cols, rows;
    int32_t tw, th, th2;  // tile texture; th2 > 0: a second, shorter one (pg_render.h compose_spans)
    float tile_scale;
    BgDraw bg;
    int32_t active;       // 0: this env takes no part (masked out, beyond n, or already known to be fat)
};

template <int GRID, int E>
struct PrepLds {
    PrepView view[E];
    int4 span[E][2][GRID];   // [env][axis][grid index]: d0, dn, s0, sn (sn = 0: nothing)
    int4 span2[E][GRID];     // rows of the second texture
    uint32_t cover[E][2][64];
    uint32_t soft_rows[E], hard_rows[E];  // bit r: grid row r shows soft / hard texels; bit 31: the background does
    uint32_t fat[E];
    uint32_t meta[E][kPrepMetaWords];
};

PG_D uint32_t pack_halves(int lo, int hi) { return (static_cast<uint32_t>(lo) & 0xffffu) | (static_cast<uint32_t>(hi) << 16); }

// Phase A — lane = (env, axis, grid index) and two more per env for the background's axes: render_texture's arithmetic
// for one axis of one tile column / row (pg_geom.h resolve_axis), then every span marks the pixels it covers with its
// grid index.  P.cover and P.fat must have been cleared (and a barrier passed); leaves a barrier to the caller.
template <int GRID, int MAXSPAN, int E>
PG_D void prep_spans(PrepLds<GRID, E>& P, int tid, int nthreads) {
    static_assert(GRID <= 30, "grid indices are bits of a word (bit 31: the background's class)");
    constexpr int kPer = 2 * GRID + 2;
    for (int q = tid; q < E * kPer; q += nthreads) {
        const int e = q / kPer, k = q - e * kPer;
        const PrepView& v = P.view[e];
        if (!v.active) continue;
        const bool is_bg = k >= 2 * GRID;
        const int axis = is_bg ? k - 2 * GRID : k / GRID;
        const int g = k - axis * GRID;
        const int count = axis == 0 ? v.cols : v.rows;
        Span sp;
        sp.d0 = sp.dn = sp.s0 = sp.sn = 0;
        bool ok = false;
        if (is_bg || g < count) {
            const int ts = is_bg ? (axis == 0 ? v.bg.desc.y : v.bg.desc.z) : (axis == 0 ? v.tw : v.th);
            const float pos = is_bg ? (axis == 0 ? v.bg.px : v.bg.py) : ((axis == 0 ? v.x0 : v.y0) + g) * kUnitPx;
            ok = resolve_axis(axis == 0 ? v.cam.px : v.cam.py, axis == 0 ? v.cam.sw : v.cam.sh, v.cam.scale, ts, pos,
                              is_bg ? v.bg.scale : v.tile_scale, false, axis == 1, sp);
        }
        if (is_bg) {  // as compose_spans hands it out: destination emptied when the axis draws nothing, source as computed
            P.meta[e][PM_BGX + 2 * axis] = pack_halves(ok ? sp.d0 : 0, ok ? sp.dn : 0);
            P.meta[e][PM_BGX + 2 * axis + 1] = pack_halves(sp.s0, sp.sn);
            continue;
        }
        P.span[e][axis][g] = ok ? make_int4(sp.d0, sp.dn, sp.s0, sp.sn) : make_int4(0, 0, 0, 0);
        bool wide = ok && sp.dn > MAXSPAN;
        if (axis == 1 && v.th2 > 0) {
            Span s2;
            s2.d0 = s2.dn = s2.s0 = s2.sn = 0;
            const bool ok2 = g < count && resolve_axis(v.cam.py, v.cam.sh, v.cam.scale, v.th2, (v.y0 + g) * kUnitPx,
                                                       v.tile_scale, false, true, s2);
            P.span2[e][g] = ok2 ? make_int4(s2.d0, s2.dn, s2.s0, s2.sn) : make_int4(0, 0, 0, 0);
            wide = wide || (ok2 && (!ok || s2.d0 != sp.d0 || s2.dn > sp.dn));  // not nested: the complete path
        }
        if (wide) {
            atomicOr(&P.fat[e], 1u);
            continue;
        }
        if (ok)
            for (int i = 0; i < MAXSPAN; i++) {
                const int p = sp.d0 + i;
                if (i < sp.dn && p >= 0 && p < 64) atomicOr(&P.cover[e][axis][p], 1u << g);
            }
    }
}

// Phase B — one wavefront pass = the 64 pixel columns or the 64 pixel rows of one env: the (at most two, adjacent)
// covering grid indices and their texel coordinates as one word, and, from the row pass, the three row-class masks
// (pg_render.h compose_hand_build: same rules).  P.soft_rows / P.hard_rows must be complete (the game's cell phase).
template <int GRID, int E>
PG_D void prep_axes(PrepLds<GRID, E>& P, const PrepOut& out, int env0, int wave, int nwaves, int lane) {
    for (int blk = wave; blk < 2 * E; blk += nwaves) {  // wave-uniform
        const int e = blk >> 1, axis = blk & 1;
        const PrepView& v = P.view[e];
        if (!v.active) continue;
        const uint32_t m = P.cover[e][axis][lane];
        const int n = __popc(m);
        const int ia = n >= 1 ? __builtin_ctz(m) : -1;
        const uint32_t m2 = m & (m - 1u);
        const int ib = n >= 2 ? __builtin_ctz(m2) : -1;
        const bool ok = n <= 2 && !(ib >= 0 && ib != ia + 1);
        int ta = 0, tb = 0;
        if (ia >= 0) {
            const int4 sp = P.span[e][axis][ia];
            ta = sample_index(sp.z, sp.w, lane - sp.x, sp.y);
        }
        if (ib >= 0) {
            const int4 sp = P.span[e][axis][ib];
            tb = sample_index(sp.z, sp.w, lane - sp.x, sp.y);
        }
        const uint32_t word = static_cast<uint32_t>(ta & 0xff) | (static_cast<uint32_t>(tb & 0xff) << 8) |
                              (static_cast<uint32_t>(ia >= 0 ? ia : 0) << 16) | (ia >= 0 ? 1u << 24 : 0u) | (ib >= 0 ? 1u << 25 : 0u);
        out.axes[size_t(env0 + e) * 128 + axis * 64 + lane] = word;
        const bool any_bad = __ballot(!ok || ta > 255 || tb > 255) != 0;
        if (axis == 1) {
            if (v.th2 > 0) {  // texel rows in the layer's second, shorter texture
                uint32_t w2 = 0;
                if (ia >= 0) {
                    const int4 sp = P.span2[e][ia];
                    const int i = lane - sp.x;
                    if (sp.w > 0 && i >= 0 && i < sp.y) w2 |= static_cast<uint32_t>(sample_index(sp.z, sp.w, i, sp.y) & 0xff) | (1u << 24);
                }
                if (ib >= 0) {
                    const int4 sp = P.span2[e][ib];
                    const int i = lane - sp.x;
                    if (sp.w > 0 && i >= 0 && i < sp.y) w2 |= (static_cast<uint32_t>(sample_index(sp.z, sp.w, i, sp.y) & 0xff) << 8) | (1u << 25);
                }
                out.axes2[size_t(env0 + e) * 64 + lane] = w2;
            }
            const uint32_t soft_bits = P.soft_rows[e], hard_bits = P.hard_rows[e];
            const bool soft_here = (soft_bits >> 31) != 0 || (ia >= 0 && ((soft_bits >> (ia & 31)) & 1u)) ||
                                   (ib >= 0 && ((soft_bits >> (ib & 31)) & 1u));
            const bool hard_here = (hard_bits >> 31) != 0 || (ia >= 0 && ((hard_bits >> (ia & 31)) & 1u)) ||
                                   (ib >= 0 && ((hard_bits >> (ib & 31)) & 1u));
            const unsigned long long m_second = __ballot(ib >= 0), m_soft = __ballot(soft_here), m_hard = __ballot(hard_here);
            if (lane == 0) {
                P.meta[e][PM_SECOND] = static_cast<uint32_t>(m_second);
                P.meta[e][PM_SECOND + 1] = static_cast<uint32_t>(m_second >> 32);
                P.meta[e][PM_SOFT] = static_cast<uint32_t>(m_soft);
                P.meta[e][PM_SOFT + 1] = static_cast<uint32_t>(m_soft >> 32);
                P.meta[e][PM_HARD] = static_cast<uint32_t>(m_hard);
                P.meta[e][PM_HARD + 1] = static_cast<uint32_t>(m_hard >> 32);
            }
        }
        if (any_bad && lane == 0) atomicOr(&P.fat[e], 1u);
    }
}

// The last phase: the envs' meta lines, coalesced.  `counts[e]` = draws of env e (> kPrepDraws: fat).  Needs a barrier
// in front (every phase has written its part of P.meta / P.fat).
template <int GRID, int E>
PG_D void prep_meta_out(PrepLds<GRID, E>& P, const PrepOut& out, int env0, const int32_t* counts, int tid, int nthreads) {
    for (int q = tid; q < E * kPrepMetaWords; q += nthreads) {
        const int e = q / kPrepMetaWords, w = q - e * kPrepMetaWords;
        if (!P.view[e].active && !P.fat[e]) continue;  // (an env the kernel sits out writes nothing; a fat one its flag)
        uint32_t word = P.meta[e][w];
        if (w == PM_FLAGS) {
            const int c = counts[e];
            const bool fat = P.fat[e] != 0 || c > kPrepDraws;
            word = (fat ? 1u : 0u) | (static_cast<uint32_t>(fat ? 0 : c) << 8);
        }
        if (w == PM_BGTEX) word = static_cast<uint32_t>(P.view[e].bg.desc.x);
        if (w == PM_WIDTHS) word = pack_halves(P.view[e].bg.desc.y, P.view[e].tw);
        out.meta[size_t(env0 + e) * kPrepMetaWords + w] = word;
    }
}

// ---- the render wave's side --------------------------------------------------------------------------------------

// One env's meta line as the render workgroup sees it (wave-uniform: scalar loads).
struct PrepMeta {
    const uint32_t* w;
    PG_D uint32_t flags() const { return w[PM_FLAGS]; }
    PG_D bool fat() const { return (w[PM_FLAGS] & 1u) != 0; }
    PG_D int draws() const { return static_cast<int>((w[PM_FLAGS] >> 8) & 0xffu); }
    PG_D unsigned long long mask(int at) const { return static_cast<unsigned long long>(w[at]) | (static_cast<unsigned long long>(w[at + 1]) << 32); }
    PG_D BgAxis bg(int axis) const {
        const uint32_t a = w[PM_BGX + 2 * axis], b = w[PM_BGX + 2 * axis + 1];
        return BgAxis{static_cast<int32_t>(a << 16) >> 16, static_cast<int32_t>(a) >> 16, static_cast<int32_t>(b & 0xffffu),
                      static_cast<int32_t>(b >> 16), static_cast<int32_t>(w[PM_BGTEX]), static_cast<int32_t>(w[PM_WIDTHS] & 0xffffu)};
    }
    PG_D int tile_w() const { return static_cast<int>(w[PM_WIDTHS] >> 16); }
};

// The composer's registers from the packed words (lane = pixel column and pixel row).
template <int GRID>
PG_D ComposeRegs prep_regs(const PrepMeta& M, uint32_t colw, uint32_t roww, uint32_t roww2, int lane) {
    ComposeRegs R;
    const uint32_t tw4 = static_cast<uint32_t>(M.tile_w()) * 4u;
    R.col_a = (colw & (1u << 24)) ? (colw & 0xffu) * 4u : kNoTexel;
    R.col_b = (colw & (1u << 25)) ? ((colw >> 8) & 0xffu) * 4u : kNoTexel;
    R.cia4 = ((colw >> 16) & 0xffu) * 4u;  // (0 when no grid column covers the pixel)
    R.row_a = (roww & (1u << 24)) ? (roww & 0xffu) * tw4 : kNoTexel;
    R.row_b = (roww & (1u << 25)) ? ((roww >> 8) & 0xffu) * tw4 : kNoTexel;
    R.cells_a = ((roww & (1u << 24)) ? ((roww >> 16) & 0xffu) * static_cast<uint32_t>(GRID) : static_cast<uint32_t>(GRID * GRID)) * 4u;
    R.row_a2 = (roww2 & (1u << 24)) ? (roww2 & 0xffu) * tw4 : kNoTexel;
    R.row_b2 = (roww2 & (1u << 25)) ? ((roww2 >> 8) & 0xffu) * tw4 : kNoTexel;
    R.col_pa = R.col_pb = 0;
    R.bg_col = bg_offset(M.bg(0), lane, 0);
    R.bg_row = bg_offset(M.bg(1), lane, 1);
    R.second_row = M.mask(PM_SECOND);
    R.soft = M.mask(PM_SOFT);
    R.hard = M.mask(PM_HARD);
    return R;
}

// A resolved draw as the pre-pass stores it: pg_render.h BlitWords, kBlitWords per draw, draws of an env back to back.
PG_D void prep_draw_store(uint32_t* at, const Blit& b) {
    const BlitWords p = blit_pack(b);
    uint2* q = reinterpret_cast<uint2*>(at);  // (24-byte records: 8-byte aligned)
    q[0] = make_uint2(p.w[0], p.w[1]);
    q[1] = make_uint2(p.w[2], p.w[3]);
    q[2] = make_uint2(p.w[4], p.w[5]);
}
PG_D Blit prep_draw_load(const uint32_t* at, bool has) {
    uint2 a = make_uint2(0, 0), b = make_uint2(0, 0), c = make_uint2(0, 0);
    if (has) {
        const uint2* q = reinterpret_cast<const uint2*>(at);
        a = q[0];
        b = q[1];
        c = q[2];
    }
    Blit d;
    d.dx = static_cast<int32_t>(a.x << 16) >> 16;
    d.dy = static_cast<int32_t>(a.x) >> 16;
    d.dw = static_cast<int32_t>(a.y & 0xffffu);
    d.dh = static_cast<int32_t>(a.y >> 16);
    d.sx = static_cast<int32_t>(b.x & 0xffffu);
    d.sy = static_cast<int32_t>(b.x >> 16);
    d.sw = static_cast<int32_t>(b.y & 0xffffu);
    d.sh = static_cast<int32_t>(b.y >> 16);
    d.tex_off = static_cast<int32_t>(c.x);
    d.tex_w = static_cast<int32_t>(c.y & 0xffffu);
    d.flip_mod = static_cast<int32_t>(c.y >> 16);
    d.rot_sn = 0;
    d.rot_cs = 65536;
    return d;
}

#endif  // __HIPCC__

}  // namespace pg
