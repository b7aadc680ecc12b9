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
#include "pg_stamps.h"

namespace pg {

constexpr int kPrepMetaWords = 32;  // one 128-byte line per env
constexpr int kPrepKinds = 8;       // tile kinds a game may have (cell byte → byte offset of the kind's texture)
constexpr int kPrepDraws = 64;      // visible draws a lean frame may have (one per lane of the render wave)
enum PrepMetaWord {
    PM_SECOND = 0,  // 64-bit row masks (lo, hi): two grid rows cover the pixel row
    PM_SOFT = 2,    //   … a covering grid row (or the background) shows texels that are not opaque
    PM_HARD = 4,    //   … so many that the one-texel attempt is not made
    PM_FLAGS = 6,   // bit 0: fat (take the complete path); bit 1: the layer has a second, shorter texture (kind 0 shows it);
                    // bits 8-15: number of draws
    PM_BGX = 7,     // background, x axis: d0 | dn << 16 (signed halves), then s0 | sn << 16
    PM_BGY = 9,     // background, y axis
    PM_BGTEX = 11,  // background: first texel in the atlas
    PM_WIDTHS = 12, // background texture width | tile texture width << 16
    PM_KINDS = 16,  // kPrepKinds words: byte offset in the atlas of each tile kind's texture
    PM_GAME = 24,   // eight words of the game's own (jumper: its compass as it lands on the observation)
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

// One tile column / row (or row of the layer's second texture) as the pixel lanes need it: destination start and
// extent, and WHICH table holds its texel coordinates — d0 (signed 16 bits) | dn << 16 | table << 24; dn = 0: nothing
// drawn, dn = kPrepInterior: filled in from the axis's template (below).
constexpr uint32_t kPrepInterior = 0xffu;
constexpr int kPrepList = 128;  // full tails a workgroup works out per frame (edge tiles, templates, backgrounds)
struct PrepTail {               // one of them, waiting: the head of render_texture's arithmetic has been done
    float d, dl;
    uint32_t code;              // env | axis << 4 | grid index << 8 | kind << 16
};
enum { kTailEdge = 0, kTailTemplate = 1, kTailBackground = 2 };

template <int GRID, int E, int MAXSPAN = kMaxSpan>
struct PrepLds {
    PrepView view[E];
    uint32_t span[E][3][GRID];   // [env][axis: 0 columns, 1 rows, 2 rows of the second texture][grid index]
    uint32_t tmpl[E][3];         // the axis's template: dn << 16 | table << 24 (0: no tile of the axis is drawn whole)
    PrepTail list[kPrepList];
    int32_t list_n;
    uint8_t texel[kPrepList][MAXSPAN];  // texel coordinate of destination offset i, per worked-out tail
    uint32_t cover[E][2][64];
    uint32_t soft_rows[E], hard_rows[E];  // bit r: grid row r shows soft / hard texels; bit 31: the background does
    uint32_t fat[E];  // why an env takes the complete path — 1: the game's view, 2: worklist full, 4: a span beyond the tables, 8: pixel candidates not adjacent
    uint32_t meta[E][kPrepMetaWords];
};

// The pre-pass's results are PLAIN stores.  (Measured and rejected, round 5: as streaming `nt` stores — the thought was
// that the 145 MB a launch leaves dirty in the L2s are written back at the kernel's end, in front of the render launch,
// which rocprofv3 shows 19 µs longer behind a pre-pass than behind none — coinrun's render kernel went from 0.333 to
// 0.398 ms, climber's 0.334 -> 0.395: what the render workgroups read a few microseconds later then comes from HBM
// instead of the Infinity Cache the write-back leaves it in.)
#define PG_PREP_STORE(value, ptr) (*(ptr) = (value))
typedef uint32_t prep_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t prep_u32x2 __attribute__((ext_vector_type(2)));

// Which group of envs a workgroup of the pre-pass takes.  The hardware deals workgroups to the eight XCDs in turn
// (workgroup b → XCD b mod 8, each with an L2 of its own), and the pre-pass reads struct-of-arrays state — a group's eight
// envs are 32 bytes of a 128-byte line of every float field, 8 bytes of every byte field — so with group = b the
// workgroups that share a line sit on different XCDs and every one of those L2s fetches the whole line.  Dealt this way the
// groups of ONE XCD are consecutive: the line is fetched once.  (Any mapping gives the same frames; n_groups not a
// multiple of 8: the tail keeps its place.)
PG_D int prep_block(int b, int n_groups) {
    const int per = n_groups >> 3, body = per << 3;
    return b < body ? (b & 7) * per + (b >> 3) : b;
}

PG_D uint32_t pack_halves(int lo, int hi) { return (static_cast<uint32_t>(lo) & 0xffffu) | (static_cast<uint32_t>(hi) << 16); }

// Phase A — the spans of the tile grid and of the background.
//
// render_texture's arithmetic per axis is a head (destination before cropping, cull: three operations) and a tail (the
// crops against the screen, five exact divisions, the source snap: ≈ 170 vector instructions).  For every tile that
// lies wholly on the screen the tail's inputs are the same — it depends on the destination's START only through the
// crops, which are not taken, and through `d -= off * (dl / sl)` with off = 0 — so all those tiles of an axis share
// one result but for d0 = (int)d: the axis's TEMPLATE, worked out once from a head with d = 0.  Only the tiles the screen
// edge cuts (two per axis) need a tail of their own.  So: step 1, lane = (env, axis, grid index): the head; whole tiles
// note their d0, cut ones queue up.  Step 2: the queue — cut tiles, one template per axis, the background's two axes —
// 64 full tails per wavefront pass, each also tabulating its texel coordinates.  Step 3: whole tiles take their
// template's extent and table; every span marks the pixels it covers with its grid index.
// P.cover must have been cleared; the caller has a barrier in front (views complete) and puts one behind.
template <int GRID, int MAXSPAN, int E, bool TWO = false>
PG_D void prep_spans(PrepLds<GRID, E, MAXSPAN>& P, int tid, int nthreads) {
    static_assert(GRID <= 30, "grid indices are bits of a word (bit 31: the background's class)");
    if (tid == 0) P.list_n = 0;
    __syncthreads();
    auto axis_params = [&](const PrepView& v, int axis, int kind, float& cam_pos, float& cam_len, int& tsize, float& scale) {
        const bool x = axis == 0;
        cam_pos = x ? v.cam.px : v.cam.py;
        cam_len = x ? v.cam.sw : v.cam.sh;
        if (kind == kTailBackground) {
            tsize = x ? v.bg.desc.y : v.bg.desc.z;
            scale = v.bg.scale;
        } else {
            tsize = x ? v.tw : (axis == 1 ? v.th : v.th2);
            scale = v.tile_scale;
        }
    };
    auto enqueue = [&](float d, float dl, uint32_t code) {
        const int at = atomicAdd(&P.list_n, 1);
        if (at < kPrepList)
            P.list[at] = PrepTail{d, dl, code};
        else
            atomicOr(&P.fat[code & 15u], 2u);  // (never seen: a frame with that many cut tiles takes the complete path)
    };
    // ---- step 1: the tiles …
    constexpr int kAxes = TWO ? 3 : 2, kTiles = kAxes * GRID;
    for (int q = tid; q < E * kTiles; q += nthreads) {
        const int e = q / kTiles, k = q - e * kTiles;
        const PrepView& v = P.view[e];
        if (!v.active) continue;
        const int axis = k / GRID, g = k - axis * GRID;
        const int count = axis == 0 ? v.cols : v.rows;
        uint32_t word = 0u;
        if (g < count && !(axis == 2 && v.th2 <= 0)) {
            float cam_pos, cam_len, scale;
            int tsize;
            axis_params(v, axis, kTailEdge, cam_pos, cam_len, tsize, scale);
            AxisHead h;
            if (axis_head(cam_pos, cam_len, v.cam.scale, tsize, ((axis == 0 ? v.x0 : v.y0) + g) * kUnitPx, scale, axis != 0, h)) {
                if (!(h.d < 0.0f) && !(h.d + h.dl > cam_len)) {  // neither crop of axis_tail is taken: a whole tile
                    const bool fits = h.d > -32768.0f && h.d < 32768.0f;   // (S1, as the tail would ask)
                    word = fits ? pack_halves(static_cast<int>(h.d), static_cast<int>(kPrepInterior)) : 0u;
                } else {
                    enqueue(h.d, h.dl, static_cast<uint32_t>(e) | (axis << 4) | (g << 8) | (kTailEdge << 16));
                }
            }
        }
        P.span[e][axis][g] = word;  // (a cut tile's word is written in step 2)
    }
    // … and per env the templates of its axes and the background's two axes
    for (int q = tid; q < E * (kAxes + 2); q += nthreads) {
        const int e = q / (kAxes + 2), j = q - e * (kAxes + 2);
        const PrepView& v = P.view[e];
        if (!v.active) continue;
        const int axis = j < kAxes ? j : j - kAxes, kind = j < kAxes ? kTailTemplate : kTailBackground;
        if (kind == kTailTemplate && axis == 2 && v.th2 <= 0) continue;
        float cam_pos, cam_len, scale;
        int tsize;
        axis_params(v, axis, kind, cam_pos, cam_len, tsize, scale);
        AxisHead h{0.0f, 0.0f};
        bool alive = true;
        if (kind == kTailBackground) {
            alive = axis_head(cam_pos, cam_len, v.cam.scale, tsize, axis == 0 ? v.bg.px : v.bg.py, scale, axis == 1, h);
            if (!alive) {  // culled on this axis: as compose_spans hands it out
                P.meta[e][PM_BGX + 2 * axis] = 0u;
                P.meta[e][PM_BGX + 2 * axis + 1] = 0u;
            }
        } else {
            h.dl = tsize * scale * v.cam.scale;  // (axis_head's expression; d = 0)
            P.tmpl[e][axis] = 0u;
            alive = !(h.d + h.dl > cam_len);  // (a tile larger than the screen is never whole: no template)
        }
        if (alive) enqueue(h.d, h.dl, static_cast<uint32_t>(e) | (axis << 4) | (kind << 16));
    }
    __syncthreads();
    // ---- step 2
    const int n_list = P.list_n < kPrepList ? P.list_n : kPrepList;
    for (int q = tid; q < n_list; q += nthreads) {
        const PrepTail t = P.list[q];
        const int e = t.code & 15u, axis = (t.code >> 4) & 15u, g = (t.code >> 8) & 0xffu, kind = (t.code >> 16) & 3u;
        const PrepView& v = P.view[e];
        float cam_pos, cam_len, scale;
        int tsize;
        axis_params(v, axis, kind, cam_pos, cam_len, tsize, scale);
        Span sp;
        sp.d0 = sp.dn = sp.s0 = sp.sn = 0;
        const bool ok = axis_tail(cam_len, v.cam.scale, tsize, scale, false, AxisHead{t.d, t.dl}, sp);
        if (kind == kTailBackground) {
            P.meta[e][PM_BGX + 2 * axis] = pack_halves(ok ? sp.d0 : 0, ok ? sp.dn : 0);
            P.meta[e][PM_BGX + 2 * axis + 1] = pack_halves(sp.s0, sp.sn);
            continue;
        }
        uint32_t word = 0u;
        if (ok) {
            bool bad = sp.dn > MAXSPAN || sp.d0 < -32768 || sp.d0 > 32767;
            for (int i = 0; i < MAXSPAN; i++) {
                const int u = i < sp.dn ? sample_index(sp.s0, sp.sn, i, sp.dn) : 0;
                bad = bad || u > 255;
                P.texel[q][i] = static_cast<uint8_t>(u);
            }
            if (bad) atomicOr(&P.fat[e], 4u);
            word = bad ? 0u : (pack_halves(sp.d0, sp.dn) | (static_cast<uint32_t>(q) << 24));
        }
        if (kind == kTailTemplate)
            P.tmpl[e][axis] = word & 0xffff0000u;
        else
            P.span[e][axis][g] = word;
    }
    __syncthreads();
    // ---- step 3
    for (int q = tid; q < E * kTiles; q += nthreads) {
        const int e = q / kTiles, k = q - e * kTiles;
        const PrepView& v = P.view[e];
        if (!v.active) continue;
        const int axis = k / GRID, g = k - axis * GRID;
        uint32_t w = P.span[e][axis][g];
        if (((w >> 16) & 0xffu) == kPrepInterior) {
            const uint32_t t = P.tmpl[e][axis];
            w = t ? ((w & 0xffffu) | t) : 0u;
            P.span[e][axis][g] = w;
        }
        const int d0 = static_cast<int32_t>(w << 16) >> 16, dn = (w >> 16) & 0xffu;
        if (axis == 2) continue;  // (the second texture's rows are nested in the first's: checked by prep_axes)
        for (int i = 0; i < MAXSPAN; i++) {
            if (__ballot(i < dn) == 0) break;  // (no span of this wave's is that long)
            const int p = d0 + i;
            if (i < dn && p >= 0 && p < 64) atomicOr(&P.cover[e][axis][p], 1u << g);
        }
    }
}

// Phase B — one wavefront pass = the 64 pixel columns or the 64 pixel rows of one env: the (at most two, adjacent)
// covering grid indices and their texel coordinates as one word, and, from the row pass, the three row-class masks
// (pg_render.h compose_hand_build: same rules).  P.soft_rows / P.hard_rows must be complete (the game's cell phase).
template <int GRID, int MAXSPAN, int E>
PG_D void prep_axes(PrepLds<GRID, E, MAXSPAN>& P, const PrepOut& out, int env0, int wave, int nwaves, int lane) {
    for (int blk = wave; blk < 2 * E; blk += nwaves) {  // wave-uniform
        const int e = blk >> 1, axis = blk & 1;
        const PrepView& v = P.view[e];
        if (!v.active) continue;
        const uint32_t m = P.cover[e][axis][lane];
        const int n = __popc(m);
        const int ia = n >= 1 ? __builtin_ctz(m) : -1;
        const uint32_t m2 = m & (m - 1u);
        const int ib = n >= 2 ? __builtin_ctz(m2) : -1;
        bool ok = n <= 2 && !(ib >= 0 && ib != ia + 1);
        const uint32_t wa = ia >= 0 ? P.span[e][axis][ia] : 0u, wb = ib >= 0 ? P.span[e][axis][ib] : 0u;
        const int da = static_cast<int32_t>(wa << 16) >> 16, db = static_cast<int32_t>(wb << 16) >> 16;
        const int ta = ia >= 0 ? P.texel[wa >> 24][lane - da] : 0;
        const int tb = ib >= 0 ? P.texel[wb >> 24][lane - db] : 0;
        const uint32_t word = static_cast<uint32_t>(ta) | (static_cast<uint32_t>(tb) << 8) |
                              (static_cast<uint32_t>(ia >= 0 ? ia : 0) << 16) | (ia >= 0 ? 1u << 24 : 0u) | (ib >= 0 ? 1u << 25 : 0u);
        PG_PREP_STORE(word, &out.axes[size_t(env0 + e) * 128 + axis * 64 + lane]);
        if (axis == 1) {
            if (v.th2 > 0) {  // texel rows in the layer's second, shorter texture: nested in the first's span, or the complete path
                uint32_t w2 = 0;
                if (ia >= 0) {
                    const uint32_t s2 = P.span[e][2][ia];
                    const int d2 = static_cast<int32_t>(s2 << 16) >> 16, n2 = (s2 >> 16) & 0xffu, i = lane - d2;
                    ok = ok && !(n2 > 0 && !span_nested(d2, n2, da, static_cast<int>((wa >> 16) & 0xffu), kObsH));
                    if (n2 > 0 && i >= 0 && i < n2) w2 |= static_cast<uint32_t>(P.texel[s2 >> 24][i]) | (1u << 24);
                }
                if (ib >= 0) {
                    const uint32_t s2 = P.span[e][2][ib];
                    const int d2 = static_cast<int32_t>(s2 << 16) >> 16, n2 = (s2 >> 16) & 0xffu, i = lane - d2;
                    ok = ok && !(n2 > 0 && !span_nested(d2, n2, db, static_cast<int>((wb >> 16) & 0xffu), kObsH));
                    if (n2 > 0 && i >= 0 && i < n2) w2 |= (static_cast<uint32_t>(P.texel[s2 >> 24][i]) << 8) | (1u << 25);
                }
                PG_PREP_STORE(w2, &out.axes2[size_t(env0 + e) * 64 + lane]);
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
        if (__ballot(!ok) != 0 && lane == 0) atomicOr(&P.fat[e], 8u);
    }
}

// ---- the cell table of a column-major byte map (tiles[ty + x * H], map row ty = H - 1 - y: coinrun/tilemap.h:62-85 and
// the other platformers alike).  The GRID = 16 cells of one window column are sixteen consecutive bytes: lane = (env, grid
// column) fetches them with ONE 16-byte load (unaligned; rows beyond the map's edge read the neighbouring column or the
// neighbouring env's map — inside the state's allocation either way — and are replaced afterwards).
// (GRID = 24, climber: 16 + 8 bytes.)
// Which of the GRID map rows ty_lo + i under a window exist (byte i = 0xff), ty_lo = H - GRID - y0: once per env.
template <int GRID, int H>
PG_D void prep_row_valid(int y0, uint32_t (&mw)[GRID / 4]) {
    const int ty_lo = H - GRID - y0;
#pragma unroll
    for (int w = 0; w < GRID / 4; w++) {
        mw[w] = 0u;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int ty = ty_lo + 4 * w + j;
            if (ty >= 0 && ty < H) mw[w] |= 0xffu << (8 * j);
        }
    }
}
template <int GRID, int W, int H>
PG_D void prep_column_fetch(const uint8_t* tiles, int x, int y0, bool& x_ok, uint32_t (&column)[GRID / 4]) {
#pragma unroll
    for (int w = 0; w < GRID / 4; w++) column[w] = 0u;
    x_ok = x >= 0 && x < W;
    if (x_ok) __builtin_memcpy(column, tiles + x * H + (H - GRID - y0), GRID);
}
// … the GRID raw bytes in window-row order (byte r of word r / 4 = window row r), `oob` where the map has no cell.
template <int GRID>
PG_D void prep_column_rows(const uint32_t (&column)[GRID / 4], const uint32_t* valid, bool x_ok, uint32_t oob, uint32_t (&rows)[GRID / 4]) {
    const uint32_t fill = 0x01010101u * oob;
    // byte i of the column is map row ty_lo + i = window row GRID - 1 - i: reverse
#pragma unroll
    for (int w = 0; w < GRID / 4; w++) {
        const uint32_t raw = column[GRID / 4 - 1 - w], ok = valid[GRID / 4 - 1 - w];
        rows[w] = __builtin_amdgcn_perm(0u, x_ok ? ((raw & ok) | (fill & ~ok)) : fill, 0x00010203u);
    }
}
template <int GRID>
PG_D void prep_column_store(uint8_t* cells_env, int c, const uint32_t (&kinds)[GRID / 4]) {
    uint32_t* at = reinterpret_cast<uint32_t*>(cells_env + c * GRID);
    if constexpr (GRID == 16) {
        prep_u32x4 v;
        v.x = kinds[0], v.y = kinds[1], v.z = kinds[2], v.w = kinds[3];
        PG_PREP_STORE(v, reinterpret_cast<prep_u32x4*>(at));
    } else {
#pragma unroll
        for (int w = 0; w < GRID / 4; w++) PG_PREP_STORE(kinds[w], &at[w]);
    }
}

#if defined(PG_FAT_WHY)
__device__ unsigned g_fat_why[6];
#endif
// The last phase: the envs' meta lines, coalesced.  `counts[e]` = draws of env e (> kPrepDraws: fat).  Needs a barrier
// in front (every phase has written its part of P.meta / P.fat).
template <int GRID, int MAXSPAN, int E>
PG_D void prep_meta_out(PrepLds<GRID, E, MAXSPAN>& P, const PrepOut& out, int env0, const int32_t* counts, int tid, int nthreads) {
    for (int q = tid; q < E * kPrepMetaWords; q += nthreads) {
        const int e = q / kPrepMetaWords, w = q - e * kPrepMetaWords;
        if (!P.view[e].active && !P.fat[e]) continue;  // (an env the kernel sits out writes nothing; a fat one its flag)
        // An env flagged fat before its view was built never had P.meta written: its line is zeros but for the words set
        // below (the render workgroup runs its lean preamble on the line before it asks whether the frame is fat).
        uint32_t word = P.view[e].active ? P.meta[e][w] : 0u;
        if (w == PM_FLAGS) {
            const int c = counts[e];
            const bool fat = P.fat[e] != 0 || c > kPrepDraws;
#if defined(PG_FAT_WHY)  // (diagnostic build, tools/build_exp.py GAME fatwhy -DPG_FAT_WHY: why frames are handed back, counted per launch)
            if (fat) {
                for (int b = 0; b < 4; b++)
                    if (P.fat[e] & (1u << b)) atomicAdd(&g_fat_why[b], 1u);
                if (c > kPrepDraws) atomicAdd(&g_fat_why[4], 1u);
                atomicAdd(&g_fat_why[5], 1u);
            }
            if (env0 == 0 && e == 0) {
                printf("handed back last launch: %u frames (view %u, worklist %u, span %u, candidates %u, draws %u)\n", g_fat_why[5], g_fat_why[0],
                       g_fat_why[1], g_fat_why[2], g_fat_why[3], g_fat_why[4]);
                for (int b = 0; b < 6; b++) g_fat_why[b] = 0u;
            }
#endif
            word = (fat ? 1u : 0u) | (P.view[e].th2 > 0 ? 2u : 0u) | (static_cast<uint32_t>(fat ? 0 : c) << 8);
        }
        if (w == PM_BGTEX) word = static_cast<uint32_t>(P.view[e].bg.desc.x);
        if (w == PM_WIDTHS) word = pack_halves(P.view[e].bg.desc.y, P.view[e].tw);
        PG_PREP_STORE(word, &out.meta[size_t(env0 + e) * kPrepMetaWords + w]);
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

PG_D void prep_draw_store(uint32_t* at, const Blit& b);

// ---- draws: cull first, finish the survivors densely ---------------------------------------------------------------
// One lane = one draw of the env's list.  Most draws of a level are off the screen (coinrun shows 13 of 64 tile columns)
// or dead particles, and render_texture's arithmetic behind its cull test is ten times that in front of it; so a lane
// first does the head of both axes (pg_geom.h axis_head: the reference's own cull, on the reference's own numbers) and
// the survivors queue up — in list order, the two envs of a wavefront one behind the other — in a small worklist, from
// which the tails are done 64 at a time.
constexpr int kPrepQueue = 64;
struct PrepDrawEntry {  // 24 bytes
    float dx, dlx, dy, dly;  // destination before cropping, both axes
    float scale;
    uint32_t misc;           // texture | flip_h << 8 | flip_v << 9 | alpha modulation << 16 | second env of the wave << 31
};
struct PrepDrawQueue {
    PrepDrawEntry e[kPrepQueue];
};
// What a game's lane knows about its draw before render_texture starts (renderer.cpp:5-7: texture, position, scale,
// alpha, flips); `go` false = no draw call at all.
struct PrepDraw {
    bool go, flip_h, flip_v;
    int tex;
    float wx, wy, scale, alpha;
};
// State of a wavefront's pass over the draw lists of its two envs.
struct PrepDrawPass {
    int queued;          // entries waiting in the worklist
    int done[2];         // draws stored so far, per env of the wave
    int touch[2];        // `cover` only: some draw that was kept reaches into the covered part's bounding box (cover[kObsH])
};
// The tails of everything queued, ranks, stores.  desc: the atlas descriptor table (in LDS); cam: the two envs' cameras.
// `cover` (optional): what a later, OPAQUE part of every frame hides — per pixel row y a word lo | hi << 8, the columns
// [lo, hi] that part overwrites (lo > hi: none), valley- / hill-shaped over the rows, so that a rectangle whose first and
// last row lie inside lies inside altogether.  A draw that lands wholly under it is dropped here: nobody sees it.
// (jumper's compass disc covers two thirds of the 64×64 frame, the bunny at its centre included.)  Word kObsH of the table
// is the covered part's bounding box, x_lo | x_hi << 8 | y_lo << 16 | y_hi << 24: whether any draw that is KEPT reaches into
// it is noted in st.touch — a frame none of whose draws does has nothing between its tile layer and the covering part
// that could have written on a covered pixel (pg_render.h compose_rows_from UNDER, overlay_rows).
// `stamps` (optional): the game's stamp table, kStampsPerTex entries per texture (pg_stamps.h) — a draw that takes its whole
// texture at a size a stamp exists for is stored with the stamp as its texture.
PG_D void prep_draws_flush(PrepDrawQueue& Q, PrepDrawPass& st, const int4* desc, const Camera& cam_a, const Camera& cam_b,
                           uint32_t* draws_a, uint32_t* draws_b, int lane, const uint32_t* cover = nullptr,
                           const uint4* stamps = nullptr) {
    if (st.queued == 0) return;  // wave-uniform
    wave_order();  // the entries were written by other lanes of this wave
    // Up to 32 draws (the usual case: a dozen or two per pair of envs): lane = (draw, axis) — both axes of every draw in
    // ONE trip through axis_tail instead of two at half the lanes; the even lane of a pair then holds x, takes y from its
    // neighbour and stores the draw.  More than 32: a lane per draw, the two axes one after the other.
    const bool paired = st.queued <= 32;
    const int j = paired ? lane >> 1 : lane, axis = paired ? lane & 1 : 0;
    const bool mine = j < st.queued;
    const PrepDrawEntry en = Q.e[mine ? j : 0];
    const bool is_b = (en.misc >> 31) != 0;
    const Camera& cam = is_b ? cam_b : cam_a;
    const int4 d = desc[en.misc & 0xffu];
    Span x, y;
    x.d0 = x.dn = x.s0 = x.sn = 0;
    y = x;
    bool has = mine;
    if (paired) {
        Span sp;
        sp.d0 = sp.dn = sp.s0 = sp.sn = 0;
        const bool ok = mine && axis_tail(axis ? cam.sh : cam.sw, cam.scale, axis ? d.z : d.y, en.scale, axis == 0 && ((en.misc >> 8) & 1u),
                                          axis ? AxisHead{en.dy, en.dly} : AxisHead{en.dx, en.dlx}, sp);
        x = sp;
        y.d0 = __shfl_xor(sp.d0, 1);
        y.dn = __shfl_xor(sp.dn, 1);
        y.s0 = __shfl_xor(sp.s0, 1);
        y.sn = __shfl_xor(sp.sn, 1);
        const bool ok_other = __shfl_xor(ok ? 1 : 0, 1) != 0;
        has = ok && ok_other && axis == 0;  // (the even lane of the pair speaks for the draw)
    } else {
        has = has && axis_tail(cam.sw, cam.scale, d.y, en.scale, (en.misc >> 8) & 1u, AxisHead{en.dx, en.dlx}, x);
        has = has && axis_tail(cam.sh, cam.scale, d.z, en.scale, false, AxisHead{en.dy, en.dly}, y);
    }
    bool touches = false;
    if (cover != nullptr && has) {  // (the part of the destination on the target: the rest is dropped anyway, raster spec S5)
        const int x0 = x.d0 < 0 ? 0 : x.d0, x1 = x.d0 + x.dn - 1 > kObsW - 1 ? kObsW - 1 : x.d0 + x.dn - 1;
        const int y0 = y.d0 < 0 ? 0 : y.d0, y1 = y.d0 + y.dn - 1 > kObsH - 1 ? kObsH - 1 : y.d0 + y.dn - 1;
        if (x0 <= x1 && y0 <= y1) {
            const uint32_t top = cover[y0], bottom = cover[y1];
            const int lo_t = static_cast<int>(top & 0xffu), lo_b = static_cast<int>(bottom & 0xffu);
            const int hi_t = static_cast<int>((top >> 8) & 0xffu), hi_b = static_cast<int>((bottom >> 8) & 0xffu);
            if (x0 >= (lo_t > lo_b ? lo_t : lo_b) && x1 <= (hi_t < hi_b ? hi_t : hi_b)) has = false;
            const uint32_t box = cover[kObsH];
            touches = has && x0 <= static_cast<int>((box >> 8) & 0xffu) && x1 >= static_cast<int>(box & 0xffu) &&
                      y0 <= static_cast<int>(box >> 24) && y1 >= static_cast<int>((box >> 16) & 0xffu);
        }
    }
    if (cover != nullptr) {  // (wave-uniform)
        if (__ballot(touches && !is_b)) st.touch[0] = 1;
        if (__ballot(touches && is_b)) st.touch[1] = 1;
    }
    const unsigned long long m_a = __ballot(has && !is_b), m_b = __ballot(has && is_b);
    const unsigned long long below = (1ull << lane) - 1ull;
    const int rank = is_b ? st.done[1] + __popcll(m_b & below) : st.done[0] + __popcll(m_a & below);
    if (has && rank < kPrepDraws) {
        Blit b;
        b.dx = x.d0, b.dw = x.dn, b.sx = x.s0, b.sw = x.sn;
        b.dy = y.d0, b.dh = y.dn, b.sy = y.s0, b.sh = y.sn;
        b.tex_off = d.x;
        b.tex_w = d.y;
        const uint32_t flips = (en.misc >> 8) & 3u;
        b.flip_mod = static_cast<int32_t>((en.misc >> 16) & 0xffu) | ((flips & 1u) ? kFlipH : ((flips & 2u) ? kFlipV : 0));  // (resolve_draw)
        b.rot_sn = 0;
        b.rot_cs = 65536;
        if (stamps != nullptr) stamp_substitute(stamps + (en.misc & 0xffu) * kStampsPerTex, d.y, d.z, b);
        prep_draw_store((is_b ? draws_b : draws_a) + size_t(rank) * kBlitWords, b);
    }
    st.done[0] += __popcll(m_a);
    st.done[1] += __popcll(m_b);
    st.queued = 0;
    wave_order();  // … and may be overwritten from here on
}
// One pass: this lane's draw (of env a or b of the wave) through the heads; survivors appended to the worklist in lane
// order (= list order).  Every lane of the wave calls this, with valid = false where there is no draw.
PG_D void prep_draws_pass(PrepDrawQueue& Q, PrepDrawPass& st, const int4* desc, const Camera& cam_a, const Camera& cam_b,
                          uint32_t* draws_a, uint32_t* draws_b, bool valid, bool is_b, const PrepDraw& p, int lane,
                          const uint32_t* cover = nullptr, const uint4* stamps = nullptr) {
    const Camera& cam = is_b ? cam_b : cam_a;
    AxisHead hx{0.0f, 0.0f}, hy{0.0f, 0.0f};
    bool alive = valid && p.go;
    if (alive) {
        const int4 d = desc[p.tex];
        alive = axis_head(cam.px, cam.sw, cam.scale, d.y, p.wx, p.scale, false, hx);
        alive = axis_head(cam.py, cam.sh, cam.scale, d.z, p.wy, p.scale, true, hy) && alive;
    }
    const unsigned long long m = __ballot(alive);
    const int n = __popcll(m);
    if (n == 0) return;  // wave-uniform
    if (st.queued + n > kPrepQueue) prep_draws_flush(Q, st, desc, cam_a, cam_b, draws_a, draws_b, lane, cover, stamps);
    if (alive) {
        int mod = 255;
        if (p.alpha != 1.0f) mod = static_cast<int>(255 * p.alpha) & 0xff;  // Uint8 parameter (renderer.cpp:56-57)
        PrepDrawEntry en;
        en.dx = hx.d, en.dlx = hx.dl, en.dy = hy.d, en.dly = hy.dl;
        en.scale = p.scale;
        en.misc = static_cast<uint32_t>(p.tex) | (p.flip_h ? 1u << 8 : 0u) | (p.flip_v ? 1u << 9 : 0u) |
                  (static_cast<uint32_t>(mod) << 16) | (is_b ? 1u << 31 : 0u);
        Q.e[st.queued + __popcll(m & ((1ull << lane) - 1ull))] = en;
    }
    st.queued += n;
}

// The cell table from the kind bytes the pre-pass stored column by column (16 × 16 tables: byte 2j, 2j + 1 = cells
// (column j / 8, rows 2j % 16 and + 1)): each of the workgroup's 128 lanes expands two cells.  `two16` = this lane's two
// bytes, kind_off = PM_KINDS word (lane & 7).  Leaves the barrier to the caller.
// Any table size: four cells (one stored word) per lane and trip; `cells_env` = the env's kind bytes in device memory.
template <int GRID>
PG_D void prep_cells_expand_any(ComposeLds<GRID>& L, const uint8_t* cells_env, uint32_t kind_off, int half, int lane) {
    const uint32_t* words = reinterpret_cast<const uint32_t*>(cells_env);
    for (int wq = half * 64 + lane; wq < GRID * GRID / 4; wq += 128) {
        const uint32_t four = words[wq];
        const int c = wq / (GRID / 4), r4 = (wq - c * (GRID / 4)) * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t k = (four >> (8 * i)) & 0xffu;
            const uint32_t off = __shfl(kind_off, static_cast<int>(k & (kPrepKinds - 1)));
            L.base[(r4 + i) * GRID + c] = static_cast<int32_t>(k == 0xffu ? kNoTexel : off);
        }
    }
    if (half == 0 && lane == 0) L.base[GRID * GRID] = L.base[GRID * GRID + 1] = static_cast<int32_t>(kNoTexel);
}
template <int GRID>
PG_D void prep_cells_expand(ComposeLds<GRID>& L, uint32_t two16, uint32_t kind_off, int half, int lane) {
    static_assert(GRID == 16, "two cells per lane of a 128-lane workgroup");
    const uint32_t k0 = two16 & 0xffu, k1 = two16 >> 8;
    const uint32_t o0 = __shfl(kind_off, static_cast<int>(k0 & (kPrepKinds - 1)));
    const uint32_t o1 = __shfl(kind_off, static_cast<int>(k1 & (kPrepKinds - 1)));
    const int j = half * 64 + lane, at = ((2 * j) & (GRID - 1)) * GRID + (j >> 3);
    L.base[at] = static_cast<int32_t>(k0 == 0xffu ? kNoTexel : o0);
    L.base[at + GRID] = static_cast<int32_t>(k1 == 0xffu ? kNoTexel : o1);
    if (half == 0 && lane == 0) L.base[GRID * GRID] = L.base[GRID * GRID + 1] = static_cast<int32_t>(kNoTexel);
}

// A resolved draw as the pre-pass stores it: pg_render.h BlitWords, kBlitWords per draw, draws of an env back to back.
PG_D void prep_draw_store(uint32_t* at, const Blit& b) {
    const BlitWords p = blit_pack(b);
    prep_u32x2* q = reinterpret_cast<prep_u32x2*>(at);  // (24-byte records: 8-byte aligned)
    prep_u32x2 w01, w23, w45;
    w01.x = p.w[0], w01.y = p.w[1], w23.x = p.w[2], w23.y = p.w[3], w45.x = p.w[4], w45.y = p.w[5];
    PG_PREP_STORE(w01, &q[0]);
    PG_PREP_STORE(w23, &q[1]);
    PG_PREP_STORE(w45, &q[2]);
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
