// Device-side rasteriser shared by the game render kernels (gfx950 only).
//
// One 64-lane wavefront owns one env; its 64×64 target lives in LDS as packed 0x00BBGGRR words (16 KiB).
// Two ways of putting pixels there, both bit-identical to replaying the reference's draw list back to front
// under raster rules S1–S5 (oracle twin: oracle/pgo_raster.cpp):
//
//  * compose_rows — background + the whole tile layer in ONE pass over the 64 rows, lane = pixel column.
//    render_texture's arithmetic is separable per axis (pg_geom.h resolve_axis), and every tile of the layer
//    shares one texture size and scale, so the layer is described by one span table per grid column and one per
//    grid row.  A pixel is covered by at most two grid columns and two grid rows (tiles overlap their
//    neighbours by the padding); the (row, column) candidates are blended in draw order
//    (y-major, x-minor).  Cost is independent of the number of tiles, every lane is busy, and the texel
//    gathers of several rows are issued together — unlike ~100–470 latency-serialised small blits.
//    If any pixel had a third covering column/row the function reports it and the caller falls back.
//  * wave_blit / wave_replay — one resolved draw executed by all lanes (flattened pixels), used for the few
//    sprites, particles and the agent, and as the fallback for tiles.
//
// The finished frame is streamed to the observation slab as contiguous 768-byte wave stores.
//
// One or two wavefronts per env.  With one, a frame's 16 KiB LDS target caps a CU at 8 resident waves (2 per SIMD),
// too few to hide the gather latency this kernel lives on.  Every helper therefore takes (half, halves): with
// halves = 2 the workgroup is two waves sharing one target — wave h composes, blends and stores the pixel rows
// [32h, 32h+32), the span tables are built one axis per wave — which doubles the waves per CU and halves the time a
// frame occupies its LDS.  Both waves run the same control flow on the same inputs, so every barrier is met by both.
#pragma once

#include "pg_engine.h"
#include "pg_geom.h"
#include "pg_probe.h"  // PG_ABL / PG_MARK / PG_TL: nothing in the product build
#include "pg_sincos.h"

namespace pg {

constexpr int kFbWords = kObsW * kObsH;

struct __attribute__((packed, aligned(4))) Rgb4 {
    uint32_t a, b, c;
};

// LDS scratch of the row composer; GRID = max tile-grid columns / rows visible in one frame.
// LDS decides how many envs a CU holds (measured: one workgroup less per CU costs 9 %), so only what the row loop
// itself reads stays in ComposeLds; the set-up tables (ComposeTmp) are read into registers before the first pixel is
// written and live in the frame target's own memory until then (compose_rows puts a barrier between the two uses).
// (alignas(16): chaser reads `base` through int4, the set-up tables that borrow the frame target are uint4 / int4.)
template <int GRID>
struct alignas(16) ComposeLds {
    int32_t base[GRID * GRID + 2];  // [row][col] byte offset of the cell's tile texture in the atlas, kNoTexel = no tile;
                                    // the two extra words are always kNoTexel (the cells of a pixel row no grid row covers)
    // flags in pairs, one word per wavefront (each wave initialises and sets its own; readers OR the two):
    int32_t too_wide[2];        // some span is wider than kMaxSpan pixels → fall back
    int32_t hard_rows[2];       // same layout as soft_rows: where compose_rows does not attempt its one-texel-per-pixel fast path
    int32_t soft_rows[2];       // bit r: grid row r shows a tile texture with texels that are not opaque (descriptor .w);
                                // bit 31: the background has some.  compose_spans sets it to its soft_init argument
                                // (default −1, "assume all"); a staging pass that knows better ORs exact bits in.
};
// … plus a second cell table for compose_rows<GRID, false, true>: the cells that show the layer's BOXED texture — one
// whose texels are transparent outside a box of texel columns and rows, so that it is a candidate only for the pixels
// that sample inside the box (chaser's points).  Same layout and sentinels as `base`; a cell is in one table or neither.
template <int GRID>
struct alignas(16) ComposeLdsBoxed {
    ComposeLds<GRID> plain;
    int32_t boxed[GRID * GRID + 2];
};
template <int GRID>
struct ComposeTmp {
    int4 col[GRID];           // per grid column: {d0, dn, s0, sn}; sn == 0 ⇒ nothing drawn
    int4 row[GRID];           // per grid row
    int4 row2[GRID];          // per grid row for the layer's second, shorter tile texture (sn == 0 ⇒ none)
    int32_t cover_n[2][64];   // [axis][pixel] how many grid columns (axis 0) / rows (axis 1) cover the pixel
    int32_t cover[2][64][2];  // the first two of them: grid index | texel coordinate << 8
};
// What the two waves hand each other before the first pixel is written (behind ComposeTmp, in the same borrowed memory).
struct ComposeHand {
    uint4 col[64];       // per pixel column (by wave 0): background column offset, column offsets a / b, 4 × cell column a
    uint4 row[64];       // per pixel row (by wave 1): background row offset, row offsets a / b, byte offset of grid row a's cells
    uint2 row2[64];      // … and the row offsets in the layer's second texture (compose_rows<GRID, true>) / boxed texture
    uint2 col2[64];      // the column offsets a / b in the boxed texture (compose_rows<GRID, false, true>)
    uint32_t masks[6];   // second_row, soft, hard (64 bits each)
    int32_t bad;         // either side found something the two-candidate scheme does not cover → fallback
};
template <int GRID>
PG_D ComposeTmp<GRID>& compose_tmp(uint32_t* fb) {
    static_assert(sizeof(ComposeTmp<GRID>) % 16 == 0 && sizeof(ComposeTmp<GRID>) + sizeof(ComposeHand) <= 64 * 64 * 4,
                  "the set-up tables borrow the frame target's memory");
    return *reinterpret_cast<ComposeTmp<GRID>*>(fb);
}
template <int GRID>
PG_D ComposeHand& compose_hand(uint32_t* fb) {
    return *reinterpret_cast<ComposeHand*>(reinterpret_cast<char*>(fb) + sizeof(ComposeTmp<GRID>));
}
constexpr int kMaxSpan = 8;  // default bound on the pixels one tile covers per axis (coinrun 5–6, maze 3); caveflyer passes 16

// The atlas descriptor table held in registers, two entries per lane (tables up to 128 textures): one pair of
// global loads at kernel start instead of a dependent global lookup in front of every draw.
struct DescRegs {
    int4 lo, hi;  // lane l: desc[l], desc[l + 64]
    PG_D static DescRegs load(const AtlasView& atlas, int lane) {
        DescRegs d;
        d.lo = lane < atlas.count ? atlas.desc[lane] : make_int4(0, 1, 1, 0);
        d.hi = lane + 64 < atlas.count ? atlas.desc[lane + 64] : make_int4(0, 1, 1, 0);
        return d;
    }
    // per-lane index (every lane must call this: cross-lane reads)
    PG_D int4 at(int t) const {
        const int l = t & 63;
        const bool up = t >= 64;
        int4 r;
        r.x = up ? __shfl(hi.x, l) : __shfl(lo.x, l);
        r.y = up ? __shfl(hi.y, l) : __shfl(lo.y, l);
        r.z = up ? __shfl(hi.z, l) : __shfl(lo.z, l);
        r.w = up ? __shfl(hi.w, l) : __shfl(lo.w, l);
        return r;
    }
    // wave-uniform index
    PG_D int4 uniform(int t) const {
        const int l = __builtin_amdgcn_readfirstlane(t) & 63;
        const bool up = __builtin_amdgcn_readfirstlane(t) >= 64;
        int4 r;
        r.x = up ? __builtin_amdgcn_readlane(hi.x, l) : __builtin_amdgcn_readlane(lo.x, l);
        r.y = up ? __builtin_amdgcn_readlane(hi.y, l) : __builtin_amdgcn_readlane(lo.y, l);
        r.z = up ? __builtin_amdgcn_readlane(hi.z, l) : __builtin_amdgcn_readlane(lo.z, l);
        r.w = up ? __builtin_amdgcn_readlane(hi.w, l) : __builtin_amdgcn_readlane(lo.w, l);
        return r;
    }
};

// Broadcast a resolved draw from lane `src` (wave-uniform) into scalar registers.
PG_D Blit blit_from_lane(const Blit& mine, int src) {
    Blit b;
    b.dx = __builtin_amdgcn_readlane(mine.dx, src);
    b.dy = __builtin_amdgcn_readlane(mine.dy, src);
    b.dw = __builtin_amdgcn_readlane(mine.dw, src);
    b.dh = __builtin_amdgcn_readlane(mine.dh, src);
    b.sx = __builtin_amdgcn_readlane(mine.sx, src);
    b.sy = __builtin_amdgcn_readlane(mine.sy, src);
    b.sw = __builtin_amdgcn_readlane(mine.sw, src);
    b.sh = __builtin_amdgcn_readlane(mine.sh, src);
    b.tex_off = __builtin_amdgcn_readlane(mine.tex_off, src);
    b.tex_w = __builtin_amdgcn_readlane(mine.tex_w, src);
    b.flip_mod = __builtin_amdgcn_readlane(mine.flip_mod, src);
    b.rot_sn = __builtin_amdgcn_readlane(mine.rot_sn, src);
    b.rot_cs = __builtin_amdgcn_readlane(mine.rot_cs, src);
    return b;
}

// The same through six packed words per lane (the halves blit_share uses below: S1 bounds destination coordinates to
// ±32767, source rectangles lie inside textures far smaller than 32768): six cross-lane reads per draw instead of
// eleven, taken apart by the scalar unit.  The angle of a rotated draw is read on top where it is needed.
// (kPacked = false in the replay templates keeps the plain form: chaser, whose render kernel the vector AND the scalar
// unit are busy in, measured 1 % slower packed; every other game 0.5–1.5 % faster.)
struct BlitWords {
    uint32_t w[6];
};
PG_D BlitWords blit_pack(const Blit& b) {
    BlitWords p;
    p.w[0] = (static_cast<uint32_t>(b.dx) & 0xffffu) | (static_cast<uint32_t>(b.dy) << 16);
    p.w[1] = static_cast<uint32_t>(b.dw) | (static_cast<uint32_t>(b.dh) << 16);
    p.w[2] = static_cast<uint32_t>(b.sx) | (static_cast<uint32_t>(b.sy) << 16);
    p.w[3] = static_cast<uint32_t>(b.sw) | (static_cast<uint32_t>(b.sh) << 16);
    p.w[4] = static_cast<uint32_t>(b.tex_off);
    p.w[5] = static_cast<uint32_t>(b.tex_w) | (static_cast<uint32_t>(b.flip_mod) << 16);
    return p;
}
PG_D Blit blit_from_lane(const BlitWords& p, const Blit& mine, int src) {
    const uint32_t w0 = __builtin_amdgcn_readlane(p.w[0], src), w1 = __builtin_amdgcn_readlane(p.w[1], src);
    const uint32_t w2 = __builtin_amdgcn_readlane(p.w[2], src), w3 = __builtin_amdgcn_readlane(p.w[3], src);
    const uint32_t w4 = __builtin_amdgcn_readlane(p.w[4], src), w5 = __builtin_amdgcn_readlane(p.w[5], src);
    Blit b;
    b.dx = static_cast<int32_t>(w0 << 16) >> 16;
    b.dy = static_cast<int32_t>(w0) >> 16;
    b.dw = static_cast<int32_t>(w1 & 0xffffu);
    b.dh = static_cast<int32_t>(w1 >> 16);
    b.sx = static_cast<int32_t>(w2 & 0xffffu);
    b.sy = static_cast<int32_t>(w2 >> 16);
    b.sw = static_cast<int32_t>(w3 & 0xffffu);
    b.sh = static_cast<int32_t>(w3 >> 16);
    b.tex_off = static_cast<int32_t>(w4);
    b.tex_w = static_cast<int32_t>(w5 & 0xffffu);
    b.flip_mod = static_cast<int32_t>(w5 >> 16);
    b.rot_sn = 0;
    b.rot_cs = 65536;
    if (b.flip_mod & kRotated) {  // wave-uniform
        b.rot_sn = __builtin_amdgcn_readlane(mine.rot_sn, src);
        b.rot_cs = __builtin_amdgcn_readlane(mine.rot_cs, src);
    }
    return b;
}

// kStamps (template parameter of everything that samples or blends a draw): whether the draws may be stamps at all.
// Only a kernel whose pre-pass substitutes stamps says true; for every other kernel the stamped forms compile to nothing.
template <bool kStamps = false>
PG_D void blend_into(uint32_t* fb, int idx, uint32_t texel, int mod);
// blend_into's `mod`: the draw's alpha modulation, or kStamped | 255 for a stamp's texels (pg_stamps.h: the modulation
// and the source half of the blend are in the texel already).
PG_D int blend_key(int flip_mod) { return flip_mod & (0xff | kStamped); }

// Consecutive draws of one wavefront may overlap, and a pixel is in general touched by a different LANE in each of
// them.  The hardware keeps a wave's LDS accesses in instruction order, but the language only orders the accesses of
// one thread: the compiler is free to let the lanes that sit out draw k (a branch on their own data) start on draw
// k + 1 before the others have done k.  wave_order() is the line it may not move anything across nor split the wave
// around: every lane's accesses above it happen before any lane's accesses below it.  No instructions.
PG_D void wave_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

PG_D uint32_t max3_u32(uint32_t a, uint32_t b, uint32_t c) {  // v_max3_u32
    const uint32_t m = a > b ? a : b;
    return m > c ? m : c;
}

// A resolved, un-rotated draw as six words per lane ([word][lane], 1536 B for a wave): the two wavefronts of an env
// need the same 64 resolved draws, and resolving them (≈ 180 vector instructions of exact float division) once and
// handing them over through LDS is cheaper than both doing it.  S1 bounds destination coordinates to ±32767 and
// texture extents are far below 32768, so halves of a word hold them; word 1 = 0 means "nothing to draw".
constexpr int kBlitWords = 6;
PG_D void blit_share(uint32_t* slots, int lane, const Blit& b, bool has) {
    slots[0 * 64 + lane] = (static_cast<uint32_t>(b.dx) & 0xffffu) | (static_cast<uint32_t>(b.dy) << 16);
    slots[1 * 64 + lane] = has ? (static_cast<uint32_t>(b.dw) | (static_cast<uint32_t>(b.dh) << 16)) : 0u;
    slots[2 * 64 + lane] = static_cast<uint32_t>(b.sx) | (static_cast<uint32_t>(b.sy) << 16);
    slots[3 * 64 + lane] = static_cast<uint32_t>(b.sw) | (static_cast<uint32_t>(b.sh) << 16);
    slots[4 * 64 + lane] = static_cast<uint32_t>(b.tex_off);
    slots[5 * 64 + lane] = static_cast<uint32_t>(b.tex_w) | (static_cast<uint32_t>(b.flip_mod) << 16);
}
PG_D bool blit_take(const uint32_t* slots, int lane, Blit& b) {
    const uint32_t w0 = slots[0 * 64 + lane], w1 = slots[1 * 64 + lane], w2 = slots[2 * 64 + lane];
    const uint32_t w3 = slots[3 * 64 + lane], w4 = slots[4 * 64 + lane], w5 = slots[5 * 64 + lane];
    b.dx = static_cast<int32_t>(w0 << 16) >> 16;
    b.dy = static_cast<int32_t>(w0) >> 16;
    b.dw = static_cast<int32_t>(w1 & 0xffffu);
    b.dh = static_cast<int32_t>(w1 >> 16);
    b.sx = static_cast<int32_t>(w2 & 0xffffu);
    b.sy = static_cast<int32_t>(w2 >> 16);
    b.sw = static_cast<int32_t>(w3 & 0xffffu);
    b.sh = static_cast<int32_t>(w3 >> 16);
    b.tex_off = static_cast<int32_t>(w4);
    b.tex_w = static_cast<int32_t>(w5 & 0xffffu);
    b.flip_mod = static_cast<int32_t>(w5 >> 16);
    b.rot_sn = 0;
    b.rot_cs = 65536;
    return w1 != 0u;
}

// The same hand-over for draws that may be rotated: two more words (16.16 sine and cosine), the flag in word 5.
constexpr int kRotBlitWords = 8;
PG_D void blit_share_rot(uint32_t* slots, int lane, const Blit& b, bool has) {
    blit_share(slots, lane, b, has);
    slots[6 * 64 + lane] = static_cast<uint32_t>(b.rot_sn);
    slots[7 * 64 + lane] = static_cast<uint32_t>(b.rot_cs);
}
PG_D bool blit_take_rot(const uint32_t* slots, int lane, Blit& b) {
    const bool has = blit_take(slots, lane, b);
    b.rot_sn = static_cast<int32_t>(slots[6 * 64 + lane]);
    b.rot_cs = static_cast<int32_t>(slots[7 * 64 + lane]);
    return has;
}

// The angle of a rotated draw as raster spec S6 uses it: 16.16 sine and cosine of deg degrees (deg != 0).
PG_D void rotation_16_16(double deg, int& sn, int& cs) {
    const float theta = static_cast<float>(deg * (3.14159265358979323846 / 180.0));
    sn = static_cast<int>(floor(static_cast<double>(sc_sinf(theta)) * 65536.0 + 0.5));
    cs = static_cast<int>(floor(static_cast<double>(sc_cosf(theta)) * 65536.0 + 0.5));
}

// Renderer::render_texture_rotated (games/*/renderer.cpp:84-101) followed by raster-spec S1 and S6: no cull, no
// crop, whole texture as source, rotation about the centre of the destination rectangle.  `rotation` is the
// float the reference passes; the angle handed to SDL is rotation * 180.0f / M_PI in double.  An angle of exactly
// zero takes the un-rotated path (S3), like the oracle.
PG_D bool resolve_rotated(const Camera& cam, int tw, int th, int tex_off, float pos_x, float pos_y, float rotation,
                          float scale, float alpha, Blit& out) {
    const float dx = (pos_x - cam.px) * cam.scale + cam.sw * 0.5f;
    const float dy = (pos_y - cam.py) * cam.scale + cam.sh * 0.5f;
    const float dw = tw * scale * cam.scale;
    const float dh = th * scale * cam.scale;
    int mod = 255;
    if (alpha != 1.0f) mod = static_cast<int>(255 * alpha) & 0xff;
    const double deg = rotation * 180.0f / 3.14159265358979323846;
    if (!(dw >= 1.0f && dh >= 1.0f && dw < 32768.0f && dh < 32768.0f)) return false;
    if (!(dx > -32768.0f && dx < 32768.0f && dy > -32768.0f && dy < 32768.0f)) return false;
    out.dx = static_cast<int>(dx);
    out.dy = static_cast<int>(dy);
    out.dw = static_cast<int>(dw);
    out.dh = static_cast<int>(dh);
    out.sx = 0;
    out.sy = 0;
    out.sw = tw;
    out.sh = th;
    out.tex_off = tex_off;
    out.tex_w = tw;
    out.flip_mod = mod;
    out.rot_sn = 0;
    out.rot_cs = 65536;
    if (deg != 0.0) {
        rotation_16_16(deg, out.rot_sn, out.rot_cs);
        out.flip_mod |= kRotated;
    }
    return true;
}

// The same with the angle in the form the raster uses, for games that know it before the frame: a bullet's rotation is
// fixed when it is fired, a ship's changes once a step — the logic kernel's lane for the env works out the sine and
// cosine once (rotation_of: exactly what resolve_rotated makes of its `rotation`) instead of every lane of both render
// wavefronts doing it every frame.  sn = cs = 0 stands for an angle of exactly zero (drawn un-rotated).
PG_D void rotation_of(float rotation, int& sn, int& cs) {
    const double deg = rotation * 180.0f / 3.14159265358979323846;
    sn = 0;
    cs = 0;
    if (deg != 0.0) rotation_16_16(deg, sn, cs);
}
PG_D bool resolve_rotated_at(const Camera& cam, int tw, int th, int tex_off, float pos_x, float pos_y, int sn, int cs,
                             float scale, float alpha, Blit& out) {
    const float dx = (pos_x - cam.px) * cam.scale + cam.sw * 0.5f;
    const float dy = (pos_y - cam.py) * cam.scale + cam.sh * 0.5f;
    const float dw = tw * scale * cam.scale;
    const float dh = th * scale * cam.scale;
    int mod = 255;
    if (alpha != 1.0f) mod = static_cast<int>(255 * alpha) & 0xff;
    if (!(dw >= 1.0f && dh >= 1.0f && dw < 32768.0f && dh < 32768.0f)) return false;
    if (!(dx > -32768.0f && dx < 32768.0f && dy > -32768.0f && dy < 32768.0f)) return false;
    out.dx = static_cast<int>(dx);
    out.dy = static_cast<int>(dy);
    out.dw = static_cast<int>(dw);
    out.dh = static_cast<int>(dh);
    out.sx = 0;
    out.sy = 0;
    out.sw = tw;
    out.sh = th;
    out.tex_off = tex_off;
    out.tex_w = tw;
    const bool rotated = sn != 0 || cs != 0;  // (sine and cosine are never both zero)
    out.flip_mod = rotated ? (mod | kRotated) : mod;
    out.rot_sn = rotated ? sn : 0;
    out.rot_cs = rotated ? cs : 65536;
    return true;
}

// A raw SDL_RenderTextureRotated(renderer, texture, NULL, &dst, angle, NULL, SDL_FLIP_NONE) in screen space (jumper's
// compass, jumper.cpp:485-508): whole texture as source, float destination rectangle, `deg` degrees about its centre.
PG_D bool resolve_screen(int tw, int th, int tex_off, float dx, float dy, float dw, float dh, double deg, Blit& out) {
    if (!(dw >= 1.0f && dh >= 1.0f && dw < 32768.0f && dh < 32768.0f)) return false;
    if (!(dx > -32768.0f && dx < 32768.0f && dy > -32768.0f && dy < 32768.0f)) return false;
    out.dx = static_cast<int>(dx);
    out.dy = static_cast<int>(dy);
    out.dw = static_cast<int>(dw);
    out.dh = static_cast<int>(dh);
    out.sx = 0;
    out.sy = 0;
    out.sw = tw;
    out.sh = th;
    out.tex_off = tex_off;
    out.tex_w = tw;
    out.flip_mod = 255;
    out.rot_sn = 0;
    out.rot_cs = 65536;
    if (deg != 0.0) {
        rotation_16_16(deg, out.rot_sn, out.rot_cs);
        out.flip_mod |= kRotated;
    }
    return true;
}
// resolve_screen with the angle already in that form; sn = cs = 0: not rotated (an angle of exactly zero).
PG_D bool resolve_screen_at(int tw, int th, int tex_off, float dx, float dy, float dw, float dh, int sn, int cs, Blit& out) {
    if (!(dw >= 1.0f && dh >= 1.0f && dw < 32768.0f && dh < 32768.0f)) return false;
    if (!(dx > -32768.0f && dx < 32768.0f && dy > -32768.0f && dy < 32768.0f)) return false;
    out.dx = static_cast<int>(dx);
    out.dy = static_cast<int>(dy);
    out.dw = static_cast<int>(dw);
    out.dh = static_cast<int>(dh);
    out.sx = 0;
    out.sy = 0;
    out.sw = tw;
    out.sh = th;
    out.tex_off = tex_off;
    out.tex_w = tw;
    const bool rotated = sn != 0 || cs != 0;  // (sine and cosine are never both zero)
    out.flip_mod = rotated ? (255 | kRotated) : 255;
    out.rot_sn = rotated ? sn : 0;
    out.rot_cs = rotated ? cs : 65536;
    return true;
}

// Raster spec S6: all lanes execute one rotated draw (wave-uniform).  The oracle scans the square of the rectangle's
// half-diagonal around its centre; a pixel is drawn only if it maps back inside the un-rotated rectangle, so any
// superset of those pixels gives the same frame.  Here: the bounding box of the rotated rectangle — in doubled
// coordinates its corners (±dw, ±dh) turn into |x| ≤ (dw·|cs| + dh·|sn|) / 65536, likewise y — widened by a pixel
// for the rounding of sn/cs, and clipped to the target.  For a needle-shaped sprite that is a fraction of the square.
// The texel fetches of four pixels per lane are issued before the first blend (one memory round trip per batch).
struct RotBox {
    int x_lo, y_lo, bw, bh;  // bw or bh ≤ 0: nothing on the target
};
PG_D RotBox rot_box(const Blit& b) {
    const int acs = b.rot_cs < 0 ? -b.rot_cs : b.rot_cs, asn = b.rot_sn < 0 ? -b.rot_sn : b.rot_sn;
    const int ex = rot_extent(b.dw, b.dh, acs, asn), ey = rot_extent(b.dh, b.dw, acs, asn);  // (pg_geom.h: why this is enough)
    int x_lo = b.dx + rot_first(b.dw, ex), x_hi = b.dx + rot_last(b.dw, ex);
    int y_lo = b.dy + rot_first(b.dh, ey), y_hi = b.dy + rot_last(b.dh, ey);
    x_lo = x_lo < 0 ? 0 : x_lo;
    y_lo = y_lo < 0 ? 0 : y_lo;
    x_hi = x_hi > kObsW - 1 ? kObsW - 1 : x_hi;
    y_hi = y_hi > kObsH - 1 ? kObsH - 1 : y_hi;
    return RotBox{x_lo, y_lo, x_hi - x_lo + 1, y_hi - y_lo + 1};
}

// kBatch: texel fetches in flight per lane — a batch costs one memory round trip.  (Four everywhere: eight or sixteen for
// jumper's needle and bunny — one trip instead of four and two — spill at its 96-register cap, render 0.94 -> 1.36 ms.)
// (Measured and rejected, round 4: for rectangles of at most 64 × 64 pixels, the two texel coordinates from per-lane tables
// read across lanes instead of two divisions a pixel, and the rotation as four 24-bit multiplies (rotated_pixel below) —
// 85 -> 45 vector instructions a pixel, bit-exact; jumper's render unchanged (0.651 ms), caveflyer's 0.429 -> 0.486: two
// more LDS round trips in front of every texel fetch cost more than the instructions they replace.)
template <int kBatch = 4, bool kStamps = false>
PG_D void wave_blit_rotated(uint32_t* fb, const AtlasView& atlas, const Blit& b, const RotBox& box, int lane,
                            int stride = 64) {
    const int x_lo = box.x_lo, y_lo = box.y_lo, bw = box.bw, bh = box.bh;
    if (bw <= 0 || bh <= 0) return;
    const uint32_t* tex = atlas.texels + b.tex_off;
    const int mod = blend_key(b.flip_mod);
    const bool stamped = kStamps && (b.flip_mod & kStamped) != 0;  // (wave-uniform: texel (i, j) itself)
    // A THIN rectangle (jumper's needle, 30 × 6) fills a fraction of its bounding box — a sixth at 45° — and every pixel of
    // the box costs the whole inverse map before it is found outside.  Of a pixel row only a short run of columns can map
    // back between the rectangle's long edges (pg_geom.h thin_run_start / thin_run_width: a superset of the row's drawn
    // pixels, swept on the host), so the box is scanned as bh rows of that width, each from its own start — and any
    // superset gives the same frame: every pixel scanned still takes the exact integer test below.  (Round 6; a flat
    // needle, whose runs are as wide as the box, and every rectangle that is not thin take the plain scan.)
    int row_w = bw;
    float run_inv = 0.0f, run_m = 0.0f;
    {
        const int asn = b.rot_sn < 0 ? -b.rot_sn : b.rot_sn;
        if (b.dw >= 3 * b.dh && asn >= 4096) {  // (wave-uniform)
            const int w = thin_run_width(b.dh, asn);
            if (w < bw) {
                row_w = w;
                run_inv = 1.0f / static_cast<float>(b.rot_sn);
                run_m = static_cast<float>(2 * b.dh) * 65536.0f;
            }
        }
    }
    const bool runs = row_w != bw;  // (wave-uniform)
    const int total = row_w * bh;
    for (int p0 = lane; p0 < total; p0 += stride * kBatch) {
        int idx[kBatch];
        uint32_t texel[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; k++) {
            idx[k] = -1;
            texel[k] = 0;
            const int p = p0 + k * stride;
            if (p >= total) continue;
            const int ry = udiv_small(p, row_w);
            const int off = p - ry * row_w;
            const int Y = y_lo + ry;
            int X = x_lo + off;
            if (runs) {  // the first column of this row's run, not left of the box
                const int xa = thin_run_start(b.dx, b.dy, b.dw, b.dh, b.rot_cs, run_m, run_inv, Y);
                X = (xa < x_lo ? x_lo : xa) + off;
                if (X >= x_lo + bw) continue;
            }
            const int px = 2 * (X - b.dx) + 1 - b.dw, py = 2 * (Y - b.dy) + 1 - b.dh;
            const long long lx = (long long)px * b.rot_cs + (long long)py * b.rot_sn + (long long)b.dw * 65536;
            const long long ly = -(long long)px * b.rot_sn + (long long)py * b.rot_cs + (long long)b.dh * 65536;
            if (lx < 0 || ly < 0 || lx >= (long long)(2 * b.dw) * 65536 || ly >= (long long)(2 * b.dh) * 65536) continue;
            const int i = static_cast<int>(lx >> 17), j = static_cast<int>(ly >> 17);
            const int u = stamped ? i : sample_index(0, b.sw, i, b.dw), v = stamped ? j : sample_index(0, b.sh, j, b.dh);
            idx[k] = Y * kObsW + X;
            texel[k] = tex[v * b.tex_w + u];
        }
#pragma unroll
        for (int k = 0; k < kBatch; k++)
            if (idx[k] >= 0) blend_into<kStamps>(fb, idx[k], texel[k], mod);
    }
}

// dst OVER-composited with one texel (raster spec S4).  Wave-level fast path: when every lane's alpha is 0 or
// 255 (opaque walls, opaque backgrounds — the common case) the blend is a select; the formula gives the same.
PG_D uint32_t over(uint32_t dst, uint32_t texel, int a) {
    if (__ballot(a != 0 && a != 255) == 0) return a ? (texel & 0x00ffffffu) : dst;
    return blend_px(dst, texel, a);
}

template <bool kStamps>
PG_D void blend_into(uint32_t* fb, int idx, uint32_t texel, int mod) {
    int a = static_cast<int>(texel >> 24);
    if (kStamps && (mod & kStamped)) {
        if (a == 0) return;
        fb[idx] = a == 255 ? (texel & 0x00ffffffu) : blend_premul(fb[idx], texel, a);
        return;
    }
    if (mod != 255) a = static_cast<int>(div255(static_cast<uint32_t>(a * mod)));
    if (a == 0) return;
    if (a == 255) {  // what the blend yields for an opaque texel, without the read-modify-write
        fb[idx] = texel & 0x00ffffffu;
        return;
    }
    fb[idx] = blend_px(fb[idx], texel, a);
}

// All 64 lanes execute one (wave-uniform) blit into the LDS target.
// Columns [xa, xa + cwa) of the clipped rectangle (x0.., y0.., ch rows), cwa ≤ 32: the wave is 64 / W rows of W = 8, 16 or
// 32 lanes.  A lane's texel column never changes; lane r holds the texel row of target row y0 + r (`row_at`, × the
// texture's pitch), fetched across lanes.
template <int kBatch, bool kStamps = false>
PG_D void wave_blit_columns(uint32_t* fb, const uint32_t* tex, const Blit& b, int xa, int cwa, int y0, int ch, int row_at, int lane,
                            int half, int halves) {
    const int mod = blend_key(b.flip_mod);
    const int wshift = cwa <= 8 ? 3 : (cwa <= 16 ? 4 : 5);  // wave-uniform
    const int rows = 64 >> wshift;
    const int rx = lane & ((1 << wshift) - 1), rsub = lane >> wshift;
    const bool on = rx < cwa;
    const int x = xa + rx;
    int i = x - b.dx;
    if (b.flip_mod & kFlipH) i = b.dw - 1 - i;
    const int u = !on ? b.sx : ((kStamps && (b.flip_mod & kStamped)) ? i : sample_index(b.sx, b.sw, i, b.dw));
    for (int t0 = half; t0 * rows < ch; t0 += halves * kBatch) {
        uint32_t texel[kBatch];
        int idx[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; k++) {
            const int r = (t0 + k * halves) * rows + rsub;
            const int v_tw = __shfl(row_at, r & 63);
            const bool ok = on && r < ch;
            idx[k] = ok ? (y0 + r) * kObsW + x : -1;
            texel[k] = 0;
            if (ok) texel[k] = tex[v_tw + u];
        }
#pragma unroll
        for (int k = 0; k < kBatch; k++)
            if (idx[k] >= 0) blend_into<kStamps>(fb, idx[k], texel[k], mod);
    }
}
template <int kBatch = 4, bool kStamps = false>
PG_D void wave_blit(uint32_t* fb, const AtlasView& atlas, const Blit& b, int lane, int half = 0, int halves = 1,
                    int row_lo = 0, int row_hi = kObsH) {
    const int x0 = b.dx > 0 ? b.dx : 0, y0 = b.dy > row_lo ? b.dy : row_lo;
    const int x1 = (b.dx + b.dw) < kObsW ? (b.dx + b.dw) : kObsW;
    const int y1 = (b.dy + b.dh) < row_hi ? (b.dy + b.dh) : row_hi;
    const int cw = x1 - x0, ch = y1 - y0;
    if (cw <= 0 || ch <= 0) return;
    const uint32_t* tex = atlas.texels + b.tex_off;
    const int tw = b.tex_w;
    const int mod = blend_key(b.flip_mod);
    const bool fh = (b.flip_mod & kFlipH) != 0, fv = (b.flip_mod & kFlipV) != 0;
    const bool stamped = kStamps && (b.flip_mod & kStamped) != 0;
    if (stamped && !fh && !fv) {
        // A stamp comes with the list of its texels that show (pg_stamps.h append_stamps, pg_geom.h stamp_list_at): lanes are dealt
        // over the entries of the stamp's rows that lie on my rows — no lane on a transparent texel (bossfight's shield is
        // a ring: 279 texels of 1 015), no sampling arithmetic, 8-byte loads side by side.  Same pixels: every texel that
        // shows is blended into its pixel once, the others never touched anything.
        const uint32_t* rows = atlas.texels + stamp_list_at(b);
        const uint2* entries = reinterpret_cast<const uint2*>(rows + ((b.dh + 2) & ~1));
        const int r0 = y0 - b.dy, r1 = y1 - b.dy;
        const int begin = static_cast<int>(__builtin_amdgcn_readfirstlane(rows[r0])), end = static_cast<int>(__builtin_amdgcn_readfirstlane(rows[r1]));
        for (int k0 = begin + half * 64; k0 < end; k0 += 64 * halves * kBatch) {
            uint2 e[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; k++) {
                const int at = k0 + 64 * halves * k + lane;
                e[k] = make_uint2(0xffffffffu, 0u);
                if (at < end) e[k] = entries[at];
            }
#pragma unroll
            for (int k = 0; k < kBatch; k++) {
                const int x = b.dx + static_cast<int>(e[k].x & 0xffu), y = b.dy + static_cast<int>((e[k].x >> 8) & 0xffu);
                if (e[k].x != 0xffffffffu && x >= 0 && x < kObsW) blend_into<kStamps>(fb, y * kObsW + x, e[k].y, mod);
            }
        }
        return;
    }
    // The texel row of target row y0 + r is the same for every lane: lane r works it out once for all of them (the
    // target has 64 rows) and the loops below pick it up with a cross-lane read — one division per lane and draw for the
    // rows, one for the lane's column, instead of three per pixel (one pixel per lane in row-major order, the narrow
    // draws' form until round 4).
    int row_at = 0;
    if (y0 + lane < y1) {
        int j = y0 + lane - b.dy;
        if (fv) j = b.dh - 1 - j;
        row_at = (stamped ? j : sample_index(b.sy, b.sh, j, b.dh)) * tw;
    }
    if (cw > 48) {
        // Wide blit (backgrounds on the fallback path, jumper's compass): lane = column, kBatch rows per iteration.
        const int x = x0 + lane;
        const bool on = lane < cw;
        int i = x - b.dx;
        if (fh) i = b.dw - 1 - i;
        const int u = !on ? b.sx : (stamped ? i : sample_index(b.sx, b.sw, i, b.dw));
        for (int yb = y0 + half; yb < y1; yb += halves * kBatch) {
            uint32_t texel[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; k++) {
                const int y = yb + k * halves;
                texel[k] = 0;
                if (y >= y1) continue;
                const int v_tw = __builtin_amdgcn_readlane(row_at, y - y0);
                if (on) texel[k] = tex[v_tw + u];
            }
#pragma unroll
            for (int k = 0; k < kBatch; k++) {
                const int y = yb + k * halves;
                if (on && y < y1) blend_into<kStamps>(fb, y * kObsW + x, texel[k], mod);
            }
        }
    } else if (cw > 32) {
        // 33 to 48 columns (bossfight's shield, 35 × 29 at 0.7 alpha — every pixel a full blend): a lane per column would
        // leave up to half the wave idle in each of `ch` trips; the first 32 columns two rows a trip, the rest apart.
        wave_blit_columns<kBatch, kStamps>(fb, tex, b, x0, 32, y0, ch, row_at, lane, half, halves);
        wave_blit_columns<kBatch, kStamps>(fb, tex, b, x0 + 32, cw - 32, y0, ch, row_at, lane, half, halves);
    } else {
        wave_blit_columns<kBatch, kStamps>(fb, tex, b, x0, cw, y0, ch, row_at, lane, half, halves);
    }
}

// Replays the draws held by the lanes flagged in `mask`, in ascending lane order.
// (Measured and rejected, twice: grouping rotated draws whose bounding box has ≤ 64 pixels like the small plain ones —
// inlined into the group loop, or in a loop of their own.  Classifying every draw's box and the extra code cost more
// than the saved round trips: bossfight 36.9 → 34–35 M env-steps/s.)
// Small draws (≤ 64 visible pixels: every sprite, particle and the agent) are taken kGroup at a time: one pixel
// per lane per draw, all texel fetches of the group issued before the first blend, so a group costs one memory
// round trip; the blends then run in draw order.  A larger draw goes through wave_blit on its own.  With two waves
// the small draws of a group alternate between them (per-draw work is mostly fixed-cost vector instructions — these
// kernels are VALU-bound — so executing every draw in both waves would double it).
PG_D void wave_replay(uint32_t* fb, const AtlasView& atlas, const Blit& mine, unsigned long long mask, int lane,
                      int half = 0, int halves = 1) {
    constexpr int kGroup = 4;
    // Which draws go alone (rotated, or more visible pixels than a wave has lanes): every lane judges its own draw,
    // so the group loop below can steer on a mask without broadcasting a draw it is not going to execute.
    bool lone = false;
    if ((mask >> lane) & 1ull) {
        const int x0 = mine.dx > 0 ? mine.dx : 0, y0 = mine.dy > 0 ? mine.dy : 0;
        const int x1 = (mine.dx + mine.dw) < kObsW ? (mine.dx + mine.dw) : kObsW;
        const int y1 = (mine.dy + mine.dh) < kObsH ? (mine.dy + mine.dh) : kObsH;
        lone = (mine.flip_mod & kRotated) || (x1 > x0 && y1 > y0 && (x1 - x0) * (y1 - y0) > 64);
    }
    const unsigned long long lones = __ballot(lone);
    int lone_turn = 0;
    while (mask) {
        uint32_t texel[kGroup];
        int idx[kGroup], mod[kGroup];
        bool stop = false;
#pragma unroll
        for (int g = 0; g < kGroup; g++) {
            idx[g] = -1;
            texel[g] = 0;
            mod[g] = 255;
            if (mask == 0 || stop) continue;
            const int src = __builtin_ctzll(mask);
            if ((lones >> src) & 1ull) {
                // a rotated or big one: alone, only at the head of a group, by the whole workgroup
                stop = true;
                if (g == 0) {
                    mask &= mask - 1;
                    const Blit b = blit_from_lane(mine, src);
                    if (b.flip_mod & kRotated) {
                        // A box that one wave covers in a single sweep (bullets, puffs) is left to one wave, the two
                        // taking turns: the other would run the same instructions with every lane idle.
                        const RotBox box = rot_box(b);
                        if (halves == 2 && box.bw * box.bh <= 64) {
                            if ((lone_turn & 1) == half) wave_blit_rotated(fb, atlas, b, box, lane, 64);
                            lone_turn++;
                        } else {
                            wave_blit_rotated(fb, atlas, b, box, lane + 64 * half, 64 * halves);
                        }
                    } else {
                        wave_blit(fb, atlas, b, lane, half, halves);
                    }
                    __syncthreads();
                }
                continue;
            }
            mask &= mask - 1;
            if (halves >= 2 && (g % halves) != half) continue;  // the small draws of a group go round the waves
            const Blit b = blit_from_lane(mine, src);
            const int x0 = b.dx > 0 ? b.dx : 0, y0 = b.dy > 0 ? b.dy : 0;
            const int x1 = (b.dx + b.dw) < kObsW ? (b.dx + b.dw) : kObsW;
            const int y1 = (b.dy + b.dh) < kObsH ? (b.dy + b.dh) : kObsH;
            const int cw = x1 - x0, ch = y1 - y0;
            if (cw <= 0 || ch <= 0 || lane >= cw * ch) continue;
            const int ry = udiv_small(lane, cw);
            const int rx = lane - ry * cw;
            const int x = x0 + rx, y = y0 + ry;
            int i = x - b.dx, j = y - b.dy;
            if (b.flip_mod & kFlipH) i = b.dw - 1 - i;
            if (b.flip_mod & kFlipV) j = b.dh - 1 - j;
            const int u = sample_index(b.sx, b.sw, i, b.dw);
            const int v = sample_index(b.sy, b.sh, j, b.dh);
            idx[g] = y * kObsW + x;
            texel[g] = atlas.texels[b.tex_off + v * b.tex_w + u];
            mod[g] = b.flip_mod & 0xff;
        }
#pragma unroll
        for (int g = 0; g < kGroup; g++) {
            if (idx[g] >= 0) blend_into(fb, idx[g], texel[g], mod[g]);
            __syncthreads();  // orders the LDS traffic of consecutive draws (they may overlap)
        }
    }
}

// The sprite passes of the two-wavefront kernels: wave h OWNS the pixel rows [row_lo, row_hi) = [32h, 32h + 32) of the
// target — it composed them, it blends every draw that reaches into them (clipped to them), in draw order, and it
// stores them.  Nothing one wave writes is read by the other, so there is no barrier between draws (wave_replay above
// hands whole draws to the waves in turn and meets at a barrier after each one) and the waves drift apart freely; a
// draw that straddles row 32 is simply done by both, each on its own rows.  Same grouping as wave_replay: small
// draws (≤ 64 pixels on my rows) kGroup at a time per memory round trip — the sprite pass is a chain of dependent
// round trips, and its length, not its instruction count, is what it costs — rotated or larger ones alone.
// One target pixel of a rotated draw (raster spec S6): pixel p of `box` (row-major) maps back into the un-rotated
// destination rectangle or misses it.  Same arithmetic as wave_blit_rotated.
template <bool kStamps = false>
PG_D bool rotated_pixel(const Blit& b, const RotBox& box, int p, int& idx, int& texel_at) {
    // A box of at most 8 × 8 pixels (every bullet and puff that gets here) is laid over the wave as an 8 × 8 grid — pixel
    // (p & 7, p >> 3), no division; a longer one row by row.  (Wave-uniform where the box is: the group path.)
    int rx, ry;
    if (box.bw <= 8 && box.bh <= 8) {
        rx = p & 7;
        ry = p >> 3;
        if (rx >= box.bw || ry >= box.bh) return false;
    } else {
        if (p >= box.bw * box.bh) return false;
        ry = udiv_small(p, box.bw);
        rx = p - ry * box.bw;
    }
    const int X = box.x_lo + rx, Y = box.y_lo + ry;
    const int px = 2 * (X - b.dx) + 1 - b.dw, py = 2 * (Y - b.dy) + 1 - b.dh;
    // Callers pass draws of at most kRotSmall pixels a side only: a pixel of the bounding box is within
    // (dw + dh) / 2 + 1 of the rectangle's centre, so |px|, |py| ≤ dw + dh + 2 < 2^10, the 16.16 sine and cosine are
    // at most 2^16 in magnitude, and every sum below stays under 2^28 — the products are exact 24-bit × 24-bit
    // multiplies, a quarter of the cost of the 64-bit form wave_blit_rotated needs for draws of any size.
    const int lx = __mul24(px, b.rot_cs) + __mul24(py, b.rot_sn) + (b.dw << 16);
    const int ly = __mul24(py, b.rot_cs) - __mul24(px, b.rot_sn) + (b.dh << 16);
    if (lx < 0 || ly < 0 || lx >= (b.dw << 17) || ly >= (b.dh << 17)) return false;
    const int i = lx >> 17, j = ly >> 17;
    idx = Y * kObsW + X;
    texel_at = (kStamps && (b.flip_mod & kStamped)) ? b.tex_off + j * b.tex_w + i
                                       : b.tex_off + sample_index(0, b.sh, j, b.dh) * b.tex_w + sample_index(0, b.sw, i, b.dw);
    return true;
}
constexpr int kRotSmall = 256;
PG_D RotBox rot_box_rows(const Blit& b, int row_lo, int row_hi) {  // rot_box clipped to the rows [row_lo, row_hi)
    RotBox box = rot_box(b);
    const int lo = box.y_lo > row_lo ? box.y_lo : row_lo;
    const int hi = (box.y_lo + box.bh) < row_hi ? (box.y_lo + box.bh) : row_hi;
    box.y_lo = lo;
    box.bh = hi - lo;
    return box;
}

// kRotInGroups: small rotated draws (a bullet, a puff: ≤ 64 pixels of bounding box on my rows) share a memory round
// trip with their neighbours in the list like the plain small ones, instead of paying one each (bossfight: dozens of
// bullets a frame).  Costs registers and code in the group loop, so only kernels with room to spare turn it on.
// The pass in two halves, for kernels that have their draws resolved before the frame is composed: replay_begin
// classifies the draws and requests the texels of the first group, the caller composes the frame — a memory round trip
// of its own — and replay_finish blends that group and goes on.  wave_replay_rows is the two back to back.
template <int kGroup>
struct ReplayState {
    unsigned long long mask, lones;  // draws still to do (those that reach my rows); which of them go alone
    // kQuarters: which draws are tiny (at most 4 × 4 pixels of my rows), and which of those touch the first, second, third
    // draw before them in the list (replay_group packs up to four tiny draws that follow and do not touch each other)
    unsigned long long tiny, near1, near2, near3;
    BlitWords packed;                // this lane's draw, packed for the cross-lane reads
    uint32_t box[2];                 // kRotInGroups / kQuarters: the draw's box on my rows (x_lo | y_lo << 16, bw | bh << 16)
    uint32_t texel[kGroup];
    int idx[kGroup], mod[kGroup];
};

// Up to four tiny draws in one slot of a group (kQuarters): sixteen lanes each — quarter q of the wave takes draw
// `src[q]`, lane l of it the pixel (l & 3, (l >> 2) & 3) of the draw's box on my rows (at most 4 × 4) —, the draws' words
// fetched per lane across lanes instead of broadcast through scalar registers.  The per-draw work of a slot (unpacking,
// the pixel's place in the texture, one gather, one blend) is then done once for four draws: a render kernel's time is its
// instruction count, and a bullet or a spark is a dozen pixels that had a 64-lane slot to themselves.  The caller has made
// sure the draws follow each other in the list and do not touch (replay_classify): one LDS write per pixel, any order.
template <bool kRotInGroups, bool kStamps>
PG_D void quarters_request(const AtlasView& atlas, const Blit& mine, const BlitWords& packed, const uint32_t (&box)[2], int lane,
                           const int (&src)[4], int count, uint32_t& texel, int& idx, int& mod) {
    const int q = lane >> 4;
    const int from = (q == 0 ? src[0] : (q == 1 ? src[1] : (q == 2 ? src[2] : src[3]))) & 63;
    const uint32_t w0 = static_cast<uint32_t>(__shfl(static_cast<int>(packed.w[0]), from));
    const uint32_t w1 = static_cast<uint32_t>(__shfl(static_cast<int>(packed.w[1]), from));
    const uint32_t w2 = static_cast<uint32_t>(__shfl(static_cast<int>(packed.w[2]), from));
    const uint32_t w3 = static_cast<uint32_t>(__shfl(static_cast<int>(packed.w[3]), from));
    const uint32_t w4 = static_cast<uint32_t>(__shfl(static_cast<int>(packed.w[4]), from));
    const uint32_t w5 = static_cast<uint32_t>(__shfl(static_cast<int>(packed.w[5]), from));
    const uint32_t b0 = static_cast<uint32_t>(__shfl(static_cast<int>(box[0]), from));
    const uint32_t b1 = static_cast<uint32_t>(__shfl(static_cast<int>(box[1]), from));
    const int dx = static_cast<int32_t>(w0 << 16) >> 16, dy = static_cast<int32_t>(w0) >> 16;
    const int dw = static_cast<int>(w1 & 0xffffu), dh = static_cast<int>(w1 >> 16);
    const int tex_off = static_cast<int>(w4), tex_w = static_cast<int>(w5 & 0xffffu), flip_mod = static_cast<int>(w5 >> 16);
    const int rx = lane & 3, ry = (lane >> 2) & 3;
    const int X = static_cast<int>(b0 & 0xffffu) + rx, Y = static_cast<int>(b0 >> 16) + ry;
    bool in = q < count && rx < static_cast<int>(b1 & 0xffffu) && ry < static_cast<int>(b1 >> 16);
    const bool stamped = kStamps && (flip_mod & kStamped) != 0;
    int i = X - dx, j = Y - dy;
    int sn = 0, cs = 65536;
    if (kRotInGroups && __ballot((flip_mod & kRotated) != 0) != 0ull) {  // (wave-uniform: a cross-lane read wants every lane there)
        sn = __shfl(mine.rot_sn, from);
        cs = __shfl(mine.rot_cs, from);
    }
    if (kRotInGroups && (flip_mod & kRotated)) {  // (per lane: the draws of a slot may differ in kind)
        const int px = 2 * i + 1 - dw, py = 2 * j + 1 - dh;  // rotated_pixel's arithmetic (tiny draws are far below kRotSmall)
        const int lx = __mul24(px, cs) + __mul24(py, sn) + (dw << 16);
        const int ly = __mul24(py, cs) - __mul24(px, sn) + (dh << 16);
        in = in && !(lx < 0 || ly < 0 || lx >= (dw << 17) || ly >= (dh << 17));
        i = lx >> 17;
        j = ly >> 17;
    } else {
        if (flip_mod & kFlipH) i = dw - 1 - i;
        if (flip_mod & kFlipV) j = dh - 1 - j;
    }
    int u = i, v = j;
    if (!stamped && in) {
        u = sample_index(static_cast<int>(w2 & 0xffffu), static_cast<int>(w3 & 0xffffu), i, dw);
        v = sample_index(static_cast<int>(w2 >> 16), static_cast<int>(w3 >> 16), j, dh);
    }
    mod = blend_key(flip_mod);
    idx = in ? Y * kObsW + X : -1;
    texel = 0u;
    if (in) texel = atlas.texels[tex_off + v * tex_w + u];
}

template <int kGroup, bool kRotInGroups, bool kPacked, int kLone = 4, bool kStamps = false, bool kQuarters = false>
PG_D void replay_group(const AtlasView& atlas, const Blit& mine, ReplayState<kGroup>& st, int lane, int row_lo, int row_hi,
                       uint32_t* fb_for_lone) {
    // requests the texels of the next ≤ kGroup small draws; a big draw at the head of the list is executed on the spot
    // (fb_for_lone; nullptr = stop in front of it instead).
    // kStamps kernels (bossfight) take the big draws in a loop of their own, in front of the unrolled one: wave_blit and
    // wave_blit_rotated are most of this function's code, and with the stamped forms on top that code, counted four times
    // against the unroller's budget inside the group loop, went over it — the loop stayed rolled and ReplayState's arrays
    // landed in scratch memory (bossfight's render kernel: 0.49 -> 0.72 ms).  The other kernels keep the round-5 form
    // (the big draw as iteration 0 of the group loop): jumper measured 0.7 % faster with it.
    constexpr bool kLonesFirst = kStamps;
    while (kLonesFirst && fb_for_lone != nullptr && st.mask != 0) {  // (wave-uniform)
        const int src = __builtin_ctzll(st.mask);
        if (!((st.lones >> src) & 1ull)) break;
        st.mask &= st.mask - 1;
        const Blit b = kPacked ? blit_from_lane(st.packed, mine, src) : blit_from_lane(mine, src);
        wave_order();
        if (b.flip_mod & kRotated) {
            wave_blit_rotated<kLone, kStamps>(fb_for_lone, atlas, b, rot_box_rows(b, row_lo, row_hi), lane, 64);
        } else {
            wave_blit<kLone, kStamps>(fb_for_lone, atlas, b, lane, 0, 1, row_lo, row_hi);
        }
        wave_order();
    }
    bool stop = false;
#pragma unroll
    for (int g = 0; g < kGroup; g++) {
        st.idx[g] = -1;
        st.texel[g] = 0;
        st.mod[g] = 255;
        if (st.mask == 0 || stop) continue;
        const int src = __builtin_ctzll(st.mask);
        if ((st.lones >> src) & 1ull) {
            stop = true;  // a big one: alone, only at the head of a group
            if (!kLonesFirst && g == 0 && fb_for_lone != nullptr) {
                st.mask &= st.mask - 1;
                const Blit b = kPacked ? blit_from_lane(st.packed, mine, src) : blit_from_lane(mine, src);
                wave_order();
                if (b.flip_mod & kRotated) {
                    wave_blit_rotated<kLone, kStamps>(fb_for_lone, atlas, b, rot_box_rows(b, row_lo, row_hi), lane, 64);
                } else {
                    wave_blit<kLone, kStamps>(fb_for_lone, atlas, b, lane, 0, 1, row_lo, row_hi);
                }
                wave_order();
            }
            continue;
        }
        if (kQuarters && ((st.tiny >> src) & 1ull)) {
            // this tiny draw and up to three that follow it in the list, tiny too and touching none of the ones taken
            static_assert(!kQuarters || kPacked, "the quarters fetch the packed words");
            int from[4] = {src, 64, 64, 64};
            int count = 1;
            unsigned long long rest = st.mask & (st.mask - 1);
#pragma unroll
            for (int k = 1; k < 4; k++) {
                if (count != k || rest == 0ull) continue;  // (wave-uniform)
                const int next = __builtin_ctzll(rest);
                const unsigned long long bit = 1ull << next;
                const bool apart = !(st.near1 & bit) && (k < 2 || !(st.near2 & bit)) && (k < 3 || !(st.near3 & bit));
                if ((st.tiny & bit) && apart) {
                    from[k] = next;
                    count = k + 1;
                    rest &= rest - 1;
                }
            }
            st.mask = rest;
            quarters_request<kRotInGroups, kStamps>(atlas, mine, st.packed, st.box, lane, from, count, st.texel[g], st.idx[g], st.mod[g]);
            continue;
        }
        st.mask &= st.mask - 1;
        const Blit b = kPacked ? blit_from_lane(st.packed, mine, src) : blit_from_lane(mine, src);
        st.mod[g] = blend_key(b.flip_mod);  // (here, where control is still wave-uniform: the key stays a scalar)
        if (kRotInGroups && (b.flip_mod & kRotated)) {
            int at = 0, where = -1;
            const uint32_t b0 = __builtin_amdgcn_readlane(st.box[0], src), b1 = __builtin_amdgcn_readlane(st.box[1], src);
            const RotBox box{static_cast<int>(b0 & 0xffffu), static_cast<int>(b0 >> 16), static_cast<int>(b1 & 0xffffu),
                             static_cast<int>(b1 >> 16)};  // (in a group: 1 ≤ bw, bh ≤ 64; x_lo, y_lo ≥ 0)
            if (rotated_pixel<kStamps>(b, box, lane, where, at)) {
                st.idx[g] = where;
                st.texel[g] = atlas.texels[at];
            }
            continue;
        }
        const int x0 = b.dx > 0 ? b.dx : 0, y0 = b.dy > row_lo ? b.dy : row_lo;
        const int x1 = (b.dx + b.dw) < kObsW ? (b.dx + b.dw) : kObsW;
        const int y1 = (b.dy + b.dh) < row_hi ? (b.dy + b.dh) : row_hi;
        const int cw = x1 - x0, ch = y1 - y0;
        int rx, ry;
        if (cw <= 8 && ch <= 8) {  // (wave-uniform) the wave as an 8 × 8 grid: no division
            rx = lane & 7;
            ry = lane >> 3;
            if (rx >= cw || ry >= ch) continue;
        } else {
            if (lane >= cw * ch) continue;
            ry = udiv_small(lane, cw);
            rx = lane - ry * cw;
        }
        const int x = x0 + rx, y = y0 + ry;
        int i = x - b.dx, j = y - b.dy;
        if (b.flip_mod & kFlipH) i = b.dw - 1 - i;
        if (b.flip_mod & kFlipV) j = b.dh - 1 - j;
        const bool stamped = kStamps && (b.flip_mod & kStamped) != 0;
        const int u = stamped ? i : sample_index(b.sx, b.sw, i, b.dw);
        const int v = stamped ? j : sample_index(b.sy, b.sh, j, b.dh);
        st.idx[g] = y * kObsW + x;
        st.texel[g] = atlas.texels[b.tex_off + v * b.tex_w + u];
    }
}

// `whole`: this lane's rotated draw's bounding box on the whole target, where somebody has worked it out already (the
// render pre-pass: rot_box is two 64-bit products per axis) — it is clipped to the wave's rows here; nullptr: rot_box now.
PG_D RotBox rot_box_clip(RotBox box, int row_lo, int row_hi) {
    const int lo = box.y_lo > row_lo ? box.y_lo : row_lo;
    const int hi = (box.y_lo + box.bh) < row_hi ? (box.y_lo + box.bh) : row_hi;
    box.y_lo = lo;
    box.bh = hi - lo;
    return box;
}
template <int kGroup, bool kRotInGroups, bool kPacked, bool kQuarters = false>
PG_D ReplayState<kGroup> replay_classify(const Blit& mine, unsigned long long mask, int lane, int row_lo, int row_hi,
                                         const RotBox* whole = nullptr) {
    bool lone = false, reaches = false, tiny = false;
    uint32_t box0 = 0, box1 = 0, edges = 0;
    if ((mask >> lane) & 1ull) {
        if (mine.flip_mod & kRotated) {
            const RotBox box = whole ? rot_box_clip(*whole, row_lo, row_hi) : rot_box_rows(mine, row_lo, row_hi);
            reaches = box.bw > 0 && box.bh > 0;
            lone = !kRotInGroups || box.bw * box.bh > 64 || mine.dw > kRotSmall || mine.dh > kRotSmall;
            // (the box of a draw that goes into a group travels with it: no second pass through rot_box's 64-bit products)
            box0 = static_cast<uint32_t>(box.x_lo) | (static_cast<uint32_t>(box.y_lo) << 16);
            box1 = static_cast<uint32_t>(box.bw & 0xffff) | (static_cast<uint32_t>(box.bh) << 16);
            tiny = kQuarters && reaches && !lone && box.bw <= 4 && box.bh <= 4;
            edges = static_cast<uint32_t>(box.x_lo) | static_cast<uint32_t>(box.y_lo - row_lo) << 8 |
                    static_cast<uint32_t>(box.x_lo + box.bw - 1) << 16 | static_cast<uint32_t>(box.y_lo - row_lo + box.bh - 1) << 24;
        } else {
            const int x0 = mine.dx > 0 ? mine.dx : 0, y0 = mine.dy > row_lo ? mine.dy : row_lo;
            const int x1 = (mine.dx + mine.dw) < kObsW ? (mine.dx + mine.dw) : kObsW;
            const int y1 = (mine.dy + mine.dh) < row_hi ? (mine.dy + mine.dh) : row_hi;
            reaches = x1 > x0 && y1 > y0;
            lone = reaches && (x1 - x0) * (y1 - y0) > 64;
            if (kQuarters) {
                tiny = reaches && x1 - x0 <= 4 && y1 - y0 <= 4;
                box0 = static_cast<uint32_t>(x0) | (static_cast<uint32_t>(y0) << 16);
                box1 = static_cast<uint32_t>(x1 - x0) | (static_cast<uint32_t>(y1 - y0) << 16);
                edges = static_cast<uint32_t>(x0) | static_cast<uint32_t>(y0 - row_lo) << 8 | static_cast<uint32_t>(x1 - 1) << 16 |
                        static_cast<uint32_t>(y1 - 1 - row_lo) << 24;
            }
        }
    }
    ReplayState<kGroup> st;
    st.mask = __ballot(reaches);
    st.lones = __ballot(lone && reaches);
    st.tiny = st.near1 = st.near2 = st.near3 = 0ull;
    if (kQuarters) {
        // A tiny draw's box as four bytes (x_lo, y_lo, x_hi, y_hi — rows counted from row_lo, all below 64), held against
        // those of the three draws before it in the list: a = hi | 0x8080, b = lo of the other — every byte of a − b has
        // bit 7 set exactly when hi ≥ lo (no borrow between the bytes: 0x80 + hi − lo ≥ 0x41), and two boxes touch when
        // that holds both ways on both axes.
        st.tiny = __ballot(tiny);
        unsigned long long before = st.mask & ((1ull << lane) - 1ull);
        const uint32_t my_hi = (edges >> 16) | 0x8080u, my_lo = edges & 0xffffu;
        bool near[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const bool there = before != 0ull;
            const int prev = there ? 63 - __builtin_clzll(before) : 0;
            before &= ~(1ull << prev);
            const uint32_t other = static_cast<uint32_t>(__shfl(static_cast<int>(edges), prev));
            const uint32_t its_hi = (other >> 16) | 0x8080u, its_lo = other & 0xffffu;
            near[k] = tiny && there && (((my_hi - its_lo) & (its_hi - my_lo) & 0x8080u) == 0x8080u);
        }
        st.near1 = __ballot(near[0]);
        st.near2 = __ballot(near[1]);
        st.near3 = __ballot(near[2]);
    }
    if (kPacked) st.packed = blit_pack(mine);
    st.box[0] = box0;
    st.box[1] = box1;
    return st;
}

template <int kGroup = 4, bool kRotInGroups = false, bool kPacked = true, bool kStamps = false, bool kQuarters = false>
PG_D ReplayState<kGroup> replay_begin(const AtlasView& atlas, const Blit& mine, unsigned long long mask, int lane,
                                      int row_lo, int row_hi, const RotBox* whole = nullptr) {
    ReplayState<kGroup> st = replay_classify<kGroup, kRotInGroups, kPacked, kQuarters>(mine, mask, lane, row_lo, row_hi, whole);
    replay_group<kGroup, kRotInGroups, kPacked, 4, kStamps, kQuarters>(atlas, mine, st, lane, row_lo, row_hi, nullptr);  // (a big draw first: nothing requested)
    return st;
}

// (Measured and rejected, round 4: software-pipelining the groups — the texels of group k + 1 requested BEFORE group k is
// blended (commit 00e86ae has the loop).  Bit-exact in all seven games, and slower: bossfight's render 0.779 -> 0.809 ms,
// jumper's 0.843 -> 0.853.  These kernels are bound by vector instructions at four clocks apiece (SQ_ACTIVE_INST_VALU:
// 4.6 clocks per SQ_INSTS_VALU), not by the length of the pass's chain of round trips; the copies that free the state for
// the next request are more instructions.)
template <int kGroup = 4, bool kRotInGroups = false, bool kPacked = true, int kLone = 4, bool kStamps = false, bool kQuarters = false>
PG_D void replay_finish(uint32_t* fb, const AtlasView& atlas, const Blit& mine, ReplayState<kGroup>& st, int lane,
                        int row_lo, int row_hi) {
    wave_order();  // what the caller put into these rows in the meantime
    for (;;) {
#pragma unroll
        for (int g = 0; g < kGroup; g++) {
            if (st.idx[g] >= 0) blend_into<kStamps>(fb, st.idx[g], st.texel[g], st.mod[g]);
            wave_order();  // draws may overlap
        }
        if (st.mask == 0) break;
        replay_group<kGroup, kRotInGroups, kPacked, kLone, kStamps, kQuarters>(atlas, mine, st, lane, row_lo, row_hi, fb);
    }
}

// kRotInGroups: small rotated draws (a bullet, a puff: ≤ 64 pixels of bounding box on my rows) share a memory round
// trip with their neighbours in the list like the plain small ones, instead of paying one each (bossfight: dozens of
// bullets a frame).  Costs registers and code in the group loop, so only kernels with room to spare turn it on.
template <int kGroup = 4, bool kRotInGroups = false, bool kPacked = true, int kLone = 4, bool kStamps = false, bool kQuarters = false>
PG_D void wave_replay_rows(uint32_t* fb, const AtlasView& atlas, const Blit& mine, unsigned long long mask, int lane,
                           int row_lo, int row_hi, const RotBox* whole = nullptr) {
    ReplayState<kGroup> st = replay_begin<kGroup, kRotInGroups, kPacked, kStamps, kQuarters>(atlas, mine, mask, lane, row_lo, row_hi, whole);
    replay_finish<kGroup, kRotInGroups, kPacked, kLone, kStamps, kQuarters>(fb, atlas, mine, st, lane, row_lo, row_hi);
}

// A draw that is the same in every frame of every env (a HUD element at a fixed place on the observation), prepared
// on the host when the atlas is loaded (Game::extend_atlas): `image` = 64×64 words, the texel that lands on each pixel
// where it is opaque, 0 elsewhere; `list` = (pixel index, texel) pairs for the texels that are translucent — the upper
// 32 rows' first, then the lower rows', each half padded to kOverlayPerLane × 64 entries with index 0xffffffff.  Opaque
// pixels are coalesced loads and masked stores, translucent ones one blend per pixel — instead of a 60×60 blit that
// samples, tests and blends every pixel it covers — and every load of the wave, the list's too, is in one round trip
// (the list walked in a loop was a round trip per 64 entries, both halves' entries by both waves: 5 000 -> ? clocks).
// Same pixels as wave_replay_rows of the draw (raster spec S1–S4).  row_lo: 0 or 32.
constexpr int kOverlayPerLane = 2;
// `rows` (wave-uniform): bit r = the opaque texels of row row_lo + r are still to be written; a row whose covered pixels
// hold them already (compose_rows_from UNDER put them there and no draw has been near since) is left out — all 32 of them
// in most frames, and then the picture is not even loaded: only the translucent texels' list remains.  Eight rows a group.
PG_D void overlay_rows(uint32_t* fb, const uint32_t* image, const uint2* list, int lane, int row_lo, uint32_t rows = 0xffffffffu) {
    constexpr int kRows = kObsH / 2;  // (the contract with the callers and with whoever made the list: a wave owns HALF the frame's rows)
    static_assert(kObsH == 64 && kRows == 32, "overlay_rows: the list is split per half frame of 32 rows");
    uint2 item[kOverlayPerLane];
#pragma unroll
    for (int k = 0; k < kOverlayPerLane; k++) item[k] = list[(row_lo ? kOverlayPerLane * 64 : 0) + k * 64 + lane];
    if (rows == 0xffffffffu) {
        uint32_t t[kRows];
#pragma unroll
        for (int k = 0; k < kRows; k++) t[k] = image[(row_lo + k) * kObsW + lane];
        wave_order();
#pragma unroll
        for (int k = 0; k < kRows; k++)
            if (t[k] >= 0xff000000u) fb[(row_lo + k) * kObsW + lane] = t[k];
    } else {
        wave_order();
#pragma unroll
        for (int g = 0; g < kRows / 8; g++) {
            if (((rows >> (8 * g)) & 0xffu) == 0u) continue;  // (wave-uniform)
            uint32_t t[8];
#pragma unroll
            for (int k = 0; k < 8; k++) t[k] = image[(row_lo + 8 * g + k) * kObsW + lane];
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (t[k] >= 0xff000000u) fb[(row_lo + 8 * g + k) * kObsW + lane] = t[k];
        }
    }
#pragma unroll
    for (int k = 0; k < kOverlayPerLane; k++)
        if (item[k].x != 0xffffffffu) blend_into(fb, static_cast<int>(item[k].x), item[k].y, 255);
    wave_order();
}

// The rows wave `half` of `halves` owns, and their share of the finished frame on its way out (no barrier needed
// between a wave's last blend and its own store).
// fb_row0: the row of `fb` that holds frame row row_lo (row_lo itself for a whole-frame target, 0 for a wave's own).
PG_D void wave_store_rows(const uint32_t* fb, uint8_t* obs_env, int lane, int row_lo, int row_hi, int fb_row0 = -1) {
    Rgb4* out = reinterpret_cast<Rgb4*>(obs_env);
    const uint4* in = reinterpret_cast<const uint4*>(fb);
    const int shift = fb_row0 < 0 ? 0 : (row_lo - fb_row0) * (kObsW / 4);
    wave_order();  // the wave's own blends, lane-to-pixel mapping of the draws
    for (int g = row_lo * (kObsW / 4) + lane; g < row_hi * (kObsW / 4); g += 64) {
        const uint4 p = in[g - shift];
        Rgb4 o;
        o.a = __builtin_amdgcn_perm(p.y, p.x, 0x04020100u);
        o.b = __builtin_amdgcn_perm(p.z, p.y, 0x05040201u);
        o.c = __builtin_amdgcn_perm(p.w, p.z, 0x06050402u);
        __builtin_nontemporal_store(o.a, &out[g].a);
        __builtin_nontemporal_store(o.b, &out[g].b);
        __builtin_nontemporal_store(o.c, &out[g].c);
    }
}

// SDL_RenderClear with (0,0,0,255): coinrun.cpp:447-448.
PG_D void wave_clear(uint32_t* fb, int lane, int half = 0, int halves = 1) {
    uint4* p = reinterpret_cast<uint4*>(fb);
    for (int k = lane + 64 * half; k < kFbWords / 4; k += 64 * halves) p[k] = make_uint4(0, 0, 0, 0);
    __syncthreads();
}

// One axis of the background draw (Renderer::render_texture is separable, pg_geom.h resolve_axis).  The two wavefronts
// of a frame split the composer's set-up: wave 0 works out everything that belongs to pixel COLUMNS (and resolves the
// background's x axis for it), wave 1 everything that belongs to pixel ROWS (y axis), and they trade the results
// through the frame target's still unused memory (ComposeHand) — instead of both doing both.  A background that one
// axis culls or empties needs no flag: that axis hands out kNoTexel offsets, and column + row offset is then out of
// the atlas.  Every game draws its background unflipped and with alpha 1 (e.g. coinrun.cpp:459-464).
struct BgAxis {
    int32_t d0, dn, s0, sn;  // destination span on the target, source span in the texture (dn = 0: nothing)
    int32_t tex_off, tex_w;
};
struct BgDraw {  // the background's draw call: texture descriptor, world position (pixels), scale
    int4 desc;
    float px, py, scale;
};
PG_D BgAxis bg_axis(const Camera& cam, const int4& desc, float pos_x, float pos_y, float scale, int axis) {
    Span sp;
    const bool ok = axis == 0 ? resolve_axis(cam.px, cam.sw, cam.scale, desc.y, pos_x, scale, false, false, sp)
                              : resolve_axis(cam.py, cam.sh, cam.scale, desc.z, pos_y, scale, false, true, sp);
    return BgAxis{ok ? sp.d0 : 0, ok ? sp.dn : 0, sp.s0, sp.sn, desc.x, desc.y};
}
// this lane's background offset on `axis`: texel column (× 4 bytes, + the texture's start) or texel row (× pitch)
PG_D uint32_t bg_offset(const BgAxis& b, int lane, int axis) {
    if (lane < b.d0 || lane >= b.d0 + b.dn) return 0x40000000u;  // kNoTexel
    const int t = sample_index(b.s0, b.sn, lane - b.d0, b.dn);
    return axis == 0 ? static_cast<uint32_t>(b.tex_off + t) * 4u : static_cast<uint32_t>(t * b.tex_w) * 4u;
}

// What a game that knows its textures passes as compose_spans' soft_init: the background's bit, and every grid row's
// if some tile texture of the layer has texels that are not opaque (descriptor .w, set when the atlas is loaded).
PG_D int32_t soft_rows_of(int bg_w, int tiles_w) {
    return static_cast<int32_t>((bg_w ? 0x80000000u : 0u) | ((tiles_w & 1) ? 0x7fffffffu : 0u));
}
// … and as hard_init: where the one-texel-per-pixel attempt is not worth making (descriptor .w bit 1: many texels of
// the texture are not opaque).
PG_D int32_t hard_rows_of(int bg_w, int tiles_w) {
    return static_cast<int32_t>(((bg_w & 2) ? 0x80000000u : 0u) | ((tiles_w & 2) ? 0x7fffffffu : 0u));
}

// Step 1 of the composer: the per-column / per-row span tables of the tile grid (lane c → column x0+c and row
// y0+c), then — after a barrier — every (span, offset) pair scatters itself to the pixel it covers, so each pixel
// column / row learns which grid columns / rows cover it and at which texel coordinate, without a search loop.
// Leaves a __syncthreads() to the caller (the staging of L.base provides it) before compose_rows.
// th2 > 0: some cells of the layer use a second texture of the same width but height th2 < th (climber's 64×53
// cap tile next to 64×64 bodies).  Its rows start where the tall ones start and end earlier, so the covering grid
// rows of a pixel stay the tall texture's; only the texel row differs per cell (compose_rows<GRID, true>).
template <int GRID, int MAXSPAN = kMaxSpan>
PG_D void compose_spans(uint32_t* fb, ComposeLds<GRID>& L, const Camera& cam, int x0, int y0, int cols, int rows, int tw,
                        int th, float tile_scale, int lane, int th2 = 0, int half = 0, int halves = 1,
                        int32_t soft_init = -1, int32_t hard_init = -1, const BgDraw* bg = nullptr, BgAxis* bga = nullptr) {
    // Two wavefronts, one axis each (wave 0: columns, wave 1: rows), and nothing of one axis is read by the other
    // wave before the barrier that follows the caller's staging of L.base — so there is no barrier in here: every
    // table below is written and read by the same wave (LDS operations of a wave complete in order), and the flags
    // come in pairs, one word per wave.
    (void)halves;
    ComposeTmp<GRID>& T = compose_tmp<GRID>(fb);
    const int axis = half;
    T.cover_n[axis][lane] = 0;
    if (lane == 0) {
        L.too_wide[half] = 0;
        L.soft_rows[half] = soft_init;
        L.hard_rows[half] = hard_init;
        if (half == 0) {
            compose_hand<GRID>(fb).bad = 0;
            L.base[GRID * GRID] = L.base[GRID * GRID + 1] = static_cast<int32_t>(0x40000000u);  // kNoTexel (declared below)
        }
    }
    // The background's draw on this axis is one more span to resolve — the same arithmetic with other numbers — so it
    // rides along in the last lane (no grid is 63 wide) instead of costing every lane a second pass through
    // resolve_axis, and comes back with cross-lane reads.
    static_assert(GRID < 63, "lane 63 resolves the background");
    const bool bg_lane = bg != nullptr && lane == 63;
    bool wide = false;
    Span sp;
    sp.d0 = sp.dn = sp.s0 = sp.sn = 0;
    bool ok = false;
    if (axis == 0) {
        if (lane < cols || bg_lane) {
            const int ts = bg_lane ? bg->desc.y : tw;
            const float pos = bg_lane ? bg->px : (x0 + lane) * kUnitPx;
            ok = resolve_axis(cam.px, cam.sw, cam.scale, ts, pos, bg_lane ? bg->scale : tile_scale, false, false, sp);
            if (!bg_lane) {
                T.col[lane] = ok ? make_int4(sp.d0, sp.dn, sp.s0, sp.sn) : make_int4(0, 0, 0, 0);
                wide = ok && sp.dn > MAXSPAN;
            }
        }
    } else if (lane < rows || bg_lane) {
        const int ts = bg_lane ? bg->desc.z : th;
        const float pos = bg_lane ? bg->py : (y0 + lane) * kUnitPx;
        ok = resolve_axis(cam.py, cam.sh, cam.scale, ts, pos, bg_lane ? bg->scale : tile_scale, false, true, sp);
        if (!bg_lane) {
            T.row[lane] = ok ? make_int4(sp.d0, sp.dn, sp.s0, sp.sn) : make_int4(0, 0, 0, 0);
            wide = ok && sp.dn > MAXSPAN;
        }
        if (th2 > 0 && !bg_lane) {
            Span s2;
            const bool ok2 =
                resolve_axis(cam.py, cam.sh, cam.scale, th2, (y0 + lane) * kUnitPx, tile_scale, false, true, s2);
            T.row2[lane] = ok2 ? make_int4(s2.d0, s2.dn, s2.s0, s2.sn) : make_int4(0, 0, 0, 0);
            wide = wide || (ok2 && !span_nested(s2.d0, s2.dn, ok ? sp.d0 : 0, ok ? sp.dn : 0, kObsH));  // not nested (pg_geom.h): take the fallback
        }
    }
    if (bga != nullptr) {
        *bga = BgAxis{__builtin_amdgcn_readlane(ok ? sp.d0 : 0, 63), __builtin_amdgcn_readlane(ok ? sp.dn : 0, 63),
                      __builtin_amdgcn_readlane(sp.s0, 63), __builtin_amdgcn_readlane(sp.sn, 63), bg->desc.x, bg->desc.y};
    }
    if (__ballot(wide)) {  // some span is wider than MAXSPAN pixels: the caller's compose_rows will decline
        if (lane == 0) L.too_wide[half] = 1;
        return;
    }
    const int4* spans = axis == 0 ? T.col : T.row;
    const int count = axis == 0 ? cols : rows;
    for (int q = lane; q < count * MAXSPAN; q += 64) {
        const int g = q / MAXSPAN, i = q % MAXSPAN;
        const int4 sp = spans[g];
        const int p = sp.x + i;
        if (sp.w > 0 && i < sp.y && p >= 0 && p < 64) {
            const int slot = atomicAdd(&T.cover_n[axis][p], 1);
            if (slot < 2) T.cover[axis][p][slot] = g | (sample_index(sp.z, sp.w, i, sp.y) << 8);
        }
    }
}

// The (at most two) covering grid indices of pixel `p` on `axis`, ascending = draw order.  Returns false when
// more than two spans cover the pixel.
template <int GRID>
PG_D bool covering(const ComposeTmp<GRID>& T, int axis, int p, int& ia, int& ib, int& ta, int& tb) {
    const int n = T.cover_n[axis][p];
    int c0 = T.cover[axis][p][0], c1 = T.cover[axis][p][1];
    if (n >= 2 && (c0 & 0xff) > (c1 & 0xff)) {
        const int t = c0;
        c0 = c1;
        c1 = t;
    }
    ia = n >= 1 ? (c0 & 0xff) : -1;
    ta = c0 >> 8;
    ib = n >= 2 ? (c1 & 0xff) : -1;
    tb = c1 >> 8;
    return n <= 2;
}

// "No texel here" marker for the composer's byte offsets: adding it to any valid offset lands beyond the
// atlas (which stays below 2^28 bytes — checked at make time), so the buffer load's hardware range
// check returns 0 = alpha 0 = pixel untouched.  Two markers added together still do not wrap 32 bits.
constexpr uint32_t kNoTexel = 0x40000000u;

// Background + tile layer in one pass.  Preconditions (set up by the game's render kernel, then a barrier):
//   compose_spans() has run; L.base[r * GRID + c] = BYTE offset of the texture of grid cell (x0+c, y0+r) in the
//   atlas, or kNoTexel; all tile textures are tw texels wide; bg / has_bg: the resolved background draw
//   (wave-uniform).
// Writes every pixel of fb (black where nothing is drawn).  Returns false — having written nothing — when the
// layer does not fit the two-candidate scheme (caller falls back to wave_replay).
// TWO: cells whose L.base has bit 0 set use the second tile texture (compose_spans th2); their texel row comes from
// L.row2 and is chosen per lane, at the price of a select and an add in front of every tile load.
// The frame of a game without a tile layer (bossfight): the background over black, nothing else — one candidate per
// pixel instead of the composer's five, no span tables.  Same arithmetic as compose_rows with four absent candidates.
// The background over black from this lane's two offsets (bg_offset of pixel column `lane` and of pixel row `lane`):
// the wave's own rows, nothing shared with the other wave — no barrier.
// kOwnTarget: `fb` holds this wave's rows only (row py_begin of the frame is row 0 of fb).
template <int kRows = kObsH / 2, bool kOwnTarget = false>
PG_D void compose_background_from(uint32_t* fb, const AtlasView& atlas, uint32_t bg_col, uint32_t bg_row, int lane, int half) {
    const __amdgpu_buffer_rsrc_t bg_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(atlas.texels), 0, static_cast<int>(atlas.texel_bytes), 0x00020000);
    // all the rows of the wave at once: 32 gathers in flight, one memory round trip
    const int py_begin = half * kRows, at_begin = kOwnTarget ? 0 : py_begin;
    uint32_t t[kRows];
#pragma unroll
    for (int k = 0; k < kRows; k++)
        t[k] = __builtin_amdgcn_raw_buffer_load_b32(bg_rsrc, bg_col, __builtin_amdgcn_readlane(bg_row, py_begin + k), 0);
    uint32_t least = t[0];
#pragma unroll
    for (int k = 1; k < kRows; k++) least = least < t[k] ? least : t[k];
    if (__ballot(least < 0xff000000u) == 0) {  // opaque everywhere: the texels are the pixels (top byte unread)
#pragma unroll
        for (int k = 0; k < kRows; k++) fb[(at_begin + k) * kObsW + lane] = t[k];
    } else {
#pragma unroll
        for (int k = 0; k < kRows; k++)
            fb[(at_begin + k) * kObsW + lane] = blend_px(0u, t[k], static_cast<int>(t[k] >> 24));
    }
}
PG_D void compose_background(uint32_t* fb, const AtlasView& atlas, const BgAxis& bga, int lane, int half, int halves) {
    // wave 0 resolved the x axis, wave 1 the y axis: trade the per-column / per-row offsets through the target's memory
    fb[64 * half + lane] = bg_offset(bga, lane, half);
    __syncthreads();
    const uint32_t bg_col = fb[lane], bg_row = fb[64 + lane];
    __syncthreads();
    compose_background_from(fb, atlas, bg_col, bg_row, lane, half);
    (void)halves;
    __syncthreads();
}

// BOX: L is the `plain` part of a ComposeLdsBoxed; the cells of its `boxed` table show a texture (same size as the layer's) that exists
// only where the sampled texel lies in columns box.x..box.y and rows box.z..box.w — outside, the cell is no candidate at
// all instead of one that turns out transparent, and the pixel's one-texel attempt goes straight to what is under it.
// Exact whenever every texel outside the box is fully transparent (the word 0 in the atlas), which the caller checks.
// The composer's hand-over tables (ComposeHand, in the frame target's still unused memory) from the span tables
// compose_spans left there: wave 0 the pixel columns, wave 1 the pixel rows.  Leaves the barrier to the caller.
template <int GRID, bool TWO, bool BOX>
PG_D void compose_hand_build(uint32_t* fb, const ComposeLds<GRID>& L, const BgAxis& bga, int tw, int lane, int ablate, int half,
                             int4 box) {
    const ComposeTmp<GRID>& T = compose_tmp<GRID>(fb);
    ComposeHand& H = compose_hand<GRID>(fb);
    if (half == 0) {
        // ---- wave 0, lane = pixel column: the covering grid columns (at most two, neighbours) and their texel columns
        int ca, cb, ua, ub;
        bool ok = covering(T, 0, lane, ca, cb, ua, ub);
        ok = ok && !(cb >= 0 && cb != ca + 1);  // the cell of (·, b) must sit one word behind that of (·, a)
        uint32_t col_a = ca >= 0 ? static_cast<uint32_t>(ua) * 4u : kNoTexel;
        uint32_t col_b = cb >= 0 ? static_cast<uint32_t>(ub) * 4u : kNoTexel;
        uint32_t bg_col = bg_offset(bga, lane, 0);
        if (PG_ABL(ablate, 1024)) {  // timing experiment: every lane of a tile samples texel column 0 (one cache line)
            col_a = ca >= 0 ? 0u : kNoTexel;
            col_b = cb >= 0 ? 0u : kNoTexel;
        }
        if (PG_ABL(ablate, 2048)) bg_col = bg_col == kNoTexel ? kNoTexel : static_cast<uint32_t>(bga.tex_off) * 4u;
        H.col[lane] = make_uint4(bg_col, col_a, col_b, static_cast<uint32_t>(ca >= 0 ? ca : 0) * 4u);
        if (BOX)
            H.col2[lane] = make_uint2(ca >= 0 && ua >= box.x && ua <= box.y ? col_a : kNoTexel,
                                      cb >= 0 && ub >= box.x && ub <= box.y ? col_b : kNoTexel);
        if (__ballot(!ok) && lane == 0) H.bad = 1;
    } else {
        // ---- wave 1, lane = pixel row: the covering grid rows and their texel rows, and the row classes as 64-bit
        // masks (bit py = pixel row py):
        //   second_row  two grid rows cover the row (the seam a tile's padding makes with the next tile);
        //   soft        a grid row covering the row shows a texture that has texels that are not opaque (L.soft_rows;
        //               bit 31: the background does): only batches with such a row look at the alphas they fetched;
        //   hard        … so many of them that the one-texel-per-pixel attempt is not made (L.hard_rows).
        int ra, rb, va, vb;
        bool ok = covering(T, 1, lane, ra, rb, va, vb);
        ok = ok && !(rb >= 0 && rb != ra + 1);  // the cells of (b, ·) must sit one table row behind those of (a, ·)
        const uint32_t row_a = ra >= 0 ? static_cast<uint32_t>(va * tw) * 4u : kNoTexel;
        const uint32_t row_b = rb >= 0 ? static_cast<uint32_t>(vb * tw) * 4u : kNoTexel;
        // byte address, inside the cell table, of grid row a's cells; a row that no grid row covers points at the
        // two sentinel words behind the table, which hold kNoTexel
        const uint32_t cells_a = static_cast<uint32_t>(ra >= 0 ? ra * GRID : GRID * GRID) * 4u;
        H.row[lane] = make_uint4(bg_offset(bga, lane, 1), row_a, row_b, cells_a);
        if (TWO) {  // texel rows of the layer's second, shorter texture
            uint32_t row_a2 = kNoTexel, row_b2 = kNoTexel;
            if (ra >= 0) {
                const int4 sp = T.row2[ra];
                const int i = lane - sp.x;
                if (sp.w > 0 && i >= 0 && i < sp.y) row_a2 = static_cast<uint32_t>(sample_index(sp.z, sp.w, i, sp.y) * tw) * 4u;
            }
            if (rb >= 0) {
                const int4 sp = T.row2[rb];
                const int i = lane - sp.x;
                if (sp.w > 0 && i >= 0 && i < sp.y) row_b2 = static_cast<uint32_t>(sample_index(sp.z, sp.w, i, sp.y) * tw) * 4u;
            }
            H.row2[lane] = make_uint2(row_a2, row_b2);
        }
        if (BOX)
            H.row2[lane] = make_uint2(ra >= 0 && va >= box.z && va <= box.w ? row_a : kNoTexel,
                                      rb >= 0 && vb >= box.z && vb <= box.w ? row_b : kNoTexel);
        const uint32_t soft_bits = static_cast<uint32_t>(L.soft_rows[0] | L.soft_rows[1]);
        const uint32_t hard_bits = static_cast<uint32_t>(L.hard_rows[0] | L.hard_rows[1]);
        // (grids beyond 31 rows alias in these masks: conservative)
        const bool soft_here = (soft_bits >> 31) != 0 || (ra >= 0 && ((soft_bits >> (ra & 31)) & 1u)) ||
                               (rb >= 0 && ((soft_bits >> (rb & 31)) & 1u));
        const bool hard_here = (hard_bits >> 31) != 0 || (ra >= 0 && ((hard_bits >> (ra & 31)) & 1u)) ||
                               (rb >= 0 && ((hard_bits >> (rb & 31)) & 1u));
        const unsigned long long m_second = __ballot(rb >= 0), m_soft = __ballot(soft_here), m_hard = __ballot(hard_here);
        const bool any_bad = __ballot(!ok) != 0;
        if (lane == 0) {
            H.masks[0] = static_cast<uint32_t>(m_second);
            H.masks[1] = static_cast<uint32_t>(m_second >> 32);
            H.masks[2] = static_cast<uint32_t>(m_soft);
            H.masks[3] = static_cast<uint32_t>(m_soft >> 32);
            H.masks[4] = static_cast<uint32_t>(m_hard);
            H.masks[5] = static_cast<uint32_t>(m_hard >> 32);
            if (any_bad) H.bad = 1;
        }
    }
}

// A game whose camera never moves (chaser: the whole world, always) has the same span and hand-over tables in every
// frame of every env: it runs compose_spans and compose_hand_build once, when the envs are made, keeps the result in
// device memory (compose_prepare, from a two-wavefront kernel of its own) and passes it to compose_rows<…, true>, which
// then only works out the background's offsets — the one thing that differs per env (texture, horizontal shift).
// The row classes of the kept tables are those of the tile layer alone; compose_rows adds the background's (bg_w: its
// descriptor's .w).
template <int GRID>
PG_D void compose_prepare(uint32_t* fb, const ComposeLds<GRID>& L, ComposeHand* out, int lane, int half) {
    ComposeHand& H = compose_hand<GRID>(fb);
    __syncthreads();
    if (half == 0 && lane == 0 && (L.too_wide[0] | L.too_wide[1])) H.bad = 1;
    __syncthreads();
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&H);
    uint32_t* dst = reinterpret_cast<uint32_t*>(out);
    for (int k = lane + 64 * half; k < static_cast<int>(sizeof(ComposeHand) / 4); k += 128) dst[k] = src[k];
}

// What the row loop works from: per lane (= pixel column, and — read across lanes — pixel row) the offsets of its
// candidates, and the three row-class masks (bit py = pixel row py).
struct ComposeRegs {
    uint32_t bg_col, col_a, col_b, cia4;     // as pixel column: background column offset, column offsets a / b, 4 × cell column a
    uint32_t bg_row, row_a, row_b, cells_a;  // as pixel row: background row offset, row offsets a / b, byte offset of grid row a's cells
    uint32_t row_a2, row_b2;                 // TWO: row offsets in the layer's second texture; BOX: in the boxed texture
    uint32_t col_pa, col_pb;                 // BOX: column offsets in the boxed texture
    unsigned long long second_row, soft, hard;
};
template <int GRID, bool TWO, bool BOX, bool UNDER = false>
PG_D bool compose_rows_from(uint32_t* fb, const ComposeLds<GRID>& L, const AtlasView& atlas, const ComposeRegs& R, int lane,
                            int ablate, int half, int halves, const unsigned long long* under = nullptr, uint32_t under_image = 0u,
                            uint32_t* whole_rows = nullptr);

template <int GRID, bool TWO = false, bool BOX = false, bool PREP = false>
PG_D bool compose_rows(uint32_t* fb, const ComposeLds<GRID>& L, const AtlasView& atlas, const BgAxis& bga, int cols, int rows,
                       int tw, int lane, int ablate, int half, int halves, int4 box = make_int4(0, -1, 0, -1),
                       const ComposeHand* prepared = nullptr, int bg_w = 0) {
    static_assert(!(TWO && BOX), "the two per-cell row tables share their hand-over slot");
    if (!PREP && (L.too_wide[0] | L.too_wide[1])) return false;
    ComposeHand& H = compose_hand<GRID>(fb);
    if (PREP) {  // the tables are there (prepared): only the background's share is worked out, and traded as usual
        if (half == 0)
            H.col[lane].x = bg_offset(bga, lane, 0);
        else
            H.row[lane].x = bg_offset(bga, lane, 1);
    } else {
        compose_hand_build<GRID, TWO, BOX>(fb, L, bga, tw, lane, ablate, half, box);
    }
    PG_MARK("g_hand");
    __syncthreads();
    const ComposeHand& K = PREP ? *prepared : H;  // (the kept tables are read from device memory: 48 bytes a lane, in the L2)
    if (K.bad) return false;  // (nothing has been written to the target yet)
    uint4 hc = K.col[lane], hr = K.row[lane];
    if (PREP) {
        hc.x = H.col[lane].x;
        hr.x = H.row[lane].x;
    }
    ComposeRegs R;
    R.bg_col = hc.x, R.col_a = hc.y, R.col_b = hc.z, R.cia4 = hc.w;
    R.bg_row = hr.x, R.row_a = hr.y, R.row_b = hr.z, R.cells_a = hr.w;
    R.row_a2 = R.row_b2 = R.col_pa = R.col_pb = 0;
    if (TWO || BOX) {
        const uint2 h2 = K.row2[lane];
        R.row_a2 = h2.x;
        R.row_b2 = h2.y;
    }
    if (BOX) {
        const uint2 h2 = K.col2[lane];
        R.col_pa = h2.x;
        R.col_pb = h2.y;
    }
    auto mask64 = [&](int k) {
        // (through uint32_t: the builtin returns int, and a low word with bit 31 set would fill the high one with ones —
        // which only ever made the complete path's masks too generous: more rows than need be judged soft or two-rowed)
        return static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(K.masks[2 * k]))) |
               (static_cast<unsigned long long>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(K.masks[2 * k + 1]))) << 32);
    };
    R.second_row = mask64(0);
    R.soft = PREP && bg_w != 0 ? ~0ull : mask64(1);
    R.hard = PREP && (bg_w & 2) ? ~0ull : mask64(2);
    __syncthreads();  // everybody has read the set-up tables out of the frame target's memory: it may be written now
    PG_MARK("h_handread");
    return compose_rows_from<GRID, TWO, BOX>(fb, L, atlas, R, lane, ablate, half, halves);
}

// The row loop itself, from the per-lane values and row masks of ComposeRegs — however the caller came by them: out of the
// hand-over tables the two waves have just built in the frame target's memory (compose_rows above), or unpacked from what
// a pre-pass kernel left in device memory (pg_prepass.h).  The cell table L.base must be complete and visible (a barrier
// behind its staging is the caller's); nothing else of L is read.
// UNDER (jumper's compass disc): `under[py]` = the pixels of row py that an opaque texel of a picture drawn over every
// frame — 64 × 64 words at byte `under_image` of the atlas, the same for every env — is going to cover whatever lies
// beneath.  The one-texel attempt fetches THAT texel for them instead of the layer's: the pixel ends up the same (the
// overlay writes the same word again when its turn in the draw order comes, over whatever a sprite left there in
// between), the row's other pixels are judged as ever (the stand-in is opaque), and what is not fetched is the backdrop
// under two thirds of the frame — a scattered texel a pixel, 600 MB a launch of a kernel that runs at the memory's rate.
// A row that takes the general form is composed whole, as ever.
template <int GRID, bool TWO, bool BOX, bool UNDER>
PG_D bool compose_rows_from(uint32_t* fb, const ComposeLds<GRID>& L, const AtlasView& atlas, const ComposeRegs& R, int lane,
                            int ablate, int half, int halves, const unsigned long long* under, uint32_t under_image,
                            uint32_t* whole_rows) {
    // (whole_rows, UNDER's companion: bit r = row r of this wave's 32 went through the general form, i.e. was composed
    // whole — its covered pixels hold what lies beneath, not the stand-in — wave-uniform)
    // All texel reads go through one buffer descriptor over the atlas: 32-bit byte offsets, and out-of-range
    // (= "no candidate") reads return 0 without a branch.
    // (`ablate` bits 5/6 are timing experiments: a descriptor with zero records drops every load through it.)
    const __amdgpu_buffer_rsrc_t bg_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(atlas.texels), 0, PG_ABL(ablate, 32) ? 0 : static_cast<int>(atlas.texel_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t atlas_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(atlas.texels), 0, PG_ABL(ablate, 64) ? 0 : static_cast<int>(atlas.texel_bytes), 0x00020000);
    const uint32_t bg_col = R.bg_col, col_a = R.col_a, col_b = R.col_b, cia4 = R.cia4;
    const uint32_t bg_row = R.bg_row, row_a = R.row_a, row_b = R.row_b, cells_a = R.cells_a;
    const uint32_t row_a2 = R.row_a2, row_b2 = R.row_b2, col_pa = R.col_pa, col_pb = R.col_pb;
    // byte distance from a cell of L.base to the same cell of the boxed table
    constexpr uint32_t kBoxed = BOX ? static_cast<uint32_t>(sizeof(ComposeLds<GRID>)) : 0u;
    static_assert(!BOX || __builtin_offsetof(ComposeLdsBoxed<GRID>, boxed) == sizeof(ComposeLds<GRID>), "boxed table right behind");
    const unsigned long long second_row = R.second_row, soft = R.soft, hard = R.hard;
    constexpr int bg_mod = 255;  // (see BgAxis)

    // Rows in batches: every texel gather of a batch is issued before any pixel is produced, so a batch costs one
    // memory round trip.  Candidates in draw order: background, (row a, col a), (row a, col b), then — on the rows
    // two grid rows cover — (row b, col a), (row b, col b).
    // This loop is written for the fewest instructions of ANY kind: the kernel issues about one instruction per
    // SIMD issue slot whatever the mix (measured: scalar and vector instructions cost the same here), so one
    // cross-lane read per row value beats a packed word that scalar code has to take apart.
#ifndef PG_BATCH
#define PG_BATCH 8
#endif
    constexpr int kBatch = PG_BATCH;
    static_assert(kBatch == 8, "the batch's slice of the row masks is taken as one byte");
#ifndef PG_GEN_ROWS
#define PG_GEN_ROWS 4
#endif
    // Rows the general form takes at a time.  Its 5 × kGen texels were the register peak of every kernel that composes
    // (eight rows: caveflyer 106 registers, climber 100, chaser 96, maze 83; four: 82, 64, 78, 46), and it is the rare path.
    constexpr int kGen = PG_GEN_ROWS;
    static_assert(kBatch % kGen == 0, "");
    if (PG_ABL(ablate, 128)) {  // timing experiment: everything but the row loop
        __syncthreads();
        return true;
    }
    const char* const cells = reinterpret_cast<const char*>(L.base) + cia4;  // per lane: its column a (b = next word)
    auto texel_of = [&](uint32_t cell, uint32_t col, uint32_t row_first, uint32_t row_second) {
        if (TWO) {  // the cell says which of the layer's two textures it shows (bit 0): its row offset is per lane
            const uint32_t row = (cell & 1u) ? row_second : row_first;
            return __builtin_amdgcn_raw_buffer_load_b32(atlas_rsrc, (cell & ~3u) + col + row, 0, 0);
        }
        return __builtin_amdgcn_raw_buffer_load_b32(atlas_rsrc, cell + col, row_first, 0);
    };
    auto boxed_texel = [&](uint32_t plain, uint32_t boxed) {
        return __builtin_amdgcn_raw_buffer_load_b32(atlas_rsrc, plain < boxed ? plain : boxed, 0, 0);
    };
    // The general form of one batch of rows: every candidate of every pixel is fetched, then resolved.  `todo`: the rows
    // of the batch it is wanted for (bit k = row py0 + k, wave-uniform); the others are left as they are.
    // The rows are named one by one (py_of[k], wave-uniform; bit k of `todo`: wanted): the rows of a wave's 32 that need this form
    // are few and scattered — one in five or six in chaser, where a wall tile's translucent edge is sampled — and taking
    // them kGen at a time wherever they lie, instead of by aligned groups of kGen, is fewer trips through memory.
    auto general_batch = [&](const int (&py_of)[kGen], uint32_t todo) {
        uint32_t t[kGen][3], u[kGen][2];
        uint32_t seconds = 0, softs = 0;
#pragma unroll
        for (int k = 0; k < kGen; k++) {
            seconds |= static_cast<uint32_t>((second_row >> py_of[k]) & 1ull) << k;
            softs |= static_cast<uint32_t>((soft >> py_of[k]) & 1ull) << k;
        }
        seconds &= todo;
        softs &= todo;
        // (opaque to the compiler: it would otherwise keep the 32 per-row tests of the attempt below alive in scalar
        // registers for this rarely taken path, and spill them)
        asm volatile("" : "+s"(seconds));
#pragma unroll
        for (int k = 0; k < kGen; k++) {
            if (!(todo & (1u << k))) continue;
            const int py = py_of[k];
            const uint32_t s_bg = __builtin_amdgcn_readlane(bg_row, py);
            const uint32_t s_a = __builtin_amdgcn_readlane(row_a, py);
            const uint32_t* cp = reinterpret_cast<const uint32_t*>(cells + __builtin_amdgcn_readlane(cells_a, py));
            const uint32_t s_a2 = (TWO || BOX) ? __builtin_amdgcn_readlane(row_a2, py) : 0u;
            t[k][0] = __builtin_amdgcn_raw_buffer_load_b32(bg_rsrc, bg_col, s_bg, 0);
            if (BOX) {  // a cell sits in one table at most: the smaller of the two offsets is the one that exists, if any
                const uint32_t* bp = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(cp) + kBoxed);
                t[k][1] = boxed_texel(cp[0] + col_a + s_a, bp[0] + col_pa + s_a2);
                t[k][2] = boxed_texel(cp[1] + col_b + s_a, bp[1] + col_pb + s_a2);
                continue;
            }
            t[k][1] = texel_of(cp[0], col_a, s_a, s_a2);
            t[k][2] = texel_of(cp[1], col_b, s_a, s_a2);
        }
        if (seconds) {
#pragma unroll
            for (int k = 0; k < kGen; k++) {
                if (seconds & (1u << k)) {
                    const int py = py_of[k];
                    const uint32_t s_b = __builtin_amdgcn_readlane(row_b, py);
                    const uint32_t* cp = reinterpret_cast<const uint32_t*>(cells + __builtin_amdgcn_readlane(cells_a, py));
                    const uint32_t s_b2 = (TWO || BOX) ? __builtin_amdgcn_readlane(row_b2, py) : 0u;
                    if (BOX) {
                        const uint32_t* bp = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(cp) + kBoxed);
                        u[k][0] = boxed_texel(cp[GRID] + col_a + s_b, bp[GRID] + col_pa + s_b2);
                        u[k][1] = boxed_texel(cp[GRID + 1] + col_b + s_b, bp[GRID + 1] + col_pb + s_b2);
                        continue;
                    }
                    u[k][0] = texel_of(cp[GRID], col_a, s_b, s_b2);
                    u[k][1] = texel_of(cp[GRID + 1], col_b, s_b, s_b2);
                }
            }
        }
        // Does the batch hold a translucent texel (alpha not in {0, 255})?  Asked only when one of its rows shows a
        // texture that has any ((a + 1) & 0xFE is zero exactly for 0 and 255).
        bool blend = bg_mod != 255;
        if (!blend && softs != 0u && !PG_ABL(ablate, 512)) {
            uint32_t translucent = 0;
#pragma unroll
            for (int k = 0; k < kGen; k++) {
                if (!(todo & (1u << k))) continue;
                translucent |= (((t[k][0] >> 24) + 1u) | ((t[k][1] >> 24) + 1u) | ((t[k][2] >> 24) + 1u)) & 0xFEu;
                if (seconds & (1u << k)) translucent |= (((u[k][0] >> 24) + 1u) | ((u[k][1] >> 24) + 1u)) & 0xFEu;
            }
            blend = __ballot(translucent != 0) != 0;
        }
        if (!blend) {
            // Opaque-or-absent everywhere in the batch: OVER is "last drawn wins" (what S4 yields for a = 0 / 255).
            // An absent or fully transparent texel is the word 0 (the atlas loader clears the colour of alpha-0
            // texels); an opaque one has 0xFF on top.  Masking the top byte down to the candidate's rank in draw
            // order turns "last drawn non-empty" into a plain unsigned maximum.  The rank stays in the target's top
            // byte, which nothing reads (blend_px, wave_store_rows).
            uint32_t pix[kGen];
#pragma unroll
            for (int k = 0; k < kGen; k++)
                if (todo & (1u << k)) pix[k] = max3_u32(t[k][0] & 0x00ffffffu, t[k][1] & 0x01ffffffu, t[k][2] & 0x02ffffffu);
            if (seconds) {
#pragma unroll
                for (int k = 0; k < kGen; k++)
                    if (seconds & (1u << k)) {  // wave-uniform: a real branch (the empty asm keeps it from becoming a select)
                        pix[k] = max3_u32(pix[k], u[k][0] & 0x03ffffffu, u[k][1] & 0x04ffffffu);
                        asm volatile("" : "+v"(pix[k]));
                    }
            }
#pragma unroll
            for (int k = 0; k < kGen; k++)
                if (todo & (1u << k)) fb[py_of[k] * kObsW + lane] = pix[k];
        } else {
#pragma unroll
            for (int k = 0; k < kGen; k++) {
                if (!(todo & (1u << k))) continue;
                uint32_t pix = 0;
                int a = static_cast<int>(t[k][0] >> 24);
                if (bg_mod != 255) a = static_cast<int>(div255(static_cast<uint32_t>(a * bg_mod)));
                pix = blend_px(pix, t[k][0], a);
                pix = blend_px(pix, t[k][1], static_cast<int>(t[k][1] >> 24));
                pix = blend_px(pix, t[k][2], static_cast<int>(t[k][2] >> 24));
                if (seconds & (1u << k)) {
                    pix = blend_px(pix, u[k][0], static_cast<int>(u[k][0] >> 24));
                    pix = blend_px(pix, u[k][1], static_cast<int>(u[k][1] >> 24));
                }
                fb[py_of[k] * kObsW + lane] = pix;
            }
        }
    };

    // The vector memory pipe takes a wavefront's 64 scattered dwords four lanes a clock — 16 clocks a gather, whatever
    // it hits — and at five gathers a row that pipe, not the ALUs, is what this kernel saturates.  So a batch first
    // fetches ONE texel per pixel: the candidate drawn LAST among those that exist there (a cell that holds a tile,
    // else the background).  If every texel it gets is opaque that is the picture — nothing below can show — and the
    // common frame (solid ground, walls, opaque backdrops) needs one gather per row instead of three to five.  Any
    // texel that is not opaque (a crate's rounded corner, lava's surface, a cut-out backdrop, no candidate at all)
    // sends the whole batch through the general form above, which gives the same pixels by construction.
    auto tile_at = [&](uint32_t cell, uint32_t col, uint32_t row_first, uint32_t row_second) {  // ≥ kNoTexel: none
        if (TWO) return (cell & ~3u) + col + ((cell & 1u) ? row_second : row_first);
        return cell + col + row_first;
    };
    // The attempt is made for all the rows of the wave at once — 32 gathers in flight, one memory round trip for the
    // whole half frame — and judged in groups of kBatch rows, the unit the general form works in.
    const int py_begin = half * (kObsH / halves);
    constexpr int kRows = kObsH / 2;
    static_assert(kRows % kBatch == 0, "");
    if (halves != 2) return false;  // (every render kernel runs two wavefronts per env)
    const uint32_t hards = PG_ABL(ablate, 16384) ? 0u : static_cast<uint32_t>(hard >> py_begin);  // this wave's 32 rows
    if (whole_rows != nullptr) *whole_rows = 0xffffffffu;
    if (bg_mod != 255 || hards == 0xffffffffu) {  // nothing worth attempting
        for (int py0 = py_begin; py0 < py_begin + kRows; py0 += kGen) {
            int py_of[kGen];
#pragma unroll
            for (int k = 0; k < kGen; k++) py_of[k] = py0 + k;
            general_batch(py_of, (1u << kGen) - 1u);
        }
        __syncthreads();
        return true;
    }
    const uint32_t seconds32 = static_cast<uint32_t>(second_row >> py_begin);
    // (UNDER: the masks of this wave's 32 rows are read as CONSTANT memory — scalar loads, the mask straight into the scalar
    // pair the select takes; as a vector load handed round with cross-lane reads they were four instructions a row, and
    // the kernel lost what the fetches saved)
    using under_ptr = const __attribute__((address_space(4))) unsigned long long*;
    const under_ptr under_rows = UNDER ? (under_ptr)(reinterpret_cast<unsigned long long>(under + py_begin)) : (under_ptr)0;
    const uint32_t under_lane = under_image + static_cast<uint32_t>(py_begin * kObsW * 4) + static_cast<uint32_t>(lane) * 4u;
    // The gathers land straight in the target: `buffer_load … lds` writes lane l's dword to LDS at M0 + 4·l, which is
    // exactly pixel (row, l) of a row-major target — no result registers (32 of them otherwise, the kernel's register
    // peak), no stores.  Out-of-range lanes write 0 (checked on the hardware: tools/probe/lds_direct_load.hip).
    using lds_ptr = __attribute__((address_space(3))) void*;
    // Eight rows at a time: first every cell word the rows may need is requested from the table, then the addresses
    // are worked out and the gathers leave.  One wait for the table per eight rows instead of one per row.
    //
    // "The candidate drawn last among those that exist" without a compare and a select per candidate: a candidate that
    // exists has a byte offset below 2^28 (the atlas is smaller: checked at make time), one that does not is at least
    // 2^30 (kNoTexel, from whichever of its three terms is missing).  Add the candidate's rank from the END of the draw
    // order, times 2^28, and the unsigned MINIMUM is the existing candidate drawn last — or, when none exists, a word
    // of at least 2^30.  The ranks sit in the per-lane column terms (loop-invariant) and in the scalar row terms.
    // Clearing bits 28-29 leaves the offset, or something beyond the atlas that reads as 0.  On rows two grid rows
    // cover the three candidates of the upper grid row are settled first and their winner enters the second minimum.
    constexpr uint32_t kRank = 1u << 28, kOffsetBits = 0xcfffffffu;  // (several missing terms may carry into bit 31)
    const uint32_t bg_col_r = bg_col + 2u * kRank, col_a_r = col_a + kRank;  // (·, b) is drawn after (·, a): rank 0
    const uint32_t col_pa_r = col_pa + kRank;  // (BOX: a boxed cell has the rank of its place, like a plain one)
    auto min3_u32 = [](uint32_t a, uint32_t b, uint32_t c) {  // v_min3_u32
        const uint32_t m = a < b ? a : b;
        return m < c ? m : c;
    };
    // (Measured and rejected, round 4 — commit bd6a4f0 has the loop: the rows walked by GRID row, min(cell a + column a,
    // cell b + column b) worked out once per grid row and the cell words read once per grid row instead of once per pixel
    // row: six vector instructions a row instead of eleven, bit-exact, and slower — coinrun's render 0.401 -> 0.408 ms,
    // maze's 0.345 -> 0.381, caveflyer's 0.540 -> 0.555: a loop of wave-uniform but dynamic length waits for its two cell
    // words once per grid row where the flat form below has eight rows' worth in flight.)
    {
#pragma unroll
    for (int g = 0; g < kRows / kBatch; g++) {
        uint32_t cell_aa[kBatch], cell_ab[kBatch], cell_ba[kBatch], cell_bb[kBatch];
        uint32_t box_aa[BOX ? kBatch : 1], box_ab[BOX ? kBatch : 1];
        unsigned long long covered[UNDER ? kBatch : 1];
        if (UNDER) {  // (the batch's eight masks leave with its cell words)
#pragma unroll
            for (int k = 0; k < kBatch; k++) covered[k] = under_rows[g * kBatch + k];
        }
#pragma unroll
        for (int k = 0; k < kBatch; k++) {
            const int py = py_begin + g * kBatch + k;
            const uint32_t* cp = reinterpret_cast<const uint32_t*>(cells + __builtin_amdgcn_readlane(cells_a, py));
            cell_aa[k] = cp[0];
            cell_ab[k] = cp[1];
            cell_ba[k] = cp[GRID];  // (used on the rows two grid rows cover; reading them anyway keeps this loop straight)
            cell_bb[k] = cp[GRID + 1];
            if (BOX) {
                const uint32_t* bp = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(cp) + kBoxed);
                box_aa[k] = bp[0];
                box_ab[k] = bp[1];
            }
        }
#pragma unroll
        for (int k = 0; k < kBatch; k++) {
            const int py = py_begin + g * kBatch + k;
            const uint32_t s_bg = __builtin_amdgcn_readlane(bg_row, py);
            const uint32_t s_a = __builtin_amdgcn_readlane(row_a, py);
            const uint32_t s_a2 = (TWO || BOX) ? __builtin_amdgcn_readlane(row_a2, py) : 0u;
            // draw order: background, (a, a), (a, b), (b, a), (b, b) — the last one there wins
            uint32_t at = min3_u32(bg_col_r + s_bg, tile_at(cell_aa[k], col_a_r, s_a, s_a2), tile_at(cell_ab[k], col_b, s_a, s_a2));
            if (BOX) at = min3_u32(at, box_aa[k] + col_pa_r + s_a2, box_ab[k] + col_pb + s_a2);
            at &= kOffsetBits;
            if (seconds32 & (1u << (g * kBatch + k))) {  // wave-uniform
                const uint32_t s_b = __builtin_amdgcn_readlane(row_b, py);
                const uint32_t s_b2 = (TWO || BOX) ? __builtin_amdgcn_readlane(row_b2, py) : 0u;
                at = min3_u32(at + 2u * kRank, tile_at(cell_ba[k], col_a_r, s_b, s_b2), tile_at(cell_bb[k], col_b, s_b, s_b2));
                if (BOX) {  // (the boxed cells of the second grid row are read here: one pixel row in five or six has one)
                    const uint32_t* bp = reinterpret_cast<const uint32_t*>(cells + __builtin_amdgcn_readlane(cells_a, py) + kBoxed);
                    at = min3_u32(at, bp[GRID] + col_pa_r + s_b2, bp[GRID + 1] + col_pb + s_b2);
                }
                at &= kOffsetBits;
                asm volatile("" : "+v"(at));  // keeps the branch a branch
            }
            if (PG_ABL(ablate, 32768)) {
                // Timing experiment (what a tile mini-atlas in LDS could buy AT MOST): the row's
                // texels come out of LDS — one ds_read_b32 at an address that scatters like a mini-atlas's would, one
                // ds_write_b32 — instead of through the gather; wrong pixels (opaque, so no row takes the general form), the
                // cost structure of a row served entirely by LDS.  Bit 16 on top: every second row only (a frame
                // half tiles, half backdrop).
                if (!PG_ABL(ablate, 65536) || (k & 1)) {
                    fb[py * kObsW + lane] = 0xff000000u | static_cast<uint32_t>(L.base[(at >> 2) % (GRID * GRID)]);
                    continue;
                }
            }
            if (UNDER) {  // (the row's mask as a scalar pair: one select on it, one add for the stand-in's address)
                const int r = g * kBatch + k;
                const uint32_t stand_in = under_lane + static_cast<uint32_t>(r * kObsW * 4);
                uint32_t chosen;
                asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(chosen) : "v"(at), "v"(stand_in), "s"(covered[k]));
                at = chosen;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(atlas_rsrc, (lds_ptr)(fb + py * kObsW), 4, static_cast<int>(at), 0, 0, 0);
        }
    }
    }
    PG_MARK("i_fast_issue");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the compiler does not track LDS-direct loads: wait for them here
    // judged in groups of kBatch rows: 8 rows = 128 × 16 bytes of the target, two reads per lane
    const uint4* const landed = reinterpret_cast<const uint4*>(fb) + py_begin * (kObsW / 4);
    uint32_t todo32 = 0;
#pragma unroll
    for (int g = 0; g < kRows / kBatch; g++) {
        const uint4 p = landed[g * (kBatch * kObsW / 4) + lane], q = landed[g * (kBatch * kObsW / 4) + 64 + lane];
        // lane l has rows l / 16 (p) and 4 + l / 16 (q) of the batch: the rows that hold a texel that is not opaque, and
        // those the attempt was not meant for, go through the general form; the others are done
        static_assert(kObsW == 64 && kBatch == 8, "sixteen lanes a row");
        uint32_t least_p = p.x < p.y ? p.x : p.y, least_q = q.x < q.y ? q.x : q.y;
        least_p = least_p < p.z ? least_p : p.z;
        least_p = least_p < p.w ? least_p : p.w;
        least_q = least_q < q.z ? least_q : q.z;
        least_q = least_q < q.w ? least_q : q.w;
        uint32_t todo = (hards >> (g * kBatch)) & 0xffu;
        const unsigned long long open_p = __ballot(least_p < 0xff000000u), open_q = __ballot(least_q < 0xff000000u);
        if ((open_p | open_q) != 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if ((open_p >> (16 * r)) & 0xffffull) todo |= 1u << r;
                if ((open_q >> (16 * r)) & 0xffffull) todo |= 16u << r;
            }
        }
        if (PG_ABL(ablate, 4096)) todo = 0xffu;
        if (!PG_ABL(ablate, 8192)) todo32 |= todo << (g * kBatch);  // (bit 13: never)
    }
    if (whole_rows != nullptr) *whole_rows = todo32;
    while (todo32 != 0u) {  // wave-uniform: the rows that go through the general form, kGen at a time
        int py_of[kGen];
        uint32_t bits = 0;
#pragma unroll
        for (int k = 0; k < kGen; k++) {
            py_of[k] = py_begin;
            if (todo32 != 0u) {
                py_of[k] = py_begin + __builtin_ctz(todo32);
                todo32 &= todo32 - 1u;
                bits |= 1u << k;
            }
        }
        general_batch(py_of, bits);
    }
    __syncthreads();
    return true;
}

// RGB pack (coinrun.cpp:377-388): obs[3k+c] = pix[4k+c]; 4 pixels → 12 bytes per lane per pass,
// 768 contiguous bytes per wave store.
PG_D void wave_store_obs(const uint32_t* fb, uint8_t* obs_env, int lane, int half = 0, int halves = 1) {
    Rgb4* out = reinterpret_cast<Rgb4*>(obs_env);
    const uint4* in = reinterpret_cast<const uint4*>(fb);
    for (int g = lane + 64 * half; g < kFbWords / 4; g += 64 * halves) {
        const uint4 p = in[g];
        Rgb4 o;  // three byte permutes; the target's top bytes (the composer leaves draw ranks there) are dropped
        o.a = __builtin_amdgcn_perm(p.y, p.x, 0x04020100u);  // x0 x1 x2 y0
        o.b = __builtin_amdgcn_perm(p.z, p.y, 0x05040201u);  // y1 y2 z0 z1
        o.c = __builtin_amdgcn_perm(p.w, p.z, 0x06050402u);  // z2 w0 w1 w2
        // Streaming stores (`global_store_dwordx3 … nt`): the 12 KB of an observation are written once and not read
        // again by this launch; keeping them out of the L2's way leaves it to the atlas and the state (measured:
        // render 0.728 → 0.681 ms, coinrun 66.5 → 70.6 M env-steps/s).
        __builtin_nontemporal_store(o.a, &out[g].a);
        __builtin_nontemporal_store(o.b, &out[g].b);
        __builtin_nontemporal_store(o.c, &out[g].c);
    }
}

}  // namespace pg
