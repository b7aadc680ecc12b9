// Device-side rasteriser shared by the game render kernels (gfx950 only).
//
// One 64-lane wavefront owns one env: its 64×64 target lives in LDS as packed 0x00BBGGRR words
// (16 KiB), draw calls are replayed back-to-front exactly in the reference's order (painter's
// algorithm, so overdraw and blend order are right by construction), and the finished frame is
// streamed to the observation slab as contiguous 768-byte wave stores (4 pixels → 3 dwords per lane).
// Raster rules S1–S5: DESIGN.md §raster-spec (oracle twin: oracle/pgo_raster.cpp spec_blit).
#pragma once

#include "pg_engine.h"
#include "pg_geom.h"

namespace pg {

constexpr int kFbWords = kObsW * kObsH;

struct __attribute__((packed, aligned(4))) Rgb4 {
    uint32_t a, b, c;
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
    b.tex = __builtin_amdgcn_readlane(mine.tex, src);
    b.flip_mod = __builtin_amdgcn_readlane(mine.flip_mod, src);
    return b;
}

PG_D void blend_into(uint32_t* fb, int idx, uint32_t texel, int mod) {
    int a = static_cast<int>(texel >> 24);
    if (mod != 255) a = a * mod / 255;
    if (a == 0) return;
    fb[idx] = blend_px(fb[idx], texel, a);
}

// All 64 lanes execute one (wave-uniform) blit into the LDS target.
PG_D void wave_blit(uint32_t* fb, const AtlasView& atlas, const Blit& b, int lane) {
    const int x0 = b.dx > 0 ? b.dx : 0, y0 = b.dy > 0 ? b.dy : 0;
    const int x1 = (b.dx + b.dw) < kObsW ? (b.dx + b.dw) : kObsW;
    const int y1 = (b.dy + b.dh) < kObsH ? (b.dy + b.dh) : kObsH;
    const int cw = x1 - x0, ch = y1 - y0;
    if (cw <= 0 || ch <= 0) return;
    const int4 d = atlas.desc[b.tex];
    const uint32_t* tex = atlas.texels + d.x;
    const int tw = d.y;
    const int mod = b.flip_mod & 0xff;
    const bool fh = (b.flip_mod & kFlipH) != 0, fv = (b.flip_mod & kFlipV) != 0;

    if (cw > 32) {
        // Wide blit (backgrounds): lane = column, source column fixed per lane, one row per iteration.
        const int x = x0 + lane;
        const bool on = lane < cw;
        int i = x - b.dx;
        if (fh) i = b.dw - 1 - i;
        const int u = on ? sample_index(b.sx, b.sw, i, b.dw) : b.sx;
        for (int y = y0; y < y1; y++) {
            int j = y - b.dy;
            if (fv) j = b.dh - 1 - j;
            const int v = sample_index(b.sy, b.sh, j, b.dh);
            if (on) blend_into(fb, y * kObsW + x, tex[v * tw + u], mod);
        }
    } else {
        const int total = cw * ch;
        for (int p = lane; p < total; p += 64) {
            const int ry = p / cw;
            const int rx = p - ry * cw;
            const int x = x0 + rx, y = y0 + ry;
            int i = x - b.dx, j = y - b.dy;
            if (fh) i = b.dw - 1 - i;
            if (fv) j = b.dh - 1 - j;
            const int u = sample_index(b.sx, b.sw, i, b.dw);
            const int v = sample_index(b.sy, b.sh, j, b.dh);
            blend_into(fb, y * kObsW + x, tex[v * tw + u], mod);
        }
    }
}

// Replays the draws held by the lanes flagged in `mask`, in ascending lane order.
PG_D void wave_replay(uint32_t* fb, const AtlasView& atlas, const Blit& mine, unsigned long long mask, int lane) {
    while (mask) {
        const int src = __builtin_ctzll(mask);
        mask &= mask - 1;
        const Blit b = blit_from_lane(mine, src);
        wave_blit(fb, atlas, b, lane);
        __syncthreads();  // single-wave workgroup: orders the LDS traffic of consecutive draws
    }
}

// SDL_RenderClear with (0,0,0,255): coinrun.cpp:447-448.
PG_D void wave_clear(uint32_t* fb, int lane) {
    uint4* p = reinterpret_cast<uint4*>(fb);
    for (int k = lane; k < kFbWords / 4; k += 64) p[k] = make_uint4(0, 0, 0, 0);
    __syncthreads();
}

// RGB pack (coinrun.cpp:377-388): obs[3k+c] = pix[4k+c]; 4 pixels → 12 bytes per lane per pass,
// 768 contiguous bytes per wave store.
PG_D void wave_store_obs(const uint32_t* fb, uint8_t* obs_env, int lane) {
    Rgb4* out = reinterpret_cast<Rgb4*>(obs_env);
    const uint4* in = reinterpret_cast<const uint4*>(fb);
    for (int g = lane; g < kFbWords / 4; g += 64) {
        const uint4 p = in[g];
        Rgb4 o;
        o.a = p.x | (p.y << 24);
        o.b = (p.y >> 8) | (p.z << 16);
        o.c = (p.z >> 16) | (p.w << 8);
        out[g] = o;
    }
}

}  // namespace pg
