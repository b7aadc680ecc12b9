// Geometry helpers shared by every game kernel: AABB tests and the
// camera / cull / crop arithmetic of the reference renderer that turns one draw call into an
// integer blit (row R1 + the raster spec S1–S2 of DESIGN.md).  All float math is written in the
// reference's operation order; the build uses -ffp-contract=off so no FMA is formed.
#pragma once

#include "pg_defs.h"

#if !defined(__HIPCC__)
#include <cmath>
#endif

namespace pg {

struct Box {
    float x, y, w, h;
};

// games/*/helpers.cpp:40-46 — strict inequality overlap.
PG_HD bool box_hit(const Box& a, const Box& b) {
    return a.x < b.x + b.w && a.x + a.w > b.x && a.y < b.y + b.h && a.y + a.h > b.y;
}

// games/*/helpers.cpp:48-108 — raylib GetCollisionRec.
PG_HD Box box_overlap(const Box& a, const Box& b) {
    Box r{0.0f, 0.0f, 0.0f, 0.0f};
    if (!box_hit(a, b)) return r;
    const float ddx = fabsf(a.x - b.x);
    const float ddy = fabsf(a.y - b.y);
    const bool left = a.x <= b.x;
    const bool top = a.y <= b.y;
    r.x = left ? b.x : a.x;
    r.y = top ? b.y : a.y;
    r.w = (left ? a.w : b.w) - ddx;
    r.h = (top ? a.h : b.h) - ddy;
    const float wcap = a.w > b.w ? b.w : a.w;
    const float hcap = a.h > b.h ? b.h : a.h;
    if (r.w >= wcap) r.w = wcap;
    if (r.h >= hcap) r.h = hcap;
    return r;
}

// One resolved draw call: destination rectangle in target pixels and the texel rectangle it
// samples, both integer (raster spec S1/S2).  `dw <= 0` marks "nothing to draw".
struct Blit {
    int32_t dx, dy, dw, dh;  // destination (may extend outside the 64×64 target; clipped per pixel)
    int32_t sx, sy, sw, sh;  // source texel rect, already intersected with the texture
    int32_t tex;             // atlas texture index
    int32_t flip_mod;        // bit 8: horizontal flip, bit 9: vertical flip, bits 0-7: alpha modulation
};

struct Camera {
    float px, py;   // gr.camera_position (pixels)
    float sw, sh;   // gr.camera_size
    float scale;    // gr.camera_scale
};

constexpr int32_t kFlipH = 1 << 8;
constexpr int32_t kFlipV = 1 << 9;

// Renderer::render_texture (games/*/renderer.cpp:5-82) followed by raster-spec S1/S2.
// tw/th: texture size; pos in world pixels.  Returns false when culled or empty.
PG_HD bool resolve_draw(const Camera& cam, int tw, int th, int tex, float pos_x, float pos_y, float scale, float alpha,
                        bool flip_h, bool flip_v, Blit& out) {
    float sx = 0.0f, sy = 0.0f;
    float sw = static_cast<float>(tw), sh = static_cast<float>(th);
    float dx = (pos_x - cam.px) * cam.scale + cam.sw * 0.5f;
    float dy = (pos_y - cam.py) * cam.scale + cam.sh * 0.5f;
    float dw = tw * scale * cam.scale;
    float dh = th * scale * cam.scale;

    if (dx > cam.sw || dy >= cam.sh || dx + dw < 0 || dy + dh < 0) return false;  // renderer.cpp:14

    if (dx < 0.0f) {  // renderer.cpp:18-26
        float ratio = -dx / dw;
        sx += sw * ratio;
        sw -= sx;
        dw += dx;
        dx = 0.0f;
    }
    if (dx + dw > cam.sw) {  // renderer.cpp:28-34
        float ratio = (dx + dw - cam.sw) / dw;
        sw = sw * (1.0f - ratio);
        dw = cam.sw - dx;
    }
    if (dy < 0.0f) {  // renderer.cpp:36-44
        float ratio = -dy / dh;
        sy += sh * ratio;
        sh -= sy;
        dh += dy;
        dy = 0.0f;
    }
    if (dy + dh > cam.sh) {  // renderer.cpp:46-52
        float ratio = (dy + dh - cam.sh) / dh;
        sh = sh * (1.0f - ratio);
        dh = cam.sh - dy;
    }

    int mod = 255;
    if (alpha != 1.0f) mod = static_cast<int>(255 * alpha) & 0xff;  // Uint8 parameter (renderer.cpp:56-57)

    int padding = static_cast<int>(ceilf(1.0f / (scale * cam.scale)));  // renderer.cpp:59
    int rx = static_cast<int>(floorf(sx));
    int ry = static_cast<int>(floorf(sy));
    int rw = static_cast<int>(ceilf(sw)) + padding;
    int rh = static_cast<int>(ceilf(sh)) + padding;

    float off_x = sx - rx, off_y = sy - ry;  // renderer.cpp:64-70
    float ratio_x = rw / sw, ratio_y = rh / sh;
    dw *= ratio_x;
    dh *= ratio_y;
    dx -= off_x * (dw / sw);
    dy -= off_y * (dh / sh);

    if (flip_h) rx = tw - rw - rx;  // renderer.cpp:72-74

    // S1: destination by truncation; non-finite, sub-pixel or absurd rectangles draw nothing.
    if (!(dw >= 1.0f && dh >= 1.0f && dw < 32768.0f && dh < 32768.0f)) return false;
    if (!(dx > -32768.0f && dx < 32768.0f && dy > -32768.0f && dy < 32768.0f)) return false;
    out.dx = static_cast<int>(dx);
    out.dy = static_cast<int>(dy);
    out.dw = static_cast<int>(dw);
    out.dh = static_cast<int>(dh);
    // S2: source rect intersected with the texture, destination untouched.
    int x0 = rx, y0 = ry, x1 = rx + rw, y1 = ry + rh;
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 > tw) x1 = tw;
    if (y1 > th) y1 = th;
    out.sx = x0;
    out.sy = y0;
    out.sw = x1 - x0;
    out.sh = y1 - y0;
    if (out.sw <= 0 || out.sh <= 0) return false;
    out.tex = tex;
    out.flip_mod = mod | (flip_h ? kFlipH : (flip_v ? kFlipV : 0));
    return true;
}

// Raster spec S3: nearest texel for destination column/row `i` of `n`, over `len` texels from `start`.
PG_HD int sample_index(int start, int len, int i, int n) { return start + ((2 * i + 1) * len) / (2 * n); }

// Raster spec S4: straight-alpha blend with truncating /255 on one packed pixel (R | G<<8 | B<<16).
// a is the source alpha after modulation; returns the new destination.
PG_HD uint32_t blend_px(uint32_t dst, uint32_t src, int a) {
    if (a >= 255) return src & 0x00ffffffu;
    const int ia = 255 - a;
    uint32_t out = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        int s = static_cast<int>((src >> (8 * c)) & 0xffu);
        int d = static_cast<int>((dst >> (8 * c)) & 0xffu);
        int v = (s * a) / 255 + (ia * d) / 255;
        out |= static_cast<uint32_t>(v) << (8 * c);
    }
    return out;
}

}  // namespace pg
