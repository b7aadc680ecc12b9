// The human-size frame of cenv_render (games/*/<game>.cpp `render_game(false)`): the same draw list as
// the observation, rasterised at W×H with camera_scale = zoom · W / 64, for ONE env.
//
// This is a debugging / viewer path, not the hot path, and it is deliberately plain: one workgroup, a W×H target of
// 0x00BBGGRR words in global memory, every draw resolved by all threads (uniform arguments → uniform control flow)
// and rasterised by all of them with the same raster spec as the observation (pg_geom.h S1–S6), a barrier after
// every draw.  No row composer, no LDS target: neither fits an arbitrary W×H, and a frame is needed once per
// keystroke, not 65 536 times per millisecond.
#pragma once

#include "pg_engine.h"
#include "pg_geom.h"
#include "pg_sincos.h"

namespace pg {

struct FrameTarget {
    uint32_t* px;  // [h][w]
    int w, h;
};

struct FramePainter {
    FrameTarget t;
    AtlasView atlas;
    Camera cam;
    int tid, nt;

    PG_D int4 desc(int tex) const { return atlas.desc[tex]; }

    PG_D void clear() {  // SDL_RenderClear with (0,0,0,255)
        for (int k = tid; k < t.w * t.h; k += nt) t.px[k] = 0;
        __syncthreads();
    }

    // All threads rasterise one resolved draw (raster spec S3–S6), then meet: the next draw may touch these pixels.
    PG_D void blit(const Blit& b) {
        int x_lo, y_lo, x_hi, y_hi;  // inclusive
        const bool rotated = (b.flip_mod & kRotated) != 0;
        if (rotated) {
            long long reach = 1;
            const long long diag2 = (long long)b.dw * b.dw + (long long)b.dh * b.dh;
            while (reach * reach * 4 < diag2) reach++;
            reach += 1;
            const long long cx2 = 2LL * b.dx + b.dw, cy2 = 2LL * b.dy + b.dh;
            x_lo = static_cast<int>((cx2 - 2 * reach) / 2 - 1);
            x_hi = static_cast<int>((cx2 + 2 * reach) / 2 + 1);
            y_lo = static_cast<int>((cy2 - 2 * reach) / 2 - 1);
            y_hi = static_cast<int>((cy2 + 2 * reach) / 2 + 1);
        } else {
            x_lo = b.dx;
            y_lo = b.dy;
            x_hi = b.dx + b.dw - 1;
            y_hi = b.dy + b.dh - 1;
        }
        if (x_lo < 0) x_lo = 0;
        if (y_lo < 0) y_lo = 0;
        if (x_hi > t.w - 1) x_hi = t.w - 1;
        if (y_hi > t.h - 1) y_hi = t.h - 1;
        const int fw = x_hi - x_lo + 1, fh = y_hi - y_lo + 1;
        const int mod = b.flip_mod & 0xff;
        if (fw > 0 && fh > 0) {
            const long long total = (long long)fw * fh;
            for (long long p = tid; p < total; p += nt) {
                const int ry = static_cast<int>(p / fw);
                const int X = x_lo + static_cast<int>(p - (long long)ry * fw), Y = y_lo + ry;
                int i, j;
                if (rotated) {
                    const long long px = 2LL * (X - b.dx) + 1 - b.dw, py = 2LL * (Y - b.dy) + 1 - b.dh;
                    const long long lx = px * b.rot_cs + py * b.rot_sn + (long long)b.dw * 65536;
                    const long long ly = -px * b.rot_sn + py * b.rot_cs + (long long)b.dh * 65536;
                    if (lx < 0 || ly < 0 || lx >= 2LL * b.dw * 65536 || ly >= 2LL * b.dh * 65536) continue;
                    i = static_cast<int>(lx >> 17);
                    j = static_cast<int>(ly >> 17);
                } else {
                    i = X - b.dx;
                    j = Y - b.dy;
                    if (b.flip_mod & kFlipH) i = b.dw - 1 - i;
                    if (b.flip_mod & kFlipV) j = b.dh - 1 - j;
                }
                const int u = b.sx + static_cast<int>(((2LL * i + 1) * b.sw) / (2LL * b.dw));
                const int v = b.sy + static_cast<int>(((2LL * j + 1) * b.sh) / (2LL * b.dh));
                const uint32_t texel = atlas.texels[b.tex_off + v * b.tex_w + u];
                int a = static_cast<int>(texel >> 24);
                if (mod != 255) a = static_cast<int>(div255(static_cast<uint32_t>(a * mod)));
                if (a == 0) continue;
                uint32_t* d = &t.px[Y * t.w + X];
                *d = blend_px(*d, texel, a);
            }
        }
        __syncthreads();
    }

    // Renderer::render_texture (renderer.cpp:5-82)
    PG_D void draw(int tex, float px, float py, float scale, float alpha = 1.0f, bool flip_h = false,
                   bool flip_v = false) {
        const int4 d = desc(tex);
        Blit b;
        if (resolve_draw(cam, d.y, d.z, d.x, px, py, scale, alpha, flip_h, flip_v, b)) blit(b);
    }
    // Renderer::render_texture_rotated (renderer.cpp:84-101)
    PG_D void draw_rotated(int tex, float px, float py, float rotation, float scale, float alpha = 1.0f) {
        const int4 d = desc(tex);
        const float dx = (px - cam.px) * cam.scale + cam.sw * 0.5f;
        const float dy = (py - cam.py) * cam.scale + cam.sh * 0.5f;
        const float dw = d.y * scale * cam.scale;
        const float dh = d.z * scale * cam.scale;
        int mod = 255;
        if (alpha != 1.0f) mod = static_cast<int>(255 * alpha) & 0xff;
        const double deg = rotation * 180.0f / 3.14159265358979323846;
        screen(tex, dx, dy, dw, dh, deg, mod);
    }
    // A raw SDL_RenderTextureRotated in screen space: whole texture, float destination, `deg` degrees about its centre.
    PG_D void screen(int tex, float dx, float dy, float dw, float dh, double deg, int mod = 255) {
        const int4 d = desc(tex);
        if (!(dw >= 1.0f && dh >= 1.0f && dw < 32768.0f && dh < 32768.0f)) return;
        if (!(dx > -32768.0f && dx < 32768.0f && dy > -32768.0f && dy < 32768.0f)) return;
        Blit b;
        b.dx = static_cast<int>(dx);
        b.dy = static_cast<int>(dy);
        b.dw = static_cast<int>(dw);
        b.dh = static_cast<int>(dh);
        b.sx = 0;
        b.sy = 0;
        b.sw = d.y;
        b.sh = d.z;
        b.tex_off = d.x;
        b.tex_w = d.y;
        b.flip_mod = mod;
        b.rot_sn = 0;
        b.rot_cs = 65536;
        if (deg != 0.0) {
            const float theta = static_cast<float>(deg * (3.14159265358979323846 / 180.0));
            b.rot_sn = static_cast<int>(floor(static_cast<double>(sc_sinf(theta)) * 65536.0 + 0.5));
            b.rot_cs = static_cast<int>(floor(static_cast<double>(sc_cosf(theta)) * 65536.0 + 0.5));
            b.flip_mod |= kRotated;
        }
        blit(b);
    }
    // The tile window of System_Tilemap::render (e.g. coinrun/tilemap.cpp:294-304): inclusive cell ranges.
    PG_D void window(int& x0, int& y0, int& x1, int& y1) const {
        const float vx = (cam.px - cam.sw * 0.5f / cam.scale) * kPxUnit;
        const float vy = (cam.py - cam.sh * 0.5f / cam.scale) * kPxUnit;
        const float vw = cam.sw * kPxUnit / cam.scale, vh = cam.sh * kPxUnit / cam.scale;
        x0 = static_cast<int>(floorf(vx));
        y0 = static_cast<int>(floorf(vy));
        x1 = static_cast<int>(ceilf(vx + vw));
        y1 = static_cast<int>(ceilf(vy + vh));
    }
};

constexpr int kFrameThreads = 256;

}  // namespace pg
