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
// samples, both integer (raster spec S1/S2).
struct Blit {
    int32_t dx, dy, dw, dh;  // destination (may extend outside the 64×64 target; clipped per pixel)
    int32_t sx, sy, sw, sh;  // source texel rect, already intersected with the texture
    int32_t tex_off;         // first texel of the texture in the atlas array
    int32_t tex_w;           // texture width (row pitch in texels)
    int32_t flip_mod;        // bit 8: horizontal flip, bit 9: vertical flip, bit 10: rotated, bits 0-7: alpha modulation
    int32_t rot_sn, rot_cs;  // rotated draws (raster spec S6): sine / cosine of the angle in 16.16 fixed point
};

struct Camera {
    float px, py;   // gr.camera_position (pixels)
    float sw, sh;   // gr.camera_size
    float scale;    // gr.camera_scale
};

constexpr int32_t kFlipH = 1 << 8;
constexpr int32_t kFlipV = 1 << 9;
constexpr int32_t kRotated = 1 << 10;
constexpr int32_t kStamped = 1 << 11;  // the texture is a pre-scaled stamp (pg_stamps.h): texel (i, j) lands on pixel (i, j), s | a << 24
// Where the list of a stamped, un-rotated, unflipped draw's visible texels is (word offset into the atlas): pg_stamps.h
// stamp_substitute leaves it in the draw's source corner, which a stamp has no use for.
PG_HD uint32_t stamp_list_at(const Blit& b) { return static_cast<uint32_t>(b.sx) | static_cast<uint32_t>(b.sy) << 16; }

// One axis of a resolved draw: destination span [d0, d0+dn) and source span [s0, s0+sn).
struct Span {
    int32_t d0, dn, s0, sn;
};

// Renderer::render_texture (games/*/renderer.cpp:5-82) is separable: every x quantity depends only on
// x inputs and every y quantity only on y inputs (the cull test is an OR of two x and two y conditions on
// the untouched values).  This is one axis of it, followed by raster-spec S1/S2 for that axis.
//   cam_pos/cam_len: camera position and viewport size on this axis; tsize: texture extent on this axis;
//   strict_far: the far-side cull is `>=` on y and `>` on x (renderer.cpp:14).
// … in two parts, for callers that cull many draws and finish few (the render pre-pass, pg_prepass.h): the head is the
// destination before any cropping and the cull test (renderer.cpp:8-14), the tail everything after it.
struct AxisHead {
    float d, dl;
};
PG_HD bool axis_head(float cam_pos, float cam_len, float cam_scale, int tsize, float pos, float scale, bool strict_far,
                     AxisHead& h) {
    h.d = (pos - cam_pos) * cam_scale + cam_len * 0.5f;
    h.dl = tsize * scale * cam_scale;
    return !((strict_far ? h.d >= cam_len : h.d > cam_len) || h.d + h.dl < 0);  // renderer.cpp:14
}
PG_HD bool axis_tail(float cam_len, float cam_scale, int tsize, float scale, bool flip, const AxisHead& h, Span& out) {
    float s = 0.0f;
    float sl = static_cast<float>(tsize);
    float d = h.d;
    float dl = h.dl;

    if (d < 0.0f) {  // renderer.cpp:18-26 / 36-44
        float ratio = -d / dl;
        s += sl * ratio;
        sl -= s;
        dl += d;
        d = 0.0f;
    }
    if (d + dl > cam_len) {  // renderer.cpp:28-34 / 46-52
        float ratio = (d + dl - cam_len) / dl;
        sl = sl * (1.0f - ratio);
        dl = cam_len - d;
    }

    int padding = static_cast<int>(ceilf(1.0f / (scale * cam_scale)));  // renderer.cpp:59
    int r0 = static_cast<int>(floorf(s));
    int rl = static_cast<int>(ceilf(sl)) + padding;
    float off = s - r0;  // renderer.cpp:64-70
    float ratio = rl / sl;
    dl *= ratio;
    d -= off * (dl / sl);
    if (flip) r0 = tsize - rl - r0;  // renderer.cpp:72-74 (the reference only flips x this way)

    // S1: destination by truncation; non-finite, sub-pixel or absurd spans draw nothing.
    if (!(dl >= 1.0f && dl < 32768.0f)) return false;
    if (!(d > -32768.0f && d < 32768.0f)) return false;
    out.d0 = static_cast<int>(d);
    out.dn = static_cast<int>(dl);
    // S2: source span intersected with the texture, destination untouched.
    int a = r0, b = r0 + rl;
    if (a < 0) a = 0;
    if (b > tsize) b = tsize;
    out.s0 = a;
    out.sn = b - a;
    return out.sn > 0;
}
PG_HD bool resolve_axis(float cam_pos, float cam_len, float cam_scale, int tsize, float pos, float scale, bool flip,
                        bool strict_far, Span& out) {
    AxisHead h;
    if (!axis_head(cam_pos, cam_len, cam_scale, tsize, pos, scale, strict_far, h)) return false;
    return axis_tail(cam_len, cam_scale, tsize, scale, flip, h, out);
}

// The full draw call = both axes.  tex_off/tw/th describe the texture; pos in world pixels.
// Returns false when culled or empty.
PG_HD bool resolve_draw(const Camera& cam, int tw, int th, int tex_off, float pos_x, float pos_y, float scale,
                        float alpha, bool flip_h, bool flip_v, Blit& out) {
    Span x, y;
    if (!resolve_axis(cam.px, cam.sw, cam.scale, tw, pos_x, scale, flip_h, false, x)) return false;
    if (!resolve_axis(cam.py, cam.sh, cam.scale, th, pos_y, scale, false, true, y)) return false;
    int mod = 255;
    if (alpha != 1.0f) mod = static_cast<int>(255 * alpha) & 0xff;  // Uint8 parameter (renderer.cpp:56-57)
    out.dx = x.d0;
    out.dw = x.dn;
    out.sx = x.s0;
    out.sw = x.sn;
    out.dy = y.d0;
    out.dh = y.dn;
    out.sy = y.s0;
    out.sh = y.sn;
    out.tex_off = tex_off;
    out.tex_w = tw;
    out.flip_mod = mod | (flip_h ? kFlipH : (flip_v ? kFlipV : 0));
    out.rot_sn = 0;
    out.rot_cs = 65536;
    return true;
}

// A tile layer with a second, shorter texture (the brown theme's 64×53 cap on 64×64 bodies): the pixel rows a grid row's
// cap covers must be among those its body covers — the composer finds a pixel row's candidates from the bodies' spans
// alone.  Only pixels ON the target count: cut by the target's top edge down to its last texel row or two, the cap's
// rectangle comes out of render_texture's crop arithmetic at a negative row, and at the bottom edge it may reach one row
// further beyond the target than the body's (until round 4 both made the frame take the draw-list replay: one
// jumper frame in forty).
PG_HD bool span_nested(int d_in, int n_in, int d_out, int n_out, int len) {
    const int lo_in = d_in > 0 ? d_in : 0, hi_in = d_in + n_in < len ? d_in + n_in : len;
    if (lo_in >= hi_in) return true;
    const int lo_out = d_out > 0 ? d_out : 0, hi_out = d_out + n_out < len ? d_out + n_out : len;
    return lo_in >= lo_out && hi_in <= hi_out;
}

// Raster spec S6, the bounding box of a rotated draw on the target (pg_render.h rot_box).  How tight it may be:
// a pixel with doubled offsets (px, py) from the rectangle's centre is drawn iff a = px·cs + py·sn lies in [−dw·2^16, dw·2^16)
// and b = py·cs − px·sn in [−dh·2^16, dh·2^16).  Then px·N = a·cs − b·sn with N = cs² + sn², so |px| ≤ 2^16·(dw·|cs| +
// dh·|sn|) / N.  sn and cs are 2^16·sin and 2^16·cos rounded to integers, each within 0.51 of the real product, so N =
// 2^32·(1 + e) with |e| < 1.5 / 2^16, and with v = (dw·|cs| + dh·|sn|) / 2^16 ≤ dw + dh:  |px| ≤ v·(1 + 1.5/2^16) ≤ v +
// 2(dw + dh)/2^16 — px being an integer, |px| ≤ rot_extent(dw, dh, |cs|, |sn|); likewise |py| ≤ rot_extent(dh, dw, …).
// px = 2(X − dx) + 1 − dw, so X − dx runs from ceil((dw − 1 − ex) / 2) = rot_first to floor((dw − 1 + ex) / 2) = rot_last.
// (Until round 4 the box had one more unit in ex and one more pixel on every side on top of that: 10 × 10 pixels around a
// 4 × 4 bullet at 45° instead of 6 × 6 — more than a wave's 64 lanes, so every bullet went alone, with a memory round trip
// of its own.  Any superset of the drawn pixels gives the same frame; tests/cpp/test_primitives.cpp sweeps this one.)
PG_HD int rot_extent(int along, int across, int acs, int asn) {
    return static_cast<int>((static_cast<long long>(along) * acs + static_cast<long long>(across) * asn + 2ll * (along + across)) >> 16);
}
PG_HD int rot_first(int dn, int extent) { return (dn - extent) >> 1; }
PG_HD int rot_last(int dn, int extent) { return (dn - 1 + extent) >> 1; }

// Raster spec S6, THIN rectangles (dw ≥ 3·dh: jumper's needle, 30 × 6).  Of a pixel row Y of the target only a short run of
// columns can map back between the rectangle's long edges: with px = 2(X − dx) + 1 − dw, py = 2(Y − dy) + 1 − dh the row
// coordinate ly = −px·sn + (py·cs + dh·2^16) must lie in [0, 2·dh·2^16) — a linear condition on px with slope −sn, met over
// L = dh·2^16 / |sn| pixels, (A, A + L] say.  thin_run_start gives a column at or left of floor(A) − 2 … floor(A): A in
// floats — the quantities are integers below 2^25, the quotient below a few thousand for |sn| ≥ 4096, so the float result
// is within 0.01 of A — rounded down, less one; thin_run_width gives floor(L) + 5 columns from there, which reach
// floor(A) + floor(L) + 1 ≥ floor(A + L), the run's last column, from wherever the start landed.  Every pixel raster spec
// S6 draws on row Y lies in [start, start + width): swept on the host against the exact 64-bit test
// (tests/cpp/test_primitives.cpp test_thin_runs).  pg_render.h wave_blit_rotated scans those runs instead of the rows of
// the bounding box.  (m = 2·dh·2^16 and inv = 1 / sn as floats, worked out once per draw.)
PG_HD int thin_run_width(int dh, int asn) { return (dh * 65536) / asn + 5; }
PG_HD int thin_run_start(int dx, int dy, int dw, int dh, int cs, float m, float inv, int Y) {
    const int qy = 2 * (Y - dy) + 1 - dh;
    const float c = static_cast<float>(qy * (cs >> 4)) * 16.0f + static_cast<float>(qy * (cs & 15)) +
                    static_cast<float>(dh) * 65536.0f;  // qy·cs + dh·2^16 (|qy| < 2^9, |cs| ≤ 2^16: the parts are exact)
    const float p_min = fminf((c - m) * inv, c * inv);
    return dx + static_cast<int>(floorf((p_min + static_cast<float>(dw - 1)) * 0.5f)) - 1;  // X = dx + (px + dw − 1) / 2
}

// Raster spec S3: nearest texel for destination column/row `i` of `n`, over `len` texels from `start`.
// floor(a / b) for 0 <= a < 2^22, 1 <= b < 2^22.  On the device: one reciprocal estimate and a ±1 fix-up
// instead of the ~30-instruction generic 32-bit division (a and b are exact in float, the estimate is off
// by at most one; checked against `/` in tests/cpp/test_primitives.cpp for the host twin of this code).
PG_HD int udiv_small(int a, int b) {
#if defined(__HIP_DEVICE_COMPILE__)
    int q = static_cast<int>(static_cast<float>(a) * __builtin_amdgcn_rcpf(static_cast<float>(b)));
#else
    int q = static_cast<int>(static_cast<float>(a) * (1.0f / static_cast<float>(b)));
#endif
    int r = a - q * b;
    if (r < 0) q--;
    if (r >= b) q++;
    return q;
}

PG_HD int sample_index(int start, int len, int i, int n) { return start + udiv_small((2 * i + 1) * len, 2 * n); }

// floor(x / 255) for 0 <= x < 65536 (exhaustively checked in tests/cpp/test_primitives.cpp).
PG_HD uint32_t div255(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return static_cast<uint32_t>(__umul24(x, 0x8081u)) >> 23;  // (the intrinsic's result is signed in this HIP: x ≥ 65280 needs the cast)
#else
    return (x * 0x8081u) >> 23;
#endif
}

// floor(x / 255) on the two 16-bit halves of a word at once, each 0 ≤ x ≤ 65 025 (a product of two bytes):
// (x + 1 + (x >> 8)) >> 8, which never leaves 16 bits (at most 65 280), so the halves do not disturb each other.
// (Exhaustively checked in tests/cpp/test_primitives.cpp and on the device in tests/hip/selftest.hip.)
PG_HD uint32_t div255_pair(uint32_t x) {
    const uint32_t t = x + 0x00010001u + ((x >> 8) & 0x00ff00ffu);
    return (t >> 8) & 0x00ff00ffu;
}
PG_HD uint32_t mul_pair(uint32_t halves, uint32_t byte) {  // both bytes of 0x00XX00YY times a byte: fits, no carries
#if defined(__HIP_DEVICE_COMPILE__)
    return static_cast<uint32_t>(__umul24(halves, byte));
#else
    return halves * byte;
#endif
}

// Raster spec S4: straight-alpha blend with truncating /255 on one packed pixel (R | G<<8 | B<<16).
// a is the source alpha after modulation (0..255); returns the new destination.  a = 0 leaves dst, a = 255
// yields src — both fall out of the formula, no special cases needed.  Red and blue go through the arithmetic
// together, as the two halves of one word (render kernels are bound by vector instructions: 25 instead of 32 a pixel).
PG_HD uint32_t blend_px(uint32_t dst, uint32_t src, int a) {
    const uint32_t ua = static_cast<uint32_t>(a), ia = 255u - ua;
#ifdef PG_BLEND_CHANNELWISE  // the three channels one by one: same result; chaser's render kernel, whose scalar unit is as
                             // busy as its vector unit, measured 5 % slower with the paired form (every other game 0–2.5 % faster)
    const uint32_t r = div255((src & 0xffu) * ua) + div255((dst & 0xffu) * ia);
    const uint32_t g = div255(((src >> 8) & 0xffu) * ua) + div255(((dst >> 8) & 0xffu) * ia);
    const uint32_t b = div255(((src >> 16) & 0xffu) * ua) + div255(((dst >> 16) & 0xffu) * ia);
    return r | (g << 8) | (b << 16);
#else
    const uint32_t rb = div255_pair(mul_pair(src & 0x00ff00ffu, ua)) + div255_pair(mul_pair(dst & 0x00ff00ffu, ia));
    const uint32_t g = div255(((src >> 8) & 0xffu) * ua) + div255(((dst >> 8) & 0xffu) * ia);
    return rb | (g << 8);
#endif
}

// A texel as a stamp holds it (pg_stamps.h): the first half of raster spec S4 applied — a = A·mod/255 (mod != 255), s =
// a < 255 ? C·a/255 : C per channel — as s | a << 24; 0 when a = 0 (nothing lands).
PG_HD uint32_t stamp_texel(uint32_t t, int mod) {
    uint32_t a = t >> 24;
    if (mod != 255) a = div255(a * static_cast<uint32_t>(mod));
    if (a == 0u) return 0u;
    if (a == 255u) return (t & 0x00ffffffu) | 0xff000000u;
    const uint32_t r = div255((t & 0xffu) * a), g = div255(((t >> 8) & 0xffu) * a), b = div255(((t >> 16) & 0xffu) * a);
    return r | g << 8 | b << 16 | a << 24;
}
// The second half of raster spec S4 for a texel whose first half is tabulated (pg_stamps.h: src = s | a << 24 with
// s = C·a/255 per channel, 0 < a < 255): D' = s + (255 − a)·D/255.  s ≤ a, so no channel exceeds 255.
PG_HD uint32_t blend_premul(uint32_t dst, uint32_t src, int a) {
    const uint32_t ia = 255u - static_cast<uint32_t>(a);
    const uint32_t rb = (src & 0x00ff00ffu) + div255_pair(mul_pair(dst & 0x00ff00ffu, ia));
    const uint32_t g = ((src >> 8) & 0xffu) + div255(((dst >> 8) & 0xffu) * ia);
    return rb | (g << 8);
}

}  // namespace pg
