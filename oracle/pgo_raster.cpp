// ORACLE — TEST INFRASTRUCTURE ONLY (see pgo_common.h header).
#include "pgo_raster.h"

#include <cstdio>
#include <cstdlib>
#include <cmath>

namespace pgo {

TextureBank& TextureBank::global() {
    static TextureBank bank;
    return bank;
}

void TextureBank::put(const std::string& name, int w, int h, const uint8_t* rgba) {
    auto t = std::make_unique<Texture>();
    t->w = w;
    t->h = h;
    t->rgba.assign(rgba, rgba + size_t(w) * h * 4);
    items_[name] = std::move(t);
}

const Texture* TextureBank::find(const std::string& name) const {
    auto it = items_.find(name);
    return it == items_.end() ? nullptr : it->second.get();
}

void Surface::clear_black() {
    for (size_t i = 0; i < px.size(); i += 4) {
        px[i + 0] = 0;
        px[i + 1] = 0;
        px[i + 2] = 0;
        px[i + 3] = 255;
    }
}

// Raster spec.
//  S1  dst rect → integers by truncation toward zero of x, y, w, h; nothing is drawn unless 1 <= w,h < 32768
//      and |x|,|y| < 32768 (NaN fails every comparison, so a non-finite rect draws nothing).
//  S2  src rect → integers by truncation, then intersected with the texture; nothing is drawn if empty.
//      The dst rect is NOT adjusted for the part of src that fell outside the texture.
//  S3  nearest sampling at pixel centres in integer arithmetic:
//        u = sx + ((2*i+1)*sw) / (2*dw),  v = sy + ((2*j+1)*sh) / (2*dh)     (floor division)
//      with i mirrored (dw-1-i) for a horizontal flip and j mirrored for a vertical flip.
//  S4  straight-alpha "blend" with truncating /255:
//        a = A;  if mod != 255: a = a*mod/255;   a == 0 → pixel untouched
//        s = (a < 255) ? C*a/255 : C;   D = s + (255-a)*D/255     for C in R,G,B (and A likewise)
//  S5  target clip: pixels outside the surface are dropped.
//  S6  rotation (angle != 0, whole texture as source, no flip): see spec_blit_rotated below.
void spec_blit(Surface& target, const Texture& tex, float fsx, float fsy, float fsw, float fsh, float fdx, float fdy,
               float fdw, float fdh, double angle_deg, int flip, int alpha_mod) {
    if (angle_deg != 0.0) {
        spec_blit_rotated(target, tex, fdx, fdy, fdw, fdh, angle_deg, alpha_mod);
        return;
    }
    // S1: a destination that is not finite, narrower than one pixel or absurdly large draws nothing
    // (the reference's crop arithmetic yields 0*inf = NaN for sprites that end exactly on the viewport edge).
    if (!(fdw >= 1.0f && fdh >= 1.0f && fdw < 32768.0f && fdh < 32768.0f)) return;
    if (!(fdx > -32768.0f && fdx < 32768.0f && fdy > -32768.0f && fdy < 32768.0f)) return;
    const int dx = static_cast<int>(fdx), dy = static_cast<int>(fdy);
    const int dw = static_cast<int>(fdw), dh = static_cast<int>(fdh);

    int sx0 = static_cast<int>(fsx), sy0 = static_cast<int>(fsy);
    int sx1 = sx0 + static_cast<int>(fsw), sy1 = sy0 + static_cast<int>(fsh);
    if (sx0 < 0) sx0 = 0;
    if (sy0 < 0) sy0 = 0;
    if (sx1 > tex.w) sx1 = tex.w;
    if (sy1 > tex.h) sy1 = tex.h;
    const int sw = sx1 - sx0, sh = sy1 - sy0;
    if (sw <= 0 || sh <= 0) return;

    for (int j = 0; j < dh; j++) {
        const int ty = dy + j;
        if (ty < 0 || ty >= target.h) continue;
        const int jj = (flip & kFlipV) ? dh - 1 - j : j;
        const int v = sy0 + static_cast<int>((int64_t(2 * jj + 1) * sh) / (2 * int64_t(dh)));
        for (int i = 0; i < dw; i++) {
            const int tx = dx + i;
            if (tx < 0 || tx >= target.w) continue;
            const int ii = (flip & kFlipH) ? dw - 1 - i : i;
            const int u = sx0 + static_cast<int>((int64_t(2 * ii + 1) * sw) / (2 * int64_t(dw)));
            const uint8_t* s = &tex.rgba[(size_t(v) * tex.w + u) * 4];
            int a = s[3];
            if (alpha_mod != 255) a = a * alpha_mod / 255;
            if (a == 0) continue;
            uint8_t* d = &target.px[(size_t(ty) * target.w + tx) * 4];
            for (int c = 0; c < 3; c++) {
                const int sc = (a < 255) ? s[c] * a / 255 : s[c];
                d[c] = static_cast<uint8_t>(sc + (255 - a) * d[c] / 255);
            }
            d[3] = static_cast<uint8_t>(a + (255 - a) * d[3] / 255);
        }
    }
}

// S6.  The destination rectangle (truncated as in S1) is rotated clockwise by `angle_deg` about its centre and
// every target pixel whose centre falls inside the rotated rectangle takes the texel that S3 assigns to the
// un-rotated column/row it maps back to.  All arithmetic is integer once sine and cosine are fixed:
//   theta  = (float)(angle_deg * (pi / 180))          (double product, rounded to float)
//   sn, cs = round(sinf(theta) * 65536), round(cosf(theta) * 65536)   (glibc sinf/cosf; 16.16 fixed point)
//   for a pixel (X, Y): px = 2*(X-dx) + 1 - dw, py = 2*(Y-dy) + 1 - dh        (doubled offsets from the centre)
//     lx = px*cs + py*sn + dw*65536,  ly = -px*sn + py*cs + dh*65536          (doubled, 16.16)
//     inside iff 0 <= lx < 2*dw*65536 and 0 <= ly < 2*dh*65536;  i = lx >> 17, j = ly >> 17
//   pixels scanned: the square of half-side ceil(sqrt(dw*dw + dh*dh) / 2) + 1 around the centre, clipped (S5).
void spec_blit_rotated(Surface& target, const Texture& tex, float fdx, float fdy, float fdw, float fdh, double angle_deg,
                       int alpha_mod) {
    if (!(fdw >= 1.0f && fdh >= 1.0f && fdw < 32768.0f && fdh < 32768.0f)) return;
    if (!(fdx > -32768.0f && fdx < 32768.0f && fdy > -32768.0f && fdy < 32768.0f)) return;
    const int dx = static_cast<int>(fdx), dy = static_cast<int>(fdy);
    const int dw = static_cast<int>(fdw), dh = static_cast<int>(fdh);
    const float theta = static_cast<float>(angle_deg * (3.14159265358979323846 / 180.0));
    const int sn = static_cast<int>(std::floor(static_cast<double>(sinf(theta)) * 65536.0 + 0.5));
    const int cs = static_cast<int>(std::floor(static_cast<double>(cosf(theta)) * 65536.0 + 0.5));
    int reach = 1;
    while (reach * reach * 4 < dw * dw + dh * dh) reach++;  // ceil(sqrt(dw²+dh²)/2)
    reach += 1;
    const int cx2 = 2 * dx + dw, cy2 = 2 * dy + dh;  // doubled centre
    const int x_lo = (cx2 - 2 * reach) / 2 - 1, x_hi = (cx2 + 2 * reach) / 2 + 1;
    const int y_lo = (cy2 - 2 * reach) / 2 - 1, y_hi = (cy2 + 2 * reach) / 2 + 1;
    for (int Y = y_lo; Y <= y_hi; Y++) {
        if (Y < 0 || Y >= target.h) continue;
        for (int X = x_lo; X <= x_hi; X++) {
            if (X < 0 || X >= target.w) continue;
            const int px = 2 * (X - dx) + 1 - dw, py = 2 * (Y - dy) + 1 - dh;
            const int64_t lx = int64_t(px) * cs + int64_t(py) * sn + int64_t(dw) * 65536;
            const int64_t ly = -int64_t(px) * sn + int64_t(py) * cs + int64_t(dh) * 65536;
            if (lx < 0 || ly < 0 || lx >= int64_t(2 * dw) * 65536 || ly >= int64_t(2 * dh) * 65536) continue;
            const int i = static_cast<int>(lx >> 17), j = static_cast<int>(ly >> 17);
            const int u = static_cast<int>((int64_t(2 * i + 1) * tex.w) / (2 * int64_t(dw)));
            const int v = static_cast<int>((int64_t(2 * j + 1) * tex.h) / (2 * int64_t(dh)));
            const uint8_t* s = &tex.rgba[(size_t(v) * tex.w + u) * 4];
            int a = s[3];
            if (alpha_mod != 255) a = a * alpha_mod / 255;
            if (a == 0) continue;
            uint8_t* d = &target.px[(size_t(Y) * target.w + X) * 4];
            for (int c = 0; c < 3; c++) {
                const int sc = (a < 255) ? s[c] * a / 255 : s[c];
                d[c] = static_cast<uint8_t>(sc + (255 - a) * d[c] / 255);
            }
            d[3] = static_cast<uint8_t>(a + (255 - a) * d[3] / 255);
        }
    }
}

void Painter::draw(const Texture* tex, V2 pos, float scale, float alpha, bool flip_h, bool flip_v) {
    draw_calls++;
    if (!enabled) return;
    float sx = 0.0f, sy = 0.0f;
    float sw = static_cast<float>(tex->w), sh = static_cast<float>(tex->h);

    float dx = (pos.x - cam_pos.x) * cam_scale + cam_size.x * 0.5f;
    float dy = (pos.y - cam_pos.y) * cam_scale + cam_size.y * 0.5f;
    float dw = tex->w * scale * cam_scale;
    float dh = tex->h * scale * cam_scale;

    // renderer.cpp:14 — note '>' on x and '>=' on y.
    if (dx > cam_size.x || dy >= cam_size.y || dx + dw < 0 || dy + dh < 0) return;

    // renderer.cpp:18-52 — crop to the viewport with a proportional source crop.
    if (dx < 0.0f) {
        float ratio = -dx / dw;
        sx += sw * ratio;
        sw -= sx;
        dw += dx;
        dx = 0.0f;
    }
    if (dx + dw > cam_size.x) {
        float ratio = (dx + dw - cam_size.x) / dw;
        sw = sw * (1.0f - ratio);
        dw = cam_size.x - dx;
    }
    if (dy < 0.0f) {
        float ratio = -dy / dh;
        sy += sh * ratio;
        sh -= sy;
        dh += dy;
        dy = 0.0f;
    }
    if (dy + dh > cam_size.y) {
        float ratio = (dy + dh - cam_size.y) / dh;
        sh = sh * (1.0f - ratio);
        dh = cam_size.y - dy;
    }

    // renderer.cpp:54-57 — SDL_SetTextureAlphaMod takes a Uint8: float → u8 truncation.
    int mod = 255;
    if (alpha != 1.0f) mod = static_cast<uint8_t>(255 * alpha);

    // renderer.cpp:59-70 — integer source snap with padding and destination compensation.
    int padding = std::ceil(1.0f / (scale * cam_scale));
    int rx = static_cast<int>(std::floor(sx));
    int ry = static_cast<int>(std::floor(sy));
    int rw = static_cast<int>(std::ceil(sw)) + padding;
    int rh = static_cast<int>(std::ceil(sh)) + padding;

    float off_x = sx - rx, off_y = sy - ry;
    float ratio_x = rw / sw, ratio_y = rh / sh;
    dw *= ratio_x;
    dh *= ratio_y;
    dx -= off_x * (dw / sw);
    dy -= off_y * (dh / sh);

    if (flip_h) rx = tex->w - rw - rx;  // renderer.cpp:72-74

    int flip = flip_h ? kFlipH : (flip_v ? kFlipV : kFlipNone);  // renderer.cpp:78
    spec_blit(*target, *tex, static_cast<float>(rx), static_cast<float>(ry), static_cast<float>(rw),
              static_cast<float>(rh), dx, dy, dw, dh, 0.0, flip, mod);
}

void Painter::draw_rotated(const Texture* tex, V2 pos, float rotation, float scale, float alpha) {
    draw_calls++;
    if (!enabled) return;
    float dx = (pos.x - cam_pos.x) * cam_scale + cam_size.x * 0.5f;
    float dy = (pos.y - cam_pos.y) * cam_scale + cam_size.y * 0.5f;
    float dw = tex->w * scale * cam_scale;
    float dh = tex->h * scale * cam_scale;
    int mod = 255;
    if (alpha != 1.0f) mod = static_cast<uint8_t>(255 * alpha);
    // renderer.cpp:97 — float * float / double(M_PI)
    double deg = rotation * 180.0f / M_PI;
    spec_blit(*target, *tex, 0.0f, 0.0f, static_cast<float>(tex->w), static_cast<float>(tex->h), dx, dy, dw, dh, deg,
              kFlipNone, mod);
}

void pack_rgb(const Surface& s, uint8_t* out) {
    const size_t n = size_t(s.w) * s.h;
    for (size_t k = 0; k < n; k++) {
        out[3 * k + 0] = s.px[4 * k + 0];
        out[3 * k + 1] = s.px[4 * k + 1];
        out[3 * k + 2] = s.px[4 * k + 2];
    }
}

}  // namespace pgo
