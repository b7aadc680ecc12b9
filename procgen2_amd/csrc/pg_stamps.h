// Pre-scaled sprite stamps (round 6).
//
// A game whose camera scale never changes draws a given texture at a given sprite scale into the SAME destination size
// every time — bossfight's bullets are always 4 × 4 pixels of a 48 × 46 texture, its shield always 35 × 29 of 143 × 119
// at alpha 0.7.  For a draw that takes the whole texture as its source (not cropped by an edge of the screen, raster
// spec S2) the texel that lands on destination pixel (i, j) is then a function of (texture, dw, dh, i, j) and of nothing
// else: tex[sample_index(0, th, j, dh)][sample_index(0, tw, i, dw)] (raster spec S3, integers only).  A *stamp* is that
// dw × dh image, made once on the host when the atlas is loaded, with the blend's first half already applied to every
// texel (raster spec S4: a = A·mod/255, s = a < 255 ? C·a/255 : C — stored as s | a << 24): the render pre-pass
// substitutes it for the texture in every draw it fits (stamp_substitute: integers compared, nothing predicted), and a
// stamped draw (flag kStamped) needs no division per pixel to find its texel (column i, row j of the stamp), fetches from
// one or two cache lines instead of one per source row, and blends with one multiplication per channel instead of two.
// Same pixels by construction: D' = s + (255 − a)·D/255 IS the spec's formula with its first half tabulated.
// A draw no stamp fits (cropped by the screen's edge, another size) goes the general way, as before.
#pragma once

#include <vector>

#include "pg_engine.h"
#include "pg_geom.h"

namespace pg {

// (Blit::flip_mod bit kStamped, pg_geom.h: the texture is a stamp — identity sampling, premultiplied texels, mod applied)
constexpr int kStampsPerTex = 4;       // sizes / modulations one texture may be stamped at

// The table the pre-pass looks a texture up in: [texture][kStampsPerTex] of {first texel of the stamp in the atlas,
// dw | dh << 16, mod, 0}; dw = 0: unused slot.
struct StampSpec {
    int tex, dw, dh, mod;
};

// (host functions; hipcc parses them in both of its passes)
// The destination size render_texture (pg_geom.h resolve_axis: renderer.cpp:5-76 + S1) gives a sprite of `tsize` texels at
// `scale` that lies wholly on the screen, or 0 when no placement draws the whole texture.
inline int stamp_len_plain(float cam_len, float cam_scale, int tsize, float scale) {
    Span sp;
    // the sprite centred on a camera at the origin: uncropped unless it is larger than the screen
    const float pos = -0.5f * tsize * scale;
    if (!resolve_axis(0.0f, cam_len, cam_scale, tsize, pos, scale, false, false, sp)) return 0;
    return (sp.s0 == 0 && sp.sn == tsize) ? sp.dn : 0;
}
// … and a rotated one (pg_render.h resolve_rotated_at: renderer.cpp:84-101 + S1; never cropped).
inline int stamp_len_rotated(float cam_scale, int tsize, float scale) {
    const float d = tsize * scale * cam_scale;
    return (d >= 1.0f && d < 32768.0f) ? static_cast<int>(d) : 0;
}
inline int stamp_mod(float alpha) { return alpha != 1.0f ? (static_cast<int>(255 * alpha) & 0xff) : 255; }

// Appends the stamps and their table to the atlas; returns the table's word offset (16-byte aligned), 0 = none.
// Specs with dw or dh 0, duplicates, and specs beyond kStampsPerTex per texture are skipped (such draws go the general way).
// (AtlasT: pg_engine.h Atlas, or the self-test's stand-in with the same four members.)
template <class AtlasT>
inline uint32_t append_stamps(AtlasT& atlas, int n_tex, const std::vector<StampSpec>& specs) {
    std::vector<uint32_t> table(size_t(n_tex) * kStampsPerTex * 4, 0u);
    for (const StampSpec& sp : specs) {
        if (sp.tex < 0 || sp.tex >= n_tex || sp.dw < 1 || sp.dh < 1 || sp.dw > 255 || sp.dh > 255) continue;
        int slot = -1;
        bool dup = false;
        for (int k = 0; k < kStampsPerTex; k++) {
            const uint32_t* e = &table[(size_t(sp.tex) * kStampsPerTex + k) * 4];
            if (e[1] == 0u) {
                if (slot < 0) slot = k;
            } else if (e[1] == (static_cast<uint32_t>(sp.dw) | static_cast<uint32_t>(sp.dh) << 16) && e[2] == static_cast<uint32_t>(sp.mod)) {
                dup = true;
            }
        }
        if (dup || slot < 0) continue;
        const int4 d = atlas.desc_host(sp.tex);
        const uint32_t* tex = atlas.texels_host(sp.tex);
        std::vector<uint32_t> img(size_t(sp.dw) * sp.dh);
        for (int j = 0; j < sp.dh; j++)
            for (int i = 0; i < sp.dw; i++) {
                const uint32_t t = tex[sample_index(0, d.z, j, sp.dh) * d.y + sample_index(0, d.y, i, sp.dw)];
                const uint32_t w = stamp_texel(t, sp.mod);  // S4's first half
                img[size_t(j) * sp.dw + i] = w;
            }
        const uint32_t at = atlas.append_words(img);  // (desc_host / texels_host pointers are not kept across this)
        uint32_t* e = &table[(size_t(sp.tex) * kStampsPerTex + slot) * 4];
        e[0] = at;
        e[1] = static_cast<uint32_t>(sp.dw) | static_cast<uint32_t>(sp.dh) << 16;
        e[2] = static_cast<uint32_t>(sp.mod);
    }
    while (atlas.texel_bytes() % 16) atlas.append_words({0u});
    return atlas.append_words(table);
}

#if defined(__HIPCC__)
// `b` is a resolved draw of texture `tex` (tw × th texels): if it takes the whole texture, unflipped sampling aside, and a
// stamp of its destination size and modulation exists, the stamp becomes its texture.  `table`: the texture's
// kStampsPerTex entries (LDS or device memory).
PG_D void stamp_substitute(const uint4* table, int tw, int th, Blit& b) {
    if (b.sx != 0 || b.sy != 0 || b.sw != tw || b.sh != th) return;
    const uint32_t size = static_cast<uint32_t>(b.dw) | static_cast<uint32_t>(b.dh) << 16;
    const uint32_t mod = static_cast<uint32_t>(b.flip_mod & 0xff);
#pragma unroll
    for (int k = 0; k < kStampsPerTex; k++) {
        const uint4 e = table[k];
        if (e.y == size && e.z == mod) {
            b.tex_off = static_cast<int32_t>(e.x);
            b.tex_w = b.dw;
            b.sw = b.dw;
            b.sh = b.dh;
            b.flip_mod = (b.flip_mod & ~0xff) | 255 | kStamped;
        }
    }
}
#endif

}  // namespace pg
