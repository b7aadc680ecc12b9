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
#include "pg_render.h"  // RotBox

namespace pg {

// (Blit::flip_mod bit kStamped, pg_geom.h: the texture is a stamp — identity sampling, premultiplied texels, mod applied)
constexpr int kStampsPerTex = 4;       // sizes / modulations one texture may be stamped at

// The table the pre-pass looks a texture up in: [texture][kStampsPerTex] of {first texel of the stamp in the atlas,
// dw | dh << 16, mod, core}; dw = 0: unused slot.  core = i0 | j0 << 8 | i1 << 16 | j1 << 24: the stamp's texels outside
// columns [i0, i1) and rows [j0, j1) are all transparent (i1 = 0: every texel is) — a 4 × 4 bullet of bossfight is 2 × 2
// opaque texels in a transparent rim, and a draw need not look at pixels that can only meet the rim (stamp_substitute).
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
        int i0 = sp.dw, j0 = sp.dh, i1 = 0, j1 = 0;
        for (int j = 0; j < sp.dh; j++)
            for (int i = 0; i < sp.dw; i++)
                if (img[size_t(j) * sp.dw + i] != 0u) {
                    i0 = i < i0 ? i : i0, j0 = j < j0 ? j : j0;
                    i1 = i + 1 > i1 ? i + 1 : i1, j1 = j + 1 > j1 ? j + 1 : j1;
                }
        if (i1 == 0) i0 = j0 = j1 = 0;
        // Behind the image (at the next even word), the stamp as a LIST of the texels that show, row by
        // row, for a big draw of it (pg_render.h wave_blit): core rows + 1 words — where each core row's entries start,
        // and where the last one's end —, padded to an even count, then {i − i0 | (j − j0) << 8, texel} per entry.  A ring
        // like bossfight's shield is 279 texels of 1 015.
        if ((img.size() & 1u) != 0u) img.push_back(0u);
        {
            const int rows = j1 - j0;
            std::vector<uint32_t> row_at(size_t(rows) + 1, 0u), entries;
            for (int j = j0; j < j1; j++) {
                row_at[j - j0] = static_cast<uint32_t>(entries.size() / 2);
                for (int i = i0; i < i1; i++)
                    if (img[size_t(j) * sp.dw + i] != 0u) {
                        entries.push_back(static_cast<uint32_t>(i - i0) | static_cast<uint32_t>(j - j0) << 8);
                        entries.push_back(img[size_t(j) * sp.dw + i]);
                    }
            }
            row_at[rows] = static_cast<uint32_t>(entries.size() / 2);
            if ((row_at.size() & 1u) != 0u) row_at.push_back(0u);
            img.insert(img.end(), row_at.begin(), row_at.end());
            img.insert(img.end(), entries.begin(), entries.end());
        }
        while (atlas.texel_bytes() % 8) atlas.append_words({0u});  // (the entries are read as 8-byte pairs)
        const uint32_t at = atlas.append_words(img);  // (desc_host / texels_host pointers are not kept across this)
        uint32_t* e = &table[(size_t(sp.tex) * kStampsPerTex + slot) * 4];
        e[0] = at;
        e[1] = static_cast<uint32_t>(sp.dw) | static_cast<uint32_t>(sp.dh) << 16;
        e[2] = static_cast<uint32_t>(sp.mod);
        e[3] = static_cast<uint32_t>(i0) | static_cast<uint32_t>(j0) << 8 | static_cast<uint32_t>(i1) << 16 | static_cast<uint32_t>(j1) << 24;
    }
    while (atlas.texel_bytes() % 16) atlas.append_words({0u});
    return atlas.append_words(table);
}

#if defined(__HIPCC__)
// `b` is a resolved draw of texture `tex` (tw × th texels): if it takes the whole texture, flipped sampling aside, and a
// stamp of its destination size and modulation exists, the stamp becomes its texture.  `table`: the texture's
// kStampsPerTex entries (LDS or device memory).  Returns what of the draw can show at all:
//   * an un-rotated, unflipped draw is cut down to the stamp's core (the rectangle outside which every texel is
//     transparent): same pixels, fewer of them looked at;
//   * a rotated draw keeps its rectangle (the rotation is about its centre) and, when the core is centred in the stamp,
//     gets the core's size back in `core_w` / `core_h` for rot_box_core below (else dw, dh);
//   * false: the stamp has no texel that shows — the draw can be dropped.
PG_D bool stamp_substitute(const uint4* table, int tw, int th, Blit& b, int& core_w, int& core_h) {
    core_w = b.dw;
    core_h = b.dh;
    if (b.sx != 0 || b.sy != 0 || b.sw != tw || b.sh != th) return true;
    const uint32_t size = static_cast<uint32_t>(b.dw) | static_cast<uint32_t>(b.dh) << 16;
    const uint32_t mod = static_cast<uint32_t>(b.flip_mod & 0xff);
    bool shows = true;
#pragma unroll
    for (int k = 0; k < kStampsPerTex; k++) {
        const uint4 e = table[k];
        if (e.y == size && e.z == mod) {
            const int i0 = static_cast<int>(e.w & 0xffu), j0 = static_cast<int>((e.w >> 8) & 0xffu);
            const int i1 = static_cast<int>((e.w >> 16) & 0xffu), j1 = static_cast<int>(e.w >> 24);
            b.tex_off = static_cast<int32_t>(e.x);
            b.tex_w = b.dw;
            b.sw = b.dw;
            b.sh = b.dh;
            b.flip_mod = (b.flip_mod & ~0xff) | 255 | kStamped;
            shows = i1 > 0;
            if (!shows) {
                b.dw = b.dh = 0;  // (an empty rectangle: whoever replays the list finds that it reaches no pixel)
            } else if (b.flip_mod & kRotated) {
                if (i0 + i1 == b.dw && j0 + j1 == b.dh) {
                    core_w = i1 - i0;
                    core_h = j1 - j0;
                }
            } else if (!(b.flip_mod & (kFlipH | kFlipV))) {
                // (… and where the stamp's list is, in the source rectangle's two free halves: pg_geom.h stamp_list_at)
                const uint32_t list = (e.x + static_cast<uint32_t>(b.dw * b.dh) + 1u) & ~1u;
                b.sx = static_cast<int32_t>(list & 0xffffu);
                b.sy = static_cast<int32_t>(list >> 16);
                b.tex_off += j0 * b.tex_w + i0;  // (the stamp's own width stays the pitch)
                b.dx += i0;
                b.dy += j0;
                b.dw = b.sw = core_w = i1 - i0;
                b.dh = b.sh = core_h = j1 - j0;
            }
        }
    }
    return shows;
}
PG_D bool stamp_substitute(const uint4* table, int tw, int th, Blit& b) {
    int cw, ch;
    return stamp_substitute(table, tw, th, b, cw, ch);
}
// The bounding box, on the target, of the pixels of rotated draw `b` that can map back into the central core_w × core_h
// texels of its dw × dh destination rectangle (raster spec S6; pg_render.h rot_box is the case core = whole).  A pixel
// with doubled offsets (px, py) from the rectangle's centre lands in destination column i = (px·cs + py·sn + dw·2^16) >> 17;
// i in [i0, i1) with i0 + i1 = dw is a = px·cs + py·sn in [−core_w·2^16, core_w·2^16) — the condition pg_geom.h
// rot_extent's bound is derived from, with core_w in dw's place; likewise rows.  px = 2(X − dx) + 1 − dw as ever, so the
// offsets to the rectangle's corner run from rot_first(dw, ex) to rot_last(dw, ex).
PG_D RotBox rot_box_core(const Blit& b, int core_w, int core_h) {
    const int acs = b.rot_cs < 0 ? -b.rot_cs : b.rot_cs, asn = b.rot_sn < 0 ? -b.rot_sn : b.rot_sn;
    const int ex = rot_extent(core_w, core_h, acs, asn), ey = rot_extent(core_h, core_w, acs, asn);
    int x_lo = b.dx + rot_first(b.dw, ex), x_hi = b.dx + rot_last(b.dw, ex);
    int y_lo = b.dy + rot_first(b.dh, ey), y_hi = b.dy + rot_last(b.dh, ey);
    x_lo = x_lo < 0 ? 0 : x_lo;
    y_lo = y_lo < 0 ? 0 : y_lo;
    x_hi = x_hi > kObsW - 1 ? kObsW - 1 : x_hi;
    y_hi = y_hi > kObsH - 1 ? kObsH - 1 : y_hi;
    return RotBox{x_lo, y_lo, x_hi - x_lo + 1, y_hi - y_lo + 1};
}
#endif

}  // namespace pg
