// ORACLE — TEST INFRASTRUCTURE ONLY (see pgo_common.h header).
//
// Camera / cull / crop arithmetic of the reference renderer (R1, R2) and the written raster
// spec that stands in for the absent SDL3 software renderer (R3), plus the RGB pack (R4).
#pragma once

#include <map>
#include <memory>

#include "pgo_common.h"

namespace pgo {

// Decoded RGBA8 (straight alpha) image.  Textures are handed to the oracle already decoded
// (tests decode the PNGs with PIL), so the oracle does not share the product's PNG decoder.
struct Texture {
    int w = 0, h = 0;
    std::vector<uint8_t> rgba;  // h*w*4, row-major
};

// Name → texture registry (asset_manager.h:13-26).  Names are the reference's relative asset
// paths, e.g. "assets/kenney/Items/coinGold.png".
class TextureBank {
   public:
    static TextureBank& global();
    void put(const std::string& name, int w, int h, const uint8_t* rgba);
    // Returns nullptr when missing; logic-only runs (render disabled) never dereference.
    const Texture* find(const std::string& name) const;
    size_t size() const { return items_.size(); }

   private:
    std::map<std::string, std::unique_ptr<Texture>> items_;
};

enum Flip { kFlipNone = 0, kFlipH = 1, kFlipV = 2 };

// 32-bit RGBA target surface (bytes R,G,B,A).
struct Surface {
    int w, h;
    std::vector<uint8_t> px;
    Surface(int w_, int h_) : w(w_), h(h_), px(size_t(w_) * h_ * 4, 0) {}
    void clear_black();  // coinrun.cpp:447-448 — opaque black
};

// DESIGN.md §raster-spec, rule S1–S6: the stand-in for SDL_RenderTextureRotated.
// src: integer-valued float rect in texel space; dst: float rect in target pixels.
void spec_blit(Surface& target, const Texture& tex, float sx, float sy, float sw, float sh, float dx, float dy,
               float dw, float dh, double angle_deg, int flip, int alpha_mod);

void spec_blit_rotated(Surface& target, const Texture& tex, float dx, float dy, float dw, float dh, double angle_deg,
                       int alpha_mod);

// renderer.h:9-31 — the global renderer's camera, one per env here.
struct Painter {
    Surface* target = nullptr;
    V2 cam_pos{0.0f, 0.0f};
    V2 cam_size{64.0f, 64.0f};
    float cam_scale = 1.0f;
    bool enabled = true;  // logic-only traces switch drawing off
    long draw_calls = 0;

    // renderer.cpp:5-82
    void draw(const Texture* tex, V2 pos, float scale = 1.0f, float alpha = 1.0f, bool flip_h = false,
              bool flip_v = false);
    // renderer.cpp:84-101
    void draw_rotated(const Texture* tex, V2 pos, float rotation, float scale = 1.0f, float alpha = 1.0f);
};

// coinrun.cpp:377-388 (identical in every game): RGBA → RGB, row-major HWC.
void pack_rgb(const Surface& s, uint8_t* out);

}  // namespace pgo
