// Minimal PNG → RGBA8 decoder for the engine's sprite atlas (host side, zlib for inflate).
//
// Replaces the IMG_Load + SDL_CreateTextureFromSurface pair of the reference
// (games/*/common_assets.cpp:3-16): every texture becomes straight-alpha RGBA8; RGB gets A=255,
// palette images are expanded through PLTE/tRNS, grey is replicated, 16-bit samples keep their
// high byte.  Non-interlaced PNGs only (all reference assets are).
#pragma once

#include <stdint.h>

#include <string>
#include <vector>

namespace pg {

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> rgba;
};

// Returns false and fills `err` on failure.
bool decode_png_file(const std::string& path, Image& out, std::string& err);
bool decode_png_memory(const uint8_t* data, size_t size, Image& out, std::string& err);

}  // namespace pg
