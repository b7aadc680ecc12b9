#include "png_decode.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace pg {
namespace {

uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

int paeth(int a, int b, int c) {
    int p = a + b - c;
    int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}

}  // namespace

bool decode_png_memory(const uint8_t* data, size_t size, Image& out, std::string& err) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (size < 8 || std::memcmp(data, sig, 8) != 0) {
        err = "not a PNG";
        return false;
    }
    int w = 0, h = 0, depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat;
    uint8_t pal[256][4];
    int npal = 0;
    for (int i = 0; i < 256; i++) pal[i][0] = pal[i][1] = pal[i][2] = 0, pal[i][3] = 255;
    bool has_trns_key = false;
    uint16_t trns_key[3] = {0, 0, 0};

    size_t pos = 8;
    bool done = false;
    while (!done && pos + 12 <= size) {
        uint32_t len = be32(data + pos);
        const uint8_t* type = data + pos + 4;
        const uint8_t* body = data + pos + 8;
        if (pos + 12 + len > size) {
            err = "truncated chunk";
            return false;
        }
        if (!std::memcmp(type, "IHDR", 4)) {
            w = static_cast<int>(be32(body));
            h = static_cast<int>(be32(body + 4));
            depth = body[8];
            ctype = body[9];
            interlace = body[12];
        } else if (!std::memcmp(type, "PLTE", 4)) {
            npal = static_cast<int>(len / 3);
            for (int i = 0; i < npal && i < 256; i++) {
                pal[i][0] = body[3 * i];
                pal[i][1] = body[3 * i + 1];
                pal[i][2] = body[3 * i + 2];
            }
        } else if (!std::memcmp(type, "tRNS", 4)) {
            if (ctype == 3) {
                for (uint32_t i = 0; i < len && i < 256; i++) pal[i][3] = body[i];
            } else if (ctype == 0 && len >= 2) {
                has_trns_key = true;
                trns_key[0] = static_cast<uint16_t>((body[0] << 8) | body[1]);
            } else if (ctype == 2 && len >= 6) {
                has_trns_key = true;
                for (int c = 0; c < 3; c++) trns_key[c] = static_cast<uint16_t>((body[2 * c] << 8) | body[2 * c + 1]);
            }
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            done = true;
        }
        pos += 12 + len;
    }
    if (w <= 0 || h <= 0 || ctype < 0) {
        err = "missing IHDR";
        return false;
    }
    if (interlace != 0) {
        err = "interlaced PNG not supported";
        return false;
    }
    int channels;
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: err = "bad colour type"; return false;
    }
    if (!(depth == 8 || depth == 16 || (depth < 8 && (ctype == 0 || ctype == 3)))) {
        err = "unsupported bit depth";
        return false;
    }
    const int bits_pp = channels * depth;
    const size_t stride = (size_t(w) * bits_pp + 7) / 8;
    const int bpp = bits_pp >= 8 ? bits_pp / 8 : 1;  // filter unit
    std::vector<uint8_t> raw((stride + 1) * size_t(h));
    uLongf raw_len = static_cast<uLongf>(raw.size());
    int zr = uncompress(raw.data(), &raw_len, idat.data(), static_cast<uLong>(idat.size()));
    if (zr != Z_OK || raw_len != raw.size()) {
        err = "inflate failed";
        return false;
    }
    // Unfilter in place.
    std::vector<uint8_t> prev(stride, 0);
    for (int y = 0; y < h; y++) {
        uint8_t* line = raw.data() + (stride + 1) * size_t(y);
        int ft = line[0];
        uint8_t* cur = line + 1;
        for (size_t i = 0; i < stride; i++) {
            int a = i >= size_t(bpp) ? cur[i - bpp] : 0;
            int b = prev[i];
            int c = i >= size_t(bpp) ? prev[i - bpp] : 0;
            int x = cur[i];
            switch (ft) {
                case 0: break;
                case 1: x += a; break;
                case 2: x += b; break;
                case 3: x += (a + b) >> 1; break;
                case 4: x += paeth(a, b, c); break;
                default: err = "bad filter"; return false;
            }
            cur[i] = static_cast<uint8_t>(x);
        }
        std::memcpy(prev.data(), cur, stride);
    }
    out.w = w;
    out.h = h;
    out.rgba.resize(size_t(w) * h * 4);
    for (int y = 0; y < h; y++) {
        const uint8_t* cur = raw.data() + (stride + 1) * size_t(y) + 1;
        uint8_t* dst = &out.rgba[size_t(y) * w * 4];
        for (int x = 0; x < w; x++) {
            uint16_t s[4] = {0, 0, 0, 0};
            if (depth == 8) {
                for (int c = 0; c < channels; c++) s[c] = cur[x * channels + c];
            } else if (depth == 16) {
                for (int c = 0; c < channels; c++)
                    s[c] = static_cast<uint16_t>((cur[(x * channels + c) * 2] << 8) | cur[(x * channels + c) * 2 + 1]);
            } else {
                int bit = x * depth;
                s[0] = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
            }
            auto to8 = [&](uint16_t v) -> uint8_t {
                if (depth == 16) return static_cast<uint8_t>(v >> 8);
                if (depth == 8) return static_cast<uint8_t>(v);
                return static_cast<uint8_t>(v * 255 / ((1 << depth) - 1));
            };
            uint8_t r, g, b, a = 255;
            switch (ctype) {
                case 0:
                    r = g = b = to8(s[0]);
                    if (has_trns_key && s[0] == trns_key[0]) a = 0;
                    break;
                case 2:
                    r = to8(s[0]);
                    g = to8(s[1]);
                    b = to8(s[2]);
                    if (has_trns_key && s[0] == trns_key[0] && s[1] == trns_key[1] && s[2] == trns_key[2]) a = 0;
                    break;
                case 3: {
                    int k = s[0] & 0xff;
                    r = pal[k][0];
                    g = pal[k][1];
                    b = pal[k][2];
                    a = pal[k][3];
                    break;
                }
                case 4:
                    r = g = b = to8(s[0]);
                    a = to8(s[1]);
                    break;
                default:
                    r = to8(s[0]);
                    g = to8(s[1]);
                    b = to8(s[2]);
                    a = to8(s[3]);
                    break;
            }
            dst[4 * x + 0] = r;
            dst[4 * x + 1] = g;
            dst[4 * x + 2] = b;
            dst[4 * x + 3] = a;
        }
    }
    return true;
}

bool decode_png_file(const std::string& path, Image& out, std::string& err) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) {
        err = "cannot open " + path;
        return false;
    }
    std::vector<uint8_t> buf;
    uint8_t tmp[65536];
    size_t n;
    while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    std::fclose(f);
    bool ok = decode_png_memory(buf.data(), buf.size(), out, err);
    if (!ok) err = path + ": " + err;
    return ok;
}

}  // namespace pg
