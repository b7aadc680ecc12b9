// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the Procgen2 hot path (step / render / reset), used as the
// checker for the HIP engine in procgen2_amd/.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load this library.  The product path never
// links, imports or calls anything in oracle/.
//
// Parity status:
//   * game logic (RNG stream, level generation, physics, reward, terminated) is PINNED
//     against the reward/terminated CRC traces in SURVEY.md Appendix C, which were
//     produced from the unmodified reference sources (tests/golden/appendix_c.json);
//   * pixels are UNPINNED at the SDL boundary: the reference rasterises through an SDL3
//     pre-release software renderer that is neither in /root/reference nor in this image.
//     The raster rules used here are the written spec in DESIGN.md §raster-spec.
//
// This restatement deliberately leans on the *real* libstdc++ containers and <random>
// (std::mt19937, std::uniform_*_distribution, std::unordered_set<int>, std::sort) wherever
// the reference's results depend on their internals (SURVEY.md §0 facts 4,5; rows T1–T5),
// so that those behaviours are exact by construction on the CPU side and the HIP side's
// hand-written emulations are checked against the genuine article.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <random>
#include <string>
#include <unordered_set>
#include <vector>

namespace pgo {

// helpers.h:8-9
constexpr float kUnitPx = 16.0f;
constexpr float kPxUnit = 1.0f / kUnitPx;

struct V2 {
    float x = 0.0f, y = 0.0f;
};

struct Box {  // helpers.h:15-17 Rectangle
    float x = 0.0f, y = 0.0f, w = 0.0f, h = 0.0f;
};

// helpers.cpp:40-46 — strict AABB overlap.
inline bool boxes_touch(const Box& a, const Box& b) {
    return a.x < b.x + b.w && a.x + a.w > b.x && a.y < b.y + b.h && a.y + a.h > b.y;
}

// helpers.cpp:48-108 — raylib-style intersection rectangle.
inline Box overlap_box(const Box& a, const Box& b) {
    Box r;
    if (!boxes_touch(a, b)) return r;
    const float ddx = std::fabs(a.x - b.x);
    const float ddy = std::fabs(a.y - b.y);
    const bool a_left = a.x <= b.x;
    const bool a_top = a.y <= b.y;
    r.x = a_left ? b.x : a.x;
    r.y = a_top ? b.y : a.y;
    r.w = (a_left ? a.w : b.w) - ddx;
    r.h = (a_top ? a.h : b.h) - ddy;
    const float wcap = a.w > b.w ? b.w : a.w;
    const float hcap = a.h > b.h ? b.h : a.h;
    if (r.w >= wcap) r.w = wcap;
    if (r.h >= hcap) r.h = hcap;
    return r;
}

// T1/T2: one mt19937 per env for its whole life; distributions are stateless so a fresh
// temporary per draw is equivalent to the reference's named distribution objects.
struct Rng {
    std::mt19937 eng;
    void seed(uint32_t s) { eng.seed(s); }
    int irange(int lo, int hi) {
        std::uniform_int_distribution<int> d(lo, hi);
        return d(eng);
    }
    float unit() {
        std::uniform_real_distribution<float> d(0.0f, 1.0f);
        return d(eng);
    }
    float frange(float lo, float hi) {
        std::uniform_real_distribution<float> d(lo, hi);
        return d(eng);
    }
};

// T5 (ecs.h:24-61, ecs.cpp:3-50): FIFO id allocator 0..999.
struct IdPool {
    static constexpr int kMax = 1000;
    std::deque<int> free_ids;
    IdPool() { refill(); }
    void refill() {
        free_ids.clear();
        for (int i = 0; i < kMax; i++) free_ids.push_back(i);
    }
    int take() {
        int e = free_ids.front();
        free_ids.pop_front();
        return e;
    }
    void give_back(int e) { free_ids.push_back(e); }
};

// T3: a System's entity set.  Real std::unordered_set<int> so iteration order (and the
// bucket count that survives clear()) is libstdc++'s own.
using IdSet = std::unordered_set<int>;

// CRC-32 (IEEE 802.3, reflected, poly 0xEDB88320) used by the Appendix-C traces.
struct Crc32 {
    uint32_t table[256];
    uint32_t state = 0xFFFFFFFFu;
    Crc32() {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
            table[i] = c;
        }
    }
    void feed(const void* p, size_t n) {
        const uint8_t* b = static_cast<const uint8_t*>(p);
        for (size_t i = 0; i < n; i++) state = table[(state ^ b[i]) & 0xFF] ^ (state >> 8);
    }
    uint32_t value() const { return state ^ 0xFFFFFFFFu; }
};

}  // namespace pgo
