// CPU check of the engine's host/device primitives (procgen2_amd/csrc/pg_*.h compiled for the host)
// against the genuine libstdc++ behaviour the reference depends on (SURVEY.md rows T1–T4).
// Built and run by tests/test_primitives.py; prints "OK <name>" per section, exits non-zero on failure.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <unordered_set>
#include <vector>

#include "pg_geom.h"
#include "pg_order.h"
#include "pg_rng.h"
#include "pg_atan2.h"
#include "pg_sincos.h"

static int fails = 0;
#define CHECK(cond, ...)                 \
    do {                                 \
        if (!(cond)) {                   \
            std::printf("FAIL: " __VA_ARGS__); \
            std::printf("\n");           \
            if (++fails > 10) std::exit(1); \
        }                                \
    } while (0)

static void test_mt() {
    // SURVEY.md T1 known answers
    uint32_t st[pg::kMtWords];
    pg::mt_seed(st, 123u);
    CHECK(pg::mt_next(st) == 2991312382u, "mt seed 123 #0");
    CHECK(pg::mt_next(st) == 3062119789u, "mt seed 123 #1");
    CHECK(pg::mt_next(st) == 1228959102u, "mt seed 123 #2");
    pg::mt_seed(st, static_cast<uint32_t>(-5));
    CHECK(pg::mt_next(st) == 1844333030u, "mt seed -5 #0");
    pg::mt_seed(st, 1u);
    uint32_t v = 0;
    for (int i = 0; i < 624; i++) v = pg::mt_next(st);  // the 624th draw (1-based), last before the first re-twist
    CHECK(v == 2006116153u, "mt seed 1 #624 got %u", v);
    CHECK(pg::mt_next(st) == 1104314680u, "mt seed 1 #625");

    for (uint32_t seed : {0u, 1u, 7u, 123u, 4294967291u, 0xdeadbeefu}) {
        std::mt19937 ref(seed);
        pg::mt_seed(st, seed);
        for (int i = 0; i < 3000; i++) {
            uint32_t a = static_cast<uint32_t>(ref()), b = pg::mt_next(st);
            CHECK(a == b, "mt stream seed %u draw %d: %u vs %u", seed, i, a, b);
        }
    }
    std::printf("OK mt19937\n");
}

static void test_distributions() {
    // SURVEY.md T2 known answers (seed 123, in this order)
    uint32_t st[pg::kMtWords];
    pg::mt_seed(st, 123u);
    CHECK(pg::rng_int(st, 1, 3) == 3, "kat int(1,3)");
    CHECK(pg::rng_int(st, 0, 48) == 34, "kat int(0,48)");
    CHECK(pg::rng_int(st, 0, 19) == 5, "kat int(0,19)");
    auto bits = [](float f) {
        uint32_t u;
        std::memcpy(&u, &f, 4);
        return u;
    };
    CHECK(bits(pg::rng_real(st, 0.0f, 1.0f)) == 0x3edb608bu, "kat real(0,1)");
    CHECK(bits(pg::rng_real(st, -1.0f, 1.0f)) == 0xbf0bda20u, "kat real(-1,1)");
    CHECK(bits(pg::rng_real(st, 0.7f, 1.2f)) == 0x3f85d10fu, "kat real(0.7,1.2)");

    std::mt19937 meta(99);
    for (int round = 0; round < 200; round++) {
        uint32_t seed = static_cast<uint32_t>(meta());
        std::mt19937 ref(seed);
        pg::mt_seed(st, seed);
        for (int i = 0; i < 400; i++) {
            int kind = static_cast<int>(meta() % 4);
            if (kind < 2) {
                int lo = static_cast<int>(meta() % 50) - 10;
                int span = (kind == 0) ? static_cast<int>(meta() % 8) : static_cast<int>(meta() % 3000);
                std::uniform_int_distribution<int> d(lo, lo + span);
                int a = d(ref), b = pg::rng_int(st, lo, lo + span);
                CHECK(a == b, "uniform_int(%d,%d): %d vs %d", lo, lo + span, a, b);
            } else {
                float lo = (kind == 2) ? 0.0f : -1.0f - static_cast<float>(meta() % 7) * 0.3f;
                float hi = (kind == 2) ? 1.0f : 1.2f + static_cast<float>(meta() % 5) * 0.7f;
                std::uniform_real_distribution<float> d(lo, hi);
                float a = d(ref), b = pg::rng_real(st, lo, hi);
                CHECK(bits(a) == bits(b), "uniform_real(%g,%g): %08x vs %08x", lo, hi, bits(a), bits(b));
            }
        }
    }
    std::printf("OK distributions\n");
}

static bool same_order(const std::unordered_set<int>& ref, const pg::HashOrder& h) {
    int16_t p = static_cast<int16_t>(h.head);
    for (int k : ref) {
        if (p != k) return false;
        p = h.next[p];
    }
    return p == pg::kNil && static_cast<int>(ref.size()) == h.count && static_cast<int>(ref.bucket_count()) == h.buckets;
}

static void test_hash_order() {
    std::vector<int16_t> next(2400), before(2400);
    // SURVEY.md T3 known answers
    {
        std::unordered_set<int> ref;
        pg::HashOrder h;
        pg::hash_init(h, next.data(), before.data());
        for (int i = 0; i < 30; i++) {
            ref.insert(i);
            pg::hash_insert(h, i);
            CHECK(same_order(ref, h), "ascending insert %d", i);
        }
        int expect[30] = {29, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28};
        int16_t p = static_cast<int16_t>(h.head);
        for (int i = 0; i < 30; i++) {
            CHECK(p == expect[i], "T3 order at %d", i);
            p = h.next[p];
        }
    }
    std::mt19937 meta(4242);
    for (int round = 0; round < 300; round++) {
        std::unordered_set<int> ref;
        pg::HashOrder h;
        pg::hash_init(h, next.data(), before.data());
        int universe = 5 + static_cast<int>(meta() % (round < 250 ? 120 : 1000));
        int ops = 50 + static_cast<int>(meta() % 400);
        for (int i = 0; i < ops; i++) {
            int r = static_cast<int>(meta() % 100);
            int key = static_cast<int>(meta() % universe);
            if (r < 60) {
                ref.insert(key);
                pg::hash_insert(h, key);
            } else if (r < 95) {
                ref.erase(key);
                pg::hash_erase(h, key);
            } else if (r < 98) {
                ref.clear();
                pg::hash_clear(h);
            } else {  // an "episode": clear then ascending ids, the ECS pattern
                ref.clear();
                pg::hash_clear(h);
                int n = static_cast<int>(meta() % universe);
                for (int k = 0; k < n; k++) {
                    ref.insert(k);
                    pg::hash_insert(h, k);
                }
            }
            CHECK(same_order(ref, h), "random ops round %d op %d", round, i);
            CHECK(pg::hash_contains(h, key) == (ref.count(key) != 0), "contains");
        }
    }
    std::printf("OK hash_order\n");
}

static void test_sort() {
    // SURVEY.md T4 known answers: all-equal keys
    {
        int expect17[17] = {8, 16, 15, 14, 13, 12, 11, 10, 9, 0, 7, 6, 5, 4, 3, 2, 1};
        pg::ZItem a[17];
        for (int i = 0; i < 17; i++) a[i] = {1.0f, i};
        pg::sort_by_key(a, 17);
        for (int i = 0; i < 17; i++) CHECK(a[i].id == expect17[i], "T4 n=17 at %d: %d", i, a[i].id);
    }
    std::mt19937 meta(777);
    for (int round = 0; round < 3000; round++) {
        int n = static_cast<int>(meta() % (round < 2500 ? 80 : 700));
        int distinct = 1 + static_cast<int>(meta() % 4);
        if (round % 7 == 0) distinct = 1000;
        std::vector<std::pair<float, int>> ref(n);
        std::vector<pg::ZItem> mine(n);
        for (int i = 0; i < n; i++) {
            float z = static_cast<float>(static_cast<int>(meta() % distinct)) - 1.0f;
            ref[i] = {z, i};
            mine[i] = {z, i};
        }
        if (round % 11 == 0)  // adversarial-ish: organ pipe to push partition depth
            for (int i = 0; i < n; i++) ref[i].first = mine[i].z = static_cast<float>(i < n / 2 ? i : n - i);
        std::sort(ref.begin(), ref.end(),
                  [](const std::pair<float, int>& l, const std::pair<float, int>& r) { return l.first < r.first; });
        pg::sort_by_key(mine.data(), n);
        for (int i = 0; i < n; i++)
            CHECK(ref[i].second == mine[i].id, "sort round %d n=%d distinct=%d at %d", round, n, distinct, i);
    }
    std::printf("OK sort\n");
}

static void test_blend() {
    for (uint32_t x = 0; x < 65536; x++) CHECK(pg::div255(x) == x / 255, "div255(%u)", x);
    // the two-at-once form: every value a product of two bytes can take, in either half, beside an extreme in the other
    for (uint32_t x = 0; x <= 65025; x++)
        for (uint32_t other : {0u, 1u, 254u, 255u, 65024u, 65025u}) {
            const uint32_t lo = pg::div255_pair(x | other << 16), hi = pg::div255_pair(other | x << 16);
            CHECK((lo & 0xffffu) == x / 255 && (lo >> 16) == other / 255, "div255_pair low half %u beside %u", x, other);
            CHECK((hi >> 16) == x / 255 && (hi & 0xffffu) == other / 255, "div255_pair high half %u beside %u", x, other);
        }
    for (int b = 1; b < 700; b++)
        for (int a = 0; a < (1 << 22); a += (b < 140 ? 1 + a / 4096 : 997)) CHECK(pg::udiv_small(a, b) == a / b, "udiv_small(%d,%d)", a, b);
    for (int n = 1; n <= 70; n++)  // every (column, width) pair a blit can see
        for (int len = 1; len <= 2100; len += (len < 200 ? 1 : 13))
            for (int i = 0; i < n; i++)
                CHECK(pg::sample_index(3, len, i, n) == 3 + ((2 * i + 1) * len) / (2 * n), "sample_index(%d,%d,%d)", len, i, n);
    // Raster spec S4 written out naively: s = a<255 ? C*a/255 : C;  D = s + (255-a)*D/255
    for (int a = 0; a < 256; a++)
        for (int s = 0; s < 256; s++)
            for (int d = 0; d < 256; d += (a % 16 == 0 ? 1 : 5)) {
                const int sc = (a < 255) ? s * a / 255 : s;
                const int want = sc + (255 - a) * d / 255;
                const uint32_t got = pg::blend_px(uint32_t(d) | uint32_t(d) << 8 | uint32_t(d) << 16,
                                                  uint32_t(s) | uint32_t(s) << 8 | uint32_t(s) << 16 | 0xab000000u, a);
                CHECK(got == (uint32_t(want) | uint32_t(want) << 8 | uint32_t(want) << 16), "blend a=%d s=%d d=%d", a, s, d);
            }
    // … and through a stamp's texel (pg_stamps.h: stamp_texel on the host, blend_premul on the device): the same pixel as
    // the naive S4 with the modulation applied first, for every texel alpha, modulation, colour and destination
    for (int mod : {255, 178, 254, 100, 1, 0})
        for (int A = 0; A < 256; A++)
            for (int s = 0; s < 256; s += (A % 8 == 0 ? 1 : 7))
                for (int d = 0; d < 256; d += (A % 16 == 0 ? 1 : 5)) {
                    const int a = mod != 255 ? A * mod / 255 : A;
                    int want = d;
                    if (a != 0) {
                        const int sc = (a < 255) ? s * a / 255 : s;
                        want = sc + (255 - a) * d / 255;
                    }
                    const uint32_t st = pg::stamp_texel(uint32_t(s) | uint32_t(s) << 8 | uint32_t(s) << 16 | uint32_t(A) << 24, mod);
                    const uint32_t dst = uint32_t(d) | uint32_t(d) << 8 | uint32_t(d) << 16;
                    const int sa = int(st >> 24);
                    const uint32_t got = sa == 0 ? dst : (sa == 255 ? (st & 0x00ffffffu) : pg::blend_premul(dst, st, sa));
                    CHECK(sa == a, "stamp alpha A=%d mod=%d", A, mod);
                    CHECK(got == (uint32_t(want) | uint32_t(want) << 8 | uint32_t(want) << 16), "stamp blend A=%d mod=%d s=%d d=%d", A, mod, s, d);
                }
    {   // channels that differ from each other (red and blue share a word inside blend_px)
        uint32_t seed = 12345u;
        auto next = [&]() { return seed = seed * 1664525u + 1013904223u; };
        for (int k = 0; k < 4000000; k++) {
            const uint32_t d = next() >> 8, sc = next(), a = next() >> 24;
            uint32_t want = 0;
            for (int sh = 0; sh < 24; sh += 8)
                want |= ((((sc >> sh) & 255u) * a) / 255u + (((d >> sh) & 255u) * (255u - a)) / 255u) << sh;
            CHECK(pg::blend_px(d, sc, static_cast<int>(a)) == want, "blend_px(%08x, %08x, %u)", d, sc, a);
        }
    }
    // resolve_draw is the composition of its two axes and reproduces the known coinrun tile geometry:
    // a 128-px tile at zoom 0.3 is 4.8 px, padded source 155 texels → destination trunc(5.8125) = 5 px.
    pg::Camera cam{100.0f, 200.0f, 64.0f, 64.0f, 0.3f};
    pg::Blit b;
    CHECK(pg::resolve_draw(cam, 128, 128, 7, 96.0f, 192.0f, 0.125f, 1.0f, false, false, b), "tile visible");
    CHECK(b.dw == 5 && b.dh == 5 && b.sx == 0 && b.sw == 128 && b.tex_off == 7 && b.tex_w == 128, "tile geometry");
    CHECK(!pg::resolve_draw(cam, 128, 128, 7, 4000.0f, 192.0f, 0.125f, 1.0f, false, false, b), "culled");
    std::printf("OK blend\n");
}

static void test_sincos() {
    // Against the sinf/cosf this process is linked with (glibc; the reference games call exactly these).
    auto bits = [](float f) {
        uint32_t u;
        std::memcpy(&u, &f, 4);
        return u;
    };
    auto check = [&](float x) {
        CHECK(bits(pg::sc_sinf(x)) == bits(sinf(x)), "sinf(%a): %a vs %a", x, pg::sc_sinf(x), sinf(x));
        CHECK(bits(pg::sc_cosf(x)) == bits(cosf(x)), "cosf(%a): %a vs %a", x, pg::sc_cosf(x), cosf(x));
    };
    // every float in [2^-13, 8) and its negative, stride 7 (≈ 21 M arguments): the games' angles live here
    for (uint32_t u = 0x39000000u; u < 0x41000000u; u += 7) {
        float x;
        std::memcpy(&x, &u, 4);
        check(x);
        check(-x);
    }
    // sparser sweep of everything else that is finite, including the |x| >= 120 reduction
    for (uint32_t u = 0; u < 0x7f800000u; u += 4099) {
        float x;
        std::memcpy(&x, &u, 4);
        check(x);
        check(-x);
    }
    std::mt19937 meta(31337);
    for (int i = 0; i < 2000000; i++) {
        float x = (static_cast<float>(meta() >> 8) / 16777216.0f - 0.5f) * 2000.0f;
        check(x);
    }
    std::printf("OK sincos\n");
}

// pg_atan2.h against the process's libm (what the reference's std::atan2(float, float) calls).
static void test_atan2() {
    std::mt19937 rng(99);
    auto bits = [](float f) {
        uint32_t u;
        std::memcpy(&u, &f, 4);
        return u;
    };
    long checked = 0;
    auto check = [&](float y, float x) {
        const float want = atan2f(y, x), got = pg::at_atan2f(y, x);
        if (bits(want) != bits(got) && !(want != want && got != got)) {
            std::printf("FAIL atan2f(%a, %a): libm %a twin %a\n", y, x, want, got);
            std::exit(1);
        }
        checked++;
    };
    std::uniform_real_distribution<float> world(-40.0f, 40.0f);
    for (int i = 0; i < 4000000; i++) check(world(rng), world(rng));  // to_goal vectors of a 40×40 level
    for (int i = 0; i < 4000000; i++) {  // arbitrary bit patterns: every branch incl. inf / nan / denormals
        uint32_t a = rng(), b = rng();
        float y, x;
        std::memcpy(&y, &a, 4);
        std::memcpy(&x, &b, 4);
        check(y, x);
    }
    const float specials[] = {0.0f, -0.0f, 1.0f, -1.0f, 0.4375f, 0.6875f, 1.1875f, 2.4375f, 1e-10f, 3e38f, 1e-40f,
                              INFINITY, -INFINITY, 0.5f, 2.0f, 33554432.0f, 67108864.0f};
    for (float y : specials)
        for (float x : specials) {
            check(y, x);
            check(-y, x);
            check(y, -x);
        }
    for (int i = 0; i < 2000000; i++) {  // atanf alone, dense around the interval boundaries
        uint32_t a = rng();
        float v;
        std::memcpy(&v, &a, 4);
        const float want = atanf(v), got = pg::at_atanf(v);
        if (bits(want) != bits(got) && !(want != want && got != got)) {
            std::printf("FAIL atanf(%a): libm %a twin %a\n", v, want, got);
            std::exit(1);
        }
    }
    std::printf("OK atan2 (%ld pairs)\n", checked);
}

// The axis arithmetic of Renderer::render_texture (games/*/renderer.cpp:5-82) + raster spec S1/S2 in one piece, as
// pg_geom.h resolve_axis was written before round 4 split it into axis_head / axis_tail for the render pre-pass.
static bool axis_in_one_piece(float cam_pos, float cam_len, float cam_scale, int tsize, float pos, float scale, bool flip,
                              bool strict_far, pg::Span& out) {
    float s = 0.0f;
    float sl = static_cast<float>(tsize);
    float d = (pos - cam_pos) * cam_scale + cam_len * 0.5f;
    float dl = tsize * scale * cam_scale;
    if ((strict_far ? d >= cam_len : d > cam_len) || d + dl < 0) return false;
    if (d < 0.0f) {
        float ratio = -d / dl;
        s += sl * ratio;
        sl -= s;
        dl += d;
        d = 0.0f;
    }
    if (d + dl > cam_len) {
        float ratio = (d + dl - cam_len) / dl;
        sl = sl * (1.0f - ratio);
        dl = cam_len - d;
    }
    int padding = static_cast<int>(ceilf(1.0f / (scale * cam_scale)));
    int r0 = static_cast<int>(floorf(s));
    int rl = static_cast<int>(ceilf(sl)) + padding;
    float off = s - r0;
    float ratio = rl / sl;
    dl *= ratio;
    d -= off * (dl / sl);
    if (flip) r0 = tsize - rl - r0;
    if (!(dl >= 1.0f && dl < 32768.0f)) return false;
    if (!(d > -32768.0f && d < 32768.0f)) return false;
    out.d0 = static_cast<int>(d);
    out.dn = static_cast<int>(dl);
    int a = r0, b = r0 + rl;
    if (a < 0) a = 0;
    if (b > tsize) b = tsize;
    out.s0 = a;
    out.sn = b - a;
    return out.sn > 0;
}

// What the render pre-pass (pg_prepass.h prep_spans) rests on: (1) head + tail IS the one-piece arithmetic, bit for bit;
// (2) a tile that neither crop touches has the tail of its axis's TEMPLATE (the same texture, scale and camera scale, d = 0)
// in everything but d0, and its d0 is (int)d.  Swept over the games' own parameter families (tile textures of 64, 53 and
// 128 texels at 16 / size pixels per texel, the zooms of all seven games and their modes, cameras at arbitrary float
// positions, tile positions on the 16-pixel grid) and over random floats.
static void test_axis_template() {
    uint32_t seed = 2024u;
    auto next = [&]() { return seed = seed * 1664525u + 1013904223u; };
    auto unit = [&]() { return static_cast<float>(next() >> 8) / 16777216.0f; };
    const int sizes[] = {64, 53, 128, 70, 99};
    const float zooms[] = {0.2f, 0.3f, 0.5f, 0.16f, 64.0f / 16.0f / 11.0f, 64.0f / 16.0f / 13.0f, 64.0f / 16.0f / 19.0f, 1.0f,
                           64.0f / (16.0f * 15.0f), 64.0f / (16.0f * 8.0f)};
    long whole = 0, cut = 0, culled = 0;
    for (int k = 0; k < 3000000; k++) {
        const bool on_grid = k % 4 != 0;
        const int tsize = sizes[next() % 5];
        const float cam_scale = on_grid ? zooms[next() % 10] : 0.05f + 2.0f * unit();
        const float scale = on_grid ? 16.0f / tsize : 0.01f + unit();
        const float cam_len = 64.0f;
        const float cam_pos = 1100.0f * unit() - 20.0f;
        // (positions around the camera: two thirds of a screen either side, so that whole, cut and culled tiles all occur)
        const float near = cam_pos + (2.0f * unit() - 1.0f) * (0.67f * cam_len / cam_scale);
        const float pos = on_grid ? 16.0f * floorf(near / 16.0f) : near;
        const bool flip = !on_grid && (next() & 1u), strict = (next() & 1u) != 0;
        pg::Span want{0, 0, 0, 0}, got{0, 0, 0, 0};
        const bool ok_want = axis_in_one_piece(cam_pos, cam_len, cam_scale, tsize, pos, scale, flip, strict, want);
        const bool ok_got = pg::resolve_axis(cam_pos, cam_len, cam_scale, tsize, pos, scale, flip, strict, got);
        CHECK(ok_want == ok_got && (!ok_want || (want.d0 == got.d0 && want.dn == got.dn && want.s0 == got.s0 && want.sn == got.sn)),
              "head + tail differs from the one-piece arithmetic (case %d)", k);
        pg::AxisHead h;
        if (!pg::axis_head(cam_pos, cam_len, cam_scale, tsize, pos, scale, strict, h)) {
            culled++;
            CHECK(!ok_want, "culled by the head, drawn by the arithmetic (case %d)", k);
            continue;
        }
        if (!(h.d < 0.0f) && !(h.d + h.dl > cam_len)) {  // prep_spans' test for "a whole tile"
            whole++;
            pg::Span tmpl{0, 0, 0, 0};
            pg::AxisHead t{0.0f, tsize * scale * cam_scale};  // the template's head: d = 0, the same dl
            const bool usable = !(t.d + t.dl > cam_len);
            CHECK(usable, "a whole tile whose template would be cropped (case %d)", k);
            const bool ok_tmpl = pg::axis_tail(cam_len, cam_scale, tsize, scale, flip, t, tmpl);
            CHECK(ok_tmpl == ok_want, "template and tile disagree on whether anything is drawn (case %d)", k);
            if (ok_want)
                CHECK(want.d0 == static_cast<int>(h.d) && want.dn == tmpl.dn && want.s0 == tmpl.s0 && want.sn == tmpl.sn,
                      "a whole tile is not its template shifted: d0 %d vs %d, dn %d vs %d, s0 %d vs %d, sn %d vs %d (case %d)", want.d0,
                      static_cast<int>(h.d), want.dn, tmpl.dn, want.s0, tmpl.s0, want.sn, tmpl.sn, k);
        } else {
            cut++;
        }
    }
    CHECK(whole > 300000 && cut > 100000 && culled > 100000, "the sweep must reach all three kinds (%ld whole, %ld cut, %ld culled)", whole, cut, culled);
    std::printf("OK axis template (%ld whole tiles, %ld cut, %ld culled)\n", whole, cut, culled);
}

// pg_geom.h rot_extent / rot_first / rot_last: every pixel raster spec S6 draws of a rotated rectangle lies inside the box
// they give.  Brute force in doubled pixel coordinates: all offsets within twice the rectangle's diagonal, the inside test
// exactly as pg_render.h wave_blit_rotated makes it (64-bit), over sizes 1..48 squared plus a few large ones, and the
// 16.16 sine / cosine of 3 000 angles (rounded the way rotation_16_16 rounds them, and off by one unit either way).
static void test_rot_box() {
    long drawn = 0, cases = 0, tight = 0;
    auto sweep = [&](int dw, int dh, int sn, int cs) {
        const int acs = cs < 0 ? -cs : cs, asn = sn < 0 ? -sn : sn;
        const int ex = pg::rot_extent(dw, dh, acs, asn), ey = pg::rot_extent(dh, dw, acs, asn);
        const int first_x = pg::rot_first(dw, ex), last_x = pg::rot_last(dw, ex), first_y = pg::rot_first(dh, ey), last_y = pg::rot_last(dh, ey);
        const int reach = dw + dh + 4;
        int seen_lo = 1 << 30, seen_hi = -(1 << 30);
        for (int Y = -reach; Y <= reach + dh; Y++)
            for (int X = -reach; X <= reach + dw; X++) {  // X, Y relative to the rectangle's corner (dx, dy)
                const long long px = 2 * X + 1 - dw, py = 2 * Y + 1 - dh;
                const long long lx = px * cs + py * sn + static_cast<long long>(dw) * 65536;
                const long long ly = -px * sn + py * cs + static_cast<long long>(dh) * 65536;
                if (lx < 0 || ly < 0 || lx >= static_cast<long long>(2 * dw) * 65536 || ly >= static_cast<long long>(2 * dh) * 65536) continue;
                drawn++;
                seen_lo = X < seen_lo ? X : seen_lo;
                seen_hi = X > seen_hi ? X : seen_hi;
                CHECK(X >= first_x && X <= last_x && Y >= first_y && Y <= last_y,
                      "a drawn pixel outside the box: %dx%d sn %d cs %d pixel (%d,%d) box x %d..%d y %d..%d", dw, dh, sn, cs, X, Y, first_x, last_x, first_y, last_y);
            }
        cases++;
        if (seen_lo <= seen_hi && last_x - first_x <= seen_hi - seen_lo + 2) tight++;
    };
    for (int k = 0; k < 3000; k++) {
        const double deg = k < 360 ? k : (k * 0.1234567 - 180.0);
        const float theta = static_cast<float>(deg * (3.14159265358979323846 / 180.0));
        const int sn = static_cast<int>(std::floor(static_cast<double>(std::sin(theta)) * 65536.0 + 0.5));
        const int cs = static_cast<int>(std::floor(static_cast<double>(std::cos(theta)) * 65536.0 + 0.5));
        const int dw = 1 + (k * 7) % 48, dh = 1 + (k * 13) % 48;
        sweep(dw, dh, sn, cs);
        sweep(dh, dw, sn + (k % 3) - 1, cs + ((k / 3) % 3) - 1);
        if (k % 100 == 0) sweep(200 + k / 20, 3 + k / 100, sn, cs);
    }
    for (int d = 1; d <= 12; d++)
        for (int deg = 0; deg < 360; deg++) {  // bullets and puffs: small squares at every whole degree
            const float theta = static_cast<float>(deg * (3.14159265358979323846 / 180.0));
            sweep(d, d, static_cast<int>(std::floor(static_cast<double>(std::sin(theta)) * 65536.0 + 0.5)),
                  static_cast<int>(std::floor(static_cast<double>(std::cos(theta)) * 65536.0 + 0.5)));
        }
    CHECK(drawn > 1000000 && tight * 10 > cases * 9, "the sweep must draw something and the box must be tight (%ld pixels, %ld of %ld cases within a pixel a side)", drawn, tight, cases);
    std::printf("OK rot box (%ld cases, %ld drawn pixels, all inside; %ld within a pixel a side of what is drawn)\n", cases, drawn, tight);
}

// pg_geom.h thin_run_start / thin_run_width: every pixel raster spec S6 draws of a THIN rotated rectangle on target row Y
// lies in [start, start + width) — the runs pg_render.h wave_blit_rotated scans instead of the rows of the bounding box.
// The exact 64-bit inside test over every pixel within the rectangle's reach, rectangles 9..64 long and 1..dw/3 high
// (jumper's needle is 30 × 6) at 1 500 angles and their 16.16 sines off by one unit either way, corners on and far off
// the target.
static void test_thin_runs() {
    long drawn = 0, cases = 0, slack = 0, rows = 0;
    for (int k = 0; k < 1500; k++) {
        const double deg = k < 360 ? k : (k * 0.2345678 - 180.0);
        const float theta = static_cast<float>(deg * (3.14159265358979323846 / 180.0));
        const int sn0 = static_cast<int>(std::floor(static_cast<double>(std::sin(theta)) * 65536.0 + 0.5));
        const int cs0 = static_cast<int>(std::floor(static_cast<double>(std::cos(theta)) * 65536.0 + 0.5));
        for (int v = 0; v < 3; v++) {
            const int sn = sn0 + (v == 1 ? 1 : (v == 2 ? -1 : 0)), cs = cs0 + (v == 2 ? 1 : 0);
            const int asn = sn < 0 ? -sn : sn;
            if (asn < 4096) continue;  // (wave_blit_rotated takes the plain scan)
            const int dw = 9 + (k * 7 + v) % 56, dh = 1 + (k * 5 + v) % (dw / 3);
            const int dx = -40 + (k * 11) % 120, dy = -40 + (k * 17) % 140;
            const int width = pg::thin_run_width(dh, asn);
            const float m = static_cast<float>(2 * dh) * 65536.0f, inv = 1.0f / static_cast<float>(sn);
            const int reach = dw + dh + 4;
            cases++;
            for (int Y = dy - reach; Y <= dy + dh + reach; Y++) {
                const int start = pg::thin_run_start(dx, dy, dw, dh, cs, m, inv, Y);
                int lo = 1 << 30, hi = -(1 << 30);
                for (int X = dx - reach; X <= dx + dw + reach; X++) {
                    const long long px = 2 * (X - dx) + 1 - dw, py = 2 * (Y - dy) + 1 - dh;
                    const long long lx = px * cs + py * sn + static_cast<long long>(dw) * 65536;
                    const long long ly = -px * sn + py * cs + static_cast<long long>(dh) * 65536;
                    if (lx < 0 || ly < 0 || lx >= static_cast<long long>(2 * dw) * 65536 || ly >= static_cast<long long>(2 * dh) * 65536) continue;
                    drawn++;
                    lo = X < lo ? X : lo, hi = X > hi ? X : hi;
                    CHECK(X >= start && X < start + width, "a drawn pixel outside its row's run: %dx%d at (%d,%d) sn %d cs %d pixel (%d,%d) run %d..%d", dw, dh,
                          dx, dy, sn, cs, X, Y, start, start + width - 1);
                }
                if (lo <= hi) rows++, slack += width - (hi - lo + 1);
            }
        }
    }
    CHECK(drawn > 500000 && cases > 3000, "the sweep must draw something (%ld pixels, %ld rectangles)", drawn, cases);
    std::printf("OK thin runs (%ld rectangles, %ld drawn pixels, all inside their row's run; %.1f spare columns a row)\n", cases, drawn,
                rows ? static_cast<double>(slack) / rows : 0.0);
}

// pg_geom.h span_nested — the rule by which a tile layer's second, shorter texture (the brown theme's 64×53 cap over 64×64
// bodies: jumper at zoom 0.3, climber at 0.2) may share the composer's per-pixel-row candidates with the first.  Swept
// over 400 000 camera heights per zoom with render_texture's own arithmetic (resolve_axis, y axis): whenever the rule
// says yes, every pixel row ON the target that the cap covers is covered by the body of the same grid row; it says yes
// for every camera height (no frame of these games has to fall back); and the rule it replaced (same start, no longer)
// said no for some — only ever for what the two rectangles do off the target.
static void test_span_nested() {
    long old_no = 0, cases = 0;
    for (float zoom : {0.3f, 0.2f}) {
        const float sh = 64.0f, tile_scale = 16.0f / 64.0f;
        for (int k = 0; k < 400000; k++) {
            const float cam_py = 40.0f + k * 0.00731f;
            const float vy = (cam_py - sh * 0.5f / zoom) / 16.0f;
            const int y0 = static_cast<int>(std::floor(vy));
            for (int r = 0; r < 24; r++) {
                pg::Span body, cap;
                const bool ok = pg::resolve_axis(cam_py, sh, zoom, 64, (y0 + r) * 16.0f, tile_scale, false, true, body);
                const bool ok2 = pg::resolve_axis(cam_py, sh, zoom, 53, (y0 + r) * 16.0f, tile_scale, false, true, cap);
                if (!ok2) continue;
                cases++;
                const bool nested = pg::span_nested(cap.d0, cap.dn, ok ? body.d0 : 0, ok ? body.dn : 0, 64);
                CHECK(nested, "a cap that is not nested: zoom %.1f camera %.4f grid row %d: body %d+%d cap %d+%d", zoom, cam_py, r, body.d0, body.dn, cap.d0, cap.dn);
                for (int p = 0; p < 64 && nested; p++) {
                    const bool in_cap = p >= cap.d0 && p < cap.d0 + cap.dn, in_body = ok && p >= body.d0 && p < body.d0 + body.dn;
                    CHECK(!in_cap || in_body, "pixel row %d under the cap but not under the body (camera %.4f grid row %d)", p, cam_py, r);
                }
                if (!ok || cap.d0 != body.d0 || cap.dn > body.dn) old_no++;
            }
        }
    }
    CHECK(old_no > 1000, "the sweep must reach the cases the old rule refused (%ld)", old_no);
    std::printf("OK span nested (%ld caps, all nested on the target; the rule before round 4 refused %ld of them)\n", cases, old_no);
}

int main() {
    test_axis_template();
    test_rot_box();
    test_thin_runs();
    test_span_nested();
    test_sincos();
    test_atan2();
    test_blend();
    test_mt();
    test_distributions();
    test_hash_order();
    test_sort();
    if (fails) {
        std::printf("%d failure(s)\n", fails);
        return 1;
    }
    std::printf("ALL OK\n");
    return 0;
}
