// climber on gfx950: vertical platformer, patrol mobs, collectible crystals.
//
// Reference:
//   step   games/climber/climber.cpp:323-376, common_systems.cpp:184-270 (agent), :109-168 (mobs), :66-107 (points),
//          :8-39 (sprite list)
//   render games/climber/climber.cpp:431-459, tilemap.cpp:172-198, common_systems.cpp:41-63, :272-298
//   reset  games/climber/climber.cpp:461-497, tilemap.cpp:40-70, :75-170
// Config = the reference's compile-time default (easy_mode = false, climber/tilemap.h:32-34).
//
// Machine mapping as in coinrun.hip: logic one lane per env over struct-of-arrays state, render one wavefront per
// env.  Collected crystals are destroyed entities in the reference: they leave the sprite System's
// std::unordered_set, and the z-sorted draw list is rebuilt from that set every sub-step.  Erasing from the
// libstdc++ node list keeps the relative order of the survivors, so the device keeps the episode's insertion-built
// order (pg_order.h, computed at reset) plus an alive bit per entity, and re-runs the introsort emulation over the
// survivors whenever the set changed.
#include "pg_engine.h"
#include "pg_frame.h"
#include "pg_gang.h"
#include "pg_geom.h"
#include "pg_order.h"
#include "pg_prefetch.h"
#include "pg_prepass.h"
#include "pg_render.h"
#include "pg_rng.h"
#include "pg_tiles.h"

namespace pg {
namespace PG_VARIANT_NS {
namespace climber {

// climber/tilemap.cpp:118 `enemy_prob = cfg.easy_mode ? .2 : .5` (tilemap.h:33: easy_mode = false is the default)
#if PG_VARIANT == 0
constexpr float kEnemyProb = 0.5f;
#elif PG_VARIANT == 1  // easy_mode
constexpr float kEnemyProb = 0.2f;
#else
#error "climber: unknown PG_VARIANT"
#endif

constexpr int W = 20, H = 64;
constexpr int kMaxEnt = 34, kEntStride = 48;  // ≤ 17 platforms (difficulty 3), each at most one mob and one crystal (tilemap.cpp:98-105)
enum Tile : uint8_t { kEmpty = 0, kWallTop, kWallMid };  // tilemap.h:12-17

enum Tex {
    kTexTop = 0,    // 4 themes
    kTexMid = 4,    // 4 themes
    kTexStand = 8,  // 4 suits each
    kTexJump = 12,
    kTexWalk1 = 16,
    kTexWalk2 = 20,
    kTexFish = 24,  // 2 frames
    kTexGem = 26,
    kTexBackdrop = 27,  // 10
    kTexCount = 37
};

enum { F_AX, F_AY, F_AVX, F_AVY, F_APHASE, F_CAMY, F_BGSHIFT, F_COUNT };
enum { I_FLAGS, I_THEMES, I_NENT, I_NDRAW, I_HASH_SPRITE, I_COUNT };
constexpr int kFlagGround = 1, kFlagForward = 2, kFlagListed = 4;
enum { EF_X, EF_Y, EF_VX, EF_ANIM_T, EF_COUNT };
enum { EB_INFO, EB_SPAWN_X, EB_ORDER, EB_DRAW, EB_COUNT };
// EB_INFO bits
constexpr int kMob = 1, kAlive = 2, kFrame = 4, kFlip = 8, kTexSet = 16;

// One generated level, as the generator leaves it in LDS and as it waits in the shadow slot (pg_prefetch.h).
struct Level {
    uint8_t tiles[W * H];
    float bgshift;
    int32_t themes, n_ent;
    float ex[kMaxEnt], ey[kMaxEnt], evx[kMaxEnt];
    uint8_t info[kMaxEnt], spawn_x[kMaxEnt], order[kMaxEnt];
    uint8_t pad[2];
};

struct GenLds {
    uint32_t mt[kMtWords];
};

struct State {
    int n;
    Level* shadow;   // [n]  next level of each env
    int32_t* slot;   // [n]  SlotState
    uint32_t* mt;    // [n][625]  generator chain: the stream position after the newest generated level
    uint8_t* tiles;  // [n][1280]  column-major y + x*H
    float* f;        // [F_COUNT][n]
    int32_t* i;      // [I_COUNT][n]
    float* ef;       // [n][EF_COUNT][kEntStride]  per-env contiguous: a gang's and the render wavefronts' lanes index them by entity
    uint8_t* eb;     // [n][EB_COUNT][kEntStride]
    const uint8_t* ranks;  // pg_order.h equal-key sort ranks
    PrepOut prep;          // what setup_kernel leaves for render_kernel (pg_prepass.h); not part of the state blob
};

PG_D float& SF(const State& s, int field, int env) { return s.f[size_t(field) * s.n + env]; }
PG_D int32_t& SI(const State& s, int field, int env) { return s.i[size_t(field) * s.n + env]; }
PG_D float& EF(const State& s, int field, int e, int env) { return s.ef[(size_t(env) * EF_COUNT + field) * kEntStride + e]; }
PG_D uint8_t& EB(const State& s, int field, int e, int env) { return s.eb[(size_t(env) * EB_COUNT + field) * kEntStride + e]; }

using Win = TileWinT<W, H, kWallMid>;  // out of bounds is a wall (tilemap.h:66-68)

// Tile writes by the whole wavefront into the map under construction (LDS); later fills overwrite earlier ones, so
// each ends with a barrier.
PG_D void fill(uint8_t* t, int x, int y, int w, int h, int id, int lane) {
    const int total = (w > 0 && h > 0) ? w * h : 0;
    for (int k = lane; k < total; k += 64) {
        const int a = udiv_small(k, h), b = k - a * h;
        const int px = x + a, py = y + b;
        if (!(px < 0 || py < 0 || px >= W || py >= H)) t[py + px * H] = static_cast<uint8_t>(id);
    }
    __syncthreads();
}
PG_D void fill_capped(uint8_t* t, int x, int y, int w, int h, int body, int cap, int lane) {
    fill(t, x, y, w, h - 1, body, lane);
    fill(t, x, y + h - 1, w, 1, cap, lane);
}

// The sprite System's set order for this episode: ids 0..n-1 inserted in creation order into a set that kept its
// bucket array across clear() (packed = buckets | next_resize << 16).
PG_D void episode_order(int32_t& packed, int n, uint8_t* out) {
    int16_t next[kMaxEnt];
    int16_t before[64];
    HashOrder h;
    h.next = next;
    h.before = before;
    h.head = kNil;
    h.buckets = packed & 0xffff;
    h.next_resize = packed >> 16;
    h.count = 0;
    if (h.buckets == 0) h.buckets = 1;
    for (int b = 0; b < h.buckets; b++) before[b] = kNil;
    for (int k = 0; k < n; k++) hash_insert(h, k);
    int16_t p = static_cast<int16_t>(h.head);
    for (int k = 0; k < n; k++) {
        out[k] = static_cast<uint8_t>(p);
        p = next[p];
    }
    packed = h.buckets | (h.next_resize << 16);
}

// reset() (climber.cpp:461-497 + tilemap.cpp:75-170) for one env by one wavefront: every lane walks the generator
// (the draws are wave-uniform), lane 0 records the entities.  Advances the env's generator chain (s.mt, the sprite
// set's bucket count) and leaves the level in `lv` (LDS).
PG_D void generate(const State& s, int env, GenLds& L, Level& lv, bool reseed, uint32_t seed, int lane) {
    uint32_t* gmt = s.mt + size_t(env) * kMtWords;
    if (reseed) {
        if (lane == 0) mt_seed(L.mt, seed);
    } else {
        for (int k = lane; k < kMtWords; k += 64) L.mt[k] = gmt[k];
    }
    __syncthreads();
    uint32_t* mt = L.mt;
    uint8_t* tiles = lv.tiles;
    const float max_jump = 1.5f, gravity = 0.2f;
    fill(tiles, 0, 0, W, H, kEmpty, lane);
    fill_capped(tiles, 0, 0, W, 1, kWallMid, kWallTop, lane);
    fill(tiles, 0, 0, 1, H, kWallMid, lane);
    fill(tiles, W - 1, 0, 1, H, kWallMid, lane);
    fill(tiles, 0, H - 1, W, 1, kWallMid, lane);
    const int difficulty = wave_rng_int(mt, 1, 3, lane);
    const int platforms = wave_rng_int(mt, difficulty * difficulty + 1, (difficulty + 1) * (difficulty + 1) + 1, lane);
    int cx = wave_rng_int(mt, 2, W - 3, lane), cy = 1;
    const int margin = 3;
    const float enemy_prob = kEnemyProb;
    const float reach_y = max_jump * max_jump / (2.0f * gravity);
    const int max_dy = static_cast<int>(reach_y - 0.5f);
    int n_ent = 0;
    auto spawn = [&](int x, int y, int info, float vx) {
        const int e = n_ent++;
        if (lane == 0) {
            lv.ex[e] = static_cast<float>(x) + 0.5f;
            lv.ey[e] = static_cast<float>(H - 1 - y) + 0.5f;
            lv.evx[e] = vx;
            lv.info[e] = static_cast<uint8_t>(info | kAlive);
            lv.spawn_x[e] = static_cast<uint8_t>(x);
        }
    };
    for (int p = 0; p < platforms; p++) {
        const int dy = wave_rng_int(mt, 3, max_dy - 1, lane);
        const bool roomy = (cx >= margin) && (cx <= W - 1 - margin);
        if (roomy && (wave_rng_real(mt, 0.0f, 1.0f, lane) < enemy_prob)) {
            const int my = cy + wave_rng_int(mt, 0, 1, lane) + 2;
            const float vx = 0.15f * (wave_rng_int(mt, 0, 1, lane) * 2.0f - 1.0f);  // tilemap.cpp:55
            spawn(cx, my, kMob, vx);
        }
        cy += dy;
        const int len = 2 + wave_rng_int(mt, 0, 9, lane);
        int vx = wave_rng_int(mt, 0, 1, lane) * 2 - 1;
        if (cx < margin) vx = 1;
        if (cx > W - margin) vx = -1;
        int spots[12];
        int n_spots = 0;
        for (int j = 0; j < len; j++) {
            const int nx = cx + (j + 1) * vx;
            if (nx <= 0 || nx >= W - 1) break;
            spots[n_spots++] = nx;
            fill_capped(tiles, nx, cy, 1, 1, kWallMid, kWallTop, lane);
        }
        if (wave_rng_real(mt, 0.0f, 1.0f, lane) < 0.5f || p == platforms - 1)
            spawn(spots[wave_rng_int(mt, 0, n_spots - 1, lane)], cy + 1, kTexSet, 0.0f);  // crystal: textured from the start
        cx = spots[wave_rng_int(mt, 0, n_spots - 1, lane)];
    }
    const int backdrop = wave_rng_int(mt, 0, 9, lane);
    const float shift = wave_rng_real(mt, 0.0f, 1.0f, lane);
    const int suit = wave_rng_int(mt, 0, 3, lane);
    const int theme = wave_rng_int(mt, 0, 3, lane);
    if (lane == 0) {
        lv.bgshift = shift;
        lv.themes = backdrop | (suit << 8) | (theme << 16);
        lv.n_ent = n_ent;
        int32_t packed = SI(s, I_HASH_SPRITE, env);
        episode_order(packed, n_ent, lv.order);
        SI(s, I_HASH_SPRITE, env) = packed;
    }
    __syncthreads();
    for (int k = lane; k < kMtWords; k += 64) gmt[k] = L.mt[k];
    __syncthreads();
}

// The level becomes the env's live state (what reset() and the component constructors initialise).
PG_D void install(const State& s, int env, const Level& lv, int lane) {
    uint32_t* tiles = reinterpret_cast<uint32_t*>(s.tiles + size_t(env) * (W * H));
    const uint32_t* src = reinterpret_cast<const uint32_t*>(lv.tiles);
    for (int k = lane; k < W * H / 4; k += 64) tiles[k] = src[k];
    const int n_ent = lv.n_ent;
    if (lane < n_ent) {
        EF(s, EF_X, lane, env) = lv.ex[lane];
        EF(s, EF_Y, lane, env) = lv.ey[lane];
        EF(s, EF_VX, lane, env) = lv.evx[lane];
        EF(s, EF_ANIM_T, lane, env) = 0.0f;
        EB(s, EB_INFO, lane, env) = lv.info[lane];
        EB(s, EB_SPAWN_X, lane, env) = lv.spawn_x[lane];
        EB(s, EB_ORDER, lane, env) = lv.order[lane];
    }
    if (lane == 0) {
        SF(s, F_BGSHIFT, env) = lv.bgshift;
        SF(s, F_AX, env) = 1.5f;
        SF(s, F_AY, env) = H - 2 + 1.0f;
        SF(s, F_AVX, env) = 0.0f;
        SF(s, F_AVY, env) = 0.0f;
        SF(s, F_APHASE, env) = 0.0f;
        SI(s, I_FLAGS, env) = kFlagForward;  // on_ground = false, face_forward = true, draw list cleared (D2)
        SI(s, I_THEMES, env) = lv.themes;
        SI(s, I_NENT, env) = n_ent;
        SI(s, I_NDRAW, env) = 0;
        // camera x is fixed (climber.cpp:466); camera y keeps the previous episode's value (D3)
    }
}

// What cenv_make leaves in an env besides the seeded RNG, split by owner: the generator chain (bucket counts of
// the sets that survive clear()) and the live state.  Level-seed mode (pg_engine.h LevelPlan) rebuilds every
// level from here.
PG_D void fresh_chain(const State& s, int env) {
    SI(s, I_HASH_SPRITE, env) = 1;
}
PG_D void fresh_live(const State& s, int env) {
    SF(s, F_CAMY, env) = 0.0f;  // Renderer::camera_position{0} (renderer.h:18)
}

struct Gen {  // pg_prefetch.h level_kernel<Gen>
    using State = climber::State;
    using Level = climber::Level;
    using GenLds = climber::GenLds;
    PG_D static void generate(const State& s, int env, GenLds& L, Level& lv, bool reseed, uint32_t seed, int lane) {
        climber::generate(s, env, L, lv, reseed, seed, lane);
    }
    PG_D static void install(const State& s, int env, const Level& lv, int lane) { climber::install(s, env, lv, lane); }
    PG_D static void fresh_chain(const State& s, int env) { climber::fresh_chain(s, env); }
    PG_D static void fresh_live(const State& s, int env) { climber::fresh_live(s, env); }
};

PG_D bool is_wall(int t) { return t == kWallMid || t == kWallTop; }

// A level has at most 17 platforms (difficulty 3: (3+1)² + 1, tilemap.cpp:98-105), each with at most one mob and one
// crystal.
constexpr int kMaxMobs = 17, kMaxGems = 17;

// One env = one gang of kGang adjacent lanes (pg_gang.h).  What costs in this game is get_collision — the agent's box
// and one probe per mob, every sub-step — so those are the gang's parallel work: collider 0 is the agent, collider
// 1 + m the m-th alive mob, kGang of them per pass through ONE copy of the collision code.  Everything else about the
// agent is uniform over the gang; mobs and crystals (alive ones, compacted in id order) live in LDS for the step.
#ifndef PG_CLIMBER_GANG
#define PG_CLIMBER_GANG 8
#endif
#ifndef PG_CLIMBER_WAVES
#define PG_CLIMBER_WAVES 4  // wavefronts per SIMD the logic kernel's registers are capped for
#endif
constexpr int kGang = PG_CLIMBER_GANG;
using Q = Gang<kGang>;

struct StepLds {  // one per gang
    float m_x[kMaxMobs], m_vx[kMaxMobs], m_t[kMaxMobs];
    unsigned long long m_win[kMaxMobs];
    int8_t m_ax[kMaxMobs], m_ay[kMaxMobs];
    uint8_t m_row[kMaxMobs], m_spawn[kMaxMobs], m_id[kMaxMobs], m_info[kMaxMobs];
    uint8_t g_col[kMaxGems], g_row[kMaxGems], g_id[kMaxGems], g_info[kMaxGems];
};

// System_Sprite_Render::update's list: the surviving sprites in set order, then std::sort on z (all 1.0).
PG_D void rebuild_draw_list(const State& s, Q q, int env, int n_ent) {
    int n = 0;
    for (int k0 = 0; k0 < n_ent; k0 += kGang) {
        const int k = k0 + q.g;
        n += __popc(q.ballot(k < n_ent && (EB(s, EB_INFO, k, env) & kAlive)));
    }
    const uint8_t* rank = s.ranks + rank_offset(n);  // equal keys: the sort is a fixed permutation for each n
    int r = 0;
    for (int k0 = 0; k0 < n_ent; k0 += kGang) {
        const int k = k0 + q.g;
        const int e = k < n_ent ? EB(s, EB_ORDER, k, env) : 0;
        const bool alive = k < n_ent && (EB(s, EB_INFO, e, env) & kAlive);
        const uint32_t m = q.ballot(alive);
        if (alive) EB(s, EB_DRAW, rank[r + __popc(m & ((1u << q.g) - 1u))], env) = static_cast<uint8_t>(e);
        r += __popc(m);
    }
    if (q.g == 0) SI(s, I_NDRAW, env) = n;
}

PG_D void advance(const State& s, StepLds& L, Q q, int env, int action, float& reward_out, bool& terminated_out) {
    const uint8_t* tiles = s.tiles + size_t(env) * (W * H);
    const int n_ent = SI(s, I_NENT, env);
    const float dt = 1.0f / 4;
    const uint32_t below = (1u << q.g) - 1u;

    // --- stage the alive entities (reference loop: `if (!alive) continue`): mobs and crystals in id order
    int n_mobs = 0, n_gems = 0;
    for (int e0 = 0; e0 < n_ent; e0 += kGang) {
        const int e = e0 + q.g;
        const bool ok = e < n_ent;
        const int info = ok ? EB(s, EB_INFO, e, env) : 0;
        const int spawn = ok ? EB(s, EB_SPAWN_X, e, env) : 0;
        const float x = ok ? EF(s, EF_X, e, env) : 0.0f, y = ok ? EF(s, EF_Y, e, env) : 0.0f;
        const float vx = ok ? EF(s, EF_VX, e, env) : 0.0f, t = ok ? EF(s, EF_ANIM_T, e, env) : 0.0f;
        const bool alive = (info & kAlive) != 0, mob = alive && (info & kMob), gem = alive && !(info & kMob);
        const uint32_t mobs = q.ballot(mob), gems = q.ballot(gem);
        const int row = H - 1 - static_cast<int>(y);  // y = (H-1-row) + 0.5 exactly (generate(): spawn)
        if (mob) {
            const int m = n_mobs + __popc(mobs & below);
            if (m >= kMaxMobs) __builtin_trap();
            L.m_x[m] = x;
            L.m_vx[m] = vx;
            L.m_t[m] = t;
            L.m_row[m] = static_cast<uint8_t>(row);
            L.m_spawn[m] = static_cast<uint8_t>(spawn);
            L.m_id[m] = static_cast<uint8_t>(e);
            L.m_info[m] = static_cast<uint8_t>(info);
            // its tile window for the step, around the probe of the first sub-step, spare column on the side it is heading
            // (mobs drift 0.15 tiles a step; `at()` falls back to a direct load outside the window: a cache, not an assumption)
            const float my = static_cast<float>(H - 1 - row) + 0.5f;
            const Box probe{x + vx * dt - 0.5f, my - 0.6f, 1.0f, 0.5f};
            const Win w = Win::around(tiles, probe, vx, 0.0f);
            L.m_win[m] = w.bits;
            L.m_ax[m] = static_cast<int8_t>(w.ax);
            L.m_ay[m] = static_cast<int8_t>(w.ay);
        }
        if (gem) {
            const int k = n_gems + __popc(gems & below);
            if (k >= kMaxGems) __builtin_trap();
            L.g_col[k] = static_cast<uint8_t>(static_cast<int>(x));  // x = col + 0.5 exactly
            L.g_row[k] = static_cast<uint8_t>(row);
            L.g_id[k] = static_cast<uint8_t>(e);
            L.g_info[k] = static_cast<uint8_t>(info);
        }
        n_mobs += __popc(mobs);
        n_gems += __popc(gems);
    }
    wave_order();

    int flags = SI(s, I_FLAGS, env);
    float ax = SF(s, F_AX, env), ay = SF(s, F_AY, env);
    float avx = SF(s, F_AVX, env), avy = SF(s, F_AVY, env);
    float phase = SF(s, F_APHASE, env), camy = SF(s, F_CAMY, env);
    bool ground = (flags & kFlagGround) != 0, forward = (flags & kFlagForward) != 0;
    const float max_jump = 1.55f, gravity = 0.2f, max_speed = 0.5f, mix = 0.2f, air_control = 0.15f;
    const float move_x = static_cast<float>((action == 6 || action == 7 || action == 8) -
                                            (action == 0 || action == 1 || action == 2));
    const bool jump = (action == 2 || action == 5 || action == 8);

    float reward = 0.0f;
    bool terminated = false, set_changed = (flags & kFlagListed) == 0;
    Win awin{tiles, 0, 0, 0};
    for (int ss = 0; ss < 4; ss++) {
        // --- System_Agent::update (common_systems.cpp:184-270), up to its get_collision
        const float mix_x = ground ? mix : (mix * air_control);
        avx += mix_x * (max_speed * move_x - avx) * dt;
        if (fabsf(avx) < mix_x * max_speed * dt) avx = 0.0f;
        if (jump && ground) avy = -max_jump;
        avy += gravity * dt;
        if (fabsf(avy) > max_jump) avy = (avy > 0.0f ? 1.0f : -1.0f) * max_jump;
        ax += avx * dt;
        ay += avy * dt;
        const Box body{ax + -0.5f, ay + -1.0f, 1.0f, 1.0f};
        if (ss == 0 || !awin.holds(body)) awin = Win::around(tiles, body, avx, avy);

        // --- the agent's get_collision: the same box in every lane of the gang, so the walk's nine cells go side by side
        // (pg_tiles.h collide_plain_gang8; until round 6 the agent was lane 0 of the mobs' pass and every lane walked)
        bool dead = false;
        Box agent{0.0f, 0.0f, 1.0f, 1.0f};
        {
            static_assert(kGang == 8, "collide_plain_gang8");
            const TileHit h = collide_plain_gang8(q, awin, body, is_wall);
            const float moved_x = h.x - body.x, moved_y = h.y - body.y;
            ground = moved_y < 0.0f && h.any;
            ax = h.x - -0.5f;
            ay = h.y - -1.0f;
            if (moved_x != 0.0f) avx = 0.0f;
            if (ground) avy = 0.0f;
            camy = (ay - 8 - 0.5f) * kUnitPx;
            phase += 0.1f * dt;
            phase = fmodf(phase, 1.0f);
            if (move_x > 0.0f)
                forward = true;
            else if (move_x < 0.0f)
                forward = false;
            agent = Box{ax + -0.5f, ay + -1.0f, 1.0f, 1.0f};
        }
        // --- the mobs, kGang per pass (common_systems.cpp:109-168; order-free per entity).  Their walk skips the cells no
        // mob of the wavefront meets (pg_tiles.h PG_WALK_SKIP: 96 -> 88 µs); a whether-test in front of it as in coinrun
        // (collide_any) then buys nothing more here — of the ≈ 60 mobs of a pass one usually does touch — and is not made.
        for (int m0 = 0; m0 < n_mobs; m0 += kGang) {
            const int m = m0 + q.g;
            const bool is_mob = m < n_mobs;
            float x = 0.0f, vx = 0.0f, y = 0.0f;
            Box probe = body;
            Win win = awin;
            if (is_mob) {
                x = L.m_x[m];
                vx = L.m_vx[m];
                y = static_cast<float>(H - 1 - L.m_row[m]) + 0.5f;
                x += vx * dt;
                probe = Box{x - 0.5f, y - 0.6f, 1.0f, 0.5f};
                win = Win{tiles, L.m_ax[m], L.m_ay[m], L.m_win[m]};
            }
            TileHit h{probe.x, probe.y, false};
            if (is_mob) h = collide_plain<true>(win, probe, is_wall);
            bool bitten = false;
            if (is_mob) {
                int info = L.m_info[m];
                x = h.x + 0.5f;
                bitten = box_hit(agent, Box{x + -0.4f, y + -0.4f, 0.8f, 0.8f});
                const int spawn_x = L.m_spawn[m];
                const bool end_patrol = x > spawn_x + 4 || x < spawn_x - 4;
                if (h.any || end_patrol) vx *= -1.0f;
                info = (info & ~kFlip) | (vx < 0.0f ? kFlip : 0);
                float t = L.m_t[m] + dt;
                const int adv = static_cast<int>(t * 0.2f);
                t -= adv / 0.2f;
                const int frame = (((info & kFrame) ? 1 : 0) + adv) % 2;
                info = (info & ~kFrame) | (frame ? kFrame : 0) | kTexSet;
                L.m_x[m] = x;
                L.m_vx[m] = vx;
                L.m_t[m] = t;
                L.m_info[m] = static_cast<uint8_t>(info);
            }
            if (q.any(bitten)) dead = true;
        }
        // --- points (:66-107)
        int delta = 0, available = 0;
        for (int k0 = 0; k0 < n_gems; k0 += kGang) {
            const int k = k0 + q.g;
            const int info = k < n_gems ? L.g_info[k] : 0;
            const bool alive = (info & kAlive) != 0;
            const float x = static_cast<float>(k < n_gems ? L.g_col[k] : 0) + 0.5f;
            const float y = static_cast<float>(H - 1 - (k < n_gems ? L.g_row[k] : 0)) + 0.5f;
            const bool taken = alive && box_hit(agent, Box{x + -0.5f, y + -0.5f, 1.0f, 1.0f});
            if (taken) L.g_info[k] = static_cast<uint8_t>(info & ~kAlive);  // destroy_entity
            delta += __popc(q.ballot(taken));
            available += __popc(q.ballot(alive && !taken));
        }
        if (delta) set_changed = true;
        reward = delta + (available == 0) * 10.0f;
        terminated = dead || (available == 0);
        if (terminated) break;
    }
    // Note: in the reference the mob loop runs before the point loop within a sub-step; the two never read each
    // other's data, so two loops per sub-step yield the same state.
    for (int m0 = 0; m0 < n_mobs; m0 += kGang) {
        const int m = m0 + q.g;
        if (m < n_mobs) {
            const int e = L.m_id[m];
            EF(s, EF_X, e, env) = L.m_x[m];
            EF(s, EF_VX, e, env) = L.m_vx[m];
            EF(s, EF_ANIM_T, e, env) = L.m_t[m];
            EB(s, EB_INFO, e, env) = L.m_info[m];
        }
    }
    if (set_changed)
        for (int k0 = 0; k0 < n_gems; k0 += kGang) {
            const int k = k0 + q.g;
            if (k < n_gems) EB(s, EB_INFO, L.g_id[k], env) = L.g_info[k];
        }
    if (q.g == 0) {
        SF(s, F_AX, env) = ax;
        SF(s, F_AY, env) = ay;
        SF(s, F_AVX, env) = avx;
        SF(s, F_AVY, env) = avy;
        SF(s, F_APHASE, env) = phase;
        SF(s, F_CAMY, env) = camy;
        SI(s, I_FLAGS, env) = kFlagListed | (ground ? kFlagGround : 0) | (forward ? kFlagForward : 0);
    }
    if (set_changed) {
        gang_fence();  // the info bytes just written, for the lanes that read them below
        rebuild_draw_list(s, q, env, n_ent);
    }
    reward_out = reward;
    terminated_out = terminated;
}

__global__ void __launch_bounds__(64) make_kernel(State s, uint32_t seed_base, int env_offset) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= s.n) return;
    fresh_chain(s, env);
    fresh_live(s, env);
}

__global__ void __launch_bounds__(64, PG_CLIMBER_WAVES) logic_kernel(State s, const int32_t* actions, uint32_t run_seed,
                                                                     uint32_t step_index, int env_offset, StepIO io, int prefetch, LevelPlan plan) {
    // One block is EITHER a row of gangs stepping their envs (StepLds each) OR the install row (a Level staged on its way from
    // the shadow slot to the live state): the same LDS serves both — as two arrays the logic blocks carried the Level's
    // 2–3 KB for nothing and fewer of them fitted a CU (caveflyer: 11.8 KB a block, 13 blocks instead of 16).
    constexpr size_t kLdsBytes = sizeof(StepLds) * (64 / kGang) > sizeof(Level) ? sizeof(StepLds) * (64 / kGang) : sizeof(Level);
    __shared__ alignas(16) unsigned char lds_bytes[kLdsBytes];
    StepLds* const lds = reinterpret_cast<StepLds*>(lds_bytes);
    if (blockIdx.y == 1) {  // (block-uniform) the auto-resets whose level lies ready: a copy, beside the envs that step (pg_prefetch.h)
        install_prefetched<Gen>(s, blockIdx.x * (64 / kGang), 64 / kGang, prefetch, io, plan, *reinterpret_cast<Level*>(lds_bytes), threadIdx.x, reset_served_mark(step_index), reset_due_mark(step_index));
        return;
    }
    const int env = (blockIdx.x * 64 + threadIdx.x) / kGang;
    if (env >= s.n) return;
    const Q q = Q::at(threadIdx.x);
    if (resets_in_step(io.pending[env], step_index)) return;  // this step is the env's reset (pg_prefetch.h: who serves it, and the byte)
    const int action =
        actions ? actions[env] : synthetic_action(run_seed, step_index, static_cast<uint32_t>(env_offset + env));
    float reward = 0.0f;
    bool terminated = false;
    advance(s, lds[(threadIdx.x & 63) / kGang], q, env, action, reward, terminated);
    if (q.g == 0) {
        io.reward[env] = reward;
        io.done[env] = terminated ? 1 : 0;
        io.pending[env] = terminated ? static_cast<uint8_t>(reset_due_mark(step_index + 1u)) : 0;  // (pg_prefetch.h: the byte)
    }
}

// render_game(true) (climber.cpp:431-459): one workgroup of two wavefronts per env (pg_render.h).
constexpr int kGrid = 24;  // 64 px / 3.2 px per tile = 20 tiles → at most 22 columns/rows in view (24² = 9·64 cells; LDS: 7 envs per CU instead of 6)

// The complete frame of one env by its workgroup, set-up included: the frames the pre-pass marks fat, the draw-list
// replay (flags bit 0) and kDebugNoPrepass.
PG_D void render_full(const State& s, const AtlasView& atlas, const StepIO& io, int flags, int env, uint32_t* fb,
                      ComposeLds<kGrid>& L) {
    const int lane = threadIdx.x & 63, half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // two wavefronts per env (pg_render.h)
    constexpr int halves = 2;

    const Camera cam{W / 2.0f * kUnitPx, SF(s, F_CAMY, env), 64.0f, 64.0f, 0.2f * 64.0f / 64.0f};
    const int themes = SI(s, I_THEMES, env), sflags = SI(s, I_FLAGS, env);
    const int backdrop = themes & 0xff, suit = (themes >> 8) & 0xff, theme = (themes >> 16) & 0xff;
    const int n_draw = (sflags & kFlagListed) ? SI(s, I_NDRAW, env) : 0;  // empty right after a reset (D2)
    const uint8_t* tiles = s.tiles + size_t(env) * (W * H);
    const DescRegs descs = DescRegs::load(atlas, lane);
    Blit mine;

    // sprite-pass inputs, requested early
    const bool is_sprite = lane < n_draw, is_agent = lane == n_draw;
    int spr_info = 0;
    float spr_x = 0.0f, spr_y = 0.0f;
    if (is_sprite) {
        const int e = EB(s, EB_DRAW, lane, env);
        spr_info = EB(s, EB_INFO, e, env);
        spr_x = EF(s, EF_X, e, env);
        spr_y = EF(s, EF_Y, e, env);
    }

    int bg_soft = 0;  // the backdrop has texels that are not opaque (descriptor .w)
    int4 bg_d;  // the background draw, climber.cpp:447-452: texture, world position, scale — each wave resolves the axis it needs (pg_render.h BgAxis)
    float bg_px, bg_py, bg_sc;
    {
        const int4 d = descs.uniform(kTexBackdrop + backdrop);
        bg_soft = d.w;
        const float aspect = static_cast<float>(d.y) / static_cast<float>(d.z);
        const float extra = aspect - 1.0f;
        bg_d = d;
        bg_px = -SF(s, F_BGSHIFT, env) * extra;
        bg_py = 0.0f;
        bg_sc = 64.0f * kUnitPx / d.z;
    }
    // tile window (tilemap.cpp:172-181)
    const float vx = (cam.px - cam.sw * 0.5f / cam.scale) * kPxUnit;
    const float vy = (cam.py - cam.sh * 0.5f / cam.scale) * kPxUnit;
    const float vw = cam.sw * kPxUnit / cam.scale, vh = cam.sh * kPxUnit / cam.scale;
    const int x0 = static_cast<int>(floorf(vx)), y0 = static_cast<int>(floorf(vy));
    const int x1 = static_cast<int>(ceilf(vx + vw)), y1 = static_cast<int>(ceilf(vy + vh));
    const int cols = x1 - x0 + 1, rows = y1 - y0 + 1, cells = cols * rows;
    const int4 top_d = descs.uniform(kTexTop + theme), mid_d = descs.uniform(kTexMid + theme);

    const BgDraw bg_draw{bg_d, bg_px, bg_py, bg_sc};
    BgAxis bga{};  // this wave's axis of it (wave 0: x, wave 1: y), resolved along with the tile spans
    bool composed = false;
    // The brown theme's cap tile is 64×53 next to 64×64 bodies (assets/platformer/tileBrown_06.png): the composer's
    // two-texture mode; a cap taller than the body, or of another width, would take the draw-list replay.
    const bool two = top_d.z != mid_d.z;
    if (!(flags & 1) && cols <= kGrid && rows <= kGrid && top_d.y == mid_d.y && top_d.z <= mid_d.z) {
        compose_spans(fb, L, cam, x0, y0, cols, rows, mid_d.y, mid_d.z, kUnitPx / mid_d.y, lane, two ? top_d.z : 0, half, halves,
                      soft_rows_of(bg_soft, top_d.w | mid_d.w), hard_rows_of(bg_soft, mid_d.w), &bg_draw, &bga);  // (cap tiles are few: always worth the attempt)
#pragma unroll
        for (int k = half; k < kGrid * kGrid / 64; k += halves) {
            const int cell = k * 64 + lane;
            const int r = cell / kGrid, c = cell % kGrid;
            const int t = (c < cols && r < rows) ? Win::direct(tiles, x0 + c, y0 + r) : kEmpty;
            L.base[cell] = (t == kEmpty) ? static_cast<int32_t>(kNoTexel)
                                         : (t == kWallTop ? (top_d.x * 4) | (two ? 1 : 0) : mid_d.x * 4);
        }
        __syncthreads();
        composed = two ? compose_rows<kGrid, true>(fb, L, atlas, bga, cols, rows, mid_d.y, lane, flags, half, halves)
                       : compose_rows<kGrid, false>(fb, L, atlas, bga, cols, rows, mid_d.y, lane, flags, half, halves);
    }
    if (!composed) {  // draw-list replay (tilemap.cpp:172-198)
        wave_clear(fb, lane, half, halves);
        const bool has_bg = resolve_draw(cam, bg_d.y, bg_d.z, bg_d.x, bg_px, bg_py, bg_sc, 1.0f, false, false, mine);
        wave_replay(fb, atlas, mine, has_bg ? 1ull : 0ull, lane, half, halves);
        for (int base = 0; base < cells; base += 64) {
            const int cell = base + lane;
            bool has = false;
            if (cell < cells) {
                const int row = cell / cols;
                const int x = x0 + (cell - row * cols), y = y0 + row;
                const int t = Win::direct(tiles, x, y);
                if (t != kEmpty) {
                    const int4 d = (t == kWallTop) ? top_d : mid_d;
                    has = resolve_draw(cam, d.y, d.z, d.x, x * kUnitPx, y * kUnitPx, kUnitPx / d.y, 1.0f, false, false,
                                       mine);
                }
            }
            wave_replay(fb, atlas, mine, __ballot(has), lane, half, halves);
        }
    }

    {  // positive-z sprites (common_systems.cpp:41-63), then the agent (:272-298): one draw per lane
        int want_tex = 0;
        if (is_sprite) {
            want_tex = (spr_info & kMob) ? kTexFish + ((spr_info & kFrame) ? 1 : 0) : kTexGem;
        } else if (is_agent) {
            const bool ground = (sflags & kFlagGround) != 0;
            if (fabsf(SF(s, F_AVX, env)) < 0.01f && ground)
                want_tex = kTexStand + suit;
            else if (!ground)
                want_tex = kTexJump + suit;
            else if (SF(s, F_APHASE, env) > 0.5f)
                want_tex = kTexWalk2 + suit;
            else
                want_tex = kTexWalk1 + suit;
        }
        const int4 d = descs.at(want_tex);
        // sprites and the agent differ in their parameters only: pick per lane, resolve once (a resolve_draw per kind
        // in its own branch is executed by the whole wave once per kind)
        bool has = false, go = false, flip = false;
        float wx = 0.0f, wy = 0.0f, scale_num = kUnitPx;
        if (is_sprite) {
            if (spr_info & kTexSet) {
                const float off = (spr_info & kMob) ? -0.4f : -0.5f;  // tilemap.cpp:53,66
                const float scale = 1.0f * 1.0f;
                wx = (spr_x + off) * kUnitPx;
                wy = (spr_y + off) * kUnitPx;
                scale_num = scale * kUnitPx;
                flip = (spr_info & kFlip) != 0;
                go = true;
            }
        } else if (is_agent) {
            const float px = SF(s, F_AX, env) - 0.5f, py = SF(s, F_AY, env) - 1.0f;
            wx = px * kUnitPx;
            wy = py * kUnitPx;
            scale_num = 0.8f * kUnitPx;
            flip = (sflags & kFlagForward) == 0;
            go = true;
        }
        if (go) has = resolve_draw(cam, d.y, d.z, d.x, wx, wy, scale_num / d.y, 1.0f, flip, false, mine);
        wave_replay_rows(fb, atlas, mine, __ballot(has), lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
    }
    // each wave stores the rows it owns (pg_render.h wave_replay_rows): no barrier
    wave_store_rows(fb, io.obs + size_t(env) * kObsBytes, lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
}

// ------------------------------------------------------------------------------------------------
// The render pre-pass (pg_prepass.h; coinrun.hip's setup_kernel is the commented model): tile spans, per-pixel
// candidates, the cell table and the resolved, culled draws of kPrepEnvs envs per workgroup.
// Reference arithmetic moved here unchanged: renderer.cpp:5-82, tilemap.cpp:172-198 (the window),
// common_systems.cpp:41-63,272-298 (sprites, agent).
// ------------------------------------------------------------------------------------------------
constexpr int kPrepEnvs = 8, kPrepThreads = 256;

struct PrepEnv {
    int32_t sflags, n_draw, suit;
    float avx, aphase, ax, ay;
};
struct SetupLds {
    PrepLds<kGrid, kPrepEnvs, kMaxSpan> P;
    PrepEnv env[kPrepEnvs];
    int4 desc[kTexCount];
    uint32_t draw_order[kPrepEnvs][kEntStride / 4];  // EB_DRAW of every env: fetched before anything needs it
    uint32_t row_valid[kPrepEnvs][kGrid / 4];
    int32_t counts[kPrepEnvs];
    PrepDrawQueue queue[kPrepThreads / 64];
};

__global__ void __launch_bounds__(kPrepThreads) setup_kernel(State s, AtlasView atlas, const uint8_t* mask, int flags) {
    __shared__ SetupLds S;
    PrepLds<kGrid, kPrepEnvs, kMaxSpan>& P = S.P;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int env0 = prep_block(blockIdx.x, gridDim.x) * kPrepEnvs;  // (pg_prepass.h: the groups of one XCD are consecutive)
    const PrepOut& out = s.prep;

    // ---- one memory round trip: descriptor table, the envs' scalars (lane = env), their draw orders
    for (int q = tid; q < kPrepEnvs * 2 * 64; q += kPrepThreads) (&P.cover[0][0][0])[q] = 0u;
    if (tid < kTexCount) S.desc[tid] = atlas.desc[tid];
    static_assert(kTexCount <= kPrepThreads, "one descriptor per thread");
    if (tid < kPrepEnvs * (kEntStride / 4)) {
        const int e = tid / (kEntStride / 4), w = tid - e * (kEntStride / 4);
        if (env0 + e < s.n) S.draw_order[e][w] = reinterpret_cast<const uint32_t*>(&EB(s, EB_DRAW, 0, env0 + e))[w];
    }
    Camera cam{};
    int themes = 0;
    float bgshift = 0.0f;
    bool active = false;
    if (tid < kPrepEnvs) {
        const int e = tid, env = env0 + e;
        active = env < s.n && (!mask || mask[env]);
        if (active) {
            cam = Camera{W / 2.0f * kUnitPx, SF(s, F_CAMY, env), 64.0f, 64.0f, 0.2f * 64.0f / 64.0f};
            themes = SI(s, I_THEMES, env);
            bgshift = SF(s, F_BGSHIFT, env);
            PrepEnv pe{};
            pe.sflags = SI(s, I_FLAGS, env);
            pe.n_draw = (pe.sflags & kFlagListed) ? SI(s, I_NDRAW, env) : 0;  // empty right after a reset (D2)
            pe.suit = (themes >> 8) & 0xff;
            pe.avx = SF(s, F_AVX, env);
            pe.aphase = SF(s, F_APHASE, env);
            pe.ax = SF(s, F_AX, env);
            pe.ay = SF(s, F_AY, env);
            S.env[e] = pe;
        }
    }
    __syncthreads();
    // ---- per env (lane = env): camera, tile window, background draw — render_full's preamble
    if (tid < kPrepEnvs) {
        const int e = tid;
        PrepView v{};
        P.fat[e] = 0;
        P.soft_rows[e] = P.hard_rows[e] = 0;
        S.counts[e] = 0;
        if (active) {
            v.cam = cam;
            const int backdrop = themes & 0xff, theme = (themes >> 16) & 0xff;
            const int4 d = S.desc[kTexBackdrop + backdrop];
            const float aspect = static_cast<float>(d.y) / static_cast<float>(d.z);
            const float extra = aspect - 1.0f;
            v.bg = BgDraw{d, -bgshift * extra, 0.0f, 64.0f * kUnitPx / d.z};  // climber.cpp:447-452
            const float vx = (cam.px - cam.sw * 0.5f / cam.scale) * kPxUnit;  // tilemap.cpp:172-181
            const float vy = (cam.py - cam.sh * 0.5f / cam.scale) * kPxUnit;
            const float vw = cam.sw * kPxUnit / cam.scale, vh = cam.sh * kPxUnit / cam.scale;
            v.x0 = static_cast<int>(floorf(vx));
            v.y0 = static_cast<int>(floorf(vy));
            v.cols = static_cast<int>(ceilf(vx + vw)) - v.x0 + 1;
            v.rows = static_cast<int>(ceilf(vy + vh)) - v.y0 + 1;
            const int4 top_d = S.desc[kTexTop + theme], mid_d = S.desc[kTexMid + theme];
            const bool two = top_d.z != mid_d.z;  // the brown theme's cap tile is 64×53 next to 64×64 bodies
            v.tw = mid_d.y;
            v.th = mid_d.z;
            v.th2 = two ? top_d.z : 0;
            v.tile_scale = kUnitPx / mid_d.y;
            if (v.cols > kGrid || v.rows > kGrid || top_d.y != mid_d.y || top_d.z > mid_d.z || ((flags & kDebugFatThirds) && (env0 + e) % 3 == 0)) {
                P.fat[e] = 1;
                active = false;
            }
            P.soft_rows[e] = static_cast<uint32_t>(soft_rows_of(d.w, top_d.w | mid_d.w));
            P.hard_rows[e] = static_cast<uint32_t>(hard_rows_of(d.w, mid_d.w));  // (cap tiles are few: always worth the attempt)
            // tile kinds: 0 = cap (bit 0 of its offset: the layer's second texture), 1 = body
#pragma unroll
            for (int k = 0; k < kPrepKinds; k++) P.meta[e][PM_KINDS + k] = kNoTexel;
            P.meta[e][PM_KINDS + 0] = (static_cast<uint32_t>(top_d.x) * 4u) | (two ? 1u : 0u);
            P.meta[e][PM_KINDS + 1] = static_cast<uint32_t>(mid_d.x) * 4u;
            prep_row_valid<kGrid, H>(v.y0, S.row_valid[e]);
        }
        v.active = active ? 1 : 0;
        P.view[e] = v;
    }
    __syncthreads();

    // ---- the cell table: lane = (env, grid column), the column's 24 bytes of the column-major map (pg_prepass.h) …
    static_assert(kPrepEnvs * kGrid <= kPrepThreads, "one window column per thread");
    const int cell_e = tid / kGrid, cell_c = tid - cell_e * kGrid;
    bool cell_lane = false, cell_x_ok = false;
    uint32_t column[kGrid / 4] = {};
    if (tid < kPrepEnvs * kGrid && P.view[cell_e].active) {
        cell_lane = true;
        prep_column_fetch<kGrid, W, H>(s.tiles + size_t(env0 + cell_e) * (W * H), P.view[cell_e].x0 + cell_c, P.view[cell_e].y0, cell_x_ok, column);
    }
    // … the spans are worked out while it travels …
    prep_spans<kGrid, kMaxSpan, kPrepEnvs, true>(P, tid, kPrepThreads);
    // … then the kind bytes: wall_top → 0, wall_mid → 1, empty → none
    if (cell_lane) {
        uint32_t in_rows[kGrid / 4], kinds[kGrid / 4];
        prep_column_rows<kGrid>(column, S.row_valid[cell_e], cell_x_ok, kWallMid, in_rows);  // out of bounds is a wall (tilemap.h:66-68)
#pragma unroll
        for (int w = 0; w < kGrid / 4; w++) {
            const uint32_t t = in_rows[w] & 0x07070707u;  // kEmpty 0, kWallTop 1, kWallMid 2
            const uint32_t lo = t & 0x01010101u, hi = (t >> 1) & 0x01010101u;
            const uint32_t wall = (lo ^ hi) * 0xffu;  // 0xff where t is 1 or 2
            kinds[w] = (hi & wall) | ~wall;           // kind = t - 1 for walls; 0xff: no tile
        }
        prep_column_store<kGrid>(out.cells + size_t(env0 + cell_e) * (kGrid * kGrid), cell_c, kinds);
    }
    __syncthreads();
    prep_axes<kGrid, kMaxSpan, kPrepEnvs>(P, out, env0, wave, kPrepThreads / 64, lane);

    // ---- the draws in the reference's order: the positive-z sprites (common_systems.cpp:41-63), then the agent
    // (:272-298).  Two envs per wavefront, cull first (pg_prepass.h prep_draws_pass).
    static_assert(kPrepEnvs == 2 * (kPrepThreads / 64), "two envs per wavefront");
    {
        const int ea = 2 * wave, eb = 2 * wave + 1;
        const bool on_a = P.view[ea].active != 0, on_b = P.view[eb].active != 0;
        const Camera cam_a = P.view[ea].cam, cam_b = P.view[eb].cam;
        const int cnt_a = on_a ? S.env[ea].n_draw + 1 : 0, cnt_b = on_b ? S.env[eb].n_draw + 1 : 0;
        uint32_t* const draws_a = out.draws + size_t(env0 + ea) * kPrepDraws * kBlitWords;
        uint32_t* const draws_b = out.draws + size_t(env0 + eb) * kPrepDraws * kBlitWords;
        PrepDrawPass st{0, {0, 0}, {0, 0}};
        PrepDrawQueue& Q = S.queue[wave];
        for (int base = 0; base < cnt_a + cnt_b; base += 64) {  // wave-uniform
            const int q = base + lane;
            const bool is_b = q >= cnt_a;
            const int e = is_b ? eb : ea, env = env0 + e;
            const int slot = is_b ? q - cnt_a : q;
            const bool valid = q < cnt_a + cnt_b;
            const PrepEnv& pe = S.env[e];
            PrepDraw p{false, false, false, kTexGem, 0.0f, 0.0f, 1.0f, 1.0f};
            float num = kUnitPx;
            if (valid && slot < pe.n_draw) {
                const int ent = (S.draw_order[e][slot >> 2] >> (8 * (slot & 3))) & 0xffu;
                const int info = EB(s, EB_INFO, ent, env);
                const float ex = EF(s, EF_X, ent, env), ey = EF(s, EF_Y, ent, env);
                if (info & kTexSet) {
                    p.tex = (info & kMob) ? kTexFish + ((info & kFrame) ? 1 : 0) : kTexGem;
                    const float off = (info & kMob) ? -0.4f : -0.5f;  // tilemap.cpp:53,66
                    const float scale = 1.0f * 1.0f;
                    p.wx = (ex + off) * kUnitPx;
                    p.wy = (ey + off) * kUnitPx;
                    num = scale * kUnitPx;
                    p.flip_h = (info & kFlip) != 0;
                    p.go = true;
                }
            } else if (valid) {
                const bool ground = (pe.sflags & kFlagGround) != 0;
                if (fabsf(pe.avx) < 0.01f && ground)
                    p.tex = kTexStand + pe.suit;
                else if (!ground)
                    p.tex = kTexJump + pe.suit;
                else if (pe.aphase > 0.5f)
                    p.tex = kTexWalk2 + pe.suit;
                else
                    p.tex = kTexWalk1 + pe.suit;
                const float px = pe.ax - 0.5f, py = pe.ay - 1.0f;
                p.wx = px * kUnitPx;
                p.wy = py * kUnitPx;
                num = 0.8f * kUnitPx;
                p.flip_h = (pe.sflags & kFlagForward) == 0;
                p.go = true;
            }
            p.scale = num / S.desc[p.tex].y;
            prep_draws_pass(Q, st, S.desc, cam_a, cam_b, draws_a, draws_b, valid, is_b, p, lane);
        }
        prep_draws_flush(Q, st, S.desc, cam_a, cam_b, draws_a, draws_b, lane);
        if (lane == 0) {
            S.counts[ea] = st.done[0];
            S.counts[eb] = st.done[1];
        }
    }
    __syncthreads();
    prep_meta_out<kGrid, kMaxSpan, kPrepEnvs>(P, out, env0, S.counts, tid, kPrepThreads);
}

// render_game(true) (climber.cpp:431-459): one workgroup of two wavefronts per env; a lean frame starts from what
// setup_kernel left (coinrun.hip's render_kernel is the commented model).
__global__ void __launch_bounds__(128, 4) render_kernel(State s, AtlasView atlas, const uint8_t* mask, StepIO io,
                                                    int flags) {
    const int env = blockIdx.x;
    if (mask && !mask[env]) return;
    const int lane = threadIdx.x & 63, half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int halves = 2;
    __shared__ alignas(16) uint32_t fb[kFbWords];
    __shared__ ComposeLds<kGrid> L;
    const PrepMeta M{s.prep.meta + size_t(env) * kPrepMetaWords};
    const uint32_t colw = s.prep.axes[size_t(env) * 128 + lane], roww = s.prep.axes[size_t(env) * 128 + 64 + lane];
    const uint32_t roww2 = s.prep.axes2[size_t(env) * 64 + lane];
    const uint32_t kind_off = M.w[PM_KINDS + (lane & (kPrepKinds - 1))];
    const int n_draws = M.draws();
    const bool has = lane < n_draws;
    const Blit mine = prep_draw_load(s.prep.draws + (size_t(env) * kPrepDraws + lane) * kBlitWords, has);
    prep_cells_expand_any<kGrid>(L, s.prep.cells + size_t(env) * (kGrid * kGrid), kind_off, half, lane);
    const ComposeRegs R = prep_regs<kGrid>(M, colw, roww, roww2, lane);
    __syncthreads();  // the cell table is complete
    if ((flags & (1 | kDebugNoPrepass)) || M.fat()) {  // (wave-uniform)
        render_full(s, atlas, io, flags, env, fb, L);
        return;
    }
    const int row_lo = half * (kObsH / halves), row_hi = (half + 1) * (kObsH / halves);
    if (M.flags() & 2u)
        compose_rows_from<kGrid, true, false>(fb, L, atlas, R, lane, flags, half, halves);
    else
        compose_rows_from<kGrid, false, false>(fb, L, atlas, R, lane, flags, half, halves);
    wave_replay_rows(fb, atlas, mine, __ballot(has), lane, row_lo, row_hi);
    wave_store_rows(fb, io.obs + size_t(env) * kObsBytes, lane, row_lo, row_hi);
}

// cenv_render's frame (render_game(false)) for one env: pg_frame.h; the draw list of render_kernel, one draw at a time.
__global__ void __launch_bounds__(kFrameThreads) frame_kernel(State s, AtlasView atlas, int env, FrameTarget t) {
    const float fw = static_cast<float>(t.w), fh = static_cast<float>(t.h);
    FramePainter P{t, atlas, Camera{W / 2.0f * kUnitPx, SF(s, F_CAMY, env), fw, fh, 0.2f * fw / 64.0f},
                   static_cast<int>(threadIdx.x), kFrameThreads};
    const int themes = SI(s, I_THEMES, env), sflags = SI(s, I_FLAGS, env);
    const int backdrop = themes & 0xff, suit = (themes >> 8) & 0xff, theme = (themes >> 16) & 0xff;
    const int n_draw = (sflags & kFlagListed) ? SI(s, I_NDRAW, env) : 0;
    const uint8_t* tiles = s.tiles + size_t(env) * (W * H);
    P.clear();
    {
        const int4 d = P.desc(kTexBackdrop + backdrop);
        const float aspect = static_cast<float>(d.y) / static_cast<float>(d.z);
        const float extra = aspect - 1.0f;
        P.draw(kTexBackdrop + backdrop, -SF(s, F_BGSHIFT, env) * extra, 0.0f, 64.0f * kUnitPx / d.z);
    }
    int x0, y0, x1, y1;
    P.window(x0, y0, x1, y1);
    for (int y = y0; y <= y1; y++)
        for (int x = x0; x <= x1; x++) {
            const int tile = Win::direct(tiles, x, y);
            if (tile == kEmpty) continue;
            const int tex = (tile == kWallTop ? kTexTop : kTexMid) + theme;
            P.draw(tex, x * kUnitPx, y * kUnitPx, kUnitPx / P.desc(tex).y);
        }
    for (int k = 0; k < n_draw; k++) {
        const int e = EB(s, EB_DRAW, k, env);
        const int info = EB(s, EB_INFO, e, env);
        if (!(info & kTexSet)) continue;
        const int tex = (info & kMob) ? kTexFish + ((info & kFrame) ? 1 : 0) : kTexGem;
        const float off = (info & kMob) ? -0.4f : -0.5f;
        const float scale = 1.0f * 1.0f;
        P.draw(tex, (EF(s, EF_X, e, env) + off) * kUnitPx, (EF(s, EF_Y, e, env) + off) * kUnitPx,
               scale * kUnitPx / P.desc(tex).y, 1.0f, (info & kFlip) != 0);
    }
    {
        const bool ground = (sflags & kFlagGround) != 0;
        int tex;
        if (fabsf(SF(s, F_AVX, env)) < 0.01f && ground)
            tex = kTexStand + suit;
        else if (!ground)
            tex = kTexJump + suit;
        else if (SF(s, F_APHASE, env) > 0.5f)
            tex = kTexWalk2 + suit;
        else
            tex = kTexWalk1 + suit;
        const float px = SF(s, F_AX, env) - 0.5f, py = SF(s, F_AY, env) - 1.0f;
        P.draw(tex, px * kUnitPx, py * kUnitPx, 0.8f * kUnitPx / P.desc(tex).y, 1.0f, (sflags & kFlagForward) == 0);
    }
}

class ClimberGame final : public Game {
   public:
    const char* name() const override { return "climber"; }
    std::vector<std::string> texture_names() const override {
        std::vector<std::string> v;
        for (const char* t : {"tileBlue_05", "tileGreen_05", "tileYellow_06", "tileBrown_06", "tileBlue_08",
                              "tileGreen_08", "tileYellow_09", "tileBrown_09"})
            v.push_back(std::string("platformer/") + t + ".png");
        for (const char* pose : {"_stand", "_walk4", "_walk1", "_walk2"})  // jump = walk4 (common_systems.cpp:178)
            for (const char* suit : {"Blue", "Green", "Grey", "Red"})
                v.push_back(std::string("platformer/player") + suit + pose + ".png");
        v.push_back("platformer/enemySwimming_1.png");
        v.push_back("platformer/enemySwimming_2.png");
        v.push_back("misc_assets/yellowCrystal.png");
        for (const char* b : {"platform_backgrounds/alien_bg", "platform_backgrounds/another_world_bg",
                              "platform_backgrounds_2/fantasy1", "platform_backgrounds_2/fantasy2",
                              "platform_backgrounds_2/fantasy3", "platform_backgrounds_2/fantasy4",
                              "platform_backgrounds_2/candy1", "platform_backgrounds_2/candy2",
                              "platform_backgrounds_2/candy3", "platform_backgrounds_2/candy4"})
            v.push_back(std::string(b) + ".png");
        return v;
    }
    std::string check_atlas(const std::vector<std::pair<int, int>>& sizes) const override {
        return static_cast<int>(sizes.size()) == kTexCount ? "" : "climber: unexpected texture count";
    }
    static size_t align256(size_t x) { return (x + 255) & ~size_t(255); }
    struct Layout {
        size_t shadow, slot, mt, tiles, f, i, ef, eb, total;
    };
    static Layout layout(int n) {
        Layout l{};
        size_t off = 0;
        auto take = [&](size_t bytes) {
            size_t at = off;
            off += align256(bytes);
            return at;
        };
        l.shadow = take(size_t(n) * sizeof(Level));
        l.slot = take(size_t(n) * 4);
        l.mt = take(size_t(n) * kMtWords * 4);
        l.tiles = take(size_t(n) * W * H);
        l.f = take(size_t(F_COUNT) * n * 4);
        l.i = take(size_t(I_COUNT) * n * 4);
        l.ef = take(size_t(EF_COUNT) * kEntStride * n * 4);
        l.eb = take(size_t(EB_COUNT) * kEntStride * n);
        l.total = off;
        return l;
    }
    size_t state_bytes(int n) const override { return layout(n).total; }
    void bind(void* d_state, int n, AtlasView atlas) override {
        uint8_t* p = static_cast<uint8_t*>(d_state);
        const Layout l = layout(n);
        s_.n = n;
        s_.shadow = reinterpret_cast<Level*>(p + l.shadow);
        s_.slot = reinterpret_cast<int32_t*>(p + l.slot);
        s_.mt = reinterpret_cast<uint32_t*>(p + l.mt);
        s_.tiles = p + l.tiles;
        s_.f = reinterpret_cast<float*>(p + l.f);
        s_.i = reinterpret_cast<int32_t*>(p + l.i);
        s_.ef = reinterpret_cast<float*>(p + l.ef);
        s_.eb = p + l.eb;
        s_.ranks = atlas.sort_ranks;
        atlas_ = atlas;
    }
    int blocks() const { return (s_.n + 63) / 64; }
    void launch_make(hipStream_t st, uint32_t seed_base, int env_offset) override {
        hipLaunchKernelGGL(make_kernel, dim3(blocks()), dim3(64), 0, st, s_, seed_base, env_offset);
        LevelLaunch<Gen>::make(st, s_, prefetch(), seed_base, env_offset, plan);
    }
    void launch_reset(hipStream_t st, const uint8_t* mask, const int32_t* seeds, StepIO io) override {
        LevelLaunch<Gen>::reset(st, s_, prefetch(), mask, seeds, io, plan);
    }
    bool launch_pregen(hipStream_t side, bool bulk) override {
        if (!prefetch()) return false;
        LevelLaunch<Gen>::pregen(side, s_, bulk, plan);
        return true;
    }
    int prefetch() const { return (debug_flags & kDebugNoPrefetch) ? 0 : 1; }
    void launch_logic(hipStream_t st, const int32_t* actions, uint32_t run_seed, uint32_t step_index, int env_offset,
                      StepIO io) override {
        // prefetched levels are installed beside the logic (its second row of blocks); the level kernel behind it
        // generates, synchronously, the levels that were not ready — none in steady state (pg_prefetch.h install_prefetched)
        const bool fused = prefetch() != 0;
        if (!fused) LevelLaunch<Gen>::auto_reset(st, s_, prefetch(), io, plan, PG_RESET_SPAN, reset_served_mark(step_index), reset_due_mark(step_index));
        hipLaunchKernelGGL(logic_kernel, dim3((s_.n * kGang + 63) / 64, fused ? 2 : 1), dim3(64), 0, st, s_, actions, run_seed, step_index,
                           env_offset, io, prefetch(), plan);
        if (fused) LevelLaunch<Gen>::auto_reset(st, s_, prefetch(), io, plan, PG_RESET_SPAN, reset_served_mark(step_index), reset_due_mark(step_index));
    }
    bool launch_frame(hipStream_t st, int env, uint32_t* d_px, int w, int h) override {
        hipLaunchKernelGGL(frame_kernel, dim3(1), dim3(kFrameThreads), 0, st, s_, atlas_, env, FrameTarget{d_px, w, h});
        return true;
    }
    size_t scratch_bytes(int n) const override { return prep_bytes(n, kGrid, kBlitWords, true); }
    void bind_scratch(void* d_scratch, int n) override { s_.prep = prep_bind(d_scratch, n, kGrid, kBlitWords, true); }
    void launch_prepass(hipStream_t st, const uint8_t* mask) override {
        if (!(debug_flags & (1 | kDebugNoPrepass)))
            hipLaunchKernelGGL(setup_kernel, dim3((s_.n + kPrepEnvs - 1) / kPrepEnvs), dim3(kPrepThreads), 0, st, s_, atlas_, mask, debug_flags);
    }
    void launch_render(hipStream_t st, const uint8_t* mask, StepIO io) override {
        hipLaunchKernelGGL(render_kernel, dim3(s_.n), dim3(128), 0, st, s_, atlas_, mask, io, debug_flags);
    }
    // Same layout as oracle/pgo_climber.cpp Climber::dump_state.
    int dump_state(hipStream_t st, int env, float* out, int cap) override {
        hipStreamSynchronize(st);
        const size_t n = s_.n;
        auto rf = [&](const float* base, size_t idx) {
            float v;
            hipMemcpy(&v, base + idx, 4, hipMemcpyDeviceToHost);
            return v;
        };
        auto ri = [&](int field) {
            int32_t v;
            hipMemcpy(&v, s_.i + size_t(field) * n + env, 4, hipMemcpyDeviceToHost);
            return v;
        };
        auto f = [&](int field) { return rf(s_.f, size_t(field) * n + env); };
        const int flags = ri(I_FLAGS), themes = ri(I_THEMES), n_ent = ri(I_NENT);
        std::vector<float> v = {f(F_AX), f(F_AY), f(F_AVX), f(F_AVY), (flags & kFlagGround) ? 1.0f : 0.0f,
                                (flags & kFlagForward) ? 1.0f : 0.0f, f(F_APHASE), W / 2.0f * kUnitPx, f(F_CAMY),
                                static_cast<float>(themes & 0xff), f(F_BGSHIFT),
                                static_cast<float>((themes >> 8) & 0xff), static_cast<float>((themes >> 16) & 0xff),
                                static_cast<float>(n_ent)};
        for (int e = 0; e < n_ent; e++) {
            uint8_t info;
            hipMemcpy(&info, s_.eb + (size_t(env) * EB_COUNT + EB_INFO) * kEntStride + e, 1, hipMemcpyDeviceToHost);
            auto ef = [&](int field) { return rf(s_.ef, (size_t(env) * EF_COUNT + field) * kEntStride + e); };
            v.push_back((info & kAlive) ? 1.0f : 0.0f);
            v.push_back(ef(EF_X));
            v.push_back(ef(EF_Y));
            v.push_back((info & kMob) ? ef(EF_VX) : 0.0f);
            v.push_back((info & kFrame) ? 1.0f : 0.0f);
            v.push_back(ef(EF_ANIM_T));
        }
        const int m = cap < static_cast<int>(v.size()) ? cap : static_cast<int>(v.size());
        for (int k = 0; k < m; k++) out[k] = v[k];
        return static_cast<int>(v.size());
    }
    int dump_tiles(hipStream_t st, int env, uint8_t* out, int cap) override {
        hipStreamSynchronize(st);
        const int m = cap < W * H ? cap : W * H;
        hipMemcpy(out, s_.tiles + size_t(env) * W * H, m, hipMemcpyDeviceToHost);
        return m;
    }

   private:
    State s_{};
    AtlasView atlas_{};
};

}  // namespace climber

}  // namespace PG_VARIANT_NS

std::unique_ptr<Game> PG_FACTORY(make_climber)() { return std::make_unique<PG_VARIANT_NS::climber::ClimberGame>(); }

}  // namespace pg
