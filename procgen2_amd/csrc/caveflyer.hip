// caveflyer on gfx950 (BASELINE.json configs[2]): free-flight ship with rotation
// and bullets in a cellular-automaton cave.
//
// Reference:
//   step   games/caveflyer/caveflyer.cpp:301-357, common_systems.cpp:90-289 (agent + bullets), :50-75 (enemies),
//          :333-372 (exhaust particles), :7-24 (sprite list)
//   render games/caveflyer/caveflyer.cpp:413-440, tilemap.cpp:280-303, common_systems.cpp:26-48, :291-327, :374-397
//   reset  games/caveflyer/caveflyer.cpp:442-460, tilemap.cpp:118-278, room_generator.cpp:4-202
// Config = the reference's compile-time default, hard_mode (40×40, pruned; caveflyer/tilemap.h:43-45).
//
// Machine mapping:
//   * logic  — a gang of kGang adjacent lanes per env (pg_gang.h): the ship's scalars uniformly in all of them, the
//              ≤60 objects dealt out over the lanes' registers, bullets and particles one slot per lane;
//   * render — two wavefronts per env sharing an LDS target (pg_render.h), tiles through the row composer, the rotated sprites
//              (particles, bullets, ship) as whole-wave rotated blits;
//   * level generation — one wavefront per env with a 23 KiB LDS workspace.  The generator's result depends on
//              the iteration order of a std::unordered_set<int> holding up to 1600 cells (the largest room,
//              tilemap.cpp:158-169), so the libstdc++ hashtable is replayed (pg_order.h) — inherently serial, hence
//              on LDS rather than HBM: the dependent-access latency is what bounds it (≈2 ms for one env, ≈1 µs per
//              env in bulk).  The game draws no random numbers during an episode, so the NEXT level of every env is
//              generated ahead of time on a side stream into a shadow slot and merely copied in when the episode
//              ends (pg_prefetch.h); the synchronous path remains as the fallback and for reseeding resets.
#include "pg_engine.h"
#include "pg_frame.h"
#include "pg_gang.h"
#include "pg_geom.h"
#include "pg_order.h"
#include "pg_prefetch.h"
#include "pg_render.h"
#include "pg_rng.h"
#include "pg_defs.h"
// caveflyer/tilemap.cpp: world_dim by Distribution_Mode — hard_mode 40 (the reference's compile-time default, tilemap.h),
// easy_mode 20, memory_mode 45 (the whole cave stays, not just the widened goal path).
#if PG_VARIANT == 0
#define PG_ROOMS_DIM 40
#elif PG_VARIANT == 1
#define PG_ROOMS_DIM 20
#elif PG_VARIANT == 2
#define PG_ROOMS_DIM 45
#else
#error "caveflyer: unknown PG_VARIANT"
#endif
#include "pg_rooms.h"
#include "pg_prepass.h"
#include "pg_sincos.h"
#include "pg_tiles.h"

namespace pg {
namespace PG_VARIANT_NS {
namespace caveflyer {

constexpr int W = rooms::W, H = rooms::H, kCells = W * H;
constexpr int kTileStride = (kCells + 3) / 4 * 4;  // tiles are copied as 32-bit words
constexpr bool kPrune = PG_VARIANT != 2;              // tilemap.cpp:203 should_prune = mode != memory_mode
// ids: 0 goal, 1 ship, 2.. objects; 3·(free/80) objects (tilemap.cpp:232-233): ≤ 60 at 40×40, ≤ 75 at 45×45
constexpr int kMaxEnt = 2 + 3 * ((kCells > 1600 ? kCells : 1600) / 80) + (kCells > 1600 ? 1 : 0);
static_assert(kMaxEnt <= 128, "two wave passes cover the sprites");
constexpr int kShots = 32, kPuffs = 10, kPuffSlots = 16;
constexpr double kPi = 3.14159265358979323846;  // M_PI
enum Tile : uint8_t { kEmpty = 0, kWall = 1 };
enum Kind { kMeteor = 0, kTarget = 1, kEnemy = 2, kGoal = 3 };

enum Tex {
    kTexWall = 0,
    kTexKind = 1,  // meteor, target, enemy, goal
    kTexShip = 5,
    kTexLaser = 6,
    kTexBoom = 7,  // 5
    kTexPuff = 12,
    kTexSpace = 13,  // 13
    kTexCount = 26
};

enum {
    F_AX, F_AY, F_AVX, F_AVY, F_ROT, F_CAMX, F_CAMY, F_BGSHIFT, F_STIMER, F_PTIMER,
    F_SHIP_SN, F_SHIP_CS,  // the ship's drawing angle as raster spec S6 takes it (pg_render.h rotation_of), int bits
    F_COUNT
};
enum { I_FLAGS, I_BACKDROP, I_NENT, I_NDRAW, I_SNEXT, I_SCOUNT, I_HASH_SPRITE, I_HASH_HAZARD, I_COUNT };
constexpr int kFlagListed = 1, kFlagPuffOn = 2;
enum { EF_X, EF_Y, EF_VX, EF_VY, EF_COUNT };
enum { EB_INFO, EB_ORDER_S, EB_ORDER_H, EB_DRAW, EB_PLACE_H, EB_COUNT };  // EB_PLACE_H[e]: where e stands in EB_ORDER_H
constexpr int kEntStride = (kMaxEnt + 15) / 16 * 16;
constexpr int kKindMask = 3, kAlive = 4;
enum { SH_X, SH_Y, SH_VX, SH_VY, SH_ROT, SH_FRAME, SH_SN, SH_CS, SH_COUNT };  // SH_SN/CS: rotation_of the drawing angle, fixed when fired
enum { PF_X, PF_Y, PF_DX, PF_DY, PF_ROT, PF_LIFE, PF_SN, PF_CS, PF_COUNT };  // likewise

// One generated level, as the generator leaves it in LDS and as it waits in the shadow slot.
struct Level {
    uint8_t tiles[kTileStride];
    float ax, ay, bgshift;
    int32_t n_ent, backdrop;
    float ex[kMaxEnt], ey[kMaxEnt], evx[kMaxEnt], evy[kMaxEnt];
    uint8_t info[kMaxEnt], order_s[kMaxEnt], order_h[kMaxEnt];
    uint8_t pad[2];
};
static_assert(sizeof(Level) % 4 == 0, "Level is copied as 32-bit words");

struct State {
    int n;
    Level* shadow;   // [n]  next level of each env (pg_prefetch.h)
    int32_t* slot;   // [n]  SlotState
    uint32_t* mt;    // [n][625]  generator chain: the stream position after the newest generated level
    uint8_t* tiles;  // [n][kCells]  column-major y + x*H
    uint64_t* cols;  // [n][W]  the same map as wall bits, one word per column (pg_tiles.h BitWinT): what the logic kernel reads
    float* f;        // [F_COUNT][n]
    int32_t* i;      // [I_COUNT][n]
    // per-env contiguous tables ([env][field][slot]): the lanes of a gang and of the render wavefronts index them by slot
    float* ef;       // [n][EF_COUNT][kEntStride]
    uint8_t* eb;     // [n][EB_COUNT][kEntStride]
    float* sh;       // [n][SH_COUNT][kShots]
    float* pf;       // [n][PF_COUNT][kPuffSlots]
    const uint8_t* ranks;  // pg_order.h equal-key sort ranks
    PrepOut prep;          // what setup_kernel leaves for render_kernel (pg_prepass.h); not part of the state blob
};

PG_D float& SF(const State& s, int field, int env) { return s.f[size_t(field) * s.n + env]; }
PG_D int32_t& SI(const State& s, int field, int env) { return s.i[size_t(field) * s.n + env]; }
PG_D float& EF(const State& s, int field, int e, int env) { return s.ef[(size_t(env) * EF_COUNT + field) * kEntStride + e]; }
PG_D uint8_t& EB(const State& s, int field, int e, int env) { return s.eb[(size_t(env) * EB_COUNT + field) * kEntStride + e]; }
// rotation_of(angle) into a pair of float slots (as int bits): where the angle is set, so the render kernel reads two
// words instead of evaluating sinf and cosf in every lane of both wavefronts, every frame.
PG_D void store_rotation(float& sn_slot, float& cs_slot, float angle) {
    int sn, cs;
    rotation_of(angle, sn, cs);
    sn_slot = __int_as_float(sn);
    cs_slot = __int_as_float(cs);
}
PG_D float& SH(const State& s, int field, int k, int env) { return s.sh[(size_t(env) * SH_COUNT + field) * kShots + k]; }
PG_D float& PF(const State& s, int field, int k, int env) { return s.pf[(size_t(env) * PF_COUNT + field) * kPuffSlots + k]; }

using Win = TileWinT<W, H, kWall>;  // out of bounds is a wall (tilemap.h:78-83)
using BitWin = BitWinT<W, H, kWall, kEmpty>;
PG_D bool is_wall(int t) { return t == kWall; }

// ------------------------------------------------------------------------------------------------
// level generation (one wavefront per env; the serial parts run on lane 0)
// ------------------------------------------------------------------------------------------------
using rooms::RoomsLds;
using GenLds = rooms::RoomsLds;
using rooms::automaton;
using rooms::best_room;
using rooms::cell_of;
using rooms::episode_order;
using rooms::goal_path;

PG_D void put_thing(Level& lv, int id, int kind, int cell, float vx, float vy) {
    const int x = cell / H, y = cell % H;
    lv.ex[id] = static_cast<float>(x) + 0.5f;
    lv.ey[id] = static_cast<float>(H - 1 - y) + 0.5f;
    lv.evx[id] = vx;
    lv.evy[id] = vy;
    lv.info[id] = static_cast<uint8_t>(kind | kAlive);
}

PG_D int check_neighbors(float x0, float y0, float x1, float y1) {  // tilemap.cpp:103-115
    const float neighborhood = 2.0f, epsilon = 0.001f;
    if (fabsf(x0 - x1) <= epsilon && fabsf(y0 - y1) <= neighborhood) return 1;
    if (fabsf(x0 - x1) <= neighborhood && fabsf(y0 - y1) <= epsilon) return 2;
    return 0;
}

// reset() (caveflyer.cpp:442-460) for one env by one wavefront: advances the env's generator chain (s.mt, the two
// bucket-count words) and leaves the level in `lv` (LDS).
PG_D void generate(const State& s, int env, GenLds& L, Level& lv, bool reseed, uint32_t seed, int lane) {
    uint32_t* gmt = s.mt + size_t(env) * kMtWords;
    if (reseed) {
        if (lane == 0) mt_seed(L.mt, seed);
    } else {
        for (int k = lane; k < kMtWords; k += 64) L.mt[k] = gmt[k];
    }
    __syncthreads();
    uint32_t* mt = L.mt;
    wave_coin_flips(mt, L.grid, kCells, lane);  // tilemap.cpp:139-140: grid[i] = dist01(rng) < 0.5f
    automaton(L.grid, L.aux, lane);
    __syncthreads();
    automaton(L.aux, L.grid, lane);
    __syncthreads();
    {
        const int n_free = best_room(L, lane);
        if (lane == 0) {  // tilemap.cpp:163-172
            const int goal_index = rng_int(mt, 0, n_free - 1);
            int agent_index = rng_int(mt, 0, n_free - 1);
            if (agent_index == goal_index) agent_index = (agent_index + 1) % n_free;
            L.goal_cell = L.cells[goal_index];
            L.agent_cell = L.cells[agent_index];
        }
        __syncthreads();
    }
    goal_path(L, L.agent_cell, L.goal_cell, lane);
    if (kPrune) {
        rooms::widen(L, lane);
    } else {  // memory_mode: every open cell of the cave stays; same encoding as widen(): 0 wall, 1 path, ≥ 2 free
        for (int c = lane; c < kCells; c += 64) L.aux[c] = L.grid[c] ? 0 : 2;
        __syncthreads();
        for (int k = lane; k < L.path_len; k += 64) L.aux[L.cells[k]] = 1;
        __syncthreads();
    }
    if (lane == 0) {
        // the four further automaton iterations (tilemap.cpp:217-222) never reach tile_ids (D13)

        // goal (id 0), ship (id 1)
        const int goal_cell = L.goal_cell, agent_cell = L.agent_cell;
        put_thing(lv, 0, kGoal, goal_cell, 0.0f, 0.0f);
        const float ax = static_cast<float>(agent_cell / H) + 0.5f;
        const float ay = static_cast<float>(H - 1 - (agent_cell % H));  // no +0.5 (tilemap.cpp:189)
        lv.ax = ax;
        lv.ay = ay;
        lv.info[1] = 0;

    }
    // objects on the open cells off the path, in index order (tilemap.cpp:224-272).  The whole wave walks the loop:
    // lane 0 owns the random stream, lane j remembers the j-th picked index so that "already taken?" is one ballot.
    int n_free = 0;
    for (int c0 = 0; c0 < kCells; c0 += 64) {
        const bool open = c0 + lane < kCells && L.aux[c0 + lane] >= 2;
        const unsigned long long m = __ballot(open);
        if (open) L.cells[n_free + __popcll(m & ((1ull << lane) - 1ull))] = static_cast<int16_t>(c0 + lane);
        n_free += __popcll(m);
    }
    __syncthreads();
    const int chunk = n_free / 80, num_objects = 3 * chunk;
    {
        const float ax = lv.ax, ay = lv.ay;
        int picked = -1, picked_hi = -1;  // lane j: the j-th and the (64 + j)-th picked index
        for (int i = 0; i < num_objects; i++) {
            int index = 0;
            if (lane == 0) index = rng_int(mt, 0, n_free - 1);
            index = __shfl(index, 0);
            while (__ballot(picked == index || picked_hi == index)) index = (index + 1) % n_free;
            if (lane == i) picked = index;
            if (lane + 64 == i) picked_hi = index;
            if (lane == 0) {
                const int cell = L.cells[index];
                const int id = 2 + i;
                if (i < chunk)
                    put_thing(lv, id, kMeteor, cell, 0.0f, 0.0f);
                else if (i < 2 * chunk)
                    put_thing(lv, id, kTarget, cell, 0.0f, 0.0f);
                else {  // spawn_enemy (tilemap.cpp:68-101)
                    const float magnitude = 0.1f * rng_real(mt, 0.0f, 1.0f) + 0.1f;
                    const float vel = magnitude * (rng_real(mt, 0.0f, 1.0f) < 0.5f ? 1.0f : -1.0f);
                    const float ex = static_cast<float>(cell / H) + 0.5f;
                    const float ey = static_cast<float>(H - 1 - cell % H) + 0.5f;
                    const int clash = check_neighbors(ex, ey, ax, ay);
                    bool along_x;
                    if (clash == 0)
                        along_x = rng_real(mt, 0.0f, 1.0f) < 0.5f;
                    else
                        along_x = clash == 1;
                    put_thing(lv, id, kEnemy, cell, along_x ? vel : 0.0f, along_x ? 0.0f : vel);
                }
            }
        }
    }
    if (lane == 0) {
        const int n_ent = 2 + num_objects;
        lv.n_ent = n_ent;
        lv.backdrop = rng_int(mt, 0, 12);
        lv.bgshift = rng_real(mt, 0.0f, 1.0f);

        // entity-set orders of the episode: sprites = goal + objects, hazards = objects
        uint8_t keys[kMaxEnt];
        int nk = 0;
        keys[nk++] = 0;
        for (int id = 2; id < n_ent; id++) keys[nk++] = static_cast<uint8_t>(id);
        int32_t packed = SI(s, I_HASH_SPRITE, env);
        episode_order(packed, keys, nk, lv.order_s, L.queue, L.before);
        SI(s, I_HASH_SPRITE, env) = packed;
        packed = SI(s, I_HASH_HAZARD, env);
        episode_order(packed, keys + 1, nk - 1, lv.order_h, L.queue, L.before);
        SI(s, I_HASH_HAZARD, env) = packed;
    }
    __syncthreads();
    for (int c = lane; c < kTileStride; c += 64) lv.tiles[c] = (c < kCells && L.aux[c]) ? kEmpty : kWall;
    for (int k = lane; k < kMtWords; k += 64) gmt[k] = L.mt[k];
    __syncthreads();
}

// The level becomes the env's live state (everything reset() and the component constructors initialise).
PG_D void install(const State& s, int env, const Level& lv, int lane) {
    uint32_t* tiles = reinterpret_cast<uint32_t*>(s.tiles + size_t(env) * kTileStride);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(lv.tiles);
    for (int k = lane; k < kTileStride / 4; k += 64) tiles[k] = src[k];
    for (int x = lane; x < W; x += 64) {  // collide() looks at tile (x, H-1-y): bit y of the column word
        uint64_t word = ~0ull << H;
        for (int y = 0; y < H; y++) word |= static_cast<uint64_t>(lv.tiles[(H - 1 - y) + x * H] == kWall ? 1 : 0) << y;
        s.cols[size_t(env) * W + x] = word;
    }
    const int n_ent = lv.n_ent;
    for (int e = lane; e < n_ent; e += 64) {
        EF(s, EF_X, e, env) = lv.ex[e];
        EF(s, EF_Y, e, env) = lv.ey[e];
        EF(s, EF_VX, e, env) = lv.evx[e];
        EF(s, EF_VY, e, env) = lv.evy[e];
        EB(s, EB_INFO, e, env) = lv.info[e];
        if (e < n_ent - 1) EB(s, EB_ORDER_S, e, env) = lv.order_s[e];
        if (e < n_ent - 2) {
            EB(s, EB_ORDER_H, e, env) = lv.order_h[e];
            EB(s, EB_PLACE_H, lv.order_h[e], env) = static_cast<uint8_t>(e);
        }
    }
    if (lane < kPuffs)
        for (int f = 0; f < PF_COUNT; f++) PF(s, f, lane, env) = 0.0f;
    if (lane == 0) {
        SF(s, F_AX, env) = lv.ax;
        SF(s, F_AY, env) = lv.ay;
        SF(s, F_AVX, env) = 0.0f;
        SF(s, F_AVY, env) = 0.0f;
        SF(s, F_ROT, env) = 0.0f;
        store_rotation(SF(s, F_SHIP_SN, env), SF(s, F_SHIP_CS, env), static_cast<float>(0.0f + kPi * 0.5f));
        SF(s, F_BGSHIFT, env) = lv.bgshift;
        SF(s, F_STIMER, env) = 0.0f;
        SF(s, F_PTIMER, env) = 0.0f;
        SI(s, I_NENT, env) = n_ent;
        SI(s, I_BACKDROP, env) = lv.backdrop;
        SI(s, I_FLAGS, env) = kFlagPuffOn;  // draw list cleared; Component_Particles::enabled = true
        SI(s, I_NDRAW, env) = 0;
        SI(s, I_SNEXT, env) = 0;  // System_Agent::reset (common_systems.h:83-87); the 32 slots keep their contents
        SI(s, I_SCOUNT, env) = 0;
        // camera keeps the previous episode's value (D3)
    }
}


// ------------------------------------------------------------------------------------------------
// step
// ------------------------------------------------------------------------------------------------
// One env = one gang (pg_gang.h).  The ship's scalars are uniform over the gang.  The env's entity table (positions, info,
// place in the hazard order, the enemies' velocities) and its map (one word of wall bits per column) are staged in LDS
// for the four sub-steps; the hazard test of the ship, the first-hit search of a flying bullet and the enemies' own
// moves all run kGang entities at a time — entity e is always handled by lane e mod kGang — and what changed goes back
// at the end.  Bullet ring slot k and particle k belong to lane k mod kGang.
#ifndef PG_CAVEFLYER_GANG
#define PG_CAVEFLYER_GANG 8
#endif
#ifndef PG_CAVEFLYER_WAVES
#define PG_CAVEFLYER_WAVES 4  // wavefronts per SIMD the logic kernel's registers are capped for
#endif
constexpr int kGang = PG_CAVEFLYER_GANG;
constexpr int kMaxEnemies = (kMaxEnt - 2) / 3 + 1;
struct StepLds {  // one per gang
    uint64_t cols[W];
    float x[kEntStride], y[kEntStride];
    float vx[kMaxEnemies], vy[kMaxEnemies];  // of enemy first_enemy + j
    uint8_t info[kEntStride], place[kEntStride];
};
constexpr int kPuffsPerLane = (kPuffs + kGang - 1) / kGang;
using Q = Gang<kGang>;
static_assert(kGang >= 8, "six lanes share a sub-step's trigonometry");

PG_D bool hit(const Box& a, const Box& b) {  // box_hit without short circuits
    return (a.x < b.x + b.w) & (a.x + a.w > b.x) & (a.y < b.y + b.h) & (a.y + a.h > b.y);
}
PG_D Box thing_box(float x, float y, int kind) {
    const bool big = (kind == kEnemy) | (kind == kGoal);
    const float half = big ? -0.4f : -0.25f, size = big ? 0.8f : 0.5f;
    return Box{x + half, y + half, size, size};
}

// System_Sprite_Render::update's list: the surviving sprites in set order, then std::sort on z (all 1.0).
PG_D void rebuild_draw_list(const State& s, int env, int n_ent) {
    int n = 0;
    for (int k = 0; k < n_ent - 1; k++) n += (EB(s, EB_INFO, EB(s, EB_ORDER_S, k, env), env) & kAlive) ? 1 : 0;
    const uint8_t* rank = s.ranks + rank_offset(n);  // equal keys: the sort is a fixed permutation for each n
    int r = 0;
    for (int k = 0; k < n_ent - 1; k++) {
        const int e = EB(s, EB_ORDER_S, k, env);
        if (EB(s, EB_INFO, e, env) & kAlive) EB(s, EB_DRAW, rank[r++], env) = static_cast<uint8_t>(e);
    }
    SI(s, I_NDRAW, env) = n;
}

PG_D void advance(const State& s, StepLds& L, Q q, int env, int action, float& reward_out, bool& terminated_out) {
    const int n_ent = SI(s, I_NENT, env);
    const int first_enemy = 2 + 2 * ((n_ent - 2) / 3);  // meteors, targets, enemies: a third of the objects each
    const uint64_t* cols = L.cols;
    for (int x = q.g; x < W; x += kGang) L.cols[x] = s.cols[size_t(env) * W + x];
    for (int e = q.g; e < n_ent; e += kGang) {  // (the goal and the ship are not in the hazard set: info 0)
        L.info[e] = e >= 2 ? EB(s, EB_INFO, e, env) : 0;
        L.place[e] = e >= 2 ? EB(s, EB_PLACE_H, e, env) : 0;
        L.x[e] = EF(s, EF_X, e, env);
        L.y[e] = EF(s, EF_Y, e, env);
        if (e >= first_enemy) {
            L.vx[e - first_enemy] = EF(s, EF_VX, e, env);
            L.vy[e - first_enemy] = EF(s, EF_VY, e, env);
        }
    }
    wave_order();
    const float goal_x = L.x[0], goal_y = L.y[0];
    float puff_life[kPuffsPerLane];
#pragma unroll
    for (int p = 0; p < kPuffsPerLane; p++) {
        const int k = q.g + kGang * p;
        puff_life[p] = k < kPuffs ? PF(s, PF_LIFE, k, env) : 1.0f;
    }
    const int flags = SI(s, I_FLAGS, env);
    float ax = SF(s, F_AX, env), ay = SF(s, F_AY, env), avx = SF(s, F_AVX, env), avy = SF(s, F_AVY, env);
    float rot = SF(s, F_ROT, env), s_timer = SF(s, F_STIMER, env), p_timer = SF(s, F_PTIMER, env);
    int s_next = SI(s, I_SNEXT, env), s_count = SI(s, I_SCOUNT, env);
    bool puff_on = (flags & kFlagPuffOn) != 0;
    bool set_changed = (flags & kFlagListed) == 0;

    const float dt = 1.0f / 4;
    const float accel = 0.05f, spin_rate = 0.05f, vel_decay = 0.1f, reverse_mul = 0.5f, bullet_time = 0.5f,
                bullet_speed = 1.0f, explosion_rate = 0.5f;
    const float movement_x = static_cast<float>((action == 6 || action == 7 || action == 8) -
                                                (action == 0 || action == 1 || action == 2));
    float movement_y = static_cast<float>((action == 2 || action == 5 || action == 8) -
                                          (action == 0 || action == 3 || action == 6));
    const bool fire = action == 9;
    if (movement_y < 0.0f) movement_y *= reverse_mul;

    float reward = 0.0f;
    bool terminated = false;
    int ship_sn = 0, ship_cs = 0;
    for (int ss = 0; ss < 4; ss++) {
        // --- System_Agent::update (common_systems.cpp:90-289)
        bool alive = true, achieved_goal = false;
        int targets_destroyed = 0;
        rot += movement_x * spin_rate * dt;
        // The sub-step's trigonometry, one value per lane instead of every value in every lane: cos / sin of the ship's
        // rotation, cos / sin of rotation + π/2 (the exhaust puff), and that angle as the raster takes it (rotation_of: a
        // fired bullet's, a puff's and — after the last sub-step — the ship's drawing angle are all this one).
        const float prot = static_cast<float>(rot + kPi * 0.5f);
        float dir_x, dir_y, puff_c, puff_s;
        int prot_sn, prot_cs;
        {
            const double deg = prot * 180.0f / 3.14159265358979323846;  // rotation_of / rotation_16_16 (pg_render.h)
            const float theta = static_cast<float>(deg * (3.14159265358979323846 / 180.0));
            const int role = q.g >> 1;
            const float mine = sc_trig(role == 0 ? rot : (role == 1 ? prot : theta), (q.g & 1) == 0);
            dir_x = __shfl(mine, 0, kGang);
            dir_y = __shfl(mine, 1, kGang);
            puff_c = __shfl(mine, 2, kGang);
            puff_s = __shfl(mine, 3, kGang);
            const int fixed = static_cast<int>(floor(static_cast<double>(mine) * 65536.0 + 0.5));
            prot_cs = deg != 0.0 ? __shfl(fixed, 4, kGang) : 0;
            prot_sn = deg != 0.0 ? __shfl(fixed, 5, kGang) : 0;
        }
        ship_sn = prot_sn;
        ship_cs = prot_cs;
        if (fire) {
            if (s_timer == 0.0f && s_count < kShots) {
                s_timer = bullet_time;
                if ((s_next & (kGang - 1)) == q.g) {
                    SH(s, SH_ROT, s_next, env) = rot;
                    SH(s, SH_SN, s_next, env) = __int_as_float(prot_sn);
                    SH(s, SH_CS, s_next, env) = __int_as_float(prot_cs);
                    SH(s, SH_VX, s_next, env) = dir_x * bullet_speed;
                    SH(s, SH_VY, s_next, env) = dir_y * bullet_speed;
                    SH(s, SH_X, s_next, env) = ax;
                    SH(s, SH_Y, s_next, env) = ay;
                    SH(s, SH_FRAME, s_next, env) = 0.0f;
                }
                s_next = (s_next + 1) % kShots;
                s_count++;
            } else {
                s_timer = fmaxf(0.0f, s_timer - dt);
            }
        }
        const float acc_x = dir_x * movement_y * accel, acc_y = dir_y * movement_y * accel;
        avx += (acc_x - avx * vel_decay) * dt;
        avy += (acc_y - avy * vel_decay) * dt;
        ax += avx * dt;
        ay += avy * dt;
        Box body{ax + -0.4f, ay + -0.4f, 0.8f, 0.8f};
        {
            const BitWin win = BitWin::fetch(cols, static_cast<int>(floorf(body.x)), static_cast<int>(floorf(body.y)));
            // (the walk's nine cells side by side over the gang's lanes, pg_tiles.h: logic kernel 101.5 -> 88 µs same-box)
            static_assert(kGang == 8, "collide_plain_gang8");
            const TileHit h = collide_plain_gang8(q, win, body, is_wall);
            const float moved_x = h.x - body.x, moved_y = h.y - body.y;
            ax = h.x - -0.4f;
            ay = h.y - -0.4f;
            body.x = ax + -0.4f;
            body.y = ay + -0.4f;
            if (moved_x != 0.0f) avx = 0.0f;
            if (moved_y != 0.0f) avy = 0.0f;
        }
        {   // hazards: any hit kills, order-free
            bool crash = false;
            for (int e0 = 0; e0 < n_ent; e0 += kGang) {
                const int e = e0 + q.g;
                const int info = e < n_ent ? L.info[e] : 0;
                const float x = e < n_ent ? L.x[e] : 0.0f, y = e < n_ent ? L.y[e] : 0.0f;
                crash = crash | (((info & kAlive) != 0) & hit(body, thing_box(x, y, info & kKindMask)));
            }
            if (q.any(crash)) alive = false;
        }
        if (hit(body, thing_box(goal_x, goal_y, kGoal))) achieved_goal = true;

        // The bullets, newest first (see bossfight.hip agent_update for how a trip stands in for the reference's loop
        // with its shrinking count).  Bullets in flight are few, and each of them looks for the FIRST hazard it touches in
        // the hazard set's iteration order, destroying it if it is a target — so they are taken one after the other,
        // each tested against all entities at once (the winner is the touched entity with the smallest place in the order).
        {
            int count = s_count;
            for (int i0 = 0; i0 < count; i0 += kGang) {
                const Q::Trip t = q.trip<kShots, (kGang < kShots ? kGang : kShots)>(s_next, i0);
                const int k = t.slot;
                const bool mine = (q.g < kShots) & (t.i < count);
                float frame = -1.0f, bx = 0.0f, by = 0.0f, bvx = 0.0f, bvy = 0.0f;
                if (mine) {
                    frame = SH(s, SH_FRAME, k, env);
                    bx = SH(s, SH_X, k, env);
                    by = SH(s, SH_Y, k, env);
                    bvx = SH(s, SH_VX, k, env);
                    bvy = SH(s, SH_VY, k, env);
                }
                const bool live = mine & (frame != -1.0f);
                const bool gone = live & (frame >= 5.0f);  // (a bullet that hits something this sub-step is at frame 1)
                const bool act = live & (t.i < count - Q::before(q.ranked<(kGang < kShots ? kGang : kShots)>(t, gone), t.rank));
                const bool flying = act & (frame == 0.0f);
                uint32_t flying_list = q.ranked<(kGang < kShots ? kGang : kShots)>(t, flying);
                if (flying_list) {
                    const Box sb{bx - 0.01f, by - 0.01f, 0.02f, 0.02f};
                    bool stops = false;
                    if (flying) {
                        const BitWin win = BitWin::fetch(cols, static_cast<int>(floorf(sb.x)), static_cast<int>(floorf(sb.y)));
                        stops = collide_any(win, sb, is_wall);
                    }
                    while (flying_list) {
                        const int rank = __ffs(flying_list) - 1;
                        flying_list &= flying_list - 1;
                        const int from = ((kGang < kShots ? kGang : kShots) - 1 - t.turn - rank) & ((kGang < kShots ? kGang : kShots) - 1);  // the lane of that rank
                        const Box other{__shfl(sb.x, from, kGang), __shfl(sb.y, from, kGang), 0.02f, 0.02f};
                        // key = place in the hazard order · 1024 + entity · 4 + kind; the smallest touched one wins
                        int best = 0x7fffffff;
                        for (int e0 = 0; e0 < n_ent; e0 += kGang) {
                            const int e = e0 + q.g;
                            const int info = e < n_ent ? L.info[e] : 0;
                            const float x = e < n_ent ? L.x[e] : 0.0f, y = e < n_ent ? L.y[e] : 0.0f;
                            const int kind = info & kKindMask;
                            const bool touched = ((info & kAlive) != 0) & hit(other, thing_box(x, y, kind));
                            const int key = ((e < n_ent ? L.place[e] : 0) << 10) | (e << 2) | kind;
                            best = (touched & (key < best)) ? key : best;
                        }
#pragma unroll
                        for (int w = 1; w < kGang; w <<= 1) {
                            const int o = __shfl_xor(best, w, kGang);
                            best = o < best ? o : best;
                        }
                        if (best != 0x7fffffff) {
                            if (q.g == from) stops = true;
                            if ((best & 3) == kTarget) {  // destroy_entity
                                const int e = (best >> 2) & 255;
                                if ((e & (kGang - 1)) == q.g) {
                                    const int info = L.info[e] & ~kAlive;
                                    L.info[e] = static_cast<uint8_t>(info);
                                    EB(s, EB_INFO, e, env) = static_cast<uint8_t>(info);
                                }
                                set_changed = true;
                                targets_destroyed++;
                            }
                        }
                    }
                    if (stops) {
                        bvx = 0.0f;
                        bvy = 0.0f;
                        frame = 1.0f;
                    }
                }
                bx += bvx * dt;
                by += bvy * dt;
                frame = gone ? -1.0f : (frame >= 1.0f ? frame + explosion_rate * dt : frame);
                if (act) {
                    SH(s, SH_X, k, env) = bx;
                    SH(s, SH_Y, k, env) = by;
                    SH(s, SH_VX, k, env) = bvx;
                    SH(s, SH_VY, k, env) = bvy;
                    SH(s, SH_FRAME, k, env) = frame;
                }
                count -= __popc(q.ballot(gone & act));
            }
            s_count = count;
        }
        puff_on = movement_y > 0.0f;

        // --- System_Mob_AI::update (common_systems.cpp:50-75).  The enemies are the last third of the objects
        // (tilemap.cpp:232-272: ids first_enemy .. n_ent-1), so consecutive lanes hold consecutive enemies: a pass moves
        // kGang of them, each lane picking the one of its slots that holds its enemy.
        for (int e0 = first_enemy; e0 < n_ent; e0 += kGang) {
            const int e = e0 + ((q.g - e0) & (kGang - 1));  // the id in [e0, e0 + kGang) that is mine
            if (e < n_ent) {
                const int j = e - first_enemy;
                float vx = L.vx[j], vy = L.vy[j];
                const float x = L.x[e] + vx * dt, y = L.y[e] + vy * dt;
                const Box box{x + -0.4f, y + -0.4f, 0.8f, 0.8f};
                const BitWin win = BitWin::fetch(cols, static_cast<int>(floorf(box.x)), static_cast<int>(floorf(box.y)));
                if (collide_any(win, box, is_wall)) {
                    L.vx[j] = -vx;
                    L.vy[j] = -vy;
                }
                L.x[e] = x;
                L.y[e] = y;
            }
        }

        // --- System_Particles::update (common_systems.cpp:333-372)
        {
            const float lifespan = 3.0f, spawn_time = 0.3f, off_x = 0.0f, off_y = 0.3f;
            int dead_index = -1;
#pragma unroll
            for (int p = 0; p < kPuffsPerLane; p++) {
                const int k = q.g + kGang * p;
                puff_life[p] -= dt;
                const uint32_t dead = q.ballot((k < kPuffs) & (puff_life[p] <= 0.0f));
                if (dead) dead_index = kGang * p + 31 - __clz(dead);
            }
            p_timer += dt;
            if (dead_index != -1 && p_timer >= spawn_time && puff_on) {
                p_timer = fmodf(p_timer, spawn_time);
                const float c = puff_c, sn = puff_s;
#pragma unroll
                for (int p = 0; p < kPuffsPerLane; p++)
                    if (q.g + kGang * p == dead_index) {
                        puff_life[p] = lifespan;
                        PF(s, PF_ROT, dead_index, env) = prot;
                        PF(s, PF_SN, dead_index, env) = __int_as_float(prot_sn);
                        PF(s, PF_CS, dead_index, env) = __int_as_float(prot_cs);
                        PF(s, PF_DX, dead_index, env) = -dir_x;
                        PF(s, PF_DY, dead_index, env) = -dir_y;
                        PF(s, PF_X, dead_index, env) = ax + (c * off_x - sn * off_y);
                        PF(s, PF_Y, dead_index, env) = ay + (sn * off_x + c * off_y);
                    }
            }
        }

        reward = achieved_goal * 10.0f + targets_destroyed * 3.0f;
        terminated = !alive || achieved_goal;
        if (terminated) break;
    }
    // what the sub-steps changed, back to memory: the enemies, the particles' lives, the scalars
    for (int e0 = first_enemy; e0 < n_ent; e0 += kGang) {
        const int e = e0 + ((q.g - e0) & (kGang - 1));
        if (e < n_ent) {
            EF(s, EF_X, e, env) = L.x[e];
            EF(s, EF_Y, e, env) = L.y[e];
            EF(s, EF_VX, e, env) = L.vx[e - first_enemy];
            EF(s, EF_VY, e, env) = L.vy[e - first_enemy];
        }
    }
#pragma unroll
    for (int p = 0; p < kPuffsPerLane; p++) {
        const int k = q.g + kGang * p;
        if (k < kPuffs) PF(s, PF_LIFE, k, env) = puff_life[p];
    }
    if (set_changed) gang_fence();  // the destroyed targets' info bytes, for the lane that rebuilds the list
    if (q.g == 0) {
        SF(s, F_CAMX, env) = ax * kUnitPx;
        SF(s, F_CAMY, env) = ay * kUnitPx;
        SF(s, F_AX, env) = ax;
        SF(s, F_AY, env) = ay;
        SF(s, F_AVX, env) = avx;
        SF(s, F_AVY, env) = avy;
        SF(s, F_ROT, env) = rot;
        SF(s, F_SHIP_SN, env) = __int_as_float(ship_sn);
        SF(s, F_SHIP_CS, env) = __int_as_float(ship_cs);
        SF(s, F_STIMER, env) = s_timer;
        SF(s, F_PTIMER, env) = p_timer;
        SI(s, I_SNEXT, env) = s_next;
        SI(s, I_SCOUNT, env) = s_count;
        SI(s, I_FLAGS, env) = kFlagListed | (puff_on ? kFlagPuffOn : 0);
        if (set_changed) rebuild_draw_list(s, env, n_ent);
    }
    reward_out = reward;
    terminated_out = terminated;
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// What cenv_make leaves in an env besides the seeded RNG, split by owner: the generator chain (bucket counts of
// the sets that survive clear()) and the live state.  Level-seed mode (pg_engine.h LevelPlan) rebuilds every
// level from here.
PG_D void fresh_chain(const State& s, int env) {
    SI(s, I_HASH_SPRITE, env) = 1;
    SI(s, I_HASH_HAZARD, env) = 1;
}
PG_D void fresh_live(const State& s, int env) {
    SF(s, F_CAMX, env) = 0.0f;  // Renderer::camera_position{0} (renderer.h:18)
    SF(s, F_CAMY, env) = 0.0f;
    for (int k = 0; k < kShots; k++)  // std::vector<Bullet>(32): frame = -1 ("dead"), rest zero
        for (int f = 0; f < SH_COUNT; f++) SH(s, f, k, env) = (f == SH_FRAME) ? -1.0f : 0.0f;
}

__global__ void __launch_bounds__(64) make_kernel(State s, uint32_t seed_base, int env_offset) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= s.n) return;
    fresh_chain(s, env);
    fresh_live(s, env);
}

struct Gen {  // pg_prefetch.h level_kernel<Gen>
    using State = caveflyer::State;
    using Level = caveflyer::Level;
    using GenLds = caveflyer::GenLds;
    PG_D static void generate(const State& s, int env, GenLds& L, Level& lv, bool reseed, uint32_t seed, int lane) {
        caveflyer::generate(s, env, L, lv, reseed, seed, lane);
    }
    PG_D static void install(const State& s, int env, const Level& lv, int lane) {
        caveflyer::install(s, env, lv, lane);
    }
    PG_D static void fresh_chain(const State& s, int env) { caveflyer::fresh_chain(s, env); }
    PG_D static void fresh_live(const State& s, int env) { caveflyer::fresh_live(s, env); }
};

__global__ void __launch_bounds__(64, PG_CAVEFLYER_WAVES) logic_kernel(State s, const int32_t* actions, uint32_t run_seed,
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


// render_game(true) (caveflyer.cpp:413-440): one workgroup of two wavefronts per env (pg_render.h).
constexpr int kGrid = 16;  // 64 px / 8 px per tile → at most 10 columns/rows in view
constexpr int kSpan = 16;  // a tile covers up to ten pixels per axis (pg_render.h compose_spans MAXSPAN)

// The complete frame of one env by its workgroup, set-up included: the frames the pre-pass marks fat, the draw-list
// replay (flags bit 0) and kDebugNoPrepass.
PG_D void render_full(const State& s, const AtlasView& atlas, const StepIO& io, int flags, int env, uint32_t* fb,
                      ComposeLds<kGrid>& L) {
    const int lane = threadIdx.x & 63, half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // two wavefronts per env (pg_render.h)
    constexpr int halves = 2;

    const Camera cam{SF(s, F_CAMX, env), SF(s, F_CAMY, env), 64.0f, 64.0f, 0.5f * 64.0f / 64.0f};
    const int sflags = SI(s, I_FLAGS, env);
    const int n_draw = (sflags & kFlagListed) ? SI(s, I_NDRAW, env) : 0;  // empty right after a reset
    const int s_next = SI(s, I_SNEXT, env), s_count = SI(s, I_SCOUNT, env);
    const uint8_t* tiles = s.tiles + size_t(env) * kTileStride;
    const DescRegs descs = DescRegs::load(atlas, lane);
    Blit mine;

    // inputs of the sprite passes, requested early
    float puff_life = 0.0f, puff_x = 0.0f, puff_y = 0.0f, puff_dx = 0.0f, puff_dy = 0.0f;
    int puff_sn = 0, puff_cs = 0;  // the angles arrive as 16.16 sine and cosine (store_rotation)
    if (lane < kPuffs) {
        puff_life = PF(s, PF_LIFE, lane, env);
        puff_x = PF(s, PF_X, lane, env);
        puff_y = PF(s, PF_Y, lane, env);
        puff_dx = PF(s, PF_DX, lane, env);
        puff_dy = PF(s, PF_DY, lane, env);
        puff_sn = __float_as_int(PF(s, PF_SN, lane, env));
        puff_cs = __float_as_int(PF(s, PF_CS, lane, env));
    }
    // When everything after the tile layer fits the wave's 64 lanes it is drawn as ONE pass, lanes in draw order:
    // particles, then the sprites, then the bullets and the ship (see below); otherwise a pass per kind, from lane 0.
    const bool one_pass = kPuffs + n_draw + s_count + 1 <= 64;
    const int spr_lane0 = one_pass ? kPuffs : 0, shot_lane0 = one_pass ? kPuffs + n_draw : 0;
    int spr_kind = 0;
    float spr_x = 0.0f, spr_y = 0.0f;
    if (lane >= spr_lane0 && lane - spr_lane0 < n_draw) {
        const int e = EB(s, EB_DRAW, lane - spr_lane0, env);
        spr_kind = EB(s, EB_INFO, e, env) & kKindMask;
        spr_x = EF(s, EF_X, e, env);
        spr_y = EF(s, EF_Y, e, env);
    }
    float shot_frame = -1.0f, shot_x = 0.0f, shot_y = 0.0f;
    int shot_sn = 0, shot_cs = 0;
    const int shot_i = lane - shot_lane0;  // bullets newest first, then the ship
    if (shot_i >= 0 && shot_i < s_count) {
        const int k = (kShots + s_next - 1 - shot_i) % kShots;
        shot_frame = SH(s, SH_FRAME, k, env);
        shot_x = SH(s, SH_X, k, env);
        shot_y = SH(s, SH_Y, k, env);
        shot_sn = __float_as_int(SH(s, SH_SN, k, env));
        shot_cs = __float_as_int(SH(s, SH_CS, k, env));
    } else if (shot_i == s_count) {
        shot_x = SF(s, F_AX, env);
        shot_y = SF(s, F_AY, env);
        shot_sn = __float_as_int(SF(s, F_SHIP_SN, env));
        shot_cs = __float_as_int(SF(s, F_SHIP_CS, env));
    }

    int bg_soft = 0;  // the backdrop has texels that are not opaque (descriptor .w)
    int4 bg_d;  // the background draw, caveflyer.cpp:427-432: texture, world position, scale — each wave resolves the axis it needs (pg_render.h BgAxis)
    float bg_px, bg_py, bg_sc;
    {
        const int4 d = descs.uniform(kTexSpace + SI(s, I_BACKDROP, env));
        bg_soft = d.w;
        const float aspect = static_cast<float>(d.y) / static_cast<float>(d.z);
        const float extra = aspect - 1.0f;
        bg_d = d;
        bg_px = -SF(s, F_BGSHIFT, env) * extra;
        bg_py = 0.0f;
        bg_sc = 64.0f * kUnitPx / d.z;
    }
    // tile window (tilemap.cpp:280-289)
    const float vx = (cam.px - cam.sw * 0.5f / cam.scale) * kPxUnit;
    const float vy = (cam.py - cam.sh * 0.5f / cam.scale) * kPxUnit;
    const float vw = cam.sw * kPxUnit / cam.scale, vh = cam.sh * kPxUnit / cam.scale;
    const int x0 = static_cast<int>(floorf(vx)), y0 = static_cast<int>(floorf(vy));
    const int x1 = static_cast<int>(ceilf(vx + vw)), y1 = static_cast<int>(ceilf(vy + vh));
    const int cols = x1 - x0 + 1, rows = y1 - y0 + 1, cells = cols * rows;
    const int4 wall_d = descs.uniform(kTexWall);

    const BgDraw bg_draw{bg_d, bg_px, bg_py, bg_sc};
    BgAxis bga{};  // this wave's axis of it (wave 0: x, wave 1: y), resolved along with the tile spans
    bool composed = false;
    if (!(flags & 1) && cols <= kGrid && rows <= kGrid) {
        compose_spans<kGrid, kSpan>(fb, L, cam, x0, y0, cols, rows, wall_d.y, wall_d.z, kUnitPx / wall_d.y, lane, 0, half, halves,
                                 soft_rows_of(bg_soft, wall_d.w), hard_rows_of(bg_soft, wall_d.w), &bg_draw, &bga);
#pragma unroll
        for (int k = half; k < kGrid * kGrid / 64; k += halves) {
            const int cell = k * 64 + lane;
            const int r = cell / kGrid, c = cell % kGrid;
            const int t = (c < cols && r < rows) ? Win::direct(tiles, x0 + c, y0 + r) : kEmpty;
            L.base[cell] = (t == kEmpty) ? static_cast<int32_t>(kNoTexel) : wall_d.x * 4;
        }
        __syncthreads();
        composed = compose_rows(fb, L, atlas, bga, cols, rows, wall_d.y, lane, flags, half, halves);
    }
    if (!composed) {  // draw-list replay (tilemap.cpp:291-302)
        wave_clear(fb, lane, half, halves);
        const bool has_bg = resolve_draw(cam, bg_d.y, bg_d.z, bg_d.x, bg_px, bg_py, bg_sc, 1.0f, false, false, mine);
        wave_replay(fb, atlas, mine, has_bg ? 1ull : 0ull, lane, half, halves);
        for (int base = 0; base < cells; base += 64) {
            const int cell = base + lane;
            bool has = false;
            if (cell < cells) {
                const int row = cell / cols;
                const int x = x0 + (cell - row * cols), y = y0 + row;
                if (Win::direct(tiles, x, y) != kEmpty)
                    has = resolve_draw(cam, wall_d.y, wall_d.z, wall_d.x, x * kUnitPx, y * kUnitPx, kUnitPx / wall_d.y,
                                       1.0f, false, false, mine);
            }
            wave_replay(fb, atlas, mine, __ballot(has), lane, half, halves);
        }
    }

    if (one_pass) {
        // System_Particles::render (common_systems.cpp:374-397: rotated, fading), the positive-z sprites (:26-48: goal,
        // meteors, targets, enemies), System_Agent::render (:291-327: bullets newest first, then the ship) — one draw
        // per lane in that order.  The rotated kinds share one trip through resolve_rotated (its sine and cosine are
        // the expensive part), the sprites take resolve_draw.
        const bool is_puff = lane < kPuffs, is_spr = lane >= spr_lane0 && lane < shot_lane0;
        int want_tex = kTexShip;
        bool go = shot_i == s_count;  // the ship
        float size = 0.15f, alpha = 1.0f, rx = 0.0f, ry = 0.0f;
        if (is_puff) {
            want_tex = kTexPuff;
            go = puff_life > 0.0f;
        } else if (is_spr) {
            want_tex = kTexKind + spr_kind;
            go = false;
        } else if (shot_i >= 0 && shot_i < s_count && shot_frame != -1.0f) {
            go = true;
            size = 0.1f;
            want_tex = (shot_frame == 0.0f) ? kTexLaser : kTexBoom + static_cast<int>(shot_frame - 1.0f);
        }
        const int4 d = descs.at(want_tex);
        if (is_puff) {
            const float lifespan = 3.0f;
            const float life_ratio = (lifespan - puff_life) / lifespan;
            alpha = 0.5f * (1.0f - life_ratio);
            const float scale = 1.0f * (0.4f * life_ratio + 0.6f);
            const float shift = life_ratio * 2.0f;
            size = scale * kUnitPx / d.y;
            rx = (puff_x + puff_dx * shift) * kUnitPx - size * d.y * 0.5f;
            ry = (puff_y + puff_dy * shift) * kUnitPx - size * d.z * 0.5f;
        } else {
            rx = shot_x * kUnitPx - size * d.y * 0.5f;
            ry = shot_y * kUnitPx - size * d.z * 0.5f;
        }
        bool has = false;
        if (go)
            has = resolve_rotated_at(cam, d.y, d.z, d.x, rx, ry, is_puff ? puff_sn : shot_sn, is_puff ? puff_cs : shot_cs, size,
                                     alpha, mine);
        if (is_spr) {
            const float scale = 1.0f * 0.8f;
            has = resolve_draw(cam, d.y, d.z, d.x, (spr_x + -0.4f) * kUnitPx, (spr_y + -0.4f) * kUnitPx,
                               scale * kUnitPx / d.y, 1.0f, false, false, mine);
        }
        wave_replay_rows(fb, atlas, mine, __ballot(has), lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
    } else {
        {  // System_Particles::render (common_systems.cpp:374-397): rotated, fading
            const int4 d = descs.uniform(kTexPuff);
            bool has = false;
            if (lane < kPuffs && puff_life > 0.0f) {
                const float lifespan = 3.0f;
                const float life_ratio = (lifespan - puff_life) / lifespan;
                const float alpha = 0.5f * (1.0f - life_ratio);
                const float scale = 1.0f * (0.4f * life_ratio + 0.6f);
                const float shift = life_ratio * 2.0f;
                const float size = scale * kUnitPx / d.y;
                has = resolve_rotated_at(cam, d.y, d.z, d.x, (puff_x + puff_dx * shift) * kUnitPx - size * d.y * 0.5f,
                                         (puff_y + puff_dy * shift) * kUnitPx - size * d.z * 0.5f, puff_sn, puff_cs, size, alpha,
                                         mine);
            }
            wave_replay_rows(fb, atlas, mine, __ballot(has), lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
        }
        {  // positive-z sprites (common_systems.cpp:26-48): goal, meteors, targets, enemies
            const int4 d = descs.at(kTexKind + spr_kind);
            bool has = false;
            if (lane < n_draw) {
                const float scale = 1.0f * 0.8f;
                has = resolve_draw(cam, d.y, d.z, d.x, (spr_x + -0.4f) * kUnitPx, (spr_y + -0.4f) * kUnitPx,
                                   scale * kUnitPx / d.y, 1.0f, false, false, mine);
            }
            wave_replay_rows(fb, atlas, mine, __ballot(has), lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
            if (kMaxEnt > 64 && n_draw > 64) {  // memory_mode: up to 76 sprites, the rest in a second pass
                const int k = 64 + lane;
                int kind2 = 0;
                float x2 = 0.0f, y2 = 0.0f;
                if (k < n_draw) {
                    const int e = EB(s, EB_DRAW, k, env);
                    kind2 = EB(s, EB_INFO, e, env) & kKindMask;
                    x2 = EF(s, EF_X, e, env);
                    y2 = EF(s, EF_Y, e, env);
                }
                const int4 d2 = descs.at(kTexKind + kind2);
                bool has2 = false;
                if (k < n_draw) {
                    const float scale = 1.0f * 0.8f;
                    has2 = resolve_draw(cam, d2.y, d2.z, d2.x, (x2 + -0.4f) * kUnitPx, (y2 + -0.4f) * kUnitPx,
                                        scale * kUnitPx / d2.y, 1.0f, false, false, mine);
                }
                wave_replay_rows(fb, atlas, mine, __ballot(has2), lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
            }
        }
        {  // System_Agent::render (common_systems.cpp:291-327): bullets newest first, then the ship
            int want_tex = kTexShip;
            bool has = lane == s_count;
            float size = 0.15f;
            if (lane < s_count && shot_frame != -1.0f) {
                has = true;
                size = 0.1f;
                want_tex = (shot_frame == 0.0f) ? kTexLaser : kTexBoom + static_cast<int>(shot_frame - 1.0f);
            }
            const int4 d = descs.at(want_tex);
            if (has)
                has = resolve_rotated_at(cam, d.y, d.z, d.x, shot_x * kUnitPx - size * d.y * 0.5f,
                                         shot_y * kUnitPx - size * d.z * 0.5f, shot_sn, shot_cs, size, 1.0f, mine);
            wave_replay_rows(fb, atlas, mine, __ballot(has), lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
        }
    }
    // each wave stores the rows it owns (pg_render.h wave_replay_rows): no barrier
    wave_store_rows(fb, io.obs + size_t(env) * kObsBytes, lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
}

// ------------------------------------------------------------------------------------------------
// The render pre-pass (pg_prepass.h; coinrun.hip's setup_kernel is the commented model): tile spans, per-pixel
// candidates, the cell table and the resolved, culled SPRITES of kPrepEnvs envs per workgroup.  The rotated draws —
// particles, bullets, the ship — stay with the render wave: resolve_rotated_at has no division and no cull, and their
// sines and cosines come from the logic kernel.  Reference arithmetic moved here unchanged: renderer.cpp:5-82,
// tilemap.cpp:280-289 (the window), common_systems.cpp:26-48 (sprites).
// ------------------------------------------------------------------------------------------------
constexpr int kPrepEnvs = 8, kPrepThreads = 256;
enum { GW_SHOTS = 0, GW_SHIP_X, GW_SHIP_Y, GW_SHIP_SN, GW_SHIP_CS, GW_CAM_X, GW_CAM_Y };  // PM_GAME words: s_next | s_count << 16, the ship, the camera

struct PrepEnv {
    int32_t n_draw, s_count;
};
struct SetupLds {
    PrepLds<kGrid, kPrepEnvs, kSpan> P;
    PrepEnv env[kPrepEnvs];
    int4 desc[kTexCount];
    uint32_t draw_order[kPrepEnvs][kEntStride / 4];  // EB_DRAW of every env: fetched before anything needs it
    uint32_t row_valid[kPrepEnvs][kGrid / 4];
    int32_t counts[kPrepEnvs];
    PrepDrawQueue queue[kPrepThreads / 64];
};

__global__ void __launch_bounds__(kPrepThreads) setup_kernel(State s, AtlasView atlas, const uint8_t* mask, int flags) {
    __shared__ SetupLds S;
    PrepLds<kGrid, kPrepEnvs, kSpan>& P = S.P;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int env0 = prep_block(blockIdx.x, gridDim.x) * kPrepEnvs;  // (pg_prepass.h: the groups of one XCD are consecutive)
    const PrepOut& out = s.prep;

    // ---- one memory round trip: descriptor table, the envs' scalars (lane = env), their draw orders
    for (int q = tid; q < kPrepEnvs * 2 * 64; q += kPrepThreads) (&P.cover[0][0][0])[q] = 0u;
    if (tid < kTexCount) S.desc[tid] = atlas.desc[tid];
    static_assert(kTexCount <= kPrepThreads && kPrepEnvs * (kEntStride / 4) <= kPrepThreads, "one word per thread");
    if (tid < kPrepEnvs * (kEntStride / 4)) {
        const int e = tid / (kEntStride / 4), w = tid - e * (kEntStride / 4);
        if (env0 + e < s.n) S.draw_order[e][w] = reinterpret_cast<const uint32_t*>(&EB(s, EB_DRAW, 0, env0 + e))[w];
    }
    Camera cam{};
    int backdrop = 0;
    float bgshift = 0.0f;
    bool active = false;
    uint32_t game_words[7] = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
    if (tid < kPrepEnvs) {
        const int e = tid, env = env0 + e;
        active = env < s.n && (!mask || mask[env]);
        if (active) {
            cam = Camera{SF(s, F_CAMX, env), SF(s, F_CAMY, env), 64.0f, 64.0f, 0.5f * 64.0f / 64.0f};
            game_words[GW_CAM_X] = __float_as_uint(cam.px);
            game_words[GW_CAM_Y] = __float_as_uint(cam.py);
            backdrop = SI(s, I_BACKDROP, env);
            bgshift = SF(s, F_BGSHIFT, env);
            const int sflags = SI(s, I_FLAGS, env);
            PrepEnv pe{};
            pe.n_draw = (sflags & kFlagListed) ? SI(s, I_NDRAW, env) : 0;  // empty right after a reset
            pe.s_count = SI(s, I_SCOUNT, env);
            S.env[e] = pe;
            game_words[GW_SHOTS] = pack_halves(SI(s, I_SNEXT, env), pe.s_count);
            game_words[GW_SHIP_X] = __float_as_uint(SF(s, F_AX, env));
            game_words[GW_SHIP_Y] = __float_as_uint(SF(s, F_AY, env));
            game_words[GW_SHIP_SN] = __float_as_uint(SF(s, F_SHIP_SN, env));
            game_words[GW_SHIP_CS] = __float_as_uint(SF(s, F_SHIP_CS, env));
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
            const int4 d = S.desc[kTexSpace + backdrop];
            const float aspect = static_cast<float>(d.y) / static_cast<float>(d.z);
            const float extra = aspect - 1.0f;
            v.bg = BgDraw{d, -bgshift * extra, 0.0f, 64.0f * kUnitPx / d.z};  // caveflyer.cpp:427-432
            const float vx = (cam.px - cam.sw * 0.5f / cam.scale) * kPxUnit;  // tilemap.cpp:280-289
            const float vy = (cam.py - cam.sh * 0.5f / cam.scale) * kPxUnit;
            const float vw = cam.sw * kPxUnit / cam.scale, vh = cam.sh * kPxUnit / cam.scale;
            v.x0 = static_cast<int>(floorf(vx));
            v.y0 = static_cast<int>(floorf(vy));
            v.cols = static_cast<int>(ceilf(vx + vw)) - v.x0 + 1;
            v.rows = static_cast<int>(ceilf(vy + vh)) - v.y0 + 1;
            const int4 wall_d = S.desc[kTexWall];
            v.tw = wall_d.y;
            v.th = wall_d.z;
            v.th2 = 0;
            v.tile_scale = kUnitPx / wall_d.y;
            if (v.cols > kGrid || v.rows > kGrid || ((flags & kDebugFatThirds) && (env0 + e) % 3 == 0)) {
                P.fat[e] = 1;
                active = false;
            }
            P.soft_rows[e] = static_cast<uint32_t>(soft_rows_of(d.w, wall_d.w));
            P.hard_rows[e] = static_cast<uint32_t>(hard_rows_of(d.w, wall_d.w));
#pragma unroll
            for (int k = 0; k < kPrepKinds; k++) P.meta[e][PM_KINDS + k] = kNoTexel;
            P.meta[e][PM_KINDS + 0] = static_cast<uint32_t>(wall_d.x) * 4u;  // the one tile kind
#pragma unroll
            for (int k = 0; k < 7; k++) P.meta[e][PM_GAME + k] = game_words[k];
            prep_row_valid<kGrid, H>(v.y0, S.row_valid[e]);
        }
        v.active = active ? 1 : 0;
        P.view[e] = v;
    }
    __syncthreads();

    // ---- the cell table: lane = (env, grid column), one 16-byte load of the column-major map (pg_prepass.h) …
    const int cell_e = tid / kGrid, cell_c = tid - cell_e * kGrid;
    bool cell_lane = false, cell_x_ok = false;
    uint32_t column[kGrid / 4] = {};
    if (tid < kPrepEnvs * kGrid && P.view[cell_e].active) {
        cell_lane = true;
        prep_column_fetch<kGrid, W, H>(s.tiles + size_t(env0 + cell_e) * kTileStride, P.view[cell_e].x0 + cell_c, P.view[cell_e].y0, cell_x_ok, column);
    }
    // … the spans are worked out while it travels …
    prep_spans<kGrid, kSpan, kPrepEnvs>(P, tid, kPrepThreads);
    // … then the kind bytes: wall → 0, empty → none
    if (cell_lane) {
        uint32_t in_rows[kGrid / 4], kinds[kGrid / 4];
        prep_column_rows<kGrid>(column, S.row_valid[cell_e], cell_x_ok, kWall, in_rows);  // out of bounds is a wall (tilemap.h:78-83)
#pragma unroll
        for (int w = 0; w < kGrid / 4; w++) {
            const uint32_t t = in_rows[w] & 0x07070707u;
            const uint32_t any = (t | (t >> 1) | (t >> 2)) & 0x01010101u;  // 1 where the cell is not empty
            kinds[w] = ~(any * 0xffu);                                    // kind 0 for walls; 0xff: no tile
        }
        prep_column_store<kGrid>(out.cells + size_t(env0 + cell_e) * (kGrid * kGrid), cell_c, kinds);
    }
    __syncthreads();
    prep_axes<kGrid, kSpan, kPrepEnvs>(P, out, env0, wave, kPrepThreads / 64, lane);

    // ---- the positive-z sprites (common_systems.cpp:26-48: goal, meteors, targets, enemies) in draw order.  Two envs per
    // wavefront, cull first (pg_prepass.h prep_draws_pass).
    static_assert(kPrepEnvs == 2 * (kPrepThreads / 64), "two envs per wavefront");
    {
        const int ea = 2 * wave, eb = 2 * wave + 1;
        const bool on_a = P.view[ea].active != 0, on_b = P.view[eb].active != 0;
        const Camera cam_a = P.view[ea].cam, cam_b = P.view[eb].cam;
        const int cnt_a = on_a ? S.env[ea].n_draw : 0, cnt_b = on_b ? S.env[eb].n_draw : 0;
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
            PrepDraw p{false, false, false, kTexKind, 0.0f, 0.0f, 1.0f, 1.0f};
            if (valid) {
                const int ent = (S.draw_order[e][slot >> 2] >> (8 * (slot & 3))) & 0xffu;
                const int kind = EB(s, EB_INFO, ent, env) & kKindMask;
                const float ex = EF(s, EF_X, ent, env), ey = EF(s, EF_Y, ent, env);
                const float scale = 1.0f * 0.8f;
                p.tex = kTexKind + kind;
                p.wx = (ex + -0.4f) * kUnitPx;
                p.wy = (ey + -0.4f) * kUnitPx;
                p.scale = scale * kUnitPx / S.desc[p.tex].y;
                p.go = true;
            }
            prep_draws_pass(Q, st, S.desc, cam_a, cam_b, draws_a, draws_b, valid, is_b, p, lane);
        }
        prep_draws_flush(Q, st, S.desc, cam_a, cam_b, draws_a, draws_b, lane);
        if (lane == 0) {  // (the render wave's one pass: particles, these sprites, the bullets, the ship — 64 lanes)
            const int room_a = 64 - kPuffs - 1 - (on_a ? S.env[ea].s_count : 0), room_b = 64 - kPuffs - 1 - (on_b ? S.env[eb].s_count : 0);
            S.counts[ea] = st.done[0] > room_a ? kPrepDraws + 1 : st.done[0];
            S.counts[eb] = st.done[1] > room_b ? kPrepDraws + 1 : st.done[1];
        }
    }
    __syncthreads();
    prep_meta_out<kGrid, kSpan, kPrepEnvs>(P, out, env0, S.counts, tid, kPrepThreads);
}

// render_game(true) (caveflyer.cpp:413-440): one workgroup of two wavefronts per env; a lean frame starts from what
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
    PG_TL_BEGIN(6);
    PG_TL(0);
    const uint32_t kind_off = M.w[PM_KINDS + (lane & (kPrepKinds - 1))];
    const uint32_t two16 = reinterpret_cast<const uint16_t*>(s.prep.cells)[size_t(env) * (kGrid * kGrid / 2) + half * 64 + lane];
    const DescRegs descs = DescRegs::load(atlas, lane);
    // the rotated draws' inputs (per-env contiguous tables): particles in lanes 0-9; bullets and the ship behind the sprites
    float puff_life = 0.0f, puff_x = 0.0f, puff_y = 0.0f, puff_dx = 0.0f, puff_dy = 0.0f;
    int puff_sn = 0, puff_cs = 0;
    if (lane < kPuffs) {
        puff_life = PF(s, PF_LIFE, lane, env);
        puff_x = PF(s, PF_X, lane, env);
        puff_y = PF(s, PF_Y, lane, env);
        puff_dx = PF(s, PF_DX, lane, env);
        puff_dy = PF(s, PF_DY, lane, env);
        puff_sn = __float_as_int(PF(s, PF_SN, lane, env));
        puff_cs = __float_as_int(PF(s, PF_CS, lane, env));
    }
    const int n_vis = M.draws();
    const int s_next = static_cast<int>(M.w[PM_GAME + GW_SHOTS] & 0xffffu), s_count = static_cast<int>(M.w[PM_GAME + GW_SHOTS] >> 16);
    const int spr_lane0 = kPuffs, shot_lane0 = kPuffs + n_vis;
    const bool is_puff = lane < kPuffs, is_spr = lane >= spr_lane0 && lane < shot_lane0;
    Blit mine = prep_draw_load(s.prep.draws + (size_t(env) * kPrepDraws + (lane - spr_lane0)) * kBlitWords, is_spr);
    float shot_frame = -1.0f, shot_x = 0.0f, shot_y = 0.0f;
    int shot_sn = 0, shot_cs = 0;
    const int shot_i = lane - shot_lane0;  // bullets newest first, then the ship
    if (shot_i >= 0 && shot_i < s_count) {
        const int k = (kShots + s_next - 1 - shot_i) % kShots;
        shot_frame = SH(s, SH_FRAME, k, env);
        shot_x = SH(s, SH_X, k, env);
        shot_y = SH(s, SH_Y, k, env);
        shot_sn = __float_as_int(SH(s, SH_SN, k, env));
        shot_cs = __float_as_int(SH(s, SH_CS, k, env));
    } else if (shot_i == s_count) {
        shot_x = __uint_as_float(M.w[PM_GAME + GW_SHIP_X]);
        shot_y = __uint_as_float(M.w[PM_GAME + GW_SHIP_Y]);
        shot_sn = static_cast<int>(M.w[PM_GAME + GW_SHIP_SN]);
        shot_cs = static_cast<int>(M.w[PM_GAME + GW_SHIP_CS]);
    }
    prep_cells_expand<kGrid>(L, two16, kind_off, half, lane);
    const ComposeRegs R = prep_regs<kGrid>(M, colw, roww, 0u, lane);
    __syncthreads();  // the cell table is complete
    if ((flags & (1 | kDebugNoPrepass)) || M.fat()) {  // (wave-uniform)
        render_full(s, atlas, io, flags, env, fb, L);
        return;
    }
    const int row_lo = half * (kObsH / halves), row_hi = (half + 1) * (kObsH / halves);
    PG_TL(1);
    compose_rows_from<kGrid, false, false>(fb, L, atlas, R, lane, flags, half, halves);
    PG_TL(2);
    // one draw per lane in the reference's order: particles (common_systems.cpp:374-397: rotated, fading), the sprites the
    // pre-pass resolved, System_Agent::render (:291-327: bullets newest first, then the ship)
    const Camera cam{0.0f, 0.0f, 64.0f, 64.0f, 0.5f * 64.0f / 64.0f};  // (position: see below)
    int want_tex = kTexShip;
    bool go = shot_i == s_count;  // the ship
    float size = 0.15f, alpha = 1.0f, rx = 0.0f, ry = 0.0f;
    if (is_puff) {
        want_tex = kTexPuff;
        go = puff_life > 0.0f;
    } else if (is_spr) {
        go = false;
    } else if (shot_i >= 0 && shot_i < s_count && shot_frame != -1.0f) {
        go = true;
        size = 0.1f;
        want_tex = (shot_frame == 0.0f) ? kTexLaser : kTexBoom + static_cast<int>(shot_frame - 1.0f);
    }
    const int4 d = descs.at(want_tex);
    if (is_puff) {
        const float lifespan = 3.0f;
        const float life_ratio = (lifespan - puff_life) / lifespan;
        alpha = 0.5f * (1.0f - life_ratio);
        const float scale = 1.0f * (0.4f * life_ratio + 0.6f);
        const float shift = life_ratio * 2.0f;
        size = scale * kUnitPx / d.y;
        rx = (puff_x + puff_dx * shift) * kUnitPx - size * d.y * 0.5f;
        ry = (puff_y + puff_dy * shift) * kUnitPx - size * d.z * 0.5f;
    } else {
        rx = shot_x * kUnitPx - size * d.y * 0.5f;
        ry = shot_y * kUnitPx - size * d.z * 0.5f;
    }
    bool has = is_spr;
    if (go) {
        const Camera at{__uint_as_float(M.w[PM_GAME + GW_CAM_X]), __uint_as_float(M.w[PM_GAME + GW_CAM_Y]), cam.sw, cam.sh, cam.scale};
        has = resolve_rotated_at(at, d.y, d.z, d.x, rx, ry, is_puff ? puff_sn : shot_sn, is_puff ? puff_cs : shot_cs, size, alpha, mine);
    }
    // The small rotated draws — exhaust particles, bullets — share their memory round trips in groups of four like the plain
    // ones (pg_render.h kRotInGroups) instead of going alone, a round trip each: 107.8 -> 118.9 M env-steps/s same-box.  (Groups of
    // two: the same; of six or eight: the kernel spills, 85.6 and 78.4 M.)  The ship, more than 64 pixels, still goes alone.
#ifndef PG_CAVEFLYER_QUARTERS
#define PG_CAVEFLYER_QUARTERS false
#endif
    PG_TL(3);
    wave_replay_rows<4, true, true, 4, false, PG_CAVEFLYER_QUARTERS>(fb, atlas, mine, __ballot(has), lane, row_lo, row_hi);
    PG_TL(4);
    wave_store_rows(fb, io.obs + size_t(env) * kObsBytes, lane, row_lo, row_hi);
    PG_TL_END(6, true, io.obs + size_t(env) * kObsBytes + half * (kObsBytes / 2));
}

// cenv_render's frame (render_game(false)) for one env: pg_frame.h; the draw list of render_kernel, one draw at a time.
__global__ void __launch_bounds__(kFrameThreads) frame_kernel(State s, AtlasView atlas, int env, FrameTarget t) {
    const float fw = static_cast<float>(t.w), fh = static_cast<float>(t.h);
    FramePainter P{t, atlas, Camera{SF(s, F_CAMX, env), SF(s, F_CAMY, env), fw, fh, 0.5f * fw / 64.0f},
                   static_cast<int>(threadIdx.x), kFrameThreads};
    const int sflags = SI(s, I_FLAGS, env);
    const int n_draw = (sflags & kFlagListed) ? SI(s, I_NDRAW, env) : 0;
    const int s_next = SI(s, I_SNEXT, env), s_count = SI(s, I_SCOUNT, env);
    const uint8_t* tiles = s.tiles + size_t(env) * kTileStride;
    P.clear();
    {
        const int4 d = P.desc(kTexSpace + SI(s, I_BACKDROP, env));
        const float aspect = static_cast<float>(d.y) / static_cast<float>(d.z);
        const float extra = aspect - 1.0f;
        P.draw(kTexSpace + SI(s, I_BACKDROP, env), -SF(s, F_BGSHIFT, env) * extra, 0.0f, 64.0f * kUnitPx / d.z);
    }
    int x0, y0, x1, y1;
    P.window(x0, y0, x1, y1);
    for (int y = y0; y <= y1; y++)
        for (int x = x0; x <= x1; x++)
            if (Win::direct(tiles, x, y) != kEmpty) P.draw(kTexWall, x * kUnitPx, y * kUnitPx, kUnitPx / P.desc(kTexWall).y);
    for (int k = 0; k < kPuffs; k++) {
        const float life = PF(s, PF_LIFE, k, env);
        if (life <= 0.0f) continue;
        const int4 d = P.desc(kTexPuff);
        const float lifespan = 3.0f;
        const float life_ratio = (lifespan - life) / lifespan;
        const float alpha = 0.5f * (1.0f - life_ratio);
        const float scale = 1.0f * (0.4f * life_ratio + 0.6f);
        const float shift = life_ratio * 2.0f;
        const float size = scale * kUnitPx / d.y;
        P.draw_rotated(kTexPuff, (PF(s, PF_X, k, env) + PF(s, PF_DX, k, env) * shift) * kUnitPx - size * d.y * 0.5f,
                       (PF(s, PF_Y, k, env) + PF(s, PF_DY, k, env) * shift) * kUnitPx - size * d.z * 0.5f,
                       PF(s, PF_ROT, k, env), size, alpha);
    }
    for (int k = 0; k < n_draw; k++) {
        const int e = EB(s, EB_DRAW, k, env);
        const int tex = kTexKind + (EB(s, EB_INFO, e, env) & kKindMask);
        const float scale = 1.0f * 0.8f;
        P.draw(tex, (EF(s, EF_X, e, env) + -0.4f) * kUnitPx, (EF(s, EF_Y, e, env) + -0.4f) * kUnitPx,
               scale * kUnitPx / P.desc(tex).y);
    }
    for (int i = 0; i < s_count; i++) {
        const int k = (kShots + s_next - 1 - i) % kShots;
        const float frame = SH(s, SH_FRAME, k, env);
        if (frame == -1.0f) continue;
        const int tex = (frame == 0.0f) ? kTexLaser : kTexBoom + static_cast<int>(frame - 1.0f);
        const int4 d = P.desc(tex);
        const float size = 0.1f;
        P.draw_rotated(tex, SH(s, SH_X, k, env) * kUnitPx - size * d.y * 0.5f, SH(s, SH_Y, k, env) * kUnitPx - size * d.z * 0.5f,
                       static_cast<float>(SH(s, SH_ROT, k, env) + kPi * 0.5f), size);
    }
    {
        const int4 d = P.desc(kTexShip);
        const float size = 0.15f;
        P.draw_rotated(kTexShip, SF(s, F_AX, env) * kUnitPx - size * d.y * 0.5f, SF(s, F_AY, env) * kUnitPx - size * d.z * 0.5f,
                       static_cast<float>(SF(s, F_ROT, env) + kPi * 0.5f), size);
    }
}

class CaveflyerGame final : public Game {
   public:
    const char* name() const override { return "caveflyer"; }
    std::vector<std::string> texture_names() const override {
        std::vector<std::string> v;
        for (const char* t : {"groundA", "meteorBrown_big1", "ufoRed2", "enemyShipBlue4", "ufoGreen2", "playerShip1_red",
                              "laserBlue02", "explosion1", "explosion2", "explosion3", "explosion4", "explosion5",
                              "towerDefense_tile295"})
            v.push_back(std::string("misc_assets/") + t + ".png");
        for (const char* b : {"deep_space_01", "spacegen_01", "milky_way_01", "ez_space_lite_01", "meyespace_v1_01",
                              "eye_nebula_01", "deep_sky_01", "space_nebula_01", "Background-1", "Background-2",
                              "Background-3", "Background-4", "parallax-space-backgound"})
            v.push_back(std::string("space_backgrounds/") + b + ".png");
        return v;
    }
    std::string check_atlas(const std::vector<std::pair<int, int>>& sizes) const override {
        return static_cast<int>(sizes.size()) == kTexCount ? "" : "caveflyer: unexpected texture count";
    }
    static size_t align256(size_t x) { return (x + 255) & ~size_t(255); }
    struct Layout {
        size_t shadow, slot, mt, tiles, cols, f, i, ef, eb, sh, pf, total;
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
        l.tiles = take(size_t(n) * kTileStride);
        l.cols = take(size_t(n) * W * 8);
        l.f = take(size_t(F_COUNT) * n * 4);
        l.i = take(size_t(I_COUNT) * n * 4);
        l.ef = take(size_t(EF_COUNT) * kEntStride * n * 4);
        l.eb = take(size_t(EB_COUNT) * kEntStride * n);
        l.sh = take(size_t(SH_COUNT) * kShots * n * 4);
        l.pf = take(size_t(PF_COUNT) * kPuffSlots * n * 4);
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
        s_.cols = reinterpret_cast<uint64_t*>(p + l.cols);
        s_.f = reinterpret_cast<float*>(p + l.f);
        s_.i = reinterpret_cast<int32_t*>(p + l.i);
        s_.ef = reinterpret_cast<float*>(p + l.ef);
        s_.eb = p + l.eb;
        s_.sh = reinterpret_cast<float*>(p + l.sh);
        s_.pf = reinterpret_cast<float*>(p + l.pf);
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
    int pregen_every() const override { return 2; }  // (pg_engine.h)
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
    size_t scratch_bytes(int n) const override { return prep_bytes(n, kGrid, kBlitWords, false); }
    void bind_scratch(void* d_scratch, int n) override { s_.prep = prep_bind(d_scratch, n, kGrid, kBlitWords, false); }
    void launch_prepass(hipStream_t st, const uint8_t* mask) override {
        if (!(debug_flags & (1 | kDebugNoPrepass)))
            hipLaunchKernelGGL(setup_kernel, dim3((s_.n + kPrepEnvs - 1) / kPrepEnvs), dim3(kPrepThreads), 0, st, s_, atlas_, mask, debug_flags);
    }
    void launch_render(hipStream_t st, const uint8_t* mask, StepIO io) override {
        hipLaunchKernelGGL(render_kernel, dim3(s_.n), dim3(128), 0, st, s_, atlas_, mask, io, debug_flags);
    }
    // Same layout as oracle/pgo_caveflyer.cpp Caveflyer::dump_state.
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
        const int flags = ri(I_FLAGS), n_ent = ri(I_NENT);
        std::vector<float> v = {f(F_AX), f(F_AY), f(F_AVX), f(F_AVY), f(F_ROT), f(F_CAMX), f(F_CAMY),
                                static_cast<float>(ri(I_BACKDROP)), f(F_BGSHIFT), static_cast<float>(ri(I_SNEXT)),
                                static_cast<float>(ri(I_SCOUNT)), f(F_STIMER), f(F_PTIMER),
                                (flags & kFlagPuffOn) ? 1.0f : 0.0f, static_cast<float>(n_ent)};
        for (int k = 0; k < kShots; k++)
            for (int fld : {SH_X, SH_Y, SH_FRAME}) v.push_back(rf(s_.sh, (size_t(env) * SH_COUNT + fld) * kShots + k));
        for (int k = 0; k < kPuffs; k++)
            for (int fld : {PF_X, PF_Y, PF_LIFE}) v.push_back(rf(s_.pf, (size_t(env) * PF_COUNT + fld) * kPuffSlots + k));
        for (int e = 0; e < n_ent; e++) {
            if (e == 1) continue;
            uint8_t info;
            hipMemcpy(&info, s_.eb + (size_t(env) * EB_COUNT + EB_INFO) * kEntStride + e, 1, hipMemcpyDeviceToHost);
            auto ef = [&](int field) { return rf(s_.ef, (size_t(env) * EF_COUNT + field) * kEntStride + e); };
            v.push_back((info & kAlive) ? 1.0f : 0.0f);
            v.push_back(static_cast<float>(info & kKindMask));
            v.push_back(ef(EF_X));
            v.push_back(ef(EF_Y));
            v.push_back(ef(EF_VX));
            v.push_back(ef(EF_VY));
        }
        const int m = cap < static_cast<int>(v.size()) ? cap : static_cast<int>(v.size());
        for (int k = 0; k < m; k++) out[k] = v[k];
        return static_cast<int>(v.size());
    }
    int dump_tiles(hipStream_t st, int env, uint8_t* out, int cap) override {
        hipStreamSynchronize(st);
        const int m = cap < kCells ? cap : kCells;
        hipMemcpy(out, s_.tiles + size_t(env) * kTileStride, m, hipMemcpyDeviceToHost);
        return m;
    }

   private:
    State s_{};
    AtlasView atlas_{};
};

}  // namespace caveflyer

}  // namespace PG_VARIANT_NS

std::unique_ptr<Game> PG_FACTORY(make_caveflyer)() { return std::make_unique<PG_VARIANT_NS::caveflyer::CaveflyerGame>(); }

}  // namespace pg
