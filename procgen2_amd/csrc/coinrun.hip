// coinrun on gfx950: level generation, physics and rendering for N concurrent envs.
//
// Reference:
//   step   games/coinrun/coinrun.cpp:341-391, common_systems.cpp:7-39,65-105,121-252,284-313
//   render games/coinrun/coinrun.cpp:443-470, tilemap.cpp:294-321, common_systems.cpp:41-63,254-278,315-337
//   reset  games/coinrun/coinrun.cpp:472-507, tilemap.cpp:52-94,97-292
//
// Mapping onto the machine:
//   * logic (reset + the 4 physics sub-steps) is SIMT across envs — one lane per env, state laid out
//     struct-of-arrays across envs so the 64 lanes of a wave touch 64 consecutive floats;
//   * render is SIMD within an env — two wavefronts per env sharing a 64×64 target in LDS (pg_render.h).
// The ECS of the reference is gone: an entity is a slot index (= the id the reference's allocator hands
// out: saws/mobs in creation order, then the coin; the agent is kept apart), and the iteration order of the
// reference's std::unordered_set-based systems is recomputed at reset with pg_order.h and stored as two
// small permutations (sprite draw order, particle-owner order).
//
// One lane per env gives only 1 024 waves at 65 536 envs (one per SIMD), so the step is cut into parts that expose more
// parallelism and few dependent memory round trips, while producing the reference's sub-step interleaving exactly:
//   A. the agents (logic_kernel row 0, lane = env): the agent's four sub-steps — they depend on tiles only — leaving a
//      snapshot per sub-step in a scratch table;
//   B. the entities (logic_kernel rows 2…, lane = env, row = entity slot — the SAME launch: they do not react to the
//      agent): every entity simulates all its sub-steps (mob walk + particles + animation) with its state in registers;
//      tile lookups of a whole collide() come from a 4×4 window fetched in one go and packed 3 bits per cell into a
//      64-bit word; results go to the *other* half of a double-buffered entity table, a mob's x per sub-step to scratch;
//   C. resolve_kernel (lane = env): holds the hazards' boxes against the agent's, sub-step by sub-step; the first sub-step
//      that terminates (hazard, lava, coin) decides how many sub-steps really happened; in the rare case that is fewer
//      than four, that env's entities are redone from the untouched half with that limit.  Writes reward / done and
//      commits the agent.
#include "../../include/procgen2_vec.h"
#include "pg_engine.h"
#include "pg_frame.h"
#include "pg_geom.h"
#include "pg_order.h"
#include "pg_prefetch.h"
#include "pg_prepass.h"
#include "pg_render.h"
#include "pg_rng.h"
#include "pg_tiles.h"  // collide_any

namespace pg {
namespace PG_VARIANT_NS {
namespace coinrun {

constexpr int W = 64, H = 64;
constexpr int kMaxEnt = 36;  // 5 sections × max 7-wide pit of saws/mobs + the coin (tilemap.cpp:126-131,176-186)
constexpr int kSparks = 10;  // tilemap.cpp:91
constexpr int kSparkRow = 12; // (a row of ten, padded to 48 bytes)

enum Tile : uint8_t { kEmpty = 0, kWallTop, kWallMid, kLavaTop, kLavaMid, kCrate };  // tilemap.h:13-21
enum Solid { kPass = 0, kFull, kOneWay };                                            // tilemap.h:23-27
enum Kind { kSaw = 0, kMob = 1, kCoin = 2 };

// Atlas order (texture_names() below must match).
enum Tex {
    kTexTop = 0,      // 6 themes: <theme>Mid.png    (tile wall_top)
    kTexMid = 6,      // 6 themes: <theme>Center.png (tile wall_mid)
    kTexLavaTop = 12,
    kTexLava = 13,
    kTexCrate = 14,   // 4
    kTexWalker = 18,  // 9 × {still, move}
    kTexSaw = 36,     // 2
    kTexCoin = 38,
    kTexStand = 39,   // 5 each
    kTexJump = 44,
    kTexWalk1 = 49,
    kTexWalk2 = 54,
    kTexSpark = 59,
    kTexBackdrop = 60,  // 49
    kTexCount = 109
};

// scalar float fields
enum { F_AX, F_AY, F_AVX, F_AVY, F_APHASE, F_CAMX, F_CAMY, F_BGSHIFT, F_COUNT };
// scalar int fields
enum { I_FLAGS, I_THEMES, I_NENT, I_NMOB, I_HASH_SPRITE, I_HASH_SPARK, I_COUNT };
constexpr int kFlagGround = 1, kFlagForward = 2, kFlagListed = 4, kFlagBuf = 8;  // kFlagBuf: live half of the table
// static per-entity bytes
enum { EB_KIND, EB_TEX, EB_DRAW_ORDER, EB_SPARK_ORDER, EB_COUNT };
// dynamic (double-buffered) per-entity floats
enum { DF_X, DF_VX, DF_ANIM_T, DF_SPAWN_T, DF_COUNT };
// dynamic per-entity byte: animation frame | flip_x | texture assigned
constexpr int kDynFrame = 1, kDynFlip = 2, kDynTexSet = 4;

// One generated level, as the generator leaves it in LDS and as it waits in the shadow slot (pg_prefetch.h).
struct Level {
    uint8_t tiles[W * H];
    float bgshift;
    int32_t themes, n_ent, n_mob;
    float ey[kMaxEnt], x[kMaxEnt], vx[kMaxEnt];
    uint8_t kind[kMaxEnt], tex[kMaxEnt], draw_order[kMaxEnt], spark_order[kMaxEnt];
};

struct State {
    int n;
    Level* shadow;   // [n]  next level of each env
    int32_t* slot;   // [n]  SlotState
    uint32_t* mt;    // [n][625]  generator chain: the stream position after the newest generated level
    uint8_t* tiles;  // [n][4096]   tile id | crate kind << 4, column-major y + x*H (tilemap.h:62-85)
    float* f;        // [F_COUNT][n]
    int32_t* i;      // [I_COUNT][n]
    float* ey;       // [kMaxEnt][n]                      entity y (never changes within an episode)
    uint8_t* eb;     // [EB_COUNT][kMaxEnt][n]
    float* df;       // [2][DF_COUNT][kMaxEnt][n]         double-buffered
    uint8_t* db;     // [2][kMaxEnt][n]
    float* spark;    // [2][n][kMaxEnt][3][kSparkRow]     x, y, life — a mob's thirty values are 144 contiguous bytes: the entity
                     //                                   kernel's (env, mob) lanes and the render wave's spark lanes both read a
                     //                                   couple of cache lines instead of one per value
    float* scratch;  // [SC_COUNT][n]                     hand-off between the logic kernels of a step
    float4* hazx;    // [kMaxEnt][n]                      a hazard's box after each of the four sub-steps: its left edges;
    float2* hazy;    // [kMaxEnt][n]                      top edge and height.  Left by the entity lanes that come near the
                     //                                   agent (hazard_near) for resolve_kernel
    uint32_t no;     // generator switches turned off (PGV_COINRUN_NO_*: coinrun/tilemap.h:42-45 allow_* = false)
    PrepOut prep;    // what setup_kernel leaves for render_kernel (pg_prepass.h); not part of the state blob
};

// scratch rows: the agent after each of the 4 sub-steps, then one word of flags
enum {
    SC_AX = 0, SC_AY = 4, SC_AVX = 8, SC_AVY = 12, SC_PHASE = 16, SC_BITS = 20, SC_REDO = 21, SC_CAND = 22, SC_COUNT = 24
};
// SC_BITS (int): per sub-step ss: ground 1<<ss, forward 1<<(4+ss), lava 1<<(8+ss), coin 1<<(12+ss); bit 31: this env
// stepped (as opposed to: performed its auto-reset) in the current vector step; bit 30 (kBitsFar): see hazard_near.
// SC_CAND, SC_CAND + 1 (int): bit e — entity slot e left its hazard boxes for resolve_kernel, which clears the words.
// SC_REDO (int): n > 0 — the step ended after n < 4 sub-steps; the entities (advanced by four, optimistically) are
// recomputed for n sub-steps from the untouched half of the table by the env's render wavefront, one lane per entity.

PG_D float& SF(const State& s, int field, int env) { return s.f[size_t(field) * s.n + env]; }
PG_D int32_t& SI(const State& s, int field, int env) { return s.i[size_t(field) * s.n + env]; }
PG_D float& EY(const State& s, int e, int env) { return s.ey[size_t(e) * s.n + env]; }
PG_D uint8_t& EB(const State& s, int field, int e, int env) { return s.eb[(size_t(field) * kMaxEnt + e) * s.n + env]; }
PG_D float& DF(const State& s, int buf, int field, int e, int env) {
    return s.df[((size_t(buf) * DF_COUNT + field) * kMaxEnt + e) * s.n + env];
}
PG_D uint8_t& DB(const State& s, int buf, int e, int env) { return s.db[(size_t(buf) * kMaxEnt + e) * s.n + env]; }
PG_D float& SC(const State& s, int row, int env) { return s.scratch[size_t(row) * s.n + env]; }
PG_D int32_t& SCI(const State& s, int row, int env) {
    return reinterpret_cast<int32_t*>(s.scratch)[size_t(row) * s.n + env];
}
PG_D float& SP(const State& s, int buf, int comp, int e, int k, int env) {
    return s.spark[(((size_t(buf) * s.n + env) * kMaxEnt + e) * 3 + comp) * kSparkRow + k];
}

// The tile map under construction (in LDS), written by the whole wavefront: a rectangle fill is spread over the
// lanes, and since later fills overwrite earlier ones every operation ends with a barrier.
struct TileMap {
    uint8_t* t;  // 4096 cells
    int lane;
    PG_D void put(int x, int y, int id) {
        if (lane == 0 && !(x < 0 || y < 0 || x >= W || y >= H)) t[y + x * H] = static_cast<uint8_t>(id);
        __syncthreads();
    }
    PG_D void fill(int x, int y, int w, int h, int id) {
        const int total = (w > 0 && h > 0) ? w * h : 0;
        for (int k = lane; k < total; k += 64) {
            const int a = udiv_small(k, h), b = k - a * h;
            const int px = x + a, py = y + b;
            if (!(px < 0 || py < 0 || px >= W || py >= H)) t[py + px * H] = static_cast<uint8_t>(id);
        }
        __syncthreads();
    }
    PG_D void fill_capped(int x, int y, int w, int h, int body, int cap) {
        fill(x, y, w, h - 1, body);
        fill(x, y + h - 1, w, 1, cap);
    }
};

struct GenLds {
    uint32_t mt[kMtWords];
};

// ------------------------------------------------------------------------------------------------
// reset: coinrun.cpp:472-507 + tilemap.cpp:97-292.  Every lane walks the builder (the draws are wave-uniform), lane 0
// records the entities.
// ------------------------------------------------------------------------------------------------
struct LevelBuilder {
    Level& lv;
    uint32_t* mt;
    int lane;
    TileMap map;
    uint32_t no;  // PGV_COINRUN_NO_* (wave-uniform); a switched-off feature also skips its draws, as `&&` / `?:` do
    int n_ent = 0, n_mob = 0;

    PG_D int spawn(float x, float y, int kind, int tex, float vx) {
        const int e = n_ent++;
        if (lane == 0) {
            lv.ey[e] = y;
            lv.kind[e] = static_cast<uint8_t>(kind);
            lv.tex[e] = static_cast<uint8_t>(tex);
            lv.x[e] = x;
            lv.vx[e] = vx;
        }
        return e;
    }
    PG_D void add_saw(int x, int y) {  // tilemap.cpp:52-68
        spawn(static_cast<float>(x) + 0.5f, static_cast<float>(H - 1 - y) + 0.5f, kSaw, kTexSaw, 0.0f);
    }
    PG_D void add_mob(int x, int y) {  // tilemap.cpp:70-94
        const int which = wave_rng_int(mt, 0, 8, lane);
        const float vx = 0.15f * ((wave_rng_real(mt, 0.0f, 1.0f, lane) < 0.5f) * 2.0f - 1.0f);
        const int e = spawn(static_cast<float>(x) + 0.5f, static_cast<float>(H - 1 - y) + 0.5f, kMob,
                            kTexWalker + 2 * which, vx);
        if (lane == 0) lv.spark_order[n_mob] = static_cast<uint8_t>(e);  // creation order; permuted below
        n_mob++;
    }

    PG_D void build() {
        const float max_jump = 1.5f, gravity = 0.2f, max_speed = 0.5f;
        map.fill(0, 0, W, H, kEmpty);
        map.fill(0, 0, W, 1, kWallTop);
        map.fill(0, 0, 1, H, kWallMid);
        map.fill(W - 1, 0, 1, H, kWallMid);
        map.fill(0, H - 1, W, 1, kWallMid);

        const int difficulty = wave_rng_int(mt, 1, 3, lane);
        const int sections = wave_rng_int(mt, difficulty, 2 * difficulty - 1, lane);
        int cx = 5, cy = 1;
        const int pit_thresh = difficulty;
        const int danger = wave_rng_int(mt, 0, 2, lane);

        const float reach_x = max_speed * 2.0f * max_jump / gravity;
        const float reach_y = max_jump * max_jump / (2.0f * gravity);
        const int max_dx = static_cast<int>(reach_x - 0.5f);
        const int max_dy = static_cast<int>(reach_y - 0.5f);

        for (int sec = 0; sec < sections; sec++) {
            if (cx + 15 >= W) break;
            const int bump = difficulty / 3;
            int dy = (no & PGV_COINRUN_NO_DY) ? 0 : wave_rng_int(mt, 1 + bump, 4 + bump, lane);  // tilemap.cpp:158
            dy = dy < max_dy ? dy : max_dy;
            if (cy >= 20 || (cy >= 5 && wave_rng_real(mt, 0.0f, 1.0f, lane) < 0.5f)) dy = -dy;
            const int dx = wave_rng_int(mt, 3 + bump, 2 * difficulty + 2 + bump, lane);
            cy = (cy + dy) > 1 ? (cy + dy) : 1;

            const bool pit = !(no & PGV_COINRUN_NO_PIT) && (dx > 7) && (cy > 3) &&  // tilemap.cpp:174
                             (wave_rng_int(mt, 0, 19, lane) >= pit_thresh);
            if (pit) {
                int x1 = wave_rng_int(mt, 1, 3, lane);
                int x2 = wave_rng_int(mt, 1, 3, lane);
                int gap = dx - x1 - x2;
                if (gap > max_dx) {
                    gap = max_dx;
                    x2 = dx - x1 - gap;
                }
                map.fill_capped(cx, 0, x1, cy, kWallMid, kWallTop);
                map.fill_capped(cx + dx - x2, 0, x2, cy, kWallMid, kWallTop);
                const int lava_h = wave_rng_int(mt, 1, cy - 3, lane);
                if (danger == 0) {
                    map.fill_capped(cx + x1, 1, gap, lava_h, kLavaMid, kLavaTop);
                } else if (danger == 1) {
                    for (int k = 0; k < gap; k++) add_saw(cx + x1 + k, 1);
                } else {
                    for (int k = 0; k < gap; k++) add_mob(cx + x1 + k, 1);
                }
                if (gap > 4) {
                    int x3, w1;
                    if (gap == 5) {
                        x3 = wave_rng_int(mt, 1, 2, lane);
                        w1 = wave_rng_int(mt, 1, 2, lane);
                    } else if (gap == 6) {
                        x3 = wave_rng_int(mt, 1, 2, lane) + 1;
                        w1 = wave_rng_int(mt, 1, 2, lane);
                    } else {
                        x3 = wave_rng_int(mt, 1, 2, lane) + 1;
                        const int x4 = wave_rng_int(mt, 1, 2, lane) + 1;
                        w1 = gap - x3 - x4;
                    }
                    map.fill_capped(cx + x1 + x3, cy - 1, w1, 1, kWallMid, kWallTop);
                }
            } else {
                map.fill_capped(cx, 0, dx, cy, kWallMid, kWallTop);
                int ob1 = -1;
                const int ob2 = -1;
                if (wave_rng_int(mt, 0, 9, lane) < 2 * difficulty && dx > 3) {
                    ob1 = cx + wave_rng_int(mt, 1, dx - 2, lane);
                    add_saw(ob1, cy);
                }
                if (!(no & PGV_COINRUN_NO_MOBS) && wave_rng_int(mt, 0, 9, lane) < difficulty && dx > 3 &&  // :250
                    max_dx >= 4) {
                    ob1 = cx + wave_rng_int(mt, 1, dx - 2, lane);
                    add_mob(ob1, cy);
                }
                for (int k = 0; k < ((no & PGV_COINRUN_NO_CRATE) ? 0 : 2); k++) {  // tilemap.cpp:258
                    const int crate_x = cx + wave_rng_int(mt, 1, dx - 2, lane);
                    if (wave_rng_real(mt, 0.0f, 1.0f, lane) < 0.5f && ob1 != crate_x && ob2 != crate_x) {
                        const int pile = wave_rng_int(mt, 1, 3, lane);
                        for (int j = 0; j < pile; j++) {
                            // crate_dist is drawn for every pile cell (tilemap.cpp:263)
                            const int kind = wave_rng_int(mt, 0, 3, lane);
                            map.put(crate_x, cy + j, kCrate | (kind << 4));
                        }
                    }
                }
            }
            cx += dx;
        }
        spawn(static_cast<float>(cx) + 0.5f, static_cast<float>(H - 1 - cy) + 0.5f, kCoin, kTexCoin, 0.0f);
        map.fill_capped(cx, 0, 1, cy, kWallMid, kWallTop);
        map.fill(cx + 1, 0, W - cx, H, kWallMid);
    }
};

// Iteration order of a System's std::unordered_set after inserting `ids[0..n)` in that order into a
// set that was clear()ed but kept its bucket array (packed = buckets | next_resize << 16).
PG_D void episode_order(int32_t& packed, const uint8_t* ids, int n, uint8_t* out) {
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
    for (int k = 0; k < n; k++) hash_insert(h, ids[k]);
    int16_t p = static_cast<int16_t>(h.head);
    for (int k = 0; k < n; k++) {
        out[k] = static_cast<uint8_t>(p);
        p = next[p];
    }
    packed = h.buckets | (h.next_resize << 16);
}

// reset() for one env by one wavefront: advances the env's generator chain (s.mt, the two bucket-count words) and
// leaves the level in `lv` (LDS).
PG_D void generate(const State& s, int env, GenLds& L, Level& lv, bool reseed, uint32_t seed, int lane) {
    uint32_t* gmt = s.mt + size_t(env) * kMtWords;
    if (reseed) {
        if (lane == 0) mt_seed(L.mt, seed);
    } else {
        for (int k = lane; k < kMtWords; k += 64) L.mt[k] = gmt[k];
    }
    __syncthreads();
    uint32_t* mt = L.mt;
    LevelBuilder lb{lv, mt, lane, TileMap{lv.tiles, lane}, s.no};
    lb.build();
    const int backdrop = wave_rng_int(mt, 0, 48, lane);
    const float shift = wave_rng_real(mt, 0.0f, 1.0f, lane);
    const int alien = wave_rng_int(mt, 0, 4, lane);
    const int ground = wave_rng_int(mt, 0, 5, lane);
    if (lane == 0) {
        lv.bgshift = shift;
        lv.themes = backdrop | (alien << 8) | (ground << 16);
        lv.n_ent = lb.n_ent;
        lv.n_mob = lb.n_mob;
        // T3/T4: sprite set = every non-agent entity in creation order; particle set = the mobs.
        uint8_t ids[kMaxEnt], order[kMaxEnt];
        for (int k = 0; k < lb.n_ent; k++) ids[k] = static_cast<uint8_t>(k);
        int32_t packed = SI(s, I_HASH_SPRITE, env);
        episode_order(packed, ids, lb.n_ent, order);
        SI(s, I_HASH_SPRITE, env) = packed;
        ZItem items[kMaxEnt];
        for (int k = 0; k < lb.n_ent; k++) items[k] = {1.0f, order[k]};  // every coinrun sprite has z = 1
        sort_by_key(items, lb.n_ent);
        for (int k = 0; k < lb.n_ent; k++) lv.draw_order[k] = static_cast<uint8_t>(items[k].id);

        for (int k = 0; k < lb.n_mob; k++) ids[k] = lv.spark_order[k];
        packed = SI(s, I_HASH_SPARK, env);
        episode_order(packed, ids, lb.n_mob, order);
        SI(s, I_HASH_SPARK, env) = packed;
        for (int k = 0; k < lb.n_mob; k++) lv.spark_order[k] = order[k];
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
    const int buf = (SI(s, I_FLAGS, env) & kFlagBuf) ? 1 : 0;  // the live half of the dynamic table stays where it is
    const int n_ent = lv.n_ent, n_mob = lv.n_mob;
    if (lane < n_ent) {
        const int e = lane;
        const int kind = lv.kind[e];
        EY(s, e, env) = lv.ey[e];
        EB(s, EB_KIND, e, env) = static_cast<uint8_t>(kind);
        EB(s, EB_TEX, e, env) = lv.tex[e];
        EB(s, EB_DRAW_ORDER, e, env) = lv.draw_order[e];
        DF(s, buf, DF_X, e, env) = lv.x[e];
        DF(s, buf, DF_VX, e, env) = lv.vx[e];
        DF(s, buf, DF_ANIM_T, e, env) = 0.0f;
        DF(s, buf, DF_SPAWN_T, e, env) = 0.0f;
        DB(s, buf, e, env) = static_cast<uint8_t>(kind == kCoin ? kDynTexSet : 0);  // D10: animated sprites start unset
        if (kind == kMob)
            for (int k = 0; k < kSparks; k++) {
                SP(s, buf, 0, e, k, env) = 0.0f;
                SP(s, buf, 1, e, k, env) = 0.0f;
                SP(s, buf, 2, e, k, env) = 0.0f;
            }
    }
    if (lane < n_mob) EB(s, EB_SPARK_ORDER, lane, env) = lv.spark_order[lane];
    __syncthreads();  // I_FLAGS is read above by every lane
    if (lane == 0) {
        SF(s, F_AX, env) = 1.5f;
        SF(s, F_AY, env) = H - 1 - 1.0f;
        SF(s, F_AVX, env) = 0.0f;
        SF(s, F_AVY, env) = 0.0f;
        SF(s, F_APHASE, env) = 0.0f;
        SF(s, F_BGSHIFT, env) = lv.bgshift;
        // on_ground=false, face_forward=true, draw list cleared (D2); the live half of the table is unchanged
        SI(s, I_FLAGS, env) = kFlagForward | (buf ? kFlagBuf : 0);
        SI(s, I_THEMES, env) = lv.themes;
        SI(s, I_NENT, env) = n_ent;
        SI(s, I_NMOB, env) = n_mob;
        // camera keeps the previous episode's value (D3)
    }
}

// What cenv_make leaves in an env besides the seeded RNG, split by owner: the generator chain (bucket counts of
// the sets that survive clear()) and the live state.  Level-seed mode (pg_engine.h LevelPlan) rebuilds every
// level from here.
PG_D void fresh_chain(const State& s, int env) {
    SI(s, I_HASH_SPRITE, env) = 1;  // empty unordered_set: one bucket, next_resize 0
    SI(s, I_HASH_SPARK, env) = 1;
}
PG_D void fresh_live(const State& s, int env) {
    SI(s, I_FLAGS, env) = 0;
    SF(s, F_CAMX, env) = 0.0f;  // Renderer::camera_position{0} (renderer.h:18)
    SF(s, F_CAMY, env) = 0.0f;
}

struct Gen {  // pg_prefetch.h level_kernel<Gen>
    using State = coinrun::State;
    using Level = coinrun::Level;
    using GenLds = coinrun::GenLds;
    PG_D static void generate(const State& s, int env, GenLds& L, Level& lv, bool reseed, uint32_t seed, int lane) {
        coinrun::generate(s, env, L, lv, reseed, seed, lane);
    }
    PG_D static void install(const State& s, int env, const Level& lv, int lane) { coinrun::install(s, env, lv, lane); }
    PG_D static void fresh_chain(const State& s, int env) { coinrun::fresh_chain(s, env); }
    PG_D static void fresh_live(const State& s, int env) { coinrun::fresh_live(s, env); }
};

// ------------------------------------------------------------------------------------------------
// tile collision, variant A (tilemap.cpp:323-396)
// ------------------------------------------------------------------------------------------------
// 4×4 window of tile ids in collide() coordinates (x, y ↦ tile (x, H-1-y)), 3 bits per cell in one 64-bit
// word: the whole window is fetched with 16 independent byte loads (one memory round trip), after which every
// lookup of a collide() is a shift and a mask.  Cells outside the window fall back to a direct load, so the
// window's placement affects speed only.
struct TileWin {
    const uint8_t* tiles;
    int ax, ay;
    uint64_t bits;

    PG_D static int direct(const uint8_t* tiles, int x, int y) {
        const int ty = H - 1 - y;
        if (x < 0 || ty < 0 || x >= W || ty >= H) return kWallMid;  // out of bounds is a wall (tilemap.h:80-81)
        return tiles[ty + x * H] & 7;
    }
    PG_D static TileWin fetch(const uint8_t* tiles, int ax, int ay) {
        TileWin w{tiles, ax, ay, 0};
        // A column of the window is four consecutive bytes of the map (index ty + x·H, ty = H − 1 − y): four loads of a
        // word — at any byte address, which global memory allows — instead of sixteen of a byte.  UNCONDITIONAL loads, so
        // that they are all in flight together (a load behind the bounds test is a load in a branch of its own, and the
        // compiler waits for each before it enters the next): a column beyond the map reads column 0, a word that would
        // stick out above or below is read from the nearest place inside and shifted, and a cell outside the map is
        // replaced afterwards.
        const int ty_lo = H - 4 - ay;  // the window's row ay + 3; row ay + j sits in byte 3 − j of the column's word
        const int ty_at = ty_lo < 0 ? 0 : (ty_lo > H - 4 ? H - 4 : ty_lo);
        const int delta = ty_at - ty_lo;  // ∈ [−3, 3] where any cell is inside
        uint32_t col[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int x = ax + c;
            uint32_t word;
            __builtin_memcpy(&word, tiles + ty_at + ((x < 0 || x >= W) ? 0 : x) * H, 4);
            col[c] = word;
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int x = ax + c;
            const uint32_t sh = 8u * static_cast<uint32_t>(delta < 0 ? -delta : delta);
            const uint32_t word = sh >= 32u ? 0u : (delta >= 0 ? col[c] << sh : col[c] >> sh);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int ty = H - 1 - (ay + j);
                const bool inside = !(x < 0 || ty < 0 || x >= W || ty >= H);
                const int tile = inside ? static_cast<int>((word >> (8 * (3 - j))) & 7u) : kWallMid;  // tilemap.h:80-81
                w.bits |= static_cast<uint64_t>(tile) << (3 * (c + 4 * j));
            }
        }
        return w;
    }
    PG_D int cell(int dx, int dy) const { return static_cast<int>((bits >> (3 * (dx + 4 * dy))) & 7u); }  // inside the window
    PG_D int at(int x, int y) const {
        const unsigned dx = static_cast<unsigned>(x - ax), dy = static_cast<unsigned>(y - ay);
        if (dx < 4u && dy < 4u) return static_cast<int>((bits >> (3 * (dx + 4 * dy))) & 7u);
        return direct(tiles, x, y);
    }
    // Does the window hold every tile collide() scans for box r (floor(x)..ceil(x+w) × floor(y)..ceil(y+h))?
    PG_D bool holds(const Box& r) const {
        const int x0 = static_cast<int>(floorf(r.x)), y0 = static_cast<int>(floorf(r.y));
        const int x1 = static_cast<int>(ceilf(r.x + r.w)), y1 = static_cast<int>(ceilf(r.y + r.h));
        return x0 >= ax && y0 >= ay && x1 < ax + 4 && y1 < ay + 4;
    }
    // The window for box r with its spare column / row on the side the box is heading (dir_x, dir_y: velocity signs).
    PG_D static TileWin around(const uint8_t* tiles, const Box& r, float dir_x, float dir_y) {
        const int x0 = static_cast<int>(floorf(r.x)), y0 = static_cast<int>(floorf(r.y));
        return fetch(tiles, x0 - (dir_x < 0.0f ? 1 : 0), y0 - (dir_y < 0.0f ? 1 : 0));
    }
};

struct Hit {
    float x, y;
    bool any;
};

// (box_overlap_flat — get_collision_overlap without its early return — is pg_tiles.h's)
template <class Pred>
PG_D Hit collide(const TileWin& win, Box r, Pred solid, bool fallthrough, float step_y) {
    bool any = false;
    const int x0 = static_cast<int>(floorf(r.x)), y0 = static_cast<int>(floorf(r.y));
    const int x1 = static_cast<int>(ceilf(r.x + r.w)), y1 = static_cast<int>(ceilf(r.y + r.h));
    const float mid_x = r.x + r.w * 0.5f, mid_y = r.y + r.h * 0.5f;
    Box cell{0.0f, 0.0f, 1.0f, 1.0f};
    if ((x1 - x0 <= 2) & (y1 - y0 <= 2) & (x0 >= win.ax) & (y0 >= win.ay) & (x1 < win.ax + 4) & (y1 < win.ay + 4)) {
        // The boxes of this game are at most a tile wide and high: three by three cells at most, and the callers keep
        // them inside the window.  Walked as nine fixed steps with every decision a select (bitwise, so that nothing
        // short-circuits into a branch) — same cells, same order, same arithmetic as the loops below — because this
        // runs in the logic kernels, one wavefront per 64 envs with its SIMD to itself: there the time goes into the
        // taken branches of the loops (a dozen per cell), not into the arithmetic.
        const int wx = x0 - win.ax, wy = y0 - win.ay;  // 0 … 3: the box's first cell inside the window
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                const bool in = (x0 + dx <= x1) & (y0 + dy <= y1);
                const int ox_ = in ? dx : 0, oy_ = in ? dy : 0;  // (a cell past the box: look at the first one, ignore it)
                const int kind = solid(static_cast<int>((win.bits >> (3 * ((wx + ox_) + 4 * (wy + oy_)))) & 7u));
                cell.x = static_cast<float>(x0 + ox_);
                cell.y = static_cast<float>(y0 + oy_);
#if PG_WALK_SKIP
                if (__ballot(in & (kind != kPass) & box_hit(r, cell)) == 0) continue;  // (pg_tiles.h: a cell nobody's box meets takes nobody)
#endif
                const Box o = box_overlap_flat(r, cell);
                const float oy = o.y + o.h * 0.5f;
                const bool inside = (r.y + r.h - step_y > cell.y);
                const bool one_way_ok = (step_y > 0.01f) & !fallthrough & !inside;
                const bool take = in & (kind != kPass) & !((o.w == 0.0f) & (o.h == 0.0f)) & (o.w > o.h) &
                                  ((kind != kOneWay) | one_way_ok);
                r.y = take ? (oy > mid_y ? cell.y - r.h : cell.y + cell.h) : r.y;
                any = any | take;
            }
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                const bool in = (x0 + dx <= x1) & (y0 + dy <= y1);
                const int ox_ = in ? dx : 0, oy_ = in ? dy : 0;
                const int kind = solid(static_cast<int>((win.bits >> (3 * ((wx + ox_) + 4 * (wy + oy_)))) & 7u));
                cell.x = static_cast<float>(x0 + ox_);
                cell.y = static_cast<float>(y0 + oy_);
#if PG_WALK_SKIP
                if (__ballot(in & (kind != kPass) & box_hit(r, cell)) == 0) continue;
#endif
                const Box o = box_overlap_flat(r, cell);
                const float ox = o.x + o.w * 0.5f;
                const bool take = in & (kind != kPass) & !((o.w == 0.0f) & (o.h == 0.0f)) & (o.w <= o.h) & (kind != kOneWay);
                r.x = take ? (ox > mid_x ? cell.x - r.w : cell.x + cell.w) : r.x;
                any = any | take;
            }
        return {r.x, r.y, any};
    }
    for (int y = y0; y <= y1; y++)
        for (int x = x0; x <= x1; x++) {
            const int kind = solid(win.at(x, y));
            if (kind == kPass) continue;
            cell.x = static_cast<float>(x);
            cell.y = static_cast<float>(y);
            const Box o = box_overlap(r, cell);
            if (o.w == 0.0f && o.h == 0.0f) continue;
            const float oy = o.y + o.h * 0.5f;
            if (o.w > o.h) {
                if (kind == kOneWay) {
                    const bool inside = (r.y + r.h - step_y > cell.y);
                    if (step_y > 0.01f && !fallthrough && !inside) {
                        r.y = (oy > mid_y ? cell.y - r.h : cell.y + cell.h);
                        any = true;
                    }
                } else {
                    r.y = (oy > mid_y ? cell.y - r.h : cell.y + cell.h);
                    any = true;
                }
            }
        }
    for (int y = y0; y <= y1; y++)
        for (int x = x0; x <= x1; x++) {
            const int kind = solid(win.at(x, y));
            if (kind == kPass) continue;
            cell.x = static_cast<float>(x);
            cell.y = static_cast<float>(y);
            const Box o = box_overlap(r, cell);
            if (o.w == 0.0f && o.h == 0.0f) continue;
            const float ox = o.x + o.w * 0.5f;
            if (o.w <= o.h && kind != kOneWay) {
                r.x = (ox > mid_x ? cell.x - r.w : cell.x + cell.w);
                any = true;
            }
        }
    return {r.x, r.y, any};
}

// ------------------------------------------------------------------------------------------------
// step: the sub-step loop of cenv_step (coinrun.cpp:356-371)
// ------------------------------------------------------------------------------------------------
struct AgentSnap {  // agent + camera after one sub-step
    float ax, ay, avx, avy, phase, camx, camy;
    bool ground, forward;
};

// One entity, `limit` sub-steps, reading half `src` and writing half `1 - src` of the dynamic table.  Nothing here looks
// at the agent: the entities of coinrun do not react to it, and whether one of them, a hazard, overlaps it in a sub-step is
// decided by resolve_kernel (hazard_bits below) from hb: the left edge of the entity's hazard box after sub-step ss
// (ss < limit), [4] its top edge — 1e30 for the coin, no hazard — and [5] its height.  kStore = false: the boxes only,
// nothing written (resolve_kernel's fallback).
// The entities that never move — saws and the coin — from their loaded inputs (x, y, anim_t, dyn of half 1 - dst): the
// same writes and the same hazard box as ever; apart from entity_step so that logic_kernel's rows for them can ask memory
// for several entities' inputs at once (static_rows below).
template <bool kStore = true>
PG_D void static_step(const State& s, int env, int e, int kind, int dst, int limit, float x, float y, float anim_t, int dyn,
                      float (&hb)[6]) {
    const float dt = 1.0f / 4;
    if (kind == kCoin) {  // no dynamic state beyond the texture flag
        if constexpr (kStore) {
            DF(s, dst, DF_X, e, env) = x;
            DB(s, dst, e, env) = static_cast<uint8_t>(dyn);
        }
        hb[0] = hb[1] = hb[2] = hb[3] = hb[5] = 0.0f;
        hb[4] = 1e30f;
        return;
    }
    hb[0] = hb[1] = hb[2] = hb[3] = x + -0.5f;  // tilemap.cpp:66: Box{x - 0.5, y - 0.5, 1, 1}
    hb[4] = y + -0.5f;
    hb[5] = 1.0f;
#pragma unroll
    for (int ss = 0; ss < 4; ss++) {
        if (ss >= limit) break;
        // System_Sprite_Render::update (common_systems.cpp:14-29), rate 1.0 (tilemap.cpp:60)
        anim_t += dt;
        const int adv = static_cast<int>(anim_t * 1.0f);
        anim_t -= adv / 1.0f;
        dyn = ((dyn & ~kDynFrame) | ((((dyn & kDynFrame) ? 1 : 0) + adv) % 2 ? kDynFrame : 0)) | kDynTexSet;
    }
    if constexpr (kStore) {
        DF(s, dst, DF_X, e, env) = x;
        DF(s, dst, DF_ANIM_T, e, env) = anim_t;
        DB(s, dst, e, env) = static_cast<uint8_t>(dyn);
    }
}

template <bool kStore = true>
PG_D void entity_step(const State& s, int env, int e, int src, int limit, float (&hb)[6]) {
    const uint8_t* tiles = s.tiles + size_t(env) * (W * H);
    const int dst = 1 - src;
    const float dt = 1.0f / 4;
    const int kind = EB(s, EB_KIND, e, env);
    if (kind == kCoin) {
        static_step<kStore>(s, env, e, kind, dst, limit, DF(s, src, DF_X, e, env), 0.0f, 0.0f, DB(s, src, e, env), hb);
        return;
    }
    float x = DF(s, src, DF_X, e, env);
    const float y = EY(s, e, env);
    float anim_t = DF(s, src, DF_ANIM_T, e, env);
    int dyn = DB(s, src, e, env);
    if (kind == kSaw) {
        static_step<kStore>(s, env, e, kind, dst, limit, x, y, anim_t, dyn, hb);
        return;
    }
    // --- mob: System_Mob_AI (common_systems.cpp:65-105) + System_Particles (:284-313) + animation, rate 0.2
    float vx = DF(s, src, DF_VX, e, env);
    float timer = DF(s, src, DF_SPAWN_T, e, env);
    float sx[kSparks], sy[kSparks], sl[kSparks];
#pragma unroll
    for (int k = 0; k < kSparks; k++) {
        sx[k] = kStore ? SP(s, src, 0, e, k, env) : 0.0f;
        sy[k] = kStore ? SP(s, src, 1, e, k, env) : 0.0f;
        sl[k] = kStore ? SP(s, src, 2, e, k, env) : 0.0f;
    }
    // both sensors of all sub-steps live inside this window (x drifts by at most 0.15 per step)
    const TileWin win = TileWin::fetch(tiles, static_cast<int>(floorf(x - 0.66f)), static_cast<int>(floorf(y - 0.6f)));
#pragma unroll
    for (int ss = 0; ss < 4; ss++) {
        if (ss >= limit) break;
        x += vx * dt;
        const Box wall_probe{x - 0.5f, y - 0.6f, 1.0f, 0.5f};
        const Box floor_probe{x - 0.5f, y + 0.6f, 1.0f, 0.5f};
        // A probe that touches nothing comes back as it went (collide moves the box only where it takes a cell), and whether
        // it touches anything is collide_any's question (see the agent's lava probe): the walk itself — two passes over nine
        // cells — is made only in the sub-steps where some mob of the wavefront does turn round (round 6; a mob meets a
        // wall or an edge once in a few hundred sub-steps).  A probe outside the window takes the walk, which looks cells up
        // wherever they are.
        const bool wall_maybe = !win.holds(wall_probe) ||
                                pg::collide_any(win, wall_probe, [](int t) { return t == kWallMid || t == kWallTop; });
        const bool gap_maybe = !win.holds(floor_probe) || pg::collide_any(win, floor_probe, [](int t) { return t == kEmpty; });
        Hit wall{wall_probe.x, wall_probe.y, false}, gap{floor_probe.x, floor_probe.y, false};
        if (__ballot(wall_maybe))  // (wave-uniform)
            wall = collide(
                win, wall_probe, [](int t) { return (t == kWallMid || t == kWallTop) ? kFull : kPass; }, false, 0.0f);
        if (__ballot(gap_maybe))
            gap = collide(
                win, floor_probe, [](int t) { return t == kEmpty ? kFull : kPass; }, false, 0.0f);
        float nx = wall.x + 0.5f;
        if (gap.any) nx = gap.x + 0.5f;
        x = nx;
        if (wall.any || gap.any) vx *= -1.0f;
        dyn = (dyn & ~kDynFlip) | (vx > 0.0f ? kDynFlip : 0);

        hb[ss] = x + -0.5f;  // tilemap.cpp:89: Box{x - 0.5, y - 0.48, 1, 0.98}

        int dead = -1;
#pragma unroll
        for (int k = 0; k < kSparks; k++) {
            sl[k] -= dt;
            if (sl[k] <= 0.0f) dead = k;
        }
        timer += dt;
        if (dead != -1 && timer >= 0.5f) {
            timer = fmodf(timer, 0.5f);
#pragma unroll
            for (int k = 0; k < kSparks; k++)
                if (k == dead) {
                    sl[k] = 5.0f;
                    sx[k] = x + 0.0f;
                    sy[k] = y + 0.34f;
                }
        }
        anim_t += dt;
        const int adv = static_cast<int>(anim_t * 0.2f);
        anim_t -= adv / 0.2f;
        dyn = ((dyn & ~kDynFrame) | ((((dyn & kDynFrame) ? 1 : 0) + adv) % 2 ? kDynFrame : 0)) | kDynTexSet;
    }
    if constexpr (kStore) {
        DF(s, dst, DF_X, e, env) = x;
        DF(s, dst, DF_VX, e, env) = vx;
        DF(s, dst, DF_ANIM_T, e, env) = anim_t;
        DF(s, dst, DF_SPAWN_T, e, env) = timer;
        DB(s, dst, e, env) = static_cast<uint8_t>(dyn);
#pragma unroll
        for (int k = 0; k < kSparks; k++) {
            SP(s, dst, 0, e, k, env) = sx[k];
            SP(s, dst, 1, e, k, env) = sy[k];
            SP(s, dst, 2, e, k, env) = sl[k];
        }
    }
    hb[4] = y + -0.48f;
    hb[5] = 0.98f;
}

// Which hazards can touch the agent at all is decided before anyone knows where the agent goes: an entity lane compares
// its boxes with the agent's box at the START of the step, grown by kReachX / kReachY to every side, and only an entity
// that comes near leaves its boxes (hazx / hazy) and its bit in the env's candidate words (SC_CAND) for resolve_kernel —
// a few in a hundred.  The agent's lane checks the other half of the argument: that none of its four sub-steps took it
// further than that from where it started (a sub-step moves it by at most 0.125 × 0.3875 — but a collision sets it against
// a cell's edge, which no line of the code bounds by less than two cells).  Where that fails — it has not in any run —
// SC_BITS says so (kBitsFar) and resolve_kernel works every box of the env out again, from the half of the table the
// entity lanes read.  Either way the bits are the reference's.  kReachSlack: the comparisons are made in different float
// expressions on the two sides (a rounding error is 1e-6 of it).
constexpr float kReachX = 1.0f, kReachY = 2.0f, kReachSlack = 0.01f;
constexpr int kBitsFar = 1 << 30;

PG_D bool hazard_near(const float (&hb)[6], float ax0, float ay0, float reach_x, float reach_y) {
    // the agent's box: Box{ax - 0.5, ay - 1, 1, 1} (common_systems.cpp:171)
    const float x0 = ax0 - 0.5f - reach_x - kReachSlack, x1 = ax0 + 0.5f + reach_x + kReachSlack;
    const float y0 = ay0 - 1.0f - reach_y - kReachSlack, y1 = ay0 + reach_y + kReachSlack;
    if (!(hb[4] < y1 && hb[4] + hb[5] > y0)) return false;  // (the coin's 1e30 ends here)
    bool near = false;
#pragma unroll
    for (int ss = 0; ss < 4; ss++) near = near || (hb[ss] < x1 && hb[ss] + 1.0f > x0);
    return near;
}

// Bit ss: the hazard box hb overlaps the agent after sub-step ss (common_systems.cpp:241-251: the reference ORs over its
// hazard set, so set order is irrelevant — App. B).
PG_D int hazard_hits(const float (&hb)[6], const float (&bx)[4], const float (&by)[4]) {
    int hits = 0;
#pragma unroll
    for (int ss = 0; ss < 4; ss++)
        if (box_hit(Box{bx[ss], by[ss], 1.0f, 1.0f}, Box{hb[ss], hb[4], 1.0f, hb[5]})) hits |= 1 << ss;
    return hits;
}

// A: System_Agent::update ×4 (common_systems.cpp:121-252) minus the hazard loop, which needs the mobs.
PG_D void agent_substeps(const State& s, int env, int action, float reach_x, float reach_y) {
    const uint8_t* tiles = s.tiles + size_t(env) * (W * H);
    const int n_ent = SI(s, I_NENT, env);
    const int flags = SI(s, I_FLAGS, env);
    const int src = (flags & kFlagBuf) ? 1 : 0;
    float ax = SF(s, F_AX, env), ay = SF(s, F_AY, env);
    const float ax0 = ax, ay0 = ay;
    bool stayed = true;  // within reach of where the step began (hazard_near)
    float avx = SF(s, F_AVX, env), avy = SF(s, F_AVY, env);
    float phase = SF(s, F_APHASE, env);
    bool ground = (flags & kFlagGround) != 0, forward = (flags & kFlagForward) != 0;
    // the coin is the last entity created before the agent (tilemap.cpp:277-287); it never moves
    const Box coin_box{DF(s, src, DF_X, n_ent - 1, env) + -0.5f, EY(s, n_ent - 1, env) + -0.5f, 1.0f, 1.0f};

    const float dt = 1.0f / 4;
    const float max_jump = 1.55f, gravity = 0.2f, max_speed = 0.5f, mix = 0.2f, air_control = 0.15f;
    const float move_x = static_cast<float>((action == 6 || action == 7 || action == 8) -
                                            (action == 0 || action == 1 || action == 2));
    const bool jump = (action == 2 || action == 5 || action == 8);
    const bool drop = (action == 0 || action == 3 || action == 6);

    int bits = static_cast<int>(0x80000000u);
    TileWin win{tiles, 0, 0, 0};
#pragma unroll
    for (int ss = 0; ss < 4; ss++) {
        const float mix_x = ground ? mix : (mix * air_control);
        avx += mix_x * (max_speed * move_x - avx) * dt;
        if (fabsf(avx) < mix_x * max_speed * dt) avx = 0.0f;
        if (jump && ground) avy = -max_jump;
        avy += gravity * dt;
        if (fabsf(avy) > max_jump) avy = (avy > 0.0f ? 1.0f : -1.0f) * max_jump;
        ax += avx * dt;
        ay += avy * dt;

        Box b{ax + -0.5f, ay + -1.0f, 1.0f, 1.0f};
        // One window serves as long as the box stays inside it (a sub-step moves the agent by ≤ 0.125 × ≤ 0.39 units):
        // the eight dependent window fetches of a step were its eight memory round trips — now usually one or two.
        if (ss == 0 || !win.holds(b)) win = TileWin::around(tiles, b, avx, avy);
        const Hit h = collide(
            win, b, [](int t) { return (t == kWallMid || t == kWallTop) ? kFull : (t == kCrate ? kOneWay : kPass); },
            drop, avy * dt);
        const float moved_x = h.x - b.x, moved_y = h.y - b.y;
        ground = moved_y < 0.0f && h.any;
        ax = h.x - -0.5f;
        ay = h.y - -1.0f;
        b.x = ax + -0.5f;
        b.y = ay + -1.0f;
        if (moved_x != 0.0f) avx = 0.0f;
        if (ground) avy = 0.0f;

        if (!win.holds(b)) win = TileWin::around(tiles, b, avx, avy);
        // the lava probe is only asked WHETHER (common_systems.cpp:221-229): `any` is "some scanned lava cell has a non-empty
        // overlap with the box as it came" — the first cell taken is met by the box unmoved, and either pass takes every
        // such cell — which pg_tiles.h collide_any works out per axis, without the walk (round 6: a quarter of this lane's
        // instructions)
        const bool in_lava = pg::collide_any(win, b, [](int t) { return t == kLavaMid || t == kLavaTop; });

        phase += 0.1f * dt;
        phase = fmodf(phase, 1.0f);
        if (move_x > 0.0f)
            forward = true;
        else if (move_x < 0.0f)
            forward = false;

        stayed = stayed && fabsf(ax - ax0) <= reach_x && fabsf(ay - ay0) <= reach_y;
        SC(s, SC_AX + ss, env) = ax;
        SC(s, SC_AY + ss, env) = ay;
        SC(s, SC_AVX + ss, env) = avx;
        SC(s, SC_AVY + ss, env) = avy;
        SC(s, SC_PHASE + ss, env) = phase;
        bits |= (ground ? 1 << ss : 0) | (forward ? 1 << (4 + ss) : 0) | (in_lava ? 1 << (8 + ss) : 0) |
                (box_hit(b, coin_box) ? 1 << (12 + ss) : 0);
    }
    SCI(s, SC_BITS, env) = bits | (stayed ? 0 : kBitsFar);
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) make_kernel(State s, uint32_t seed_base, int env_offset) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= s.n) return;
    fresh_chain(s, env);
    fresh_live(s, env);
}

// A + B — the agent's sub-steps and the entities' in ONE launch (round 5): neither needs the other.  The agent depends on
// tiles, the entities on tiles and their own state; what joins them — does a hazard's box overlap the agent's after
// sub-step ss — is held by resolve_kernel, behind both.  As two launches they were two latency chains end to end (34 and
// 46 µs) and a kernel boundary; side by side the step pays for the longer one.  Rows of blocks (row = blockIdx.y):
//   row 0 — lane = env: the agent's sub-steps into the scratch table (envs that reset in this step sit it out);
//   row 1 — the auto-reset of those envs whose next level lies ready in its shadow slot (pg_prefetch.h install_prefetched:
//     a copy), beside the agents instead of in a launch in front of them.  The rows share nothing of a resetting env but
//     its pending byte, which the others only read and row 1 only ever turns from "due in this step" into "served in this
//     step" — a reset either way (pg_prefetch.h resets_in_step);
//   rows 2 + y — the entities.  One workgroup = one wavefront = a block of 64 envs.
//     y ≥ kMobRows: lane = env, entity ids kStaticPerLane · (y − kMobRows) … unless they are mobs (saws and coins: cheap,
//     coalesced; four to a lane since round 6: the 36 rows of one were 36 864 wavefronts of three dependent round trips).
//     y < kMobRows: the mobs.  Their path is ~5 000 instructions (tile window, two collision probes per
//     sub-step, sparks) and a level has between none and a dozen of them, so "row y = the y-th mob of every env" ran
//     that path max-over-the-block times with mostly idle lanes (SQ counters: 78 M wave instructions per launch, the
//     kernel is issue-bound).  Instead the (env, mob) pairs of the block are numbered densely — a wave prefix sum over
//     the 64 mob counts — and row y takes pairs [64y, 64y + 64): full waves, about mean-instead-of-max many of them.
// (Measured and rejected, round 5: a grid of 8 + 12 entity rows whose waves loop on to their next share of the pairs /
// entity slots instead of 2 × 36 rows of which nine in ten find nothing to do — 20 480 wavefronts instead of 73 728.
// Bit-exact and slower: 45.5 -> 79.0 µs.  The kernel is the length of its longest wave, a wave that exits early costs the
// dispatcher 0.2 ns, and a second share is a second chain of round trips behind the first.)
// (Wavefronts per SIMD the registers are capped for: the kernel wants 131, which is three; at 128 (three spilled) it is four — 66 -> 61 µs; five, at 96: 70 µs.)
#ifndef PG_COINRUN_LOGIC_WAVES
#define PG_COINRUN_LOGIC_WAVES 4
#endif
#ifndef PG_COINRUN_STATIC_PER_LANE
#define PG_COINRUN_STATIC_PER_LANE 4
#endif
constexpr int kStaticPerLane = PG_COINRUN_STATIC_PER_LANE;  // entity slots a lane of the saws' and coins' rows takes
constexpr int kStaticRows = (kMaxEnt + kStaticPerLane - 1) / kStaticPerLane;
constexpr int kMobRows = kMaxEnt;  // rows of 64 (env, mob) pairs a block of 64 envs is given: every entity of every env could be a mob
__global__ void __launch_bounds__(64, PG_COINRUN_LOGIC_WAVES) logic_kernel(State s, const int32_t* actions, uint32_t run_seed, uint32_t step_index,
                                                   int env_offset, StepIO io, int prefetch, LevelPlan plan, int install_row,
                                                   float reach_x, float reach_y) {
    const int row = static_cast<int>(blockIdx.y);
    const int lane = threadIdx.x;
    int env = blockIdx.x * 64 + lane;
    if (row == 1) {  // (block-uniform)
        __shared__ Level lv;
        if (install_row)
            install_prefetched<Gen>(s, blockIdx.x * 64, 64, prefetch, io, plan, lv, lane, reset_served_mark(step_index),
                                    reset_due_mark(step_index));
        return;
    }
    // the caller's `if term: env.reset()` (game_test.py:38-40): for such an env this step is its reset
    const bool stepping = env < s.n && !resets_in_step(io.pending[env], step_index);
    if (row == 0) {
        if (env >= s.n) return;
        if (!stepping) {
            SCI(s, SC_BITS, env) = 0;  // did not step: resolve_kernel leaves this env alone
            return;
        }
        const int action =
            actions ? actions[env] : synthetic_action(run_seed, step_index, static_cast<uint32_t>(env_offset + env));
        // (Measured, round 5: s_setprio(3) for this row — one long chain beside the entities' many shorter ones — changes
        // nothing: 59.5 against 59.4 µs.)
        agent_substeps(s, env, action, reach_x, reach_y);
        return;
    }
    const int y = row - 2;
    if (y < kMobRows) {
        // (Measured and rejected, round 6: twelve rows that stride on over the pairs instead of 36 of which nine in ten find
        // nothing — the loop around the mob's path alone, never taken twice, made the kernel 47.6 -> 52.0 µs with 36 rows,
        // and sixteen rows instead of 36 were worth 0.7 µs: the empty rows cost next to nothing.)
        const int count = stepping ? SI(s, I_NMOB, env) : 0;
        int upto = count;  // inclusive prefix sum over the block
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(upto, off);
            if (lane >= off) upto += t;
        }
        const int total = __shfl(upto, 63);
        const int pair = y * 64 + lane;
        if (y * 64 >= total) return;  // wave-uniform
        int lo = 0, hi = 63;  // owner of `pair`: the first lane whose inclusive sum exceeds it
#pragma unroll
        for (int it = 0; it < 6; it++) {
            const int mid = (lo + hi) >> 1;
            const bool right = __shfl(upto, mid) <= pair;
            lo = right ? mid + 1 : lo;
            hi = right ? hi : mid;
        }
        const int before = __shfl(upto, lo) - __shfl(count, lo);
        if (pair >= total) return;
        env = blockIdx.x * 64 + lo;
        const int e = EB(s, EB_SPARK_ORDER, pair - before, env);
        // the agent's start, for hazard_near (row 0 writes neither: resolve_kernel does)
        const float ax0 = SF(s, F_AX, env), ay0 = SF(s, F_AY, env);
        const int src = (SI(s, I_FLAGS, env) & kFlagBuf) ? 1 : 0;
        float hb[6];
        entity_step(s, env, e, src, 4, hb);
        if (hazard_near(hb, ax0, ay0, reach_x, reach_y)) {
            s.hazx[size_t(e) * s.n + env] = float4{hb[0], hb[1], hb[2], hb[3]};
            s.hazy[size_t(e) * s.n + env] = float2{hb[4], hb[5]};
            atomicOr(&SCI(s, SC_CAND + (e >> 5), env), 1 << (e & 31));
        }
    } else {
        // saws and the coin: kStaticPerLane entity slots a lane, everything they need asked for at once (the slots beyond
        // the env's count read slot kMaxEnt - 1 and write nothing) — a quarter of the wavefronts, one chain of round trips
        if (!stepping) return;
        const int n_ent = SI(s, I_NENT, env);
        const int e0 = (y - kMobRows) * kStaticPerLane;
        if (e0 >= n_ent) return;
        const float ax0 = SF(s, F_AX, env), ay0 = SF(s, F_AY, env);
        const int src = (SI(s, I_FLAGS, env) & kFlagBuf) ? 1 : 0;
        int kind[kStaticPerLane], dyn[kStaticPerLane];
        float x[kStaticPerLane], ey[kStaticPerLane], anim_t[kStaticPerLane];
#pragma unroll
        for (int k = 0; k < kStaticPerLane; k++) {
            const int ek = e0 + k < kMaxEnt ? e0 + k : kMaxEnt - 1;
            kind[k] = EB(s, EB_KIND, ek, env);
            x[k] = DF(s, src, DF_X, ek, env);
            ey[k] = EY(s, ek, env);
            anim_t[k] = DF(s, src, DF_ANIM_T, ek, env);
            dyn[k] = DB(s, src, ek, env);
        }
#pragma unroll
        for (int k = 0; k < kStaticPerLane; k++) {
            const int ek = e0 + k;
            if (ek >= n_ent || kind[k] == kMob) continue;
            float hb[6];
            static_step(s, env, ek, kind[k], 1 - src, 4, x[k], ey[k], anim_t[k], dyn[k], hb);
            if (hazard_near(hb, ax0, ay0, reach_x, reach_y)) {
                s.hazx[size_t(ek) * s.n + env] = float4{hb[0], hb[1], hb[2], hb[3]};
                s.hazy[size_t(ek) * s.n + env] = float2{hb[4], hb[5]};
                atomicOr(&SCI(s, SC_CAND + (ek >> 5), env), 1 << (ek & 31));
            }
        }
    }
}

// C — lane = env: which sub-step ended the step, rare redo, commit (coinrun.cpp:356-371).
// (Measured and rejected, round 5: no launch of its own — every wavefront of entity_kernel counts itself off in its block's
// word and the last one resolves the block's 64 envs.  Bit-exact; with relaxed agent-scope atomics and a workgroup-scope
// release 125.6 against 128.3 M env-steps/s — 73 728 atomics cost more than a 6-µs kernel and its boundary; with
// __threadfence() and an acq_rel count, an L2 write-back per wavefront, the step took 2.86 ms.)
// blockIdx.y == 1 (64 lanes): the auto-resets that logic_kernel's install row could not serve — no level lay ready, an
// episode shorter than the generator's latency — generated here, synchronously (pg_prefetch.h level_serve, mode 2), in this
// launch instead of in one of its own in front of it: in steady state that launch found nothing and cost the step its 6 µs
// and a kernel boundary.  Nothing in between needs those envs' new levels (their agents and entities sit the step out, row 0
// here leaves them alone); the pre-pass behind this launch does.  The byte carries the step's parity (pg_prefetch.h
// reset_due_mark): row 0 writes "due in step t + 1" for an env that terminates now, which row 1, looking for "due in
// step t" in the same launch, does not take for its own.
__global__ void __launch_bounds__(64) resolve_kernel(State s, StepIO io, uint32_t step_index, int prefetch, LevelPlan plan) {
    if (blockIdx.y == 1) {  // (block-uniform)
        level_serve<Gen>(s, 2, 64, prefetch, 0u, 0, nullptr, nullptr, io, plan, reset_served_mark(step_index),
                         reset_due_mark(step_index), static_cast<int>(blockIdx.x) * 64, static_cast<int>(threadIdx.x));
        return;
    }
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= s.n) return;
    const int bits = SCI(s, SC_BITS, env);
    if (bits >= 0) return;  // reset this step: reward / done are written with the install, the byte's mark expires by itself
    const int flags = SI(s, I_FLAGS, env);
    const int src = (flags & kFlagBuf) ? 1 : 0;
    // the hazards: the few boxes that came near (hazard_near), each against the agent's four
    uint32_t cand0 = static_cast<uint32_t>(SCI(s, SC_CAND, env)), cand1 = static_cast<uint32_t>(SCI(s, SC_CAND + 1, env));
    float bx[4], by[4];
#pragma unroll
    for (int ss = 0; ss < 4; ss++) {
        bx[ss] = SC(s, SC_AX + ss, env) + -0.5f;
        by[ss] = SC(s, SC_AY + ss, env) + -1.0f;
    }
    int hazard = 0;
    if (cand0 | cand1) {
        SCI(s, SC_CAND, env) = 0;
        SCI(s, SC_CAND + 1, env) = 0;
    }
    if (bits & kBitsFar) {  // the agent left the region the entity lanes looked at: every box again, from the table
        const int n_ent = SI(s, I_NENT, env);
        for (int e = 0; e < n_ent; e++) {
            float hb[6];
            entity_step<false>(s, env, e, src, 4, hb);
            hazard |= hazard_hits(hb, bx, by);
        }
    } else {
        unsigned long long cand = cand0 | (static_cast<unsigned long long>(cand1) << 32);
        while (cand) {
            const int e = __builtin_ctzll(cand);
            cand &= cand - 1;
            const float4 x = s.hazx[size_t(e) * s.n + env];
            const float2 y = s.hazy[size_t(e) * s.n + env];
            const float hb[6] = {x.x, x.y, x.z, x.w, y.x, y.y};
            hazard |= hazard_hits(hb, bx, by);
        }
    }
    const int lava = (bits >> 8) & 15, coin = (bits >> 12) & 15;
    const int ending = hazard | lava | coin;
    const int last = ending ? __builtin_ctz(ending) : 3;  // first terminating sub-step, or all four took place
    // Rare: fewer than four sub-steps happened.  The hazard bits of the sub-steps that did happen are already right
    // (entities do not react to the agent), so reward/done need nothing more; the entity table does, and that is left
    // to the render wavefront of this env (SC_REDO) instead of 36 serial entity updates on this one lane.
    SCI(s, SC_REDO, env) = last < 3 ? last + 1 : 0;
    const bool alive = !(((hazard | lava) >> last) & 1);
    const bool got_coin = ((coin >> last) & 1) != 0;
    const float ax = SC(s, SC_AX + last, env), ay = SC(s, SC_AY + last, env);
    SF(s, F_AX, env) = ax;
    SF(s, F_AY, env) = ay;
    SF(s, F_AVX, env) = SC(s, SC_AVX + last, env);
    SF(s, F_AVY, env) = SC(s, SC_AVY + last, env);
    SF(s, F_APHASE, env) = SC(s, SC_PHASE + last, env);
    SF(s, F_CAMX, env) = ax * kUnitPx;  // common_systems.cpp:238-239
    SF(s, F_CAMY, env) = (ay - 0.5f) * kUnitPx;
    SI(s, I_FLAGS, env) = kFlagListed | (((bits >> last) & 1) ? kFlagGround : 0) |
                          (((bits >> (4 + last)) & 1) ? kFlagForward : 0) | (src ? 0 : kFlagBuf);
    const bool terminated = !alive || got_coin;  // coinrun.cpp:366
    io.reward[env] = got_coin * 10.0f;           // coinrun.cpp:364, last executed sub-step only (D4)
    io.done[env] = terminated ? 1 : 0;
    io.pending[env] = terminated ? reset_due_mark(step_index + 1u) : 0;
}

// render_game(true) (coinrun.cpp:443-470): one workgroup of two wavefronts per env (pg_render.h).
// flags bit 0: force the draw-list replay for background + tiles (fallback path; parity tests run both).
// Higher bits are timing experiments (tools/ablate_render.py), compiled in only with -DPG_ABLATE (pg_render.h PG_ABL).
constexpr int kRenderWaves = 2;  // wavefronts per env (pg_render.h: two waves share one frame's LDS target)
#ifndef PG_COINRUN_QUARTERS
#define PG_COINRUN_QUARTERS false
#endif
constexpr bool kQuarters = PG_COINRUN_QUARTERS;  // the lean kernel's tiny draws (sparks) four to a slot (pg_render.h)

#ifndef PG_COINRUN_RENDER_WAVES
// Wavefronts per SIMD the registers are capped for.  Five = 96 registers = nine envs a CU, which the LDS allows since the
// shared draws left it: render 0.435 -> 0.414 ms.  (The kernel wants 121; the 57 it spills sit in the rare paths.  With the
// general form of the row composer still eight rows at a time it spilled 77 and was 7 % slower than at four.)
#define PG_COINRUN_RENDER_WAVES 5
#endif
constexpr int kGrid = 16;  // 64 px / 4.8 px per tile = 13.3 tiles → at most 16 columns/rows in view

// The complete frame of one env by its workgroup, set-up included: every frame before the pre-pass existed, and still
// the frames the pre-pass marks fat, the draw-list replay (flags bit 0) and kDebugNoPrepass.
PG_D void render_full(const State& s, const AtlasView& atlas, const StepIO& io, int flags, int env, uint32_t* fb,
                      ComposeLds<kGrid>& L) {
    const int lane = threadIdx.x & 63, half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int halves = kRenderWaves;
    // The 64 resolved draws of the sprite pass, from wave 1 to wave 0, travel through the END of the frame target's
    // memory (the composer's set-up tables borrow its start): they are written before the frame is composed and taken
    // before its first pixel is (render 0.440 -> 0.435 ms against an array of their own: 1.5 KB less LDS per env).
    uint32_t* const slots = fb + kFbWords - kBlitWords * 64;
    static_assert(sizeof(ComposeTmp<kGrid>) + sizeof(ComposeHand) + kBlitWords * 64 * 4 <= kFbWords * 4, "room behind the set-up tables");

    const Camera cam{SF(s, F_CAMX, env), SF(s, F_CAMY, env), 64.0f, 64.0f, 0.3f * 64.0f / 64.0f};
    const int themes = SI(s, I_THEMES, env);
    const int sflags = SI(s, I_FLAGS, env);
    const int buf = (sflags & kFlagBuf) ? 1 : 0;
    const int backdrop = PG_ABL(flags, 16) ? 9 : (themes & 0xff);  // (bit 4: timing experiment — one shared background)
    const int alien = (themes >> 8) & 0xff, ground_theme = (themes >> 16) & 0xff;
    const int n_ent = SI(s, I_NENT, env);
    const int n_mob = SI(s, I_NMOB, env);
    const uint8_t* tiles = s.tiles + size_t(env) * (W * H);
    const DescRegs descs = DescRegs::load(atlas, lane);  // the whole descriptor table, two entries per lane
    Blit mine;
    PG_MARK("a_state");

    {   // the step ended after fewer than four sub-steps (resolve_kernel): bring the entity table to that sub-step,
        // one lane per entity, from the half of the table the optimistic pass left untouched
        const int redo = SCI(s, SC_REDO, env);
        __syncthreads();  // every wave has read the flag before it is cleared
        if (redo) {
            if (half == 0 && lane < n_ent) {
                float hb[6];
                entity_step(s, env, lane, 1 - buf, redo, hb);
            }
            if (threadIdx.x == 0) SCI(s, SC_REDO, env) = 0;
            __threadfence();
            __syncthreads();
        }
    }

    // The per-lane inputs of the sprite pass (one draw per lane: particles, then sprites, then the agent) are
    // requested now so their memory latency hides behind the background/tile composition.
    const int n_parts = n_mob * kSparks;
    const int n_sprites = (sflags & kFlagListed) ? n_ent : 0;  // the draw list is empty until the first update (D2)
    const bool one_pass = n_parts + n_sprites + 1 <= 64;
    const int first_sprite = one_pass ? n_parts : 0;
    const bool is_part = one_pass && lane < n_parts;
    const bool is_sprite = lane >= first_sprite && lane < first_sprite + n_sprites;
    const bool is_agent = lane == first_sprite + n_sprites;
    int spr_e = 0, spr_dyn = 0, spr_tex = 0;
    float spr_x = 0.0f, spr_y = 0.0f, part_life = 0.0f, part_x = 0.0f, part_y = 0.0f;
    if (half != 1) {
        // wave 1 resolves the 64 draws of the sprite pass — before the frame is composed, while wave 0 sets up the
        // column side of the composer — and hands them over (pg_render.h blit_share)
    } else if (is_sprite) {
        spr_e = EB(s, EB_DRAW_ORDER, lane - first_sprite, env);
        spr_dyn = DB(s, buf, spr_e, env);
        spr_tex = EB(s, EB_TEX, spr_e, env);
        spr_x = DF(s, buf, DF_X, spr_e, env);
        spr_y = EY(s, spr_e, env);
    } else if (is_part) {
        const int m = lane / kSparks, k = lane - m * kSparks;
        const int e = EB(s, EB_SPARK_ORDER, m, env);
        part_life = SP(s, buf, 2, e, k, env);
        part_x = SP(s, buf, 0, e, k, env);
        part_y = SP(s, buf, 1, e, k, env);
    }
    PG_MARK("b_inputs");

    // background (coinrun.cpp:459-464)
    int4 bg_d;  // the background draw: texture, world position, scale — each wave resolves the axis it needs (pg_render.h BgAxis)
    float bg_px, bg_py, bg_sc;
    {
        const int4 d = descs.uniform(kTexBackdrop + backdrop);
        bg_d = d;
        const float aspect = static_cast<float>(d.y) / static_cast<float>(d.z);
        const float extra = aspect - 1.0f;
        bg_d = d;
        bg_px = -SF(s, F_BGSHIFT, env) * extra;
        bg_py = 0.0f;
        bg_sc = 64.0f * kUnitPx / d.z;
    }
    // negative-z sprites: none — every coinrun sprite has z = 1 (tilemap.cpp:63,88,283)

    // tile window (tilemap.cpp:294-304)
    const float vx = (cam.px - cam.sw * 0.5f / cam.scale) * kPxUnit;
    const float vy = (cam.py - cam.sh * 0.5f / cam.scale) * kPxUnit;
    const float vw = cam.sw * kPxUnit / cam.scale, vh = cam.sh * kPxUnit / cam.scale;
    const int x0 = static_cast<int>(floorf(vx)), y0 = static_cast<int>(floorf(vy));
    const int x1 = static_cast<int>(ceilf(vx + vw)), y1 = static_cast<int>(ceilf(vy + vh));
    const int cols = x1 - x0 + 1, rows = y1 - y0 + 1, cells = cols * rows;
    const int4 tile_desc = descs.uniform(kTexMid);  // every tile texture is 128×128 (checked at make time)

    if (half == 1 && !PG_ABL(flags, 2)) {  // the sprite pass's draws, resolved ahead of the frame (see above)
        {
            bool has = false;
            const int4 spark_d = descs.uniform(kTexSpark);
            // every lane asks for the descriptor it needs with a cross-lane read, so do that outside the branches
            int want_tex = 0;
            if (is_sprite) {
                want_tex = spr_tex + ((spr_dyn & kDynFrame) ? 1 : 0);
            } else if (is_agent) {
                const bool ground = (sflags & kFlagGround) != 0;
                if (fabsf(SF(s, F_AVX, env)) < 0.01f && ground)
                    want_tex = kTexStand + alien;
                else if (!ground)
                    want_tex = kTexJump + alien;
                else if (SF(s, F_APHASE, env) > 0.5f)
                    want_tex = kTexWalk2 + alien;
                else
                    want_tex = kTexWalk1 + alien;
            }
            const int4 d = descs.at(want_tex);
            // The three kinds of draw differ only in their parameters: pick them per lane, then resolve once.  (One
            // resolve_draw per kind in its own branch made every wave run its ~180 vector instructions three times.)
            bool go = false, flip = false;
            int tw = d.y, th = d.z, tex_at = d.x;
            float wx = 0.0f, wy = 0.0f, scale_num = kUnitPx, alpha = 1.0f;
            if (is_part) {  // System_Particles::render (common_systems.cpp:315-337), as `particle` above
                if (part_life > 0.0f) {
                    const float lr = (5.0f - part_life) / 5.0f;
                    alpha = 0.5f * (1.0f - lr);
                    const float scale = 0.45f * (0.4f * lr + 0.6f);
                    const float oy = -lr * 0.17f;
                    tw = spark_d.y;
                    th = spark_d.z;
                    tex_at = spark_d.x;
                    wx = part_x * kUnitPx - 0.5f * spark_d.y * scale;
                    wy = (part_y + oy) * kUnitPx - 0.5f * spark_d.z * scale;
                    scale_num = scale * kUnitPx;
                    go = true;
                }
            } else if (is_sprite) {
                if (spr_dyn & kDynTexSet) {
                    const float scale = 1.0f * 1.0f;
                    wx = (spr_x + -0.5f) * kUnitPx;
                    wy = (spr_y + -0.5f) * kUnitPx;
                    scale_num = scale * kUnitPx;
                    flip = (spr_dyn & kDynFlip) != 0;
                    go = true;
                }
            } else if (is_agent) {
                const float px = SF(s, F_AX, env) - 0.5f, py = SF(s, F_AY, env) - 2.0f;
                wx = px * kUnitPx;
                wy = py * kUnitPx;
                flip = (sflags & kFlagForward) == 0;
                go = true;
            }
            if (go)
                has = resolve_draw(cam, tw, th, tex_at, wx, wy, scale_num / static_cast<float>(tw), alpha, flip, false,
                                   mine);
            blit_share(slots, lane, mine, has);
        }
    }
    PG_MARK("c_resolve");
    const BgDraw bg_draw{bg_d, bg_px, bg_py, bg_sc};
    BgAxis bga{};  // this wave's axis of it (wave 0: x, wave 1: y), resolved along with the tile spans
    bool composed = false, sprites_ready = false;
    ReplayState<4> sprite_pass;
    if (!(flags & 1) && !PG_ABL(flags, 4) && cols <= kGrid && rows <= kGrid) {
        compose_spans(fb, L, cam, x0, y0, cols, rows, tile_desc.y, tile_desc.z, kUnitPx / tile_desc.y, lane, 0, half, halves,
                      /*soft_init=*/0, /*hard_init=*/0, &bg_draw, &bga);  // exact bits are ORed in below
    PG_MARK("d_spans");
        // Texel offset of each tile kind's texture, one per lane (0..7), looked up with a cross-lane read:
        // lanes 0-3 = wall_top, wall_mid, lava_top, lava_mid (tile id - 1), lanes 4-7 = the four crates.
        int kind_tex = kTexCrate + ((lane - 4) & 3);
        if (lane == 0) kind_tex = kTexTop + ground_theme;
        if (lane == 1) kind_tex = kTexMid + ground_theme;
        if (lane == 2) kind_tex = kTexLavaTop;
        if (lane == 3) kind_tex = kTexLava;
        const int4 kind_d = descs.at(kind_tex);
        const int kind_base = kind_d.x;
        // which grid rows show a texture with translucent texels (descriptor .w; crates and lava caps are the only
        // soft-edged tiles, 9 of the 49 backdrops have some): only there does the composer look at alphas
        uint32_t soft_rows = (threadIdx.x == 0 && bg_d.w != 0) ? 0x80000000u : 0u;
        // (the one-texel-per-pixel attempt is made everywhere unless the backdrop is mostly cut-out: crates and lava are few)
        const uint32_t hard_rows = (threadIdx.x == 0 && (bg_d.w & 2)) ? 0x80000000u : 0u;
        // the whole kGrid×kGrid table, 64 cells per pass — all by wave 0: wave 1 has resolved the sprite pass's draws above,
        // and an env's frame is done when its slower wave is (measured: render 0.456 -> 0.440 ms against an even split)
#pragma unroll
        for (int k = 0; k < kGrid * kGrid / 64; k++) {
            if (half != 0) break;
            const int cell = k * 64 + lane;
            const int r = cell / kGrid, c = cell % kGrid;
            const int x = x0 + c, ty = H - 1 - (y0 + r);
            int raw = kWallMid;  // out of bounds is a wall (tilemap.h:80-81)
            if (x >= 0 && ty >= 0 && x < W && ty < H) raw = tiles[ty + x * H];
            const int t = raw & 7;
            const int slot = t < kCrate ? t - 1 : 4 + (raw >> 4);
            const int off = __shfl(kind_base, slot < 0 ? 0 : slot);
            const int soft = __shfl(kind_d.w, slot < 0 ? 0 : slot);
            L.base[cell] = (t == kEmpty) ? static_cast<int32_t>(kNoTexel) : off * 4;
            if (t != kEmpty && soft != 0 && r < rows && c < cols) soft_rows |= 1u << r;
        }
        if (soft_rows) atomicOr(&L.soft_rows[half], static_cast<int32_t>(soft_rows));
        if (hard_rows) atomicOr(&L.hard_rows[half], static_cast<int32_t>(hard_rows));
        __syncthreads();
        PG_MARK("e_cells");
        // the draws wave 1 resolved: the texels of the first few are requested now and arrive while the frame is composed
        sprites_ready = !PG_ABL(flags, 2);
        if (sprites_ready) {
            const bool has = blit_take(slots, lane, mine);
            sprite_pass = replay_begin(atlas, mine, __ballot(has), lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
        }
        PG_MARK("f_begin");
        composed = compose_rows(fb, L, atlas, bga, cols, rows, tile_desc.y, lane, flags, half, halves);
        PG_MARK("k_composed");
    }
    if (PG_ABL(flags, 4)) composed = true;  // (bit 2: timing experiment — no background/tiles at all)
    bool taken = false;
    if (!composed) {  // draw-list replay of background and tiles (tilemap.cpp:294-321)
        if (!sprites_ready) {  // the draws wave 1 resolved leave the target's memory before it is written
            __syncthreads();
            taken = blit_take(slots, lane, mine);
            __syncthreads();
        }
        Blit tile;  // (`mine` holds the sprite pass's draws)
        wave_clear(fb, lane, half, halves);
        const bool has_bg = resolve_draw(cam, bg_d.y, bg_d.z, bg_d.x, bg_px, bg_py, bg_sc, 1.0f, false, false, tile);
        wave_replay(fb, atlas, tile, has_bg ? 1ull : 0ull, lane, half, halves);
        for (int base = 0; base < cells; base += 64) {
            const int cell = base + lane;
            bool has = false;
            if (cell < cells) {
                const int row = cell / cols;
                const int x = x0 + (cell - row * cols), y = y0 + row;
                const int ty = H - 1 - y;
                int t = kWallMid, crate = 0;
                if (x >= 0 && ty >= 0 && x < W && ty < H) {
                    const int raw = tiles[ty + x * H];
                    t = raw & 7;
                    crate = raw >> 4;
                }
                if (t != kEmpty) {
                    int tex;
                    if (t == kWallMid)
                        tex = kTexMid + ground_theme;
                    else if (t == kWallTop)
                        tex = kTexTop + ground_theme;
                    else if (t == kLavaMid)
                        tex = kTexLava;
                    else if (t == kLavaTop)
                        tex = kTexLavaTop;
                    else
                        tex = kTexCrate + crate;
                    const int4 d = atlas.desc[tex];  // fallback path: plain global lookup
                    has = resolve_draw(cam, d.y, d.z, d.x, x * kUnitPx, y * kUnitPx, kUnitPx / d.y, 1.0f, false, false,
                                       tile);
                }
            }
            wave_replay(fb, atlas, tile, __ballot(has), lane, half, halves);
        }
    }

    if (!PG_ABL(flags, 2)) {  // (bit 1: timing experiment — skips particles, sprites and the agent)
        // One draw per lane, in the reference's order: particles (owners in the particle system's set order,
        // common_systems.cpp:315-337), then the sprites of the draw list (positive z, :41-63; empty until the
        // first update — D2), then the agent (:254-278).
        const int4 spark_d = descs.uniform(kTexSpark);
        auto particle = [&](float life, float px, float py, Blit& out) {
            if (life <= 0.0f) return false;
            const float lr = (5.0f - life) / 5.0f;
            const float alpha = 0.5f * (1.0f - lr);
            const float scale = 0.45f * (0.4f * lr + 0.6f);
            const float oy = -lr * 0.17f;
            return resolve_draw(cam, spark_d.y, spark_d.z, spark_d.x, px * kUnitPx - 0.5f * spark_d.y * scale,
                                (py + oy) * kUnitPx - 0.5f * spark_d.z * scale, scale * kUnitPx / spark_d.y, alpha,
                                false, false, out);
        };
        if (!one_pass) {  // many mobs: particles in rounds of 64 first
            Blit spark;  // (`mine` holds the draws of the pass that is already under way)
            for (int base = 0; base < n_parts; base += 64) {
                bool has = false;
                if (base + lane < n_parts) {
                    const int idx = base + lane, m = idx / kSparks, k = idx - m * kSparks;
                    const int e = EB(s, EB_SPARK_ORDER, m, env);
                    has = particle(SP(s, buf, 2, e, k, env), SP(s, buf, 0, e, k, env), SP(s, buf, 1, e, k, env), spark);
                }
                wave_replay_rows(fb, atlas, spark, __ballot(has), lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
            }
        }
        // the draws wave 1 resolved before the frame was composed (several barriers ago)
        if (!sprites_ready)  // (the draw-list replay path did not get that far)
            sprite_pass = replay_begin(atlas, mine, __ballot(taken), lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
        PG_MARK("m_fallback_and_rounds");
        replay_finish(fb, atlas, mine, sprite_pass, lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
        PG_MARK("n_sprites");
    }
    // each wave stores the rows it owns (pg_render.h wave_replay_rows): no barrier
    if (!PG_ABL(flags, 8))
        wave_store_rows(fb, io.obs + size_t(env) * kObsBytes, lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
    PG_MARK("o_store");
}

// ------------------------------------------------------------------------------------------------
// The render pre-pass (pg_prepass.h): tile spans, per-pixel candidates, the cell table and the resolved, culled draws
// of kPrepEnvs envs per workgroup, every phase with its lanes dealt densely over (env, thing).
// Reference arithmetic moved here unchanged: renderer.cpp:5-82 (render_texture), tilemap.cpp:294-321 (the tile window),
// common_systems.cpp:41-63,254-278,315-337 (sprites, agent, particles).
// ------------------------------------------------------------------------------------------------
constexpr int kPrepEnvs = 8, kPrepThreads = 256;
constexpr int kSlotTop = 0, kSlotMid = 1, kSlotLavaTop = 2, kSlotLava = 3, kSlotCrate = 4;  // cell bytes: tile kinds

struct PrepEnv {  // per env of the workgroup: what the draws' lanes need, loaded once
    int32_t sflags, buf, n_ent, n_mob, n_sprites, alien;
    float avx, aphase, ax, ay;
};
struct SetupLds {
    PrepLds<kGrid, kPrepEnvs, kMaxSpan> P;
    PrepEnv env[kPrepEnvs];
    int4 desc[kTexCount];                    // the atlas descriptor table: every later lookup is an LDS read
    uint8_t order[kPrepEnvs][2][kMaxEnt + 4];  // EB_SPARK_ORDER, EB_DRAW_ORDER of every env (fetched before anything needs them)
    int32_t kind_soft[kPrepEnvs];            // bit k: tile kind k's texture has texels that are not opaque
    uint32_t row_valid[kPrepEnvs][kGrid / 4];              // byte i: 0xff if map row ty_lo + i of the env's window exists (cells below)
    int32_t counts[kPrepEnvs];
    PrepDrawQueue queue[kPrepThreads / 64];  // one worklist per wavefront
};

// (Measured and rejected, round 5: the pre-pass in two halves — views / cells / spans / axes, which need the step only
// through the camera, i.e. through the agent, as further workgroups of the ENTITY launch (camera taken optimistically from
// the agent's scratch row of sub-step 3), the draws in a kernel behind resolve_kernel as now.  Bit-exact in the 24 coinrun
// tests, and slower: the fused launch 146 µs, the draws' kernel 57, against 46 + 93 one after the other.  A kernel has one
// register allocation: with the entity path's 132 registers three wavefronts fit a SIMD, the mobs' 3 600 long chains take
// every one of those slots for their whole 40 µs, and the pre-pass workgroups, dealt behind them, start when they end; and
// the draws alone are a chain of round trips that the other phases of the same kernel used to hide.)
__global__ void __launch_bounds__(kPrepThreads) setup_kernel(State s, AtlasView atlas, const uint8_t* mask, int flags) {
    __shared__ SetupLds S;
    PrepLds<kGrid, kPrepEnvs, kMaxSpan>& P = S.P;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int env0 = prep_block(blockIdx.x, gridDim.x) * kPrepEnvs;
    const PrepOut& out = s.prep;

    // ---- everything whose address does not depend on another load is requested first: the descriptor table, the envs'
    // scalars (lane = env), the two entity orders (lane = (env, slot)) — one memory round trip for all of it
    for (int q = tid; q < kPrepEnvs * 2 * 64; q += kPrepThreads) (&P.cover[0][0][0])[q] = 0u;
    if (tid < kTexCount) S.desc[tid] = atlas.desc[tid];
    static_assert(kTexCount <= kPrepThreads, "one descriptor per thread");
    for (int q = tid; q < kPrepEnvs * 2 * kMaxEnt; q += kPrepThreads) {
        const int e = q / (2 * kMaxEnt), r = q - e * (2 * kMaxEnt), which = r / kMaxEnt, j = r - which * kMaxEnt;
        if (env0 + e < s.n) S.order[e][which][j] = EB(s, which ? EB_DRAW_ORDER : EB_SPARK_ORDER, j, env0 + e);
    }
    Camera cam{};
    int themes = 0, redo = 0;
    float bgshift = 0.0f;
    bool active = false;
    if (tid < kPrepEnvs) {
        const int e = tid, env = env0 + e;
        active = env < s.n && (!mask || mask[env]);
        if (active) {
            redo = SCI(s, SC_REDO, env);
            cam = Camera{SF(s, F_CAMX, env), SF(s, F_CAMY, env), 64.0f, 64.0f, 0.3f * 64.0f / 64.0f};
            themes = SI(s, I_THEMES, env);
            bgshift = SF(s, F_BGSHIFT, env);
            PrepEnv pe{};
            pe.sflags = SI(s, I_FLAGS, env);
            pe.buf = (pe.sflags & kFlagBuf) ? 1 : 0;
            pe.n_ent = SI(s, I_NENT, env);
            pe.n_mob = SI(s, I_NMOB, env);
            pe.n_sprites = (pe.sflags & kFlagListed) ? pe.n_ent : 0;  // the draw list is empty until the first update (D2)
            pe.alien = (themes >> 8) & 0xff;
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
        S.kind_soft[e] = 0;
        if (active && redo != 0) {  // the step ended early: the entity table is not final (resolve_kernel)
            P.fat[e] = 1;
            active = false;
        }
        if (active) {
            v.cam = cam;
            // (bit 4: timing experiment of the -DPG_ABLATE build — every env shows backdrop 9, the bound on what keeping
            // the backdrops' texels closer to the workgroups that sample them could buy: render FETCH_SIZE 309 -> 45 MB,
            // render 0.350 -> 0.333 ms)
            const int4 d = S.desc[kTexBackdrop + (PG_ABL(flags, 16) ? 9 : (themes & 0xff))];
            const float aspect = static_cast<float>(d.y) / static_cast<float>(d.z);
            const float extra = aspect - 1.0f;
            v.bg = BgDraw{d, -bgshift * extra, 0.0f, 64.0f * kUnitPx / d.z};  // coinrun.cpp:459-464
            const float vx = (cam.px - cam.sw * 0.5f / cam.scale) * kPxUnit;  // tilemap.cpp:294-304
            const float vy = (cam.py - cam.sh * 0.5f / cam.scale) * kPxUnit;
            const float vw = cam.sw * kPxUnit / cam.scale, vh = cam.sh * kPxUnit / cam.scale;
            v.x0 = static_cast<int>(floorf(vx));
            v.y0 = static_cast<int>(floorf(vy));
            v.cols = static_cast<int>(ceilf(vx + vw)) - v.x0 + 1;
            v.rows = static_cast<int>(ceilf(vy + vh)) - v.y0 + 1;
            const int4 tile_desc = S.desc[kTexMid];  // every tile texture is 128×128 (checked at make time)
            v.tw = tile_desc.y;
            v.th = tile_desc.z;
            v.th2 = 0;
            v.tile_scale = kUnitPx / tile_desc.y;
            if (v.cols > kGrid || v.rows > kGrid || ((flags & kDebugFatThirds) && (env0 + e) % 3 == 0)) {
                P.fat[e] = 1;
                active = false;
            }
            P.soft_rows[e] = d.w != 0 ? 0x80000000u : 0u;   // 9 of the 49 backdrops have texels that are not opaque
            P.hard_rows[e] = (d.w & 2) ? 0x80000000u : 0u;  // … mostly cut-out: no one-texel attempt
            // the tile kinds' textures
            const int ground_theme = (themes >> 16) & 0xff;
            int soft = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                int tex = kTexCrate + (k - kSlotCrate);
                if (k == kSlotTop) tex = kTexTop + ground_theme;
                if (k == kSlotMid) tex = kTexMid + ground_theme;
                if (k == kSlotLavaTop) tex = kTexLavaTop;
                if (k == kSlotLava) tex = kTexLava;
                const int4 kd = S.desc[tex];
                P.meta[e][PM_KINDS + k] = static_cast<uint32_t>(kd.x) * 4u;
                    if (kd.w != 0) soft |= 1 << k;
            }
            S.kind_soft[e] = soft;
            prep_row_valid<kGrid, H>(v.y0, S.row_valid[e]);  // which of the 16 map rows under the window exist
        }
        v.active = active ? 1 : 0;
        P.view[e] = v;
    }
    __syncthreads();
    if (PG_ABL(flags, 0x1000000)) return;  // (instruction inventory, -DPG_ABLATE builds only: the views alone)

    // ---- the cell table, first half: lane = (env, grid column).  The map is column-major (tilemap.h:62-85), so the sixteen
    // cells of a window column are sixteen consecutive bytes — one 16-byte load (unaligned; rows beyond the map's edge
    // read the neighbouring column or the neighbouring env's map and are replaced below) …
    static_assert(kGrid == 16 && H == 64, "a window column is one 16-byte load");
    const int cell_e = tid / kGrid, cell_c = tid - cell_e * kGrid;
    bool cell_lane = false, cell_x_ok = false;
    uint32_t column[kGrid / 4] = {};
    if (tid < kPrepEnvs * kGrid && P.view[cell_e].active) {
        cell_lane = true;
        prep_column_fetch<kGrid, W, H>(s.tiles + size_t(env0 + cell_e) * (W * H), P.view[cell_e].x0 + cell_c, P.view[cell_e].y0, cell_x_ok, column);
    }
    // … the spans are worked out while it travels …
    if (!PG_ABL(flags, 0x2000000)) prep_spans<kGrid, kMaxSpan, kPrepEnvs>(P, tid, kPrepThreads);
    // … second half: sixteen kind bytes, four cells at a time
    if (cell_lane) {
        const PrepView& v = P.view[cell_e];
        uint32_t in_rows[4];
        prep_column_rows<kGrid>(column, S.row_valid[cell_e], cell_x_ok, kWallMid, in_rows);  // out of bounds is a wall (tilemap.h:80-81)
        // soft kinds as a byte table for v_perm: byte k = 0xff if kind k's texture has texels that are not opaque
        const uint32_t soft_kinds = static_cast<uint32_t>(S.kind_soft[cell_e]);
        uint32_t soft_lo = 0u, soft_hi = 0u;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            soft_lo |= ((soft_kinds >> k) & 1u) ? 0xffu << (8 * k) : 0u;
            soft_hi |= ((soft_kinds >> (4 + k)) & 1u) ? 0xffu << (8 * k) : 0u;
        }
        uint32_t kinds[4], soft_rows = 0u;
#pragma unroll
        for (int w = 0; w < 4; w++) {  // window rows 4w .. 4w + 3: raw byte = tile id | crate kind << 4 (crates only)
            const uint32_t raw = in_rows[w];
            const uint32_t t = raw & 0x07070707u, crate = (raw >> 4) & 0x03030303u;
            const uint32_t y = (t | 0x80808080u) - 0x01010101u;          // per byte: 0x7f if empty, else 0x80 + t - 1
            const uint32_t there = ((y & 0x80808080u) >> 7) * 0xffu;      // 0xff where there is a tile
            const uint32_t slots = (y & 0x07070707u) + crate;            // kSlotTop .. kSlotLava = t - 1, crates 4 + kind
            kinds[w] = (slots & there) | ~there;                         // 0xff: no tile
            // the rows that show a soft texture (only the window's own rows and columns count)
            const uint32_t sel = (slots & there) | (0x0c0c0c0cu & ~there);  // v_perm selector 0x0c: the constant 0x00
            const uint32_t soft = __builtin_amdgcn_perm(soft_hi, soft_lo, sel) & 0x01010101u;
            soft_rows |= ((soft * 0x00204081u >> 21) & 0xfu) << (4 * w);
        }
        if (cell_c >= v.cols) soft_rows = 0u;
        soft_rows &= (v.rows >= 32 ? ~0u : (1u << v.rows) - 1u);
        prep_column_store<kGrid>(out.cells + size_t(env0 + cell_e) * (kGrid * kGrid), cell_c, kinds);
        if (soft_rows) atomicOr(&P.soft_rows[cell_e], soft_rows);
    }
    __syncthreads();
    if (PG_ABL(flags, 0x4000000)) return;
    if (!PG_ABL(flags, 0x8000000)) prep_axes<kGrid, kMaxSpan, kPrepEnvs>(P, out, env0, wave, kPrepThreads / 64, lane);
    if (PG_ABL(flags, 0x10000000)) return;

    // ---- the draws, in the reference's order — particles (owners in the particle system's set order), the sprites of
    // the draw list, the agent.  A wavefront takes two envs of the workgroup, their lists dealt densely over its passes;
    // a draw that survives render_texture's cull waits in the wave's worklist for the rest of the arithmetic
    // (pg_prepass.h prep_draws_pass).
    static_assert(kPrepEnvs == 2 * (kPrepThreads / 64), "two envs per wavefront");
    {
        const int ea = 2 * wave, eb = 2 * wave + 1;
        const bool on_a = P.view[ea].active != 0, on_b = P.view[eb].active != 0;
        const Camera cam_a = P.view[ea].cam, cam_b = P.view[eb].cam;
        const int cnt_a = on_a ? S.env[ea].n_mob * kSparks + S.env[ea].n_sprites + 1 : 0;
        const int cnt_b = on_b ? S.env[eb].n_mob * kSparks + S.env[eb].n_sprites + 1 : 0;
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
            const int n_parts = pe.n_mob * kSparks;
            PrepDraw p{false, false, false, kTexSpark, 0.0f, 0.0f, 1.0f, 1.0f};
            if (valid && slot < n_parts) {  // System_Particles::render (common_systems.cpp:315-337)
                const int m = slot / kSparks, k = slot - m * kSparks;
                const int ent = S.order[e][0][m];
                const float life = SP(s, pe.buf, 2, ent, k, env);
                const float px = SP(s, pe.buf, 0, ent, k, env), py = SP(s, pe.buf, 1, ent, k, env);
                if (life > 0.0f) {
                    const int4 spark_d = S.desc[kTexSpark];
                    const float lr = (5.0f - life) / 5.0f;
                    p.alpha = 0.5f * (1.0f - lr);
                    const float scale = 0.45f * (0.4f * lr + 0.6f);
                    const float oy = -lr * 0.17f;
                    p.wx = px * kUnitPx - 0.5f * spark_d.y * scale;
                    p.wy = (py + oy) * kUnitPx - 0.5f * spark_d.z * scale;
                    p.scale = scale * kUnitPx / static_cast<float>(spark_d.y);
                    p.go = true;
                }
            } else if (valid && slot < n_parts + pe.n_sprites) {  // System_Sprite_Render::render (:41-63)
                const int ent = S.order[e][1][slot - n_parts];
                const int dyn = DB(s, pe.buf, ent, env);
                const int tex0 = EB(s, EB_TEX, ent, env);
                const float ex = DF(s, pe.buf, DF_X, ent, env), ey = EY(s, ent, env);
                if (dyn & kDynTexSet) {
                    p.tex = tex0 + ((dyn & kDynFrame) ? 1 : 0);
                    const float scale = 1.0f * 1.0f;
                    p.wx = (ex + -0.5f) * kUnitPx;
                    p.wy = (ey + -0.5f) * kUnitPx;
                    p.scale = scale * kUnitPx / static_cast<float>(S.desc[p.tex].y);
                    p.flip_h = (dyn & kDynFlip) != 0;
                    p.go = true;
                }
            } else if (valid) {  // the agent (:254-278)
                const bool ground = (pe.sflags & kFlagGround) != 0;
                if (fabsf(pe.avx) < 0.01f && ground)
                    p.tex = kTexStand + pe.alien;
                else if (!ground)
                    p.tex = kTexJump + pe.alien;
                else if (pe.aphase > 0.5f)
                    p.tex = kTexWalk2 + pe.alien;
                else
                    p.tex = kTexWalk1 + pe.alien;
                p.wx = (pe.ax - 0.5f) * kUnitPx;
                p.wy = (pe.ay - 2.0f) * kUnitPx;
                p.scale = kUnitPx / static_cast<float>(S.desc[p.tex].y);
                p.flip_h = (pe.sflags & kFlagForward) == 0;
                p.go = true;
            }
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

// render_game(true) (coinrun.cpp:443-470): one workgroup of two wavefronts per env.  A lean frame starts from what
// setup_kernel left: one word per pixel column and row, the cell table as kind bytes, its visible draws resolved.
// (Measured and rejected, round 5 — the frame's wavefront waits 6 600 of its 22 000 clocks for its hand-off,
// profiles/r05_wave_timelines.txt, and two ways of hiding that were built, both bit-exact:
//  * a PERSISTENT form — as many workgroups as the device holds, each drawing env after env and asking for the next env's
//    hand-off while it draws this one; handed-back frames drawn first, from a list: 0.41 ms against 0.33 at any register
//    budget.  A wavefront's vector-memory counter retires in issue order, stores included, so the next frame's first wait
//    for a texel also waits for this frame's 24 streaming stores to be acknowledged — what a workgroup that ends never does;
//  * the draws fetched before their count is known (the first 16 / 24 / 64 lanes unconditionally, the rest in a second trip
//    when there are more): 0.352 / 0.368 / 0.400 ms against 0.335 — most frames have more than 24 draws, sparks mostly, and
//    then make both trips.)
__global__ void __launch_bounds__(64 * kRenderWaves, PG_COINRUN_RENDER_WAVES) render_kernel(State s, AtlasView atlas, const uint8_t* mask,
                                                                   StepIO io, int flags) {
    const int env = blockIdx.x;
    if (mask && !mask[env]) return;
    const int lane = threadIdx.x & 63, half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int halves = kRenderWaves;
    __shared__ alignas(16) uint32_t fb[kFbWords];
    __shared__ ComposeLds<kGrid> L;  // the composer's cell table
    PG_TL_BEGIN(6);
    PG_TL(0);
    // One round trip for everything the frame starts from: the packed axes, the kind offsets, this wave's half of the
    // cell bytes (vector loads) and the meta line (scalar loads) leave together; only the draws wait for their count.
    // Whether the frame is a fat one is asked AFTER the lean preamble (a few wasted instructions for the rare fat frame,
    // no load waiting behind a branch for all the others).
    const PrepMeta M{s.prep.meta + size_t(env) * kPrepMetaWords};
    const uint32_t colw = s.prep.axes[size_t(env) * 128 + lane], roww = s.prep.axes[size_t(env) * 128 + 64 + lane];
    const uint32_t kind_off = M.w[PM_KINDS + (lane & (kPrepKinds - 1))];
    const uint32_t two = reinterpret_cast<const uint16_t*>(s.prep.cells)[size_t(env) * (kGrid * kGrid / 2) + half * 64 + lane];
    const int n_draws = M.draws();
    const bool has = lane < n_draws;
    const Blit mine = prep_draw_load(s.prep.draws + (size_t(env) * kPrepDraws + lane) * kBlitWords, has);
    prep_cells_expand<kGrid>(L, two, kind_off, half, lane);  // kind bytes → byte offsets of the kinds' textures
    const ComposeRegs R = prep_regs<kGrid>(M, colw, roww, 0u, lane);
    __syncthreads();  // the cell table is complete
    PG_TL(1);
    if ((flags & (1 | kDebugNoPrepass)) || M.fat()) {  // (wave-uniform)
        render_full(s, atlas, io, flags, env, fb, L);
        return;
    }
    const int row_lo = half * (kObsH / halves), row_hi = (half + 1) * (kObsH / halves);
    // (the PG_ABL tests are timing experiments of the -DPG_ABLATE build: constants 0 in the product)
    ReplayState<4> sprite_pass = replay_begin<4, false, true, false, kQuarters>(atlas, mine, PG_ABL(flags, 2) ? 0ull : __ballot(has), lane, row_lo, row_hi);
    PG_TL(2);
    if (!PG_ABL(flags, 4)) compose_rows_from<kGrid, false, false>(fb, L, atlas, R, lane, flags, half, halves);
    PG_TL(3);
    replay_finish<4, false, true, 4, false, kQuarters>(fb, atlas, mine, sprite_pass, lane, row_lo, row_hi);
    PG_TL(4);
    if (!PG_ABL(flags, 8)) wave_store_rows(fb, io.obs + size_t(env) * kObsBytes, lane, row_lo, row_hi);
    PG_TL_END(6, true, io.obs + size_t(env) * kObsBytes + half * (kObsBytes / 2));
}

// cenv_render's frame (coinrun.cpp:393-411 → render_game(false), :443-470) for one env: pg_frame.h.
__global__ void __launch_bounds__(kFrameThreads) frame_kernel(State s, AtlasView atlas, int env, FrameTarget t) {
    FramePainter P{t, atlas,
                   Camera{SF(s, F_CAMX, env), SF(s, F_CAMY, env), static_cast<float>(t.w), static_cast<float>(t.h),
                          0.3f * static_cast<float>(t.w) / 64.0f},
                   static_cast<int>(threadIdx.x), kFrameThreads};
    const int themes = SI(s, I_THEMES, env), sflags = SI(s, I_FLAGS, env);
    const int buf = (sflags & kFlagBuf) ? 1 : 0;
    const int backdrop = themes & 0xff, alien = (themes >> 8) & 0xff, ground_theme = (themes >> 16) & 0xff;
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
            const int ty = H - 1 - y;
            int t_id = kWallMid, crate = 0;
            if (x >= 0 && ty >= 0 && x < W && ty < H) {
                const int raw = tiles[ty + x * H];
                t_id = raw & 7;
                crate = raw >> 4;
            }
            if (t_id == kEmpty) continue;
            const int tex = t_id == kWallMid   ? kTexMid + ground_theme
                            : t_id == kWallTop ? kTexTop + ground_theme
                            : t_id == kLavaMid ? kTexLava
                            : t_id == kLavaTop ? kTexLavaTop
                                               : kTexCrate + crate;
            P.draw(tex, x * kUnitPx, y * kUnitPx, kUnitPx / P.desc(tex).y);
        }
    const int n_mob = SI(s, I_NMOB, env);
    for (int m = 0; m < n_mob; m++) {  // particles, owners in the particle system's set order
        const int e = EB(s, EB_SPARK_ORDER, m, env);
        for (int k = 0; k < kSparks; k++) {
            const float life = SP(s, buf, 2, e, k, env);
            if (life <= 0.0f) continue;
            const int4 d = P.desc(kTexSpark);
            const float lr = (5.0f - life) / 5.0f;
            const float alpha = 0.5f * (1.0f - lr);
            const float scale = 0.45f * (0.4f * lr + 0.6f);
            const float oy = -lr * 0.17f;
            P.draw(kTexSpark, SP(s, buf, 0, e, k, env) * kUnitPx - 0.5f * d.y * scale,
                   (SP(s, buf, 1, e, k, env) + oy) * kUnitPx - 0.5f * d.z * scale, scale * kUnitPx / d.y, alpha);
        }
    }
    const int n_sprites = (sflags & kFlagListed) ? SI(s, I_NENT, env) : 0;
    for (int k = 0; k < n_sprites; k++) {
        const int e = EB(s, EB_DRAW_ORDER, k, env);
        const int dyn = DB(s, buf, e, env);
        if (!(dyn & kDynTexSet)) continue;
        const int tex = EB(s, EB_TEX, e, env) + ((dyn & kDynFrame) ? 1 : 0);
        const float scale = 1.0f * 1.0f;
        P.draw(tex, (DF(s, buf, DF_X, e, env) + -0.5f) * kUnitPx, (EY(s, e, env) + -0.5f) * kUnitPx,
               scale * kUnitPx / P.desc(tex).y, 1.0f, (dyn & kDynFlip) != 0);
    }
    {
        const bool ground = (sflags & kFlagGround) != 0;
        int tex;
        if (fabsf(SF(s, F_AVX, env)) < 0.01f && ground)
            tex = kTexStand + alien;
        else if (!ground)
            tex = kTexJump + alien;
        else if (SF(s, F_APHASE, env) > 0.5f)
            tex = kTexWalk2 + alien;
        else
            tex = kTexWalk1 + alien;
        const float px = SF(s, F_AX, env) - 0.5f, py = SF(s, F_AY, env) - 2.0f;
        P.draw(tex, px * kUnitPx, py * kUnitPx, kUnitPx / P.desc(tex).y, 1.0f, (sflags & kFlagForward) == 0);
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
class CoinrunGame final : public Game {
   public:
    const char* name() const override { return "coinrun"; }
    bool set_game_flags(uint32_t flags) override {
        const uint32_t known = PGV_COINRUN_NO_PIT | PGV_COINRUN_NO_CRATE | PGV_COINRUN_NO_DY | PGV_COINRUN_NO_MOBS;
        s_.no = flags & known;
        return (flags & ~known) == 0;
    }

    std::vector<std::string> texture_names() const override {
        static const char* themes[6] = {"Dirt", "Grass", "Planet", "Sand", "Snow", "Stone"};
        static const char* walkers[9] = {"slimeBlock", "slimePurple", "slimeBlue", "slimeGreen", "mouse",
                                         "snail",      "ladybug",     "wormGreen", "wormPink"};
        static const char* crates[4] = {"boxCrate", "boxCrate_double", "boxCrate_single", "boxCrate_warning"};
        static const char* aliens[5] = {"Beige", "Blue", "Green", "Pink", "Yellow"};
        std::vector<std::string> v;
        auto lower = [](std::string t) {
            for (auto& ch : t) ch = static_cast<char>(tolower(ch));
            return t;
        };
        for (auto t : themes) v.push_back(std::string("kenney/Ground/") + t + "/" + lower(t) + "Mid.png");
        for (auto t : themes) v.push_back(std::string("kenney/Ground/") + t + "/" + lower(t) + "Center.png");
        v.push_back("kenney/Tiles/lavaTop_low.png");
        v.push_back("kenney/Tiles/lava.png");
        for (auto c : crates) v.push_back(std::string("kenney/Tiles/") + c + ".png");
        for (auto w : walkers) {
            v.push_back(std::string("kenney/Enemies/") + w + ".png");
            v.push_back(std::string("kenney/Enemies/") + w + "_move.png");
        }
        v.push_back("kenney/Enemies/sawHalf.png");
        v.push_back("kenney/Enemies/sawHalf_move.png");
        v.push_back("kenney/Items/coinGold.png");
        for (auto pose : {"_stand.png", "_jump.png", "_walk1.png", "_walk2.png"})
            for (auto a : aliens) v.push_back(std::string("kenney/Players/128x256/") + a + "/alien" + a + pose);
        v.push_back("misc_assets/iconCircle_white.png");
        for (const char* b : {"alien_bg", "another_world_bg", "back_cave", "caverns", "cyberpunk_bg", "parallax_forest",
                              "scifi_bg", "scifi2_bg", "living_tissue_bg", "airadventurelevel1", "airadventurelevel2",
                              "airadventurelevel3", "airadventurelevel4", "cave_background", "blue_desert",
                              "blue_grass", "blue_land", "blue_shroom", "colored_desert", "colored_grass",
                              "colored_land", "colored_shroom", "landscape1", "landscape2", "landscape3",
                              "landscape4", "battleback1", "battleback2", "battleback3", "battleback4", "battleback5",
                              "battleback6", "battleback7", "battleback8", "battleback9", "battleback10", "sunrise"})
            v.push_back(std::string("platform_backgrounds/") + b + ".png");
        for (const char* fam : {"beach", "fantasy", "candy"})
            for (int k = 1; k <= 4; k++)
                v.push_back(std::string("platform_backgrounds_2/") + fam + std::to_string(k) + ".png");
        return v;
    }

    std::string check_atlas(const std::vector<std::pair<int, int>>& sizes) const override {
        if (static_cast<int>(sizes.size()) != kTexCount) return "coinrun: unexpected texture count";
        for (int t = kTexTop; t < kTexWalker; t++)  // the row composer assumes one tile texture size
            if (sizes[t] != sizes[kTexMid]) return "coinrun: tile textures differ in size";
        return "";
    }

    static size_t align256(size_t x) { return (x + 255) & ~size_t(255); }
    struct Layout {
        size_t shadow, slot, mt, tiles, f, i, ey, eb, df, db, spark, scratch, total;
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
        l.ey = take(size_t(kMaxEnt) * n * 4);
        l.eb = take(size_t(EB_COUNT) * kMaxEnt * n);
        l.df = take(size_t(2) * DF_COUNT * kMaxEnt * n * 4);
        l.db = take(size_t(2) * kMaxEnt * n);
        l.spark = take(size_t(2) * 3 * kMaxEnt * kSparkRow * n * 4);
        l.scratch = take(size_t(SC_COUNT) * n * 4);
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
        s_.ey = reinterpret_cast<float*>(p + l.ey);
        s_.eb = p + l.eb;
        s_.df = reinterpret_cast<float*>(p + l.df);
        s_.db = p + l.db;
        s_.spark = reinterpret_cast<float*>(p + l.spark);
        s_.scratch = reinterpret_cast<float*>(p + l.scratch);
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
        // the auto-resets: a prefetched level is installed beside the agents (logic_kernel's row 1); the levels that were
        // not ready — none in steady state — are generated synchronously behind it
        const bool fused = prefetch() != 0;
        const int served = reset_served_mark(step_index), due = reset_due_mark(step_index);
        if (!fused) LevelLaunch<Gen>::auto_reset(st, s_, prefetch(), io, plan, PG_RESET_SPAN, served, due);
        // agents, prefetched installs and entities side by side (logic_kernel)
        const dim3 blocks((s_.n + 63) / 64, 2 + kMobRows + kStaticRows);
        // (bit 24: no reach — the tests' way to resolve_kernel's fallback, see hazard_near)
        const float reach_x = (debug_flags & kDebugCoinrunNoReach) ? 0.0f : kReachX;
        const float reach_y = (debug_flags & kDebugCoinrunNoReach) ? 0.0f : kReachY;
        hipLaunchKernelGGL(logic_kernel, blocks, dim3(64), 0, st, s_, actions, run_seed, step_index, env_offset, io,
                           prefetch(), plan, fused ? 1 : 0, reach_x, reach_y);
        // (Measured and rejected, round 5, when agents and entities were two launches: that level kernel on a stream of
        // its own beside the entities — forked behind the agents, joined in front of resolve_kernel: 123.6 against 126.7 M
        // env-steps/s, three same-box pairs; the two event hand-overs cost more than they hide.)
        // The resets no prefetched level lay ready for ride in resolve_kernel's second row (see there: +0.6 to 1 %).
        hipLaunchKernelGGL(resolve_kernel, dim3((s_.n + 63) / 64, fused ? 2 : 1), dim3(64), 0, st, s_, io, step_index, prefetch(),
                           plan);
    }
    bool launch_frame(hipStream_t st, int env, uint32_t* d_px, int w, int h) override {
        hipLaunchKernelGGL(frame_kernel, dim3(1), dim3(kFrameThreads), 0, st, s_, atlas_, env, FrameTarget{d_px, w, h});
        return true;
    }
    void launch_prepass(hipStream_t st, const uint8_t* mask) override {
        if (!(debug_flags & (1 | kDebugNoPrepass)) && !PG_ABL(debug_flags, 1 << 22))  // (experiment: the last frame's pre-pass again)
            hipLaunchKernelGGL(setup_kernel, dim3((s_.n + kPrepEnvs - 1) / kPrepEnvs), dim3(kPrepThreads), 0, st, s_, atlas_, mask, debug_flags);
    }
    void launch_render(hipStream_t st, const uint8_t* mask, StepIO io) override {
        hipLaunchKernelGGL(render_kernel, dim3(s_.n), dim3(64 * kRenderWaves), 0, st, s_, atlas_, mask, io,
                           debug_flags);
    }
    // Scratch (not state, not in snapshots): the pre-pass's hand-over, then the hazards' boxes — written by logic_kernel's
    // entity lanes and read by resolve_kernel of the same step, only where a candidate bit of that step says so.
    static size_t up256(size_t b) { return (b + 255) & ~size_t(255); }
    size_t scratch_bytes(int n) const override {
        return up256(prep_bytes(n, kGrid, kBlitWords, false)) + up256(size_t(kMaxEnt) * n * sizeof(float4)) +
               up256(size_t(kMaxEnt) * n * sizeof(float2));
    }
    void bind_scratch(void* d_scratch, int n) override {
        s_.prep = prep_bind(d_scratch, n, kGrid, kBlitWords, false);
        uint8_t* p = static_cast<uint8_t*>(d_scratch) + up256(prep_bytes(n, kGrid, kBlitWords, false));
        s_.hazx = reinterpret_cast<float4*>(p);
        s_.hazy = reinterpret_cast<float2*>(p + up256(size_t(kMaxEnt) * n * sizeof(float4)));
    }

    // Same layout as oracle/pgo_coinrun.cpp Coinrun::dump_state.
    int dump_state(hipStream_t st, int env, float* out, int cap) override {
        hipStreamSynchronize(st);
        auto rd_f = [&](const float* base, size_t idx) {
            float v;
            hipMemcpy(&v, base + idx, 4, hipMemcpyDeviceToHost);
            return v;
        };
        auto rd_i = [&](size_t idx) {
            int32_t v;
            hipMemcpy(&v, s_.i + idx, 4, hipMemcpyDeviceToHost);
            return v;
        };
        auto rd_b = [&](const uint8_t* base, size_t idx) {
            uint8_t v;
            hipMemcpy(&v, base + idx, 1, hipMemcpyDeviceToHost);
            return v;
        };
        const size_t n = s_.n;
        auto f = [&](int field) { return rd_f(s_.f, size_t(field) * n + env); };
        const int flags = rd_i(size_t(I_FLAGS) * n + env), themes = rd_i(size_t(I_THEMES) * n + env);
        const int n_ent = rd_i(size_t(I_NENT) * n + env);
        const int buf = (flags & kFlagBuf) ? 1 : 0;
        std::vector<float> v;
        v.push_back(f(F_AX));
        v.push_back(f(F_AY));
        v.push_back(f(F_AVX));
        v.push_back(f(F_AVY));
        v.push_back((flags & kFlagGround) ? 1.0f : 0.0f);
        v.push_back((flags & kFlagForward) ? 1.0f : 0.0f);
        v.push_back(f(F_APHASE));
        v.push_back(f(F_CAMX));
        v.push_back(f(F_CAMY));
        v.push_back(static_cast<float>(themes & 0xff));
        v.push_back(f(F_BGSHIFT));
        v.push_back(static_cast<float>((themes >> 8) & 0xff));
        v.push_back(static_cast<float>((themes >> 16) & 0xff));
        v.push_back(static_cast<float>(n_ent));
        for (int e = 0; e < n_ent; e++) {
            const int kind = rd_b(s_.eb, (size_t(EB_KIND) * kMaxEnt + e) * n + env);
            const int dyn = rd_b(s_.db, (size_t(buf) * kMaxEnt + e) * n + env);
            auto df = [&](int field) { return rd_f(s_.df, ((size_t(buf) * DF_COUNT + field) * kMaxEnt + e) * n + env); };
            v.push_back(df(DF_X));
            v.push_back(rd_f(s_.ey, size_t(e) * n + env));
            v.push_back(kind == kMob ? df(DF_VX) : 0.0f);
            v.push_back((dyn & kDynFrame) ? 1.0f : 0.0f);
            v.push_back(kind == kCoin ? 0.0f : df(DF_ANIM_T));
        }
        const int m = cap < static_cast<int>(v.size()) ? cap : static_cast<int>(v.size());
        for (int k = 0; k < m; k++) out[k] = v[k];
        return static_cast<int>(v.size());
    }
    int dump_tiles(hipStream_t st, int env, uint8_t* out, int cap) override {
        hipStreamSynchronize(st);
        const int m = cap < W * H ? cap : W * H;
        hipMemcpy(out, s_.tiles + size_t(env) * W * H, m, hipMemcpyDeviceToHost);
        for (int k = 0; k < m; k++) out[k] &= 7;
        return m;
    }

   private:
    State s_{};
    AtlasView atlas_{};
};

}  // namespace coinrun

}  // namespace PG_VARIANT_NS

std::unique_ptr<Game> PG_FACTORY(make_coinrun)() { return std::make_unique<PG_VARIANT_NS::coinrun::CoinrunGame>(); }

}  // namespace pg
