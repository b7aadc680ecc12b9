// chaser on gfx950: pac-man-like on an 11×11 maze, three egg-born enemies, orbs and points.
//
// Reference:
//   step   games/chaser/chaser.cpp:282-334, common_systems.cpp:305-444 (agent), :117-295 (enemies), :66-106 (points),
//          :8-39 (sprite list)
//   render games/chaser/chaser.cpp:390-416, tilemap.cpp:245-267, common_systems.cpp:41-63, :446-460
//   reset  games/chaser/chaser.cpp:418-443, tilemap.cpp:80-243, maze_generator.cpp:47-130
// Config = the reference's compile-time default, easy_mode (11×11, 3 enemies; chaser/tilemap.h:39-41).
//
// Quirk kept on purpose (D21, oracle/pgo_chaser.cpp): the reference's `abs(<float>)` calls in common_systems.cpp
// compile to glibc's `int abs(int)` — the argument is truncated first — so the "close to the cell centre" tests always
// pass and the chase heuristic compares sums of truncated integers.  qabs(s, ) below is that.
//
// Machine mapping: logic one lane per env (SoA across envs) — the enemies draw from the env's mt19937 inside the step,
// in the iteration order of the enemy System's std::unordered_set; render two wavefronts per env; level generation one
// wavefront per env.  Because of the in-step draws the next level cannot be generated ahead of time (pg_prefetch.h
// is used with prefetch off: every reset carves its maze inside the step).
#include "../../include/procgen2_vec.h"
#ifndef PG_CHASER_BLEND_PAIRS
#define PG_BLEND_CHANNELWISE 1  // pg_geom.h blend_px
#endif
#include "pg_engine.h"
#include "pg_frame.h"
#include "pg_gang.h"
#include "pg_geom.h"
#include "pg_kruskal.h"
#include "pg_order.h"
#include "pg_prefetch.h"
#include "pg_render.h"
#include "pg_prepass.h"
#include "pg_rng.h"
#include "pg_setorder.h"

namespace pg {
namespace PG_VARIANT_NS {
namespace chaser {

// tilemap.cpp:85-99: world side, enemies, and the orb count of one random quadrant (1 + extra_orb_sign)
#if PG_VARIANT == 0  // easy_mode — the reference's compile-time default (tilemap.h:40)
constexpr int W = 11, kMobs = 3, kExtraOrbSign = 0;
#elif PG_VARIANT == 1  // hard_mode
constexpr int W = 13, kMobs = 3, kExtraOrbSign = -1;
#elif PG_VARIANT == 2  // extreme_mode
constexpr int W = 19, kMobs = 5, kExtraOrbSign = 1;
#else
#error "chaser: unknown PG_VARIANT"
#endif
constexpr int H = W, kCells = W * H, kTileStride = (kCells + 127) / 128 * 128;
constexpr int kOrbs = 4 + kExtraOrbSign, kFirstPoint = kOrbs + kMobs;
// a perfect maze on the even-even cells of an odd side has 2·((W+1)/2)² − 1 open cells (71 at 11×11); every one of
// them but the agent's holds an entity: orbs, eggs, and a point on the rest
constexpr int kOpenCells = 2 * ((W + 1) / 2) * ((W + 1) / 2) - 1;
constexpr int kMaxEnt = (kOpenCells - 1 + 7) / 8 * 8;
constexpr bool kWideCells = kCells > 256;  // cell indices need a second byte
constexpr int kDueBlocks = 1024;            // wavefronts of the in-step level kernel (due_level_kernel): one each for the envs that are due, usually
static_assert(kMaxEnt <= 255 && kMaxEnt <= kRankMax, "entity ids are bytes; draw lists use the equal-key rank table");
enum Tile : uint8_t { kEmpty = 0, kWall = 1, kMarker = 2 };
enum Kind { kOrb = 0, kPoint = 1, kEgg = 2 };

enum Tex {
    kTexWall = 0,
    kTexOrb = 1,
    kTexPoint = 2,
    kTexEnemy = 3,  // egg, flying 1-3, walking
    kTexAgent = 8,
    kTexFloor = 9,  // 9
    kTexCount = 18
};

enum { F_AX, F_AY, F_AVX, F_AVY, F_NVX, F_NVY, F_INPUT_T, F_ANIM_T, F_EAT_T, F_BGSHIFT, F_COUNT };
enum { I_FLAGS, I_ANIM_I, I_BG, I_NENT, I_NDRAW, I_HASH_SPRITE, I_HASH_MOB, I_COUNT };
constexpr int kFlagListed = 1;
enum { MF_X, MF_Y, MF_VX, MF_VY, MF_HATCH, MF_COUNT };
enum { EB_INFO, EB_CELL, EB_ORDER, EB_DRAW, EB_CELL_HI, EB_COUNT };
constexpr int kKindMask = 3, kAlive = 4;

// One generated level (LDS → live state; the shadow slots stay unused in this game).
struct Level {
    uint8_t tiles[kTileStride];
    float ax, ay, bgshift;
    int32_t bg, n_ent;
    uint16_t cell[kMaxEnt];
    uint8_t order[kMaxEnt];
    uint8_t mob_order[8];
};
static_assert(sizeof(Level) % 4 == 0, "Level is copied as 32-bit words");

struct GenLds {
    uint32_t mt[kMtWords];
    KruskalLds k;
    int32_t touch[260], chain[260], tail_sum[kMaxEnt + 4];  // pg_setorder.h scratch (≤ 257 buckets, ≤ kMaxEnt keys)
    int16_t link[kMaxEnt], tmp[kMaxEnt], keys[kMaxEnt];
    uint16_t free_cells[kCells + 7];
};

struct State {
    int n;
    Level* shadow;   // [n]  (pg_prefetch.h interface; never filled)
    int32_t* slot;   // [n]
    uint32_t* mt;    // [n][625]
    uint8_t* tiles;  // [n][128]  column-major y + x*H
    float* f;        // [F_COUNT][n]
    int32_t* i;      // [I_COUNT][n]
    float* mf;       // [MF_COUNT][kMobs][n]
    uint8_t* mb;     // [2][kMobs][n]   texture index (0 egg, 1-3 flying, 4 walking), iteration order of the enemy set
    uint8_t* eb;     // [n][EB_COUNT][kMaxEnt]  per-env contiguous: a gang's and the render wavefronts' lanes index it by entity
    const uint8_t* ranks;  // pg_order.h equal-key sort ranks
    int float_abs;         // game_flags PGV_CHASER_FLOAT_ABS: which `abs` the reference's abs(<float>) calls are (D21)
    // The envs the level kernel has reset in this step, for the late render pass (render_list_kernel): a list and, per
    // step parity, its length — the late pass of a step zeroes the other parity's, which the next step fills.
    int32_t* reset_list;   // [n]
    int32_t* reset_count;  // [2]
    // … and the envs whose episode ended in the step before (pending 3 → 1, settled by that step's late pass), per step
    // parity: what the in-step level kernel walks (due_level_kernel) instead of looking at every env's byte.
    int32_t* due_list;     // [2][n]
    int32_t* due_count;    // [2]
    // A second buffer for every env's random stream and its selector (pg_gang.h GangRng): the stream's next 624 words are
    // worked out ahead of the step that needs them (the step's late pass, render_list_kernel → pg_rng.h
    // mt_next_block_wave) into whichever buffer is not current, and the gang that runs out of numbers changes buffers.
    // Scratch memory, not state: whoever takes the stream out of the engine (snapshots: prepare_save) or hands it to a
    // level generator (generate) first makes mt[env] the current block again.
    uint32_t* mt_other;    // [n][kMtN]
    uint8_t* mt_sel;       // [n]  bit 1: the current words are in mt_other[env]; bit 0: the other buffer holds the next block
    int parity, listing;   // host-set per launch: step & 1; whether the list is kept at all (resets on their own stream)
    // The point sprite's texels that are not fully transparent lie in columns point_box.x..y and rows .z..w (found when
    // the atlas is loaded); point_solid: every texel in that box is opaque.
    int4 point_box;
    int point_solid;
    // The row composer's span and hand-over tables: the camera shows the whole world, always, so they are worked out
    // once (prepare_kernel, when the envs are made) and every frame reads them from here (pg_render.h compose_prepare).
    ComposeHand* prepared;
    // The render pre-pass's tables (scratch memory, not state).  The camera never moves, so an orb's or a point's draw is a
    // function of its cell: worked out once for every cell (prepare_kernel); what moves — the enemies and the agent — and
    // the background's two axes once per frame and env by a dense kernel (setup_kernel) instead of by both wavefronts of
    // the env's render workgroup, 64 lanes at a time whatever the number of draws.
    // The BASE LAYER of every env's frame (round 5): background + walls as they land on the 64 × 64 target, 0x00BBGGRR
    // words, 16 KB an env.  The camera never moves, the walls and the backdrop are the level's: that layer is the same
    // picture in every frame of an episode — it IS the episode's first frame (the draw list is empty then, D2) — and
    // working it out again every frame was 55 % of the render kernel (the row loop's one-texel pass 510 vector instructions
    // per wave, its general form, which the walls' translucent corners send one pixel row in five through, 443; 75 texel
    // gathers).  Whoever renders a frame the complete way — the late pass over the envs reset in a step, pgv_reset's
    // render, the debug paths — leaves the layer here; the step's frames start from a 16-KB copy of it and only stamp the
    // points that are still there (overlay_points).  Scratch memory, not state: derived, rebuilt after pgv_load_state.
    uint32_t* base;  // [n][64 * 64]
    // A point's stamp at every cell: the pixels of the frame its draw leaves something on (at most kStampMax: the opaque
    // middle of its texture as it lands there), as pairs {pixel = 64 · row + column, or kNoStamp; 0x00BBGGRR} — worked out
    // once, on the host, when the atlas is loaded (ChaserGame::extend_atlas: Renderer::render_texture's arithmetic and
    // raster rules S1–S3 through pg_geom.h, one call per cell) and kept behind the textures in the atlas array.
    uint32_t stamps;  // word offset in AtlasView::texels of [kCells][kStampMax][2]
    struct Prep {
        uint32_t* cell_blits;  // [2][kCells][kBlitWords]   orb, point at each cell (pg_render.h BlitWords; word 1 = 0: not drawn)
        uint32_t* movers;      // [n][kMovers][kBlitWords]  enemy 0 … kMobs − 1, the agent
        uint32_t* bg;          // [n][8]                    background, x axis then y axis: d0 | dn << 16, s0 | sn << 16, first texel, width
    } prep;
};
constexpr int kMovers = kMobs + 1;
constexpr int kStampMax = 16;
constexpr uint32_t kNoStamp = 0xffffffffu;

PG_D float& SF(const State& s, int field, int env) { return s.f[size_t(field) * s.n + env]; }
PG_D int32_t& SI(const State& s, int field, int env) { return s.i[size_t(field) * s.n + env]; }
PG_D float& MF(const State& s, int field, int m, int env) { return s.mf[(size_t(field) * kMobs + m) * s.n + env]; }
PG_D uint8_t& MB(const State& s, int field, int m, int env) { return s.mb[(size_t(field) * kMobs + m) * s.n + env]; }
PG_D uint8_t& EB(const State& s, int field, int e, int env) { return s.eb[(size_t(env) * EB_COUNT + field) * kMaxEnt + e]; }
PG_D int ent_cell(const State& s, int e, int env) {
    int c = EB(s, EB_CELL, e, env);
    if (kWideCells) c |= EB(s, EB_CELL_HI, e, env) << 8;
    return c;
}

PG_D int tile_at(const uint8_t* t, int x, int y) {  // tilemap.h:79-84: out of bounds is neither empty nor wall
    if (x < 0 || y < 0 || x >= W || y >= H) return -1;
    return t[y + x * H];
}
// D21: the unqualified abs(<float>) of common_systems.cpp:165-166,206,346-420.  Default: glibc's int abs(int) on the
// truncated argument (a small integer, exact as a float); with PGV_CHASER_FLOAT_ABS: float std::abs(float).
// See oracle/pgo_chaser.cpp qabs and tests/golden/appendix_c.json for why both exist.
PG_D float qabs(const State& s, float v) {
    const int t = static_cast<int>(v);
    return s.float_abs ? fabsf(v) : static_cast<float>(t < 0 ? -t : t);
}
PG_D int sign_of(float x) { return x == 0.0f ? 0 : (x > 0.0f) * 2 - 1; }  // helpers.h:31-36

// Iteration order of a std::unordered_set<int> that kept its bucket array across clear() after inserting
// L.keys[0..n) (packed = buckets | next_resize << 16; a fresh set is packed = 1).  Result in L.keys.  All lanes.
PG_D void set_order(GenLds& L, int32_t& packed, int n, int lane) {
    int32_t buckets = packed & 0xffff, next_resize = packed >> 16;
    wave_set_order(L.keys, n, buckets, next_resize, SetOrderScratch{L.touch, L.chain, L.link, L.tail_sum, L.tmp}, lane);
    packed = buckets | (next_resize << 16);
}

// reset() (chaser.cpp:418-443 + tilemap.cpp:80-243) for one env by one wavefront; the level is left in `lv` (LDS).
PG_D void generate(const State& s, int env, GenLds& L, Level& lv, bool reseed, uint32_t seed, int lane) {
    uint32_t* gmt = s.mt + size_t(env) * kMtWords;
    if (reseed) {
        if (lane == 0) mt_seed(L.mt, seed);
    } else {
        // (the stream's current words are wherever its gang left them: State::mt_sel; they go back into mt[env] below)
        const uint32_t* cur = (s.mt_sel[env] & 2) ? s.mt_other + size_t(env) * kMtN : gmt;
        for (int k = lane; k < kMtN; k += 64) L.mt[k] = cur[k];
        if (lane == 0) L.mt[kMtN] = gmt[kMtN];
    }
    __syncthreads();
    uint32_t* mt = L.mt;
    Carver carver{L.k, 0, 0, 0, 0};
    carver.carve(W, mt, lane);
    const int extra_quad = wave_rng_int(mt, 0, 3, lane);  // tilemap.cpp:121-122; without effect when the sign is 0
    for (int c = lane; c < kTileStride; c += 64) {
        const int x = c / H, y = c % H;
        lv.tiles[c] = (c < kCells && carver.get(x + 1, y + 1) == 0) ? kEmpty : kWall;
    }
    __syncthreads();
    // orbs per quadrant (tilemap.cpp:124-170): 1, or 1 + extra_orb_sign in the extra quadrant; positions are drawn
    // among the quadrant's open cells (x-major) with linear probing, collected in a fresh std::unordered_set<int> and
    // spawned in ITS iteration order — for two elements always the second insertion first.
    int n_orbs = 0;
    for (int q = 0; q < 4; q++) {
        const int want = 1 + (q == extra_quad ? kExtraOrbSign : 0);
        if (want == 0) continue;  // the distribution is constructed, nothing is drawn
        int count = 0;
        for (int c = 0; c < kCells; c++) {
            const int x = c / H, y = c % H;
            if (lv.tiles[c] != kWall && ((x >= W / 2) * 2 + (y >= H / 2)) == q) count++;  // markers were open cells
        }
        int pos[2];
        pos[0] = wave_rng_int(mt, 0, count - 1, lane);
        pos[1] = -1;
        if (want == 2) {
            int p = wave_rng_int(mt, 0, count - 1, lane);
            while (p == pos[0]) p = (p + 1) % count;
            pos[1] = pos[0];  // iteration order: the second insertion, then the first
            pos[0] = p;
        }
        if (lane == 0) {
            int cells_at[2] = {-1, -1};
            int seen = 0;
            for (int c = 0; c < kCells; c++) {
                const int x = c / H, y = c % H;
                if (lv.tiles[c] != kWall && ((x >= W / 2) * 2 + (y >= H / 2)) == q) {
                    if (seen == pos[0]) cells_at[0] = c;
                    if (seen == pos[1]) cells_at[1] = c;
                    seen++;
                }
            }
            for (int k = 0; k < want; k++) {
                lv.cell[n_orbs + k] = static_cast<uint16_t>(cells_at[k]);
                lv.tiles[cells_at[k]] = kMarker;
            }
        }
        n_orbs += want;
        __syncthreads();
    }
    // the agent's cell and the three eggs (tilemap.cpp:172-213): four distinct free cells, taken from a
    // std::unordered_set<int> in ITS iteration order — first the agent, then the eggs
    int n_free = 0;
    for (int c = 0; c < kCells; c++)
        if (lv.tiles[c] == kEmpty) {
            if (lane == 0) L.free_cells[n_free] = static_cast<uint16_t>(c);
            n_free++;
        }
    __syncthreads();
    int16_t picked[kMobs + 1];
#pragma unroll
    for (int j = 0; j < kMobs + 1; j++) {
        int pos = wave_rng_int(mt, 0, n_free - 1, lane);
        for (bool again = true; again;) {
            again = false;
            for (int k = 0; k < j; k++)
                if (picked[k] == pos) {
                    pos = (pos + 1) % n_free;
                    again = true;
                    break;
                }
        }
        picked[j] = static_cast<int16_t>(pos);
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < kMobs + 1; j++) L.keys[j] = picked[j];
    }
    __syncthreads();
    int32_t fresh = 1;
    set_order(L, fresh, kMobs + 1, lane);
    if (lane == 0) {
        const int16_t* order = L.keys;
        const int start = L.free_cells[order[0]];
        lv.tiles[start] = kMarker;
        lv.ax = static_cast<float>(start / H) + 0.5f;
        lv.ay = static_cast<float>(H - 1 - start % H) + 0.5f;
        for (int m = 0; m < kMobs; m++) {
            const int cell = L.free_cells[order[1 + m]];
            lv.cell[kOrbs + m] = static_cast<uint16_t>(cell);
            lv.tiles[cell] = kMarker;
        }
        int n_ent = kFirstPoint;  // a point on every cell still free (tilemap.cpp:215-225)
        for (int c = 0; c < kCells; c++)
            if (lv.tiles[c] == kEmpty) lv.cell[n_ent++] = static_cast<uint16_t>(c);
        lv.n_ent = n_ent;
        for (int c = 0; c < kCells; c++)
            if (lv.tiles[c] == kMarker) lv.tiles[c] = kEmpty;
    }
    __syncthreads();
    const int bg = wave_rng_int(mt, 0, 8, lane);
    const float shift = wave_rng_real(mt, 0.0f, 1.0f, lane);
    if (lane == 0) {
        lv.bg = bg;
        lv.bgshift = shift;
    }
    {   // entity-set orders of the episode: sprites = every non-agent entity, enemies = the eggs
        const int n_ent = lv.n_ent;
        for (int k = lane; k < n_ent; k += 64) L.keys[k] = static_cast<int16_t>(k);
        __syncthreads();
        int32_t packed = SI(s, I_HASH_SPRITE, env);
        set_order(L, packed, n_ent, lane);
        if (lane == 0) SI(s, I_HASH_SPRITE, env) = packed;
        for (int k = lane; k < n_ent; k += 64) lv.order[k] = static_cast<uint8_t>(L.keys[k]);
        __syncthreads();
        if (lane < kMobs) L.keys[lane] = static_cast<int16_t>(kOrbs + lane);
        __syncthreads();
        packed = SI(s, I_HASH_MOB, env);
        set_order(L, packed, kMobs, lane);
        if (lane == 0) SI(s, I_HASH_MOB, env) = packed;
        if (lane < kMobs) lv.mob_order[lane] = static_cast<uint8_t>(L.keys[lane]);
    }
    __syncthreads();
    for (int k = lane; k < kMtWords; k += 64) gmt[k] = L.mt[k];
    __syncthreads();
}

PG_HD float cell_x(int cell) { return static_cast<float>(cell / H) + 0.5f; }
PG_HD float cell_y(int cell) { return static_cast<float>(H - 1 - cell % H) + 0.5f; }

// The level becomes the env's live state (what reset() and the component constructors initialise).
PG_D void install(const State& s, int env, const Level& lv, int lane) {
    if (lane == 0) s.mt_sel[env] = 0;  // the generator has left the stream in mt[env]; what was made ahead of it is stale
    uint32_t* tiles = reinterpret_cast<uint32_t*>(s.tiles + size_t(env) * kTileStride);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(lv.tiles);
    for (int k = lane; k < kTileStride / 4; k += 64) tiles[k] = src[k];
    const int n_ent = lv.n_ent;
    for (int e = lane; e < n_ent; e += 64) {
        const int kind = e < kOrbs ? kOrb : (e < kFirstPoint ? kEgg : kPoint);
        EB(s, EB_INFO, e, env) = static_cast<uint8_t>(kind | kAlive);
        EB(s, EB_CELL, e, env) = static_cast<uint8_t>(lv.cell[e] & 0xff);
        if (kWideCells) EB(s, EB_CELL_HI, e, env) = static_cast<uint8_t>(lv.cell[e] >> 8);
        EB(s, EB_ORDER, e, env) = lv.order[e];
    }
    if (lane < kMobs) {
        const int cell = lv.cell[kOrbs + lane];
        MF(s, MF_X, lane, env) = cell_x(cell);
        MF(s, MF_Y, lane, env) = cell_y(cell);
        MF(s, MF_VX, lane, env) = 0.0f;
        MF(s, MF_VY, lane, env) = 0.0f;
        MF(s, MF_HATCH, lane, env) = 0.0f;
        MB(s, 0, lane, env) = 0;
        MB(s, 1, lane, env) = lv.mob_order[lane];
    }
    if (lane == 0) {
        SF(s, F_AX, env) = lv.ax;
        SF(s, F_AY, env) = lv.ay;
        SF(s, F_AVX, env) = 0.0f;
        SF(s, F_AVY, env) = 0.0f;
        SF(s, F_NVX, env) = 0.0f;
        SF(s, F_NVY, env) = 0.0f;
        SF(s, F_INPUT_T, env) = 0.0f;
        SF(s, F_ANIM_T, env) = 0.0f;
        SF(s, F_EAT_T, env) = 0.0f;
        SF(s, F_BGSHIFT, env) = lv.bgshift;
        SI(s, I_FLAGS, env) = 0;  // draw list cleared
        SI(s, I_ANIM_I, env) = 0;
        SI(s, I_BG, env) = lv.bg;
        SI(s, I_NENT, env) = n_ent;
        SI(s, I_NDRAW, env) = 0;
    }
}

// What cenv_make leaves in an env besides the seeded RNG, split by owner: the generator chain (bucket counts of
// the sets that survive clear()) and the live state.  Level-seed mode (pg_engine.h LevelPlan) rebuilds every
// level from here.
PG_D void fresh_chain(const State& s, int env) {
    SI(s, I_HASH_SPRITE, env) = 1;  // empty unordered_set: one bucket, next_resize 0
    SI(s, I_HASH_MOB, env) = 1;
}
PG_D void fresh_live(const State& s, int env) {
}

struct Gen {  // pg_prefetch.h level_kernel<Gen>
    using State = chaser::State;
    using Level = chaser::Level;
    using GenLds = chaser::GenLds;
    PG_D static void generate(const State& s, int env, GenLds& L, Level& lv, bool reseed, uint32_t seed, int lane) {
        chaser::generate(s, env, L, lv, reseed, seed, lane);
    }
    PG_D static void install(const State& s, int env, const Level& lv, int lane) { chaser::install(s, env, lv, lane); }
    PG_D static void fresh_chain(const State& s, int env) { chaser::fresh_chain(s, env); }
    PG_D static void fresh_live(const State& s, int env) { chaser::fresh_live(s, env); }
    PG_D static void served(const State& s, int env) {  // (lane 0, after an auto-reset)
        if (s.listing) s.reset_list[atomicAdd(&s.reset_count[s.parity], 1)] = env;
    }
};

// The in-step auto-reset (pg_prefetch.h level_serve, mode 2) for the envs on the step's due list — two or three in a
// thousand — by a fixed, small grid: workgroup b takes entries b, b + gridDim.x, …  (Until round 6 a wavefront per env
// looked at its env's pending byte: 65 536 workgroups a step beside the logic kernel, 0.2 ms of dispatch for two hundred
// levels, and a fourteenth of the machine's launch slots in the seven-game slab.)  An entry whose env is no longer due —
// an explicit pgv_reset came in between — is skipped by level_serve's own test of the byte.
__global__ void __launch_bounds__(64) due_level_kernel(State s, StepIO io, LevelPlan plan) {
    __builtin_amdgcn_s_setprio(3);  // (pg_prefetch.h level_kernel: one long chain beside many short ones)
    const int count = s.due_count[s.parity];
    for (int item = blockIdx.x; item < count; item += gridDim.x) {  // (workgroup-uniform)
        const int env = s.due_list[size_t(s.parity) * s.n + item];
        level_serve<Gen>(s, 2, 1, 0, 0u, 0, nullptr, nullptr, io, plan, 2, 1, env, static_cast<int>(threadIdx.x));
        __syncthreads();
    }
}

// One env = one gang of kGang adjacent lanes (pg_gang.h).  The agent and the enemies — three to five of them, visited in
// the enemy set's order because their junction choices draw from the env's stream one after the other — are uniform
// over the gang; the orbs and points (up to 198 of them), each tested against the agent every sub-step, are dealt out
// over the lanes: entity e belongs to lane e mod kGang.  The env's tiles and entity table are staged in LDS once a step.
#ifndef PG_CHASER_GANG
#define PG_CHASER_GANG (kMobs <= 4 && W <= 11 ? 4 : 8)  // (the larger worlds' sixteen gangs a wavefront would not fit four wavefronts a SIMD in the LDS)  // (8 until round 6; with the streams changing buffers 4 lanes an env: 101.7 -> 104.8 M; 16: 95.2; an enemy needs a lane)
#endif
#ifndef PG_CHASER_WAVES
#define PG_CHASER_WAVES 4  // wavefronts per SIMD the logic kernel's registers are capped for
#endif
constexpr int kGang = PG_CHASER_GANG;
using Q = Gang<kGang>;
using Rng = GangRng<kGang>;

struct StepLds {  // one per gang
    uint8_t tiles[kTileStride];
    uint16_t cell[kMaxEnt];
    uint8_t info[kMaxEnt];
    uint8_t who[kTileStride];  // the orb or point that sits on a cell (each has its own), kNobody for none
};
constexpr int kNobody = 255;
static_assert(kMaxEnt <= kNobody, "entity ids fit a byte");

// System_Sprite_Render::update's list: the surviving sprites in set order, then std::sort on z (all 0.0).
PG_D void rebuild_draw_list(const State& s, const StepLds& L, Q q, int env, int n_ent) {
    int n = 0;
    for (int e0 = 0; e0 < n_ent; e0 += kGang) {
        const int e = e0 + q.g;
        n += __popc(q.ballot(e < n_ent && (L.info[e] & kAlive)));
    }
    const uint8_t* rank = s.ranks + rank_offset(n);  // equal keys: the sort is a fixed permutation for each n
    int r = 0;
    for (int k0 = 0; k0 < n_ent; k0 += kGang) {
        const int k = k0 + q.g;
        const int e = k < n_ent ? EB(s, EB_ORDER, k, env) : 0;
        const bool alive = k < n_ent && (L.info[e] & kAlive);
        const uint32_t m = q.ballot(alive);
        if (alive) EB(s, EB_DRAW, rank[r + __popc(m & ((1u << q.g) - 1u))], env) = static_cast<uint8_t>(e);
        r += __popc(m);
    }
    if (q.g == 0) SI(s, I_NDRAW, env) = n;
}

PG_D int tile_at(const StepLds& L, int x, int y) { return tile_at(L.tiles, x, y); }

// One enemy's turn of System_Mob_AI::update (common_systems.cpp:117-295) past its hatch test, in two parts (mob_head,
// mob_tail).  `draw`: where its random numbers come from — the env's stream itself (SerialDraws: every lane of the gang runs the same enemy) or three outputs
// peeked at the place in the stream this enemy is thought to start at (PeekDraws: one enemy per lane, see advance).
// Returns true when it caught the agent.
struct Mob {
    float px, py, vx, vy, hatch;
    int tex;
};
template <class Draw>
PG_D uint32_t draw_range(Draw& draw, uint32_t range) {  // = rng_int(0, range − 1): Lemire's nearly divisionless form (pg_rng.h)
    uint64_t product = static_cast<uint64_t>(draw.next()) * range;
    uint32_t low = static_cast<uint32_t>(product);
    if (low < range) {
        const uint32_t threshold = (0u - range) % range;
        while (low < threshold) {
            product = static_cast<uint64_t>(draw.next()) * range;
            low = static_cast<uint32_t>(product);
            if (draw.over()) break;  // (PeekDraws only: more outputs than were peeked — the caller takes the long way)
        }
    }
    return static_cast<uint32_t>(product >> 32);
}
// The part of a turn that draws nothing, done once: is the enemy at a junction (or standing), which ways are open, which
// of them an aggressive enemy would take.
struct MobHead {
    bool junction;
    int open, n_open, toward;  // bit j of `open`: direction j (−x, +x, −y, +y) is possible; toward: the aggressive choice
    float speed;
    int tex;
};
PG_D MobHead mob_head(const State& s, const StepLds& L, const Mob& m, float ax, float ay, float eat_t, int anim_i, float dt) {
    const float speed_low = 0.125f, speed_high = 0.25f;
    MobHead h;
    const float px = m.px, py = m.py, vx = m.vx, vy = m.vy;
    if (eat_t == 0.0f) {
        h.tex = anim_i < 3 ? 1 + anim_i : 1 + (5 - anim_i);
        h.speed = speed_high;
    } else {
        h.tex = 4;
        h.speed = speed_low;
    }
    const float fx = qabs(s, px - (static_cast<int>(px) + 0.5f)), fy = qabs(s, py - (static_cast<int>(py) + 0.5f));
    const bool at_junction = (fx < fy ? fy : fx) < h.speed * dt;  // std::max
    h.junction = (vx == 0.0f && vy == 0.0f) || at_junction;
    h.open = 0;
    h.n_open = 0;
    h.toward = 0;
    if (h.junction) {
        bool possible[4];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int dx = 2 * j - 1;
            const int id = tile_at(L, static_cast<int>(px) + dx, H - 1 - static_cast<int>(py));
            possible[j] = (id == kEmpty && dx != -sign_of(vx));
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int dy = 2 * j - 1;
            const int id = tile_at(L, static_cast<int>(px), H - 1 - (static_cast<int>(py) + dy));
            possible[2 + j] = (id == kEmpty && dy != -sign_of(vy));
        }
        float min_dist = 999999.0f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            h.open |= possible[j] ? 1 << j : 0;
            h.n_open += possible[j] ? 1 : 0;
            if (possible[j]) {
                const float dir_x = j == 0 ? -1.0f : (j == 1 ? 1.0f : 0.0f);
                const float dir_y = j == 2 ? -1.0f : (j == 3 ? 1.0f : 0.0f);
                float d = qabs(s, px + dir_x - ax) + qabs(s, py + dir_y - ay);
                if (eat_t > 0.0f) d = -d;
                if (d < min_dist) {
                    min_dist = d;
                    h.toward = j;
                }
            }
        }
    }
    return h;
}
// … and the part that does: which way (one number: aggressive or not; a second if not: the cusp-th open direction), the
// move, the agent met (a third: where the egg goes).  Returns true when the enemy caught the agent.
template <class Draw>
PG_D bool mob_tail(const StepLds& L, Mob& m, const MobHead& h, Draw& draw, const Box& agent_rect, float eat_t, int n_ent, float dt) {
    bool player_hit = false;
    float px = m.px, py = m.py;
    float vx = m.vx, vy = m.vy;
    int tex = h.tex;
    if (h.junction) {
        const bool be_aggressive = canonical_of(draw.next()) * (1.0f - 0.0f) + 0.0f < 0.5f;  // rng_real(0, 1)
        int select = 0;
        if (be_aggressive) {
            select = h.toward;
        } else if (h.n_open > 0) {
            const int cusp = static_cast<int>(draw_range(draw, static_cast<uint32_t>(h.n_open)));
            int sum = 0;
            bool found = false;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                sum += (h.open >> j) & 1;
                if (!found && sum > cusp) {
                    select = j;
                    found = true;
                }
            }
        }
        const float dir_x = select == 0 ? -1.0f : (select == 1 ? 1.0f : 0.0f);
        const float dir_y = select == 2 ? -1.0f : (select == 3 ? 1.0f : 0.0f);
        vx = dir_x * h.speed;
        vy = dir_y * h.speed;
        if (dir_x == 0.0f) px = static_cast<int>(px) + 0.5f;
        if (dir_y == 0.0f) py = static_cast<int>(py) + 0.5f;
    }
    px += vx * dt;
    py += vy * dt;
    if (box_hit(agent_rect, Box{-0.5f + px, -0.5f + py, 1.0f, 1.0f})) {
        if (eat_t == 0.0f) {
            player_hit = true;
        } else {  // back to an egg on a random point cell, without the world-y flip (D16)
            m.hatch = 0.0f;
            const int n_free = n_ent - kFirstPoint;
            const int cell = L.cell[kFirstPoint + static_cast<int>(draw_range(draw, static_cast<uint32_t>(n_free)))];
            px = cell / H + 0.5f;
            py = cell % H + 0.5f;
            tex = 0;
        }
    }
    m.px = px;
    m.py = py;
    m.vx = vx;
    m.vy = vy;
    m.tex = tex;
    return player_hit;
}
struct SerialDraws {  // the env's stream, one output after the other (all lanes of the gang call, with the same enemy)
    Rng& rng;
    PG_D uint32_t next() { return rng.next(); }
    PG_D bool over() const { return false; }
};
constexpr int kPeek = 3;  // an enemy's turn takes at most three outputs unless a range draw is rejected (≈ range / 2^32)
struct PeekDraws {        // kPeek outputs from a place in the stream, already tempered; asked for more, it says so (over)
    uint32_t w[kPeek];
    int taken;
    PG_D uint32_t next() {
        const uint32_t v = taken == 0 ? w[0] : (taken == 1 ? w[1] : w[2]);
        taken++;
        return v;
    }
    PG_D bool over() const { return taken > kPeek; }
};

PG_D void advance(const State& s, StepLds& L, Q q, int env, int action, bool serial_mobs, float& reward_out, bool& terminated_out) {
    const int n_ent = SI(s, I_NENT, env);
    int left = 0;  // orbs and points still there
    {   // stage: tiles as 32-bit words, the entity table, who sits where
        const uint32_t* src = reinterpret_cast<const uint32_t*>(s.tiles + size_t(env) * kTileStride);
        uint32_t* dst = reinterpret_cast<uint32_t*>(L.tiles);
        for (int k = q.g; k < kTileStride / 4; k += kGang) dst[k] = src[k];
        uint32_t* who = reinterpret_cast<uint32_t*>(L.who);
        for (int k = q.g; k < kTileStride / 4; k += kGang) who[k] = 0xffffffffu;
        wave_order();
        for (int e0 = 0; e0 < n_ent; e0 += kGang) {
            const int e = e0 + q.g;
            const bool ok = e < n_ent;
            const int info = ok ? EB(s, EB_INFO, e, env) : 0;
            const int cell = ok ? ent_cell(s, e, env) : 0;
            const bool edible = ok && !(e >= kOrbs && e < kFirstPoint);
            if (ok) {
                L.info[e] = static_cast<uint8_t>(info);
                L.cell[e] = static_cast<uint16_t>(cell);
            }
            if (edible) L.who[cell] = static_cast<uint8_t>(e);
            left += __popc(q.ballot(edible && (info & kAlive)));
        }
    }
    wave_order();
    Rng rng = Rng::open(s.mt + size_t(env) * kMtWords, q, s.mt_other + size_t(env) * kMtN, s.mt_sel + env);
    float ax = SF(s, F_AX, env), ay = SF(s, F_AY, env), avx = SF(s, F_AVX, env), avy = SF(s, F_AVY, env);
    float nvx = SF(s, F_NVX, env), nvy = SF(s, F_NVY, env);
    float input_t = SF(s, F_INPUT_T, env), anim_t = SF(s, F_ANIM_T, env), eat_t = SF(s, F_EAT_T, env);
    int anim_i = SI(s, I_ANIM_I, env);
    bool set_changed = (SI(s, I_FLAGS, env) & kFlagListed) == 0;
    const float dt = 1.0f / 4;
    // The enemies: lane k of the gang (k < kMobs) holds the k-th of the enemy set's iteration order for the step.
    static_assert(kMobs <= kGang, "one enemy per lane of the gang");
    const bool has_mob = q.g < kMobs;
    const int mob_id = MB(s, 1, has_mob ? q.g : 0, env) - kOrbs;
    Mob mob{MF(s, MF_X, mob_id, env), MF(s, MF_Y, mob_id, env), MF(s, MF_VX, mob_id, env), MF(s, MF_VY, mob_id, env),
            MF(s, MF_HATCH, mob_id, env), MB(s, 0, mob_id, env)};

    float movement_x = static_cast<float>((action == 7) - (action == 1));
    float movement_y = static_cast<float>((action == 3) - (action == 5));
    if (movement_x != 0.0f && movement_y != 0.0f) movement_y = 0.0f;

    float reward = 0.0f;
    bool terminated = false;
    for (int ss = 0; ss < 4; ss++) {
#ifndef PG_CHASER_SKIP  // (instruction inventory of the logic kernel, tools/probe/chaser_logic_phases.sh: experiment builds leave parts out)
#define PG_CHASER_SKIP 0
#endif
        if (!(PG_CHASER_SKIP & 1)) {  // --- System_Agent::update (common_systems.cpp:305-444)
            const float speed = 0.2f;
            const float input_reset_time = 1.0f / speed * 0.5f;
            if (movement_x != 0.0f || movement_y != 0.0f) {
                nvx = movement_x;
                nvy = movement_y;
                input_t = 0.0f;
            }
            if (nvx > 0.0f) {
                if (qabs(s, ay - (static_cast<int>(ay) + 0.5f)) <= speed * dt &&
                    tile_at(L, static_cast<int>(ax) + 1, H - 1 - static_cast<int>(ay)) == kEmpty) {
                    ay = static_cast<int>(ay) + 0.5f;
                    avx = nvx;
                    avy = nvy;
                }
            } else if (nvx < 0.0f) {
                if (qabs(s, ay - (static_cast<int>(ay) + 0.5f)) <= speed * dt &&
                    tile_at(L, static_cast<int>(ax) - 1, H - 1 - static_cast<int>(ay)) == kEmpty) {
                    ay = static_cast<int>(ay) + 0.5f;
                    avx = nvx;
                    avy = nvy;
                }
            }
            if (nvy > 0.0f) {
                if (qabs(s, ax - (static_cast<int>(ax) + 0.5f)) <= speed * dt &&
                    tile_at(L, static_cast<int>(ax), H - 1 - (static_cast<int>(ay) + 1)) == kEmpty) {
                    ax = static_cast<int>(ax) + 0.5f;
                    avx = nvx;
                    avy = nvy;
                }
            } else if (nvy < 0.0f) {
                if (qabs(s, ax - (static_cast<int>(ax) + 0.5f)) <= speed * dt &&
                    tile_at(L, static_cast<int>(ax), H - 1 - (static_cast<int>(ay) - 1)) == kEmpty) {
                    ax = static_cast<int>(ax) + 0.5f;
                    avx = nvx;
                    avy = nvy;
                }
            }
            if (avx < 0.0f) {
                if (qabs(s, ax - (static_cast<int>(ax) + 0.5f)) <= speed * dt &&
                    tile_at(L, static_cast<int>(ax) - 1, H - 1 - static_cast<int>(ay)) != kEmpty) {
                    ax = static_cast<int>(ax) + 0.5f;
                    avx = 0.0f;
                }
            } else if (avx > 0.0f) {
                if (qabs(s, ax - (static_cast<int>(ax) + 0.5f)) <= speed * dt &&
                    tile_at(L, static_cast<int>(ax) + 1, H - 1 - static_cast<int>(ay)) != kEmpty) {
                    ax = static_cast<int>(ax) + 0.5f;
                    avx = 0.0f;
                }
            }
            if (avy < 0.0f) {
                if (qabs(s, ay - (static_cast<int>(ay) + 0.5f)) <= speed * dt &&
                    tile_at(L, static_cast<int>(ax), H - 1 - (static_cast<int>(ay) - 1)) != kEmpty) {
                    ay = static_cast<int>(ay) + 0.5f;
                    avy = 0.0f;
                }
            } else if (avy > 0.0f) {
                if (qabs(s, ay - (static_cast<int>(ay) + 0.5f)) <= speed * dt &&
                    tile_at(L, static_cast<int>(ax), H - 1 - (static_cast<int>(ay) + 1)) != kEmpty) {
                    ay = static_cast<int>(ay) + 0.5f;
                    avy = 0.0f;
                }
            }
            ax += avx * speed * dt;
            ay += avy * speed * dt;
            if (input_t >= input_reset_time) {
                nvx = 0.0f;
                nvy = 0.0f;
            } else {
                input_t += dt;
            }
        }
        const Box agent_rect{-0.5f + ax, -0.5f + ay, 1.0f, 1.0f};

        // --- System_Mob_AI::update (common_systems.cpp:117-295), enemies in the set's iteration order
        // The enemies do not see each other: all that joins them is the env's random stream, which they draw from one
        // after the other — an enemy at a junction one number (aggressive or not), a second if it was not (which way), a
        // third when the agent, having eaten an orb, catches it (where its egg goes).  So they take their turns side by
        // side, one per lane (round 6; until then one after the other on every lane of the gang: 118 of the kernel's
        // 157 µs), each on outputs peeked at the place in the stream where it is THOUGHT to start — the numbers taken by
        // the enemies before it, as far as known — and the turns are taken again until every start is right: lane 0's is
        // from the outset, lane k's after at most k rounds.  With the int reading of `abs` (D21, the default) every
        // enemy counts as at a junction in every sub-step, so every one draws and it is kMobs rounds each time — which is why
        // a round is only the part of the turn that draws (mob_tail: which way, the move, the agent met), a third of the
        // turn's instructions; the part that does not (mob_head: the junction test, the open ways, the aggressive choice) is
        // done once in front.  With the float reading a junction comes every sixteenth sub-step and one round settles it.  The long way — one enemy after the other on the stream
        // itself, every lane running the same one (SerialDraws) — is taken where the peeked outputs do not reach:
        // the stream's 624 words run out inside the window, or a range draw is rejected (probability ≈ range / 2^32);
        // `serial_mobs` (pgv_set_debug bit 25) forces it for the tests.
        bool player_hit = false;
        {
            const float hatch_time = 50.0f, anim_time = 1.0f;
            const bool hatched = has_mob && mob.hatch >= hatch_time;
            bool long_way = serial_mobs || rng.idx + kMobs * kPeek > kMtN;  // (gang-uniform)
            MobHead head{false, 0, 0, 0, 0.0f, 0};
            if (hatched && !(PG_CHASER_SKIP & 2) && !long_way) head = mob_head(s, L, mob, ax, ay, eat_t, anim_i, dt);
            if (!(PG_CHASER_SKIP & 2) && !long_way) {
                Mob after = mob;
                bool hit = false, over = false;
                int start = 0, taken = 0;
                for (int round = 0; round < kMobs; round++) {
                    after = mob;
                    PeekDraws draw{{0u, 0u, 0u}, 0};
                    if (hatched) {
#pragma unroll
                        for (int j = 0; j < kPeek; j++) draw.w[j] = mt_temper(rng.x[rng.idx + start + j]);
                        hit = mob_tail(L, after, head, draw, agent_rect, eat_t, n_ent, dt);
                    }
                    over = draw.over();
                    taken = over ? kPeek : draw.taken;
                    // where every lane's enemy starts, given what the ones before it took this round (0 to 3 each)
                    const uint32_t below = (1u << q.g) - 1u;
                    const int now = __popc(q.ballot((taken & 1) != 0) & below) + 2 * __popc(q.ballot((taken & 2) != 0) & below);
                    // (an enemy that drew nothing this round never looked at its outputs: wherever it starts, its turn stands)
                    const bool moved = taken != 0 && now != start;
                    start = now;
                    if (!q.any(moved)) break;
                }
                long_way = q.any(over);  // (a rejected range draw: never in practice)
                if (!long_way) {
                    player_hit = q.any(hit);
                    if (hatched) mob = after;
                    const int total = __popc(q.ballot((taken & 1) != 0)) + 2 * __popc(q.ballot((taken & 2) != 0));
                    rng.idx += total;  // (next() refills its window by itself)
                }
            }
            if (!(PG_CHASER_SKIP & 2) && long_way) {
                for (int k = 0; k < kMobs; k++) {  // enemy k on every lane of the gang, committed by lane k
                    const int src = q.shift + k;
                    if (__shfl(mob.hatch, src) >= hatch_time) {
                        Mob one{__shfl(mob.px, src), __shfl(mob.py, src), __shfl(mob.vx, src), __shfl(mob.vy, src),
                                __shfl(mob.hatch, src), __shfl(mob.tex, src)};
                        SerialDraws draw{rng};
                        const MobHead h = mob_head(s, L, one, ax, ay, eat_t, anim_i, dt);
                        player_hit = mob_tail(L, one, h, draw, agent_rect, eat_t, n_ent, dt) || player_hit;
                        if (q.g == k) mob = one;
                    }
                }
            }
            if (has_mob && !hatched && !(PG_CHASER_SKIP & 2)) mob.hatch += dt;
            if (anim_t < anim_time) {
                anim_t += dt;
            } else {
                anim_t -= anim_time;
                anim_i = (anim_i + 1) % 6;
            }
            if (eat_t > 0.0f) eat_t = fmaxf(0.0f, eat_t - dt);
        }

        // --- System_Point::update (common_systems.cpp:66-106): order-free.  The reference tests every orb and point; the
        // agent's 1×1 box can only reach those on the nine cells around its own (their boxes are at most 1×1 about a cell
        // centre), so the lanes take one of those cells each and test — with the same box arithmetic — whoever sits there.
        int delta = 0;
        bool orb_eaten = false;
        {
            const int cx = static_cast<int>(ax), cy = H - 1 - static_cast<int>(ay);  // cell (x, tile row) under the agent's centre
#pragma unroll
            for (int n0 = 0; n0 < ((PG_CHASER_SKIP & 4) ? 0 : 9); n0 += kGang) {
                const int nb = n0 + q.g;
                const int x = cx + nb % 3 - 1, ty = cy + (nb / 3) % 3 - 1;
                const bool inside = nb < 9 && x >= 0 && ty >= 0 && x < W && ty < H;
                const int cell = inside ? ty + x * H : 0;
                const int e = inside ? L.who[cell] : kNobody;
                const int info = e != kNobody ? L.info[e] : 0;
                const bool alive = (info & kAlive) != 0;
                const float px = cell_x(cell), py = cell_y(cell);
                const bool orb = e < kOrbs;
                const Box rect = orb ? Box{-0.5f + px, -0.5f + py, 1.0f, 1.0f} : Box{-0.3f + px, -0.3f + py, 0.6f, 0.6f};
                const bool eaten = alive && box_hit(agent_rect, rect);
                if (eaten) {  // destroy_entity
                    L.info[e] = static_cast<uint8_t>(info & ~kAlive);
                    EB(s, EB_INFO, e, env) = static_cast<uint8_t>(info & ~kAlive);
                }
                delta += __popc(q.ballot(eaten));
                orb_eaten = orb_eaten | q.any(eaten & orb);
            }
        }
        left -= delta;
        const int available = left;
        if (orb_eaten) eat_t = 75.0f;
        if (delta) set_changed = true;
        reward = delta * 0.04f + (available == 0) * 10.0f;
        terminated = player_hit || (available == 0);
        if (terminated) break;
    }
    rng.close();
    if (has_mob) {
        MF(s, MF_X, mob_id, env) = mob.px;
        MF(s, MF_Y, mob_id, env) = mob.py;
        MF(s, MF_VX, mob_id, env) = mob.vx;
        MF(s, MF_VY, mob_id, env) = mob.vy;
        MF(s, MF_HATCH, mob_id, env) = mob.hatch;
        MB(s, 0, mob_id, env) = static_cast<uint8_t>(mob.tex);
    }
    if (q.g == 0) {
        SF(s, F_AX, env) = ax;
        SF(s, F_AY, env) = ay;
        SF(s, F_AVX, env) = avx;
        SF(s, F_AVY, env) = avy;
        SF(s, F_NVX, env) = nvx;
        SF(s, F_NVY, env) = nvy;
        SF(s, F_INPUT_T, env) = input_t;
        SF(s, F_ANIM_T, env) = anim_t;
        SF(s, F_EAT_T, env) = eat_t;
        SI(s, I_ANIM_I, env) = anim_i;
        SI(s, I_FLAGS, env) = kFlagListed;
    }
    if (set_changed && !(PG_CHASER_SKIP & 8)) {
        wave_order();
        rebuild_draw_list(s, L, q, env, n_ent);
    }
    reward_out = reward;
    terminated_out = terminated;
}

__global__ void __launch_bounds__(64) make_kernel(State s) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= s.n) return;
    fresh_chain(s, env);
    fresh_live(s, env);
}

__global__ void __launch_bounds__(64, PG_CHASER_WAVES) logic_kernel(State s, const int32_t* actions, uint32_t run_seed,
                                                                    uint32_t step_index, int env_offset, StepIO io, int serial_mobs) {
    __shared__ StepLds lds[64 / kGang];
    const int env = (blockIdx.x * 64 + threadIdx.x) / kGang;
    if (env >= s.n) return;
    if (io.pending[env] != 0) return;  // reset by the level kernel in this step, maybe right now (pg_engine.h StepIO)
    const Q q = Q::at(threadIdx.x);
    const int action =
        actions ? actions[env] : synthetic_action(run_seed, step_index, static_cast<uint32_t>(env_offset + env));
    float reward = 0.0f;
    bool terminated = false;
    advance(s, lds[(threadIdx.x & 63) / kGang], q, env, action, serial_mobs != 0, reward, terminated);
    if (q.g == 0) {
        io.reward[env] = reward;
        io.done[env] = terminated ? 1 : 0;
        io.pending[env] = terminated ? 3 : 0;  // (3, not 1: the level kernel may be running beside this one — pg_engine.h StepIO)
    }
}

// render_game(true) (chaser.cpp:390-416): one workgroup of two wavefronts per env (pg_render.h).
// pass 0: every env (of the mask); the step's level and logic kernels are both done, the flag is settled here
// (pg_engine.h StepIO: 3 → 1, 2 → 0).  A step whose resets run on their own stream (pg_engine.h launch_render_step /
// _late) renders in two passes: 1 = beside the level kernel, every env that is NOT being reset (pending 1 or 2: the level
// kernel may be anywhere) — and it leaves the flags alone: a 1 written now could still be picked up by a late wavefront of
// the level kernel running beside it, which would reset the env a step early; 2 = after the level kernel, the envs it
// reset (all 2 by then), and the flags are settled.
// The wall tile's corners are translucent (16 of its 256 texels, enough for the atlas loader's "hard" mark), but a frame
// samples at most two or three of them per wall, all in one pixel row of the tile: the composer's one-texel attempt
// settles the other rows, so it is made regardless.
// Wavefronts per SIMD the render kernels' registers are capped for (five = 96 registers; they take 80).  Only the default
// mode's env is small enough for the LDS to hold nine of them per CU; the larger worlds hold eight either way.
#ifndef PG_CHASER_RENDER_WAVES
#if PG_VARIANT == 0
#define PG_CHASER_RENDER_WAVES 5
#else
#define PG_CHASER_RENDER_WAVES 4
#endif
#endif
#ifndef PG_CHASER_HARD_WALLS
#define PG_CHASER_HARD_WALLS 0
#endif
// Composer grid: W tiles + the border cells of the inclusive window, not a cell more — LDS decides how many envs a CU
// holds, and with 13 × 13 cell tables (16 × 16: 700 bytes more) and the draw list in registers the default mode's env
// takes 17.4 KB instead of 19.1: nine per CU instead of eight.
constexpr int kGrid = W + 2;
constexpr int kEntRegs = (kMaxEnt + 63) / 64;  // the draw list, entry lane + 64·j in register j of every wavefront

// What the sprite pass needs of the entity tables, fetched when the kernel starts — the draw list's entities and the
// enemies' places are two and three dependent global loads deep, and a wavefront that goes for them only when it gets
// to its sprites waits that long with nothing else to do (the composer's work is behind it by then).
struct SpriteLds {
    float mob[kMobs + 1][4];  // per enemy: x, y, texture
};

// What every frame shows: the whole world (chaser.cpp:401), and the wall window of it (tilemap.cpp:245-254).
struct View {
    Camera cam;
    int x0, y0, cols, rows;
};
PG_HD View view_of_world() {
    const float zoom = 64.0f * kPxUnit / static_cast<float>(W);  // chaser.cpp:401
    View v{Camera{W * 0.5f * kUnitPx, H * 0.5f * kUnitPx, 64.0f, 64.0f, zoom}, 0, 0, 0, 0};
    const Camera& cam = v.cam;
    const float vx = (cam.px - cam.sw * 0.5f / cam.scale) * kPxUnit;
    const float vy = (cam.py - cam.sh * 0.5f / cam.scale) * kPxUnit;
    const float vw = cam.sw * kPxUnit / cam.scale, vh = cam.sh * kPxUnit / cam.scale;
    v.x0 = static_cast<int>(floorf(vx));
    v.y0 = static_cast<int>(floorf(vy));
    v.cols = static_cast<int>(ceilf(vx + vw)) - v.x0 + 1;
    v.rows = static_cast<int>(ceilf(vy + vh)) - v.y0 + 1;
    return v;
}
// descriptor .w of what the tile layer shows, as the composer's row classes want it (a point is a candidate only
// inside its box: if the box is solid it brings nothing that is not opaque)
PG_D int layer_flags(const State& s, const int4& wall_d, const int4& point_d, bool points_in_layer) {
    return wall_d.w | (points_in_layer && !s.point_solid ? point_d.w : 0);
}
// Needs: the wall's texture size (check_atlas), texels all opaque or all clear, a clear rim (descriptor .w bits 2, 3).
PG_D bool points_join_layer(const int4& point_d, int flags) { return !(flags & 1) && (point_d.w & 12) == 8; }

// One env's frame by its workgroup (two wavefronts, pg_render.h); fb, LB and S are the workgroup's LDS.
// kPrepped: the draws and the background's axes come from the pre-pass's tables (State::Prep) instead of being resolved
// here — the frames of the step's main pass; the late pass (envs reset in this step: their level did not exist when the
// pre-pass ran) and the debug paths resolve their own.
// The points of the draw list over a frame that already holds the base layer: lane = point.  A point is a stamp — a handful
// of opaque pixels whose places and colours depend on its cell alone (State::stamps) — and no two points' stamps meet, nor a
// wall's padded seam (they lie in the middle of free cells): written in any order, each wave the pixels of its own rows.
// `ent`: the draw list as render_env holds it (entry lane + 64·j in register j: entity | kind << 8 | cell << 16).
PG_D void stamp_points(uint32_t* fb, const State& s, const AtlasView& atlas, const uint32_t (&ent)[kEntRegs], int n_draw, int lane, int half) {
    const uint4* table = reinterpret_cast<const uint4*>(atlas.texels + s.stamps);
    const uint32_t row_lo = static_cast<uint32_t>(half) * (kObsH / 2) * kObsW, row_hi = row_lo + (kObsH / 2) * kObsW;
#pragma unroll
    for (int j = 0; j < kEntRegs; j++) {
        const uint32_t v = ent[j];
        const bool point = lane + 64 * j < n_draw && ((v >> 8) & kKindMask) == kPoint;
        if (__ballot(point) == 0) continue;  // wave-uniform
        const uint4* mine = table + size_t(point ? (v >> 16) : 0u) * (kStampMax / 2);
        uint4 e[kStampMax / 2];
#pragma unroll
        for (int k = 0; k < kStampMax / 2; k++) e[k] = mine[k];  // (two stamps a load; the table is 15 KB for every env of the engine)
#pragma unroll
        for (int k = 0; k < kStampMax / 2; k++) {
            if (point && e[k].x >= row_lo && e[k].x < row_hi) fb[e[k].x] = e[k].y;
            if (point && e[k].z >= row_lo && e[k].z < row_hi) fb[e[k].z] = e[k].w;
        }
    }
}

// kBase (kPrepped frames only): the frame starts from the env's base layer (State::base) and stamps its points, instead of
// composing background and walls.  write_base (the other frames): background and walls are composed WITHOUT the points
// (the sprite pass draws those), and the layer is left in State::base on the way.
template <bool kPrepped>
PG_D void render_env(const State& s, const AtlasView& atlas, const StepIO& io, int flags, int env, uint32_t* fb,
                     ComposeLdsBoxed<kGrid>& LB, SpriteLds& S, bool from_base = false, bool write_base = false) {
    ComposeLds<kGrid>& L = LB.plain;
    const int lane = threadIdx.x & 63, half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // two wavefronts per env (pg_render.h)
    constexpr int halves = 2;

    PG_TL_BEGIN(7);
    PG_TL(0);
    const View view = view_of_world();
    const Camera& cam = view.cam;
    const int sflags = SI(s, I_FLAGS, env);
    const int n_draw = (sflags & kFlagListed) ? SI(s, I_NDRAW, env) : 0;  // empty right after a reset
    const uint8_t* tiles = s.tiles + size_t(env) * kTileStride;
    DescRegs descs{};
    if (!kPrepped) descs = DescRegs::load(atlas, lane);
    Blit mine;
    uint32_t ent[kEntRegs];  // per draw-list place lane + 64·j: entity | kind << 8 | cell << 16
#pragma unroll
    for (int j = 0; j < kEntRegs; j++) {
        const int k = lane + 64 * j;
        ent[j] = 0u;
        if (k < n_draw) {
            const int e = EB(s, EB_DRAW, k, env);
            ent[j] = static_cast<uint32_t>(e | (EB(s, EB_INFO, e, env) & kKindMask) << 8 | ent_cell(s, e, env) << 16);
        }
    }
    if (!kPrepped && half == 1 && lane < kMobs) {
        S.mob[lane][0] = MF(s, MF_X, lane, env);
        S.mob[lane][1] = MF(s, MF_Y, lane, env);
        S.mob[lane][2] = static_cast<float>(MB(s, 0, lane, env));
    }
    float agent_x = 0.0f, agent_y = 0.0f;
    if (!kPrepped) agent_x = SF(s, F_AX, env), agent_y = SF(s, F_AY, env);
    // (the barriers of the composer, or of the replay that stands in for it, come between these writes and their readers)

    int bg_soft = 0;  // the backdrop has texels that are not opaque (descriptor .w)
    int4 bg_d;  // the background draw, chaser.cpp:404-409: texture, world position, scale — each wave resolves the axis it needs (pg_render.h BgAxis)
    float bg_px, bg_py, bg_sc;
    {
        const int4 d = atlas.desc[kTexFloor + __builtin_amdgcn_readfirstlane(SI(s, I_BG, env))];  // (wave-uniform: scalar loads)
        bg_soft = d.w;
        const float aspect = static_cast<float>(d.y) / static_cast<float>(d.z);
        const float extra = aspect - 1.0f;
        bg_d = d;
        bg_px = -SF(s, F_BGSHIFT, env) * extra;
        bg_py = 0.0f;
        bg_sc = 64.0f * kUnitPx / d.z;
    }
    const int x0 = view.x0, y0 = view.y0, cols = view.cols, rows = view.rows, cells = cols * rows;
    const int4 wall_d = atlas.desc[kTexWall], point_d = atlas.desc[kTexPoint];
    const bool points_in_layer = points_join_layer(point_d, flags) && !write_base;

    bool composed = false;
    PG_TL(1);
    if (kPrepped && from_base) {
        // (Measured and rejected, round 5: the layer as 12 KB of RGB, four pixels = twelve bytes a lane and trip, unpacked into the
        // target's words — a quarter less to read: 0.496 against 0.485 ms.  The copy is not what this kernel waits for.)
        // the base layer: a straight copy of this wave's 32 rows (8 KB), memory to LDS without passing through registers —
        // `buffer_load_dwordx4 … lds`: lane l's 16 bytes land at M0 + 16·l, 1 KB an instruction
        {
            using lds_ptr = __attribute__((address_space(3))) void*;
            const __amdgpu_buffer_rsrc_t base_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint32_t*>(s.base + size_t(env) * kFbWords + half * (kFbWords / 2)), 0, kFbWords / 2 * 4, 0x00020000);
#pragma unroll
            for (int k = 0; k < kFbWords / 8 / 64; k++)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(base_rsrc, (lds_ptr)(fb + half * (kFbWords / 2) + k * 256), 16, lane * 16, k * 1024, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler does not track LDS-direct loads: the copy has landed)
        wave_order();  // the copy, lane by lane, before the stamps of other lanes on the same words
        PG_TL(2);
        // … and the points that are still there, stamped over it (each wave its own rows: no barrier)
        stamp_points(fb, s, atlas, ent, n_draw, lane, half);
        wave_order();
        composed = true;
    } else if (!(flags & 1) && cols <= kGrid && rows <= kGrid) {
        // this wave's axis of the background (wave 0: x, wave 1: y); the tile spans are the prepared ones
        BgAxis bga;
        if (kPrepped) {
            const uint32_t* w = s.prep.bg + size_t(env) * 8 + half * 4;
            bga = BgAxis{static_cast<int32_t>(w[0] << 16) >> 16, static_cast<int32_t>(w[0]) >> 16, static_cast<int32_t>(w[1] & 0xffffu),
                         static_cast<int32_t>(w[1] >> 16), static_cast<int32_t>(w[2]), static_cast<int32_t>(w[3])};
        } else {
            bga = bg_axis(cam, bg_d, bg_px, bg_py, bg_sc, half);
        }
        if (half == 0 && lane < 2) L.base[kGrid * kGrid + lane] = static_cast<int32_t>(kNoTexel);
        for (int cell = lane + 64 * half; cell < kGrid * kGrid; cell += 64 * halves) {
            const int r = cell / kGrid, c = cell % kGrid;
            const bool wall = c < cols && r < rows && tile_at(tiles, x0 + c, H - 1 - (y0 + r)) == kWall;
            L.base[cell] = wall ? wall_d.x * 4 : static_cast<int32_t>(kNoTexel);
            LB.boxed[cell] = static_cast<int32_t>(kNoTexel);
        }
        if (half == 1 && lane < 2) LB.boxed[kGrid * kGrid + lane] = static_cast<int32_t>(kNoTexel);
        if (points_in_layer) {
            // The points of the draw list join the tile layer: a point is drawn with exactly a tile's arithmetic
            // (world position = its cell's corner, scale 16 / texture width — common_systems.cpp:41-63 vs
            // tilemap.cpp:256-266), its texture is the wall's size, and the only texels of it that are not fully
            // transparent are the opaque 4×4 in the middle — nowhere near the one-pixel seam it shares with its
            // neighbours.  So where a point lands in the picture does not depend on when it is drawn relative to
            // walls and other points; relative to the sprites that are not points it does, and the sprite pass below
            // re-draws (idempotently: opaque texels) the points that follow such a sprite in the list and touch it.
            __syncthreads();  // the wall pass has written every cell
#pragma unroll
            for (int j = 0; j < kEntRegs; j++) {  // (wave j mod 2 takes register j's entries)
                const uint32_t v = ent[j];
                if ((j & 1) == half && lane + 64 * j < n_draw && ((v >> 8) & kKindMask) == kPoint) {
                    const int cell = static_cast<int>(v >> 16);
                    const int c = cell / H - x0, r = (H - 1 - cell % H) - y0;
                    if (c >= 0 && r >= 0 && c < cols && r < rows) LB.boxed[r * kGrid + c] = point_d.x * 4;
                }
            }
        }
        __syncthreads();
        PG_TL(2);
        composed = compose_rows<kGrid, false, true, true>(fb, L, atlas, bga, cols, rows, wall_d.y, lane, flags, half, halves,
                                                          s.point_box, s.prepared, bg_soft);
    }
    if (!composed) {  // draw-list replay (tilemap.cpp:256-266)
        wave_clear(fb, lane, half, halves);
        const bool has_bg = resolve_draw(cam, bg_d.y, bg_d.z, bg_d.x, bg_px, bg_py, bg_sc, 1.0f, false, false, mine);
        wave_replay(fb, atlas, mine, has_bg ? 1ull : 0ull, lane, half, halves);
        for (int base = 0; base < cells; base += 64) {
            const int cell = base + lane;
            bool has = false;
            if (cell < cells) {
                const int row = cell / cols;
                const int x = x0 + (cell - row * cols), y = y0 + row;
                if (tile_at(tiles, x, H - 1 - y) == kWall)
                    has = resolve_draw(cam, wall_d.y, wall_d.z, wall_d.x, x * kUnitPx, y * kUnitPx, kUnitPx / wall_d.y,
                                       1.0f, false, false, mine);
            }
            wave_replay(fb, atlas, mine, __ballot(has), lane, half, halves);
        }
    }
    if (write_base) {  // background + walls are in the target, nothing else yet: the env's base layer (State::base)
        wave_order();
        const uint4* src = reinterpret_cast<const uint4*>(fb) + half * (kFbWords / 8);
        uint4* dst = reinterpret_cast<uint4*>(s.base + size_t(env) * kFbWords) + half * (kFbWords / 8);
#pragma unroll
        for (int k = 0; k < kFbWords / 8 / 64; k++) dst[k * 64 + lane] = src[k * 64 + lane];
    }
    // every sprite has z = 0: the positive pass (common_systems.cpp:41-63), then the agent (:446-460)
    // Points that the composer has already put into the picture are drawn again only if an earlier draw of the list
    // that is not a point (orb, egg, enemy) touches them: those would otherwise end up on top of the point instead of
    // under it.  The rectangles of the non-point draws seen so far live in the dead cell table, one region per wave
    // (the waves run this pass independently, each on the rows it owns).
    int4* const seen = reinterpret_cast<int4*>(L.base) + half * (kOrbs + kMobs + 2);
    static_assert(2 * (kOrbs + kMobs + 2) * 4 <= kGrid * kGrid, "scratch inside the cell table");
    int n_seen = 0;
    PG_TL(3);
    PG_MARK("s_sprites");
    const bool skip_points = composed && points_in_layer;
    for (int first = 0; first < n_draw + 1 && !PG_ABL(flags, 0x10000); first += 64) {  // (bit 16: timing experiment — no sprite pass)
        const int k = first + lane;
        int want_tex = kTexAgent;
        float x = 0.0f, y = 0.0f;
        bool has = false, is_point = false;
        if (kPrepped) {  // the lane's draw as the pre-pass left it: by cell for what stays put, by env for what moves
            const uint32_t* at = nullptr;
            if (k < n_draw) {
                uint32_t v = ent[0];
#pragma unroll
                for (int j = 1; j < kEntRegs; j++) v = first == 64 * j ? ent[j] : v;
                const int e = static_cast<int>(v & 0xffu), kind = static_cast<int>((v >> 8) & kKindMask);
                is_point = kind == kPoint;
                at = kind == kEgg ? s.prep.movers + (size_t(env) * kMovers + (e - kOrbs)) * kBlitWords
                                  : s.prep.cell_blits + (size_t(is_point ? kCells : 0) + (v >> 16)) * kBlitWords;
            } else if (k == n_draw) {
                at = s.prep.movers + (size_t(env) * kMovers + kMobs) * kBlitWords;
            }
            mine = prep_draw_load(at, at != nullptr);
            has = mine.dw > 0;  // (a draw that render_texture culls is stored as zeros)
        } else if (k < n_draw) {
            uint32_t v = ent[0];
#pragma unroll
            for (int j = 1; j < kEntRegs; j++) v = first == 64 * j ? ent[j] : v;
            const int e = static_cast<int>(v & 0xffu), kind = static_cast<int>((v >> 8) & kKindMask);
            has = true;
            if (kind == kEgg) {
                const int m = e - kOrbs;
                want_tex = kTexEnemy + static_cast<int>(S.mob[m][2]);
                x = S.mob[m][0];
                y = S.mob[m][1];
            } else {
                const int cell = static_cast<int>(v >> 16);
                want_tex = kind == kOrb ? kTexOrb : kTexPoint;
                is_point = kind == kPoint;
                x = cell_x(cell);
                y = cell_y(cell);
            }
        } else if (k == n_draw) {
            has = true;
            x = agent_x;
            y = agent_y;
        }
        if (!kPrepped) {
            const int4 d = descs.at(want_tex);
            if (has) {
                const float scale = (k == n_draw) ? kUnitPx / d.y * 1.0f : (1.0f * 1.0f) * kUnitPx / d.y;
                has = resolve_draw(cam, d.y, d.z, d.x, (x + -0.5f) * kUnitPx, (y + -0.5f) * kUnitPx, scale, 1.0f, false, false,
                                   mine);
            }
        }
        PG_MARK("t_resolved");
        if (skip_points) {
            const unsigned long long others = __ballot(has && !is_point);
            if (has && !is_point) {  // append in draw order
                const int slot = n_seen + __popcll(others & ((1ull << lane) - 1ull));
                seen[slot] = make_int4(mine.dx, mine.dy, mine.dx + mine.dw, mine.dy + mine.dh);
            }
            wave_order();  // (the rectangles of earlier passes are read by other lanes than wrote them)
            // a point is kept if a non-point draw EARLIER in the list touches it: those of earlier passes (all of
            // them precede it), and those of this pass at lower lanes
            bool keep = !is_point;
            if (is_point && has) {
                for (int q = 0; q < n_seen; q++) {
                    const int4 r = seen[q];
                    keep = keep || (mine.dx < r.z && mine.dx + mine.dw > r.x && mine.dy < r.w && mine.dy + mine.dh > r.y);
                }
            }
            unsigned long long m = others;
            while (m) {  // wave-uniform loop over this pass's non-point draws
                const int q = __builtin_ctzll(m);
                m &= m - 1;
                const int rx0 = __builtin_amdgcn_readlane(mine.dx, q), ry0 = __builtin_amdgcn_readlane(mine.dy, q);
                const int rx1 = rx0 + __builtin_amdgcn_readlane(mine.dw, q), ry1 = ry0 + __builtin_amdgcn_readlane(mine.dh, q);
                if (is_point && lane > q)
                    keep = keep || (mine.dx < rx1 && mine.dx + mine.dw > rx0 && mine.dy < ry1 && mine.dy + mine.dh > ry0);
            }
            n_seen += __popcll(others);
            has = has && keep;
        }
        PG_MARK("u_kept");
        if (PG_ABL(flags, 0x20000)) has = false;  // (bit 17: timing experiment — the pass without its draws)
        if (first == 0) PG_TL(4);
        // (Measured and rejected, round 5, on the kernel that starts from the base layer — 50 of its 96 registers in use:
        // groups of six or eight draws a round trip, 0.470 -> 0.870 ms; the draw list as one packed word per place, written
        // by rebuild_draw_list, instead of three loads two levels deep: 0.4702 -> 0.4699 ms.)
        wave_replay_rows<4, false, false>(fb, atlas, mine, __ballot(has), lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
    }
    PG_TL(5);
    PG_MARK("v_drawn");
    // each wave stores the rows it owns (pg_render.h wave_replay_rows): no barrier
    wave_store_rows(fb, io.obs + size_t(env) * kObsBytes, lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
    PG_TL_END(7, composed, io.obs + size_t(env) * kObsBytes + half * (kObsBytes / 2));
}

// Once per engine: the composer's tables for the one view there is (State::prepared).
__global__ void __launch_bounds__(128) prepare_kernel(State s, AtlasView atlas) {
    __shared__ alignas(16) uint32_t fb[kFbWords];
    __shared__ ComposeLdsBoxed<kGrid> LB;
    const int lane = threadIdx.x & 63, half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const View view = view_of_world();
    const int4 wall_d = atlas.desc[kTexWall], point_d = atlas.desc[kTexPoint];
    const int layer_w = layer_flags(s, wall_d, point_d, points_join_layer(point_d, 0));
    compose_spans<kGrid, 16>(fb, LB.plain, view.cam, view.x0, view.y0, view.cols, view.rows, wall_d.y, wall_d.z, kUnitPx / wall_d.y,
                             lane, 0, half, 2, soft_rows_of(0, layer_w), hard_rows_of(0, PG_CHASER_HARD_WALLS ? layer_w : 0));
    __syncthreads();
    compose_hand_build<kGrid, false, true>(fb, LB.plain, BgAxis{}, wall_d.y, lane, 0, half, s.point_box);
    compose_prepare<kGrid>(fb, LB.plain, s.prepared, lane, half);
    // an orb's and a point's draw at every cell (common_systems.cpp:41-63 as render_env's sprite pass states it)
    for (int q = threadIdx.x; q < 2 * kCells; q += blockDim.x) {
        const int is_point = q / kCells, cell = q - is_point * kCells;
        const int4 d = atlas.desc[is_point ? kTexPoint : kTexOrb];
        const float scale = (1.0f * 1.0f) * kUnitPx / d.y;
        Blit b;
        const bool has = resolve_draw(view.cam, d.y, d.z, d.x, (cell_x(cell) + -0.5f) * kUnitPx, (cell_y(cell) + -0.5f) * kUnitPx, scale,
                                      1.0f, false, false, b);
        BlitWords w = blit_pack(b);
        if (!has) w.w[0] = w.w[1] = w.w[2] = w.w[3] = w.w[4] = w.w[5] = 0u;
        for (int k = 0; k < kBlitWords; k++) s.prep.cell_blits[size_t(q) * kBlitWords + k] = w.w[k];
    }
}

// The render pre-pass: eight lanes an env — its enemies (at most five) and the agent resolved (their positions are
// floats: nothing to tabulate), and the two axes of the background's draw (chaser.cpp:404-409; pg_render.h bg_axis).
constexpr int kPrepLanes = 8;
static_assert(kMovers + 2 <= kPrepLanes, "enemies, agent, two background axes");
__global__ void __launch_bounds__(256) setup_kernel(State s, AtlasView atlas) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int env = gid / kPrepLanes, item = gid - env * kPrepLanes;
    if (env >= s.n) return;
    const View view = view_of_world();
    if (item < kMovers) {
        const bool agent = item == kMobs;
        const int tex = agent ? kTexAgent : kTexEnemy + MB(s, 0, item, env);
        const float x = agent ? SF(s, F_AX, env) : MF(s, MF_X, item, env), y = agent ? SF(s, F_AY, env) : MF(s, MF_Y, item, env);
        const int4 d = atlas.desc[tex];
        const float scale = agent ? kUnitPx / d.y * 1.0f : (1.0f * 1.0f) * kUnitPx / d.y;
        Blit b;
        const bool has = resolve_draw(view.cam, d.y, d.z, d.x, (x + -0.5f) * kUnitPx, (y + -0.5f) * kUnitPx, scale, 1.0f, false, false, b);
        BlitWords w = blit_pack(b);
        if (!has) w.w[0] = w.w[1] = w.w[2] = w.w[3] = w.w[4] = w.w[5] = 0u;
        uint2* at = reinterpret_cast<uint2*>(s.prep.movers + (size_t(env) * kMovers + item) * kBlitWords);
        at[0] = make_uint2(w.w[0], w.w[1]);
        at[1] = make_uint2(w.w[2], w.w[3]);
        at[2] = make_uint2(w.w[4], w.w[5]);
    } else if (item < kMovers + 2) {
        const int axis = item - kMovers;
        const int4 d = atlas.desc[kTexFloor + SI(s, I_BG, env)];
        const float aspect = static_cast<float>(d.y) / static_cast<float>(d.z);
        const float extra = aspect - 1.0f;
        const BgAxis a = bg_axis(view.cam, d, -SF(s, F_BGSHIFT, env) * extra, 0.0f, 64.0f * kUnitPx / d.z, axis);
        uint4* at = reinterpret_cast<uint4*>(s.prep.bg + size_t(env) * 8 + axis * 4);
        *at = make_uint4(pack_halves(a.d0, a.dn), pack_halves(a.s0, a.sn), static_cast<uint32_t>(a.tex_off), static_cast<uint32_t>(a.tex_w));
    }
}

template <bool kPrepped>
__global__ void __launch_bounds__(128, PG_CHASER_RENDER_WAVES) render_kernel(State s, AtlasView atlas, const uint8_t* mask, StepIO io,
                                                    int flags, int pass) {
    const int env = blockIdx.x;
    if (mask && !mask[env]) return;
    {
        const int p = io.pending[env];
        __syncthreads();  // every thread has read the flag before thread 0 settles it
        if (pass == 1 && (p == 1 || p == 2)) return;
        if (pass != 1 && threadIdx.x == 0 && p >= 2) io.pending[env] = p == 3 ? 1 : 0;
    }
    __shared__ alignas(16) uint32_t fb[kFbWords];
    __shared__ ComposeLdsBoxed<kGrid> L;
    __shared__ SpriteLds S;
    // kPrepped: from the env's base layer (the host launches this form only when every env that is drawn here has one:
    // ChaserGame::from_base).  Otherwise the complete frame, which leaves the base layer behind.
    if (kPrepped)
        render_env<true>(s, atlas, io, flags, env, fb, L, S, true, false);
    else
        render_env<false>(s, atlas, io, flags, env, fb, L, S, false, true);
}

// The late pass of a step whose resets ran on their own stream: the frames of the envs on the level kernel's list (a
// few hundred of 65 536: a workgroup takes every gridDim.x-th of them), and the flags of everyone settled (3 → 1 for the
// envs that ended an episode in this step, 2 → 0 for the ones just reset).  A full-size launch that exits at once for
// all but the listed envs costs 30 µs in dispatch alone.
// Every stream whose current words are in the second buffer back into mt[env] (ChaserGame::prepare_save); a wavefront per
// 64 envs.  The block made ahead, if any, is dropped: the late pass makes it again.
__global__ void __launch_bounds__(64) streams_home_kernel(State s) {
    const int lane = threadIdx.x, env0 = blockIdx.x * 64, e = env0 + lane;
    unsigned long long todo = __ballot(e < s.n && (s.mt_sel[e] & 2));
    while (todo) {  // (wave-uniform)
        const int env = env0 + __builtin_ctzll(todo);
        todo &= todo - 1;
        uint32_t* home = s.mt + size_t(env) * kMtWords;
        const uint32_t* other = s.mt_other + size_t(env) * kMtN;
        for (int k = lane; k < kMtN; k += 64) home[k] = other[k];
        if (lane == 0) s.mt_sel[env] = 0;
    }
}

// Workgroups from `groups` on: the random streams' next blocks (State::mt_other, mt_sel) for the envs that have none — the two or
// three in a hundred whose gang took its block in this step, and those given a level — a wavefront per 64 envs.  Nothing
// else of this engine runs beside the late pass, so the streams stand still; an env whose byte is not 0 (it resets in the
// next step, or is being settled right now) is left for a later step.
__global__ void __launch_bounds__(128, PG_CHASER_RENDER_WAVES) render_list_kernel(State s, AtlasView atlas, StepIO io, int flags, int groups) {
    if (static_cast<int>(blockIdx.x) >= groups) {  // (workgroup-uniform)
        const int lane = threadIdx.x & 63;
        const int env0 = ((static_cast<int>(blockIdx.x) - groups) * 2 + static_cast<int>(threadIdx.x >> 6)) * 64;
        const int e = env0 + lane;
        const int sel = e < s.n ? s.mt_sel[e] : 1;
        unsigned long long todo = __ballot(e < s.n && !(sel & 1) && io.pending[e] == 0);
        while (todo) {  // (wave-uniform)
            const int k = __builtin_ctzll(todo), env = env0 + k;
            todo &= todo - 1;
            uint32_t* home = s.mt + size_t(env) * kMtWords;
            uint32_t* other = s.mt_other + size_t(env) * kMtN;
            const int sel_env = __shfl(sel, k);  // (every lane: a cross-lane read inside `if (lane == 0)` would read an idle lane)
            const bool in_other = (sel_env & 2) != 0;
            mt_next_block_wave(in_other ? other : home, in_other ? home : other, lane);
            if (lane == 0) s.mt_sel[env] = static_cast<uint8_t>(sel_env | 1);
        }
        return;
    }
    // (the bytes four at a time: a word that holds no 3 — nearly every word — is left alone)
    const int next = 1 - s.parity;
    for (int e4 = blockIdx.x * 128 + threadIdx.x; e4 * 4 < s.n; e4 += groups * 128) {
        if (e4 * 4 + 4 <= s.n) {
            const uint32_t four = reinterpret_cast<const uint32_t*>(io.pending)[e4];
            if (!(((four ^ 0x03030303u) - 0x01010101u) & ~(four ^ 0x03030303u) & 0x80808080u)) continue;  // no byte equals 3
        }
        for (int e = e4 * 4; e < e4 * 4 + 4 && e < s.n; e++)
            if (io.pending[e] == 3) {
                io.pending[e] = 1;
                s.due_list[size_t(next) * s.n + atomicAdd(&s.due_count[next], 1)] = e;  // the next step's level kernel resets it
            }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        s.reset_count[1 - s.parity] = 0;
        s.due_count[s.parity] = 0;  // (this step's level kernel has walked it: the late pass runs behind the join)
    }
    __shared__ alignas(16) uint32_t fb[kFbWords];
    __shared__ ComposeLdsBoxed<kGrid> L;
    __shared__ SpriteLds S;
    const int count = s.reset_count[s.parity];
    for (int item = blockIdx.x; item < count; item += groups) {
        const int env = s.reset_list[item];
        if (threadIdx.x == 0) io.pending[env] = 0;
        render_env<false>(s, atlas, io, flags, env, fb, L, S, false, true);  // (… and leaves the new level's base layer)
        __syncthreads();  // the next env of this workgroup reuses the LDS
    }
}

// cenv_render's frame (render_game(false)) for one env: pg_frame.h; the draw list of render_kernel, one draw at a time.
__global__ void __launch_bounds__(kFrameThreads) frame_kernel(State s, AtlasView atlas, int env, FrameTarget t) {
    const float fw = static_cast<float>(t.w), fh = static_cast<float>(t.h);
    const float zoom = fw * kPxUnit / static_cast<float>(W);  // chaser.cpp:401
    FramePainter P{t, atlas, Camera{W * 0.5f * kUnitPx, H * 0.5f * kUnitPx, fw, fh, zoom}, static_cast<int>(threadIdx.x),
                   kFrameThreads};
    const int sflags = SI(s, I_FLAGS, env);
    const int n_draw = (sflags & kFlagListed) ? SI(s, I_NDRAW, env) : 0;
    const uint8_t* tiles = s.tiles + size_t(env) * kTileStride;
    P.clear();
    {
        const int4 d = P.desc(kTexFloor + SI(s, I_BG, env));
        const float aspect = static_cast<float>(d.y) / static_cast<float>(d.z);
        const float extra = aspect - 1.0f;
        P.draw(kTexFloor + SI(s, I_BG, env), -SF(s, F_BGSHIFT, env) * extra, 0.0f, 64.0f * kUnitPx / d.z);
    }
    int x0, y0, x1, y1;
    P.window(x0, y0, x1, y1);
    for (int y = y0; y <= y1; y++)
        for (int x = x0; x <= x1; x++)
            if (tile_at(tiles, x, H - 1 - y) == kWall) P.draw(kTexWall, x * kUnitPx, y * kUnitPx, kUnitPx / P.desc(kTexWall).y);
    for (int k = 0; k < n_draw; k++) {
        const int e = EB(s, EB_DRAW, k, env);
        const int kind = EB(s, EB_INFO, e, env) & kKindMask;
        int tex;
        float x, y;
        if (kind == kEgg) {
            const int m = e - kOrbs;
            tex = kTexEnemy + MB(s, 0, m, env);
            x = MF(s, MF_X, m, env);
            y = MF(s, MF_Y, m, env);
        } else {
            const int cell = ent_cell(s, e, env);
            tex = kind == kOrb ? kTexOrb : kTexPoint;
            x = cell_x(cell);
            y = cell_y(cell);
        }
        const float scale = 1.0f * 1.0f;
        P.draw(tex, (x + -0.5f) * kUnitPx, (y + -0.5f) * kUnitPx, scale * kUnitPx / P.desc(tex).y);
    }
    P.draw(kTexAgent, (SF(s, F_AX, env) + -0.5f) * kUnitPx, (SF(s, F_AY, env) + -0.5f) * kUnitPx,
           kUnitPx / P.desc(kTexAgent).y * 1.0f);
}

class ChaserGame final : public Game {
   public:
    const char* name() const override { return "chaser"; }
    std::vector<std::string> texture_names() const override {
        std::vector<std::string> v = {"misc_assets/tileStone_slope.png", "misc_assets/yellowCrystal.png",
                                      "custom/chaser_point.png",         "misc_assets/enemySpikey_1b.png",
                                      "misc_assets/enemyFlying_1.png",   "misc_assets/enemyFlying_2.png",
                                      "misc_assets/enemyFlying_3.png",   "misc_assets/enemyWalking_1b.png",
                                      "misc_assets/enemyFloating_1b.png", "topdown_backgrounds/floortiles.png"};
        for (int k = 1; k <= 8; k++) v.push_back("topdown_backgrounds/backgrounddetailed" + std::to_string(k) + ".png");
        return v;
    }
    std::string check_atlas(const std::vector<std::pair<int, int>>& sizes) const override {
        if (static_cast<int>(sizes.size()) != kTexCount) return "chaser: unexpected texture count";
        if (sizes[kTexPoint] != sizes[kTexWall]) return "chaser: the point texture must have the wall tile's size (row composer)";
        return "";
    }
    // The box of the point sprite's texels that are not fully transparent (atlas texels with alpha 0 are the word 0):
    // inside it the sprite is a candidate of the row composer's tile layer, outside it is not there at all.
    void extend_atlas(Atlas& atlas) override {
        const int4 d = atlas.desc_host(kTexPoint);
        const uint32_t* tex = atlas.texels_host(kTexPoint);
        int4 box = make_int4(d.y, -1, d.z, -1);
        for (int y = 0; y < d.z; y++)
            for (int x = 0; x < d.y; x++)
                if (tex[size_t(y) * d.y + x] != 0u) {
                    box.x = std::min(box.x, x);
                    box.y = std::max(box.y, x);
                    box.z = std::min(box.z, y);
                    box.w = std::max(box.w, y);
                }
        bool solid = true;
        for (int y = box.z; y <= box.w; y++)
            for (int x = box.x; x <= box.y; x++) solid = solid && (tex[size_t(y) * d.y + x] >> 24) == 255u;
        s_.point_box = box;
        s_.point_solid = solid ? 1 : 0;
        stamp_ok_ = solid && (d.w & 12) == 8;  // (points_join_layer: a clear rim around an opaque box)
        // A point's stamp at every cell (State::stamps): its draw call through render_texture's arithmetic and raster rules
        // S1–S3 (pg_geom.h resolve_draw / sample_index — the very functions the kernels run, compiled for the host with the
        // same -ffp-contract=off), the pixels whose texel is not clear.  More than kStampMax of them, or one that is not
        // opaque: no stamps — every frame is then composed the complete way (from_base() false).
        std::vector<uint32_t> words(size_t(kCells) * kStampMax * 2, kNoStamp);
        const View view = view_of_world();
        for (int cell = 0; cell < kCells; cell++) {
            Blit b;
            const float scale = (1.0f * 1.0f) * kUnitPx / d.y;
            if (!resolve_draw(view.cam, d.y, d.z, 0, (cell_x(cell) + -0.5f) * kUnitPx, (cell_y(cell) + -0.5f) * kUnitPx, scale, 1.0f, false,
                              false, b))
                continue;
            int count = 0;
            for (int j = 0; j < b.dh; j++)
                for (int i = 0; i < b.dw; i++) {
                    const int X = b.dx + i, Y = b.dy + j;
                    if (X < 0 || Y < 0 || X >= kObsW || Y >= kObsH) continue;  // S5
                    const uint32_t texel = tex[size_t(sample_index(b.sy, b.sh, j, b.dh)) * d.y + sample_index(b.sx, b.sw, i, b.dw)];
                    if ((texel >> 24) == 0u) continue;
                    if ((texel >> 24) != 255u || count == kStampMax) {
                        stamp_ok_ = false;
                        continue;
                    }
                    words[(size_t(cell) * kStampMax + count) * 2] = static_cast<uint32_t>(Y * kObsW + X);
                    words[(size_t(cell) * kStampMax + count) * 2 + 1] = texel & 0x00ffffffu;
                    count++;
                }
        }
        while (atlas.texel_bytes() % 16) atlas.append_words({0u});  // (the kernels read the table 16 bytes at a time)
        s_.stamps = atlas.append_words(words);
    }
    static size_t align256(size_t x) { return (x + 255) & ~size_t(255); }
    struct Layout {
        size_t shadow, slot, mt, tiles, f, i, mf, mb, eb, reset_list, reset_count, due_list, due_count, prepared, total;
    };
    static Layout layout(int n) {
        Layout l{};
        size_t off = 0;
        auto take = [&](size_t bytes) {
            size_t at = off;
            off += align256(bytes);
            return at;
        };
        l.shadow = take(sizeof(Level));  // unused: this game never prefetches
        l.slot = take(size_t(n) * 4);
        l.mt = take(size_t(n) * kMtWords * 4);
        l.tiles = take(size_t(n) * kTileStride);
        l.f = take(size_t(F_COUNT) * n * 4);
        l.i = take(size_t(I_COUNT) * n * 4);
        l.mf = take(size_t(MF_COUNT) * kMobs * n * 4);
        l.mb = take(size_t(2) * kMobs * n);
        l.eb = take(size_t(EB_COUNT) * kMaxEnt * n);  // (same size either way round)
        l.reset_list = take(size_t(n) * 4);
        l.reset_count = take(8);
        l.due_list = take(size_t(2) * n * 4);
        l.due_count = take(8);
        l.prepared = take(sizeof(ComposeHand));
        l.total = off;
        return l;
    }
    size_t state_bytes(int n) const override { return layout(n).total; }
    bool set_game_flags(uint32_t flags) override {  // include/procgen2_vec.h PGV_CHASER_FLOAT_ABS
        s_.float_abs = (flags & PGV_CHASER_FLOAT_ABS) ? 1 : 0;
        return (flags & ~PGV_CHASER_FLOAT_ABS) == 0;
    }
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
        s_.mf = reinterpret_cast<float*>(p + l.mf);
        s_.mb = p + l.mb;
        s_.eb = p + l.eb;
        s_.reset_list = reinterpret_cast<int32_t*>(p + l.reset_list);
        s_.reset_count = reinterpret_cast<int32_t*>(p + l.reset_count);
        s_.due_list = reinterpret_cast<int32_t*>(p + l.due_list);
        s_.due_count = reinterpret_cast<int32_t*>(p + l.due_count);
        s_.prepared = reinterpret_cast<ComposeHand*>(p + l.prepared);
        s_.ranks = atlas.sort_ranks;
        atlas_ = atlas;
    }
    int blocks() const { return (s_.n + 63) / 64; }
    void launch_make(hipStream_t st, uint32_t seed_base, int env_offset) override {
        hipLaunchKernelGGL(make_kernel, dim3(blocks()), dim3(64), 0, st, s_);
        hipLaunchKernelGGL(prepare_kernel, dim3(1), dim3(128), 0, st, s_, atlas_);
        LevelLaunch<Gen>::make(st, s_, 0, seed_base, env_offset, plan);
    }
    void launch_reset(hipStream_t st, const uint8_t* mask, const int32_t* seeds, StepIO io) override {
        LevelLaunch<Gen>::reset(st, s_, 0, mask, seeds, io, plan);
    }
    bool resets_beside_logic() const override { return true; }
    void launch_logic(hipStream_t st, const int32_t* actions, uint32_t run_seed, uint32_t step_index, int env_offset,
                      StepIO io) override {
        s_.parity = static_cast<int>(step_index & 1u);
        s_.listing = 1;  // (the envs reset in this step are listed for the late pass)
        // the auto-resets, beside the logic kernel (engine.hip: reset_stream is always there for a game that resets so)
        hipLaunchKernelGGL(due_level_kernel, dim3(s_.n < kDueBlocks ? s_.n : kDueBlocks), dim3(64), 0, reset_stream, s_, io, plan);
        hipLaunchKernelGGL(logic_kernel, dim3((s_.n * kGang + 63) / 64), dim3(64), 0, st, s_, actions, run_seed, step_index,
                           env_offset, io, (debug_flags & kDebugChaserSerialMobs) ? 1 : 0);
    }
    bool launch_frame(hipStream_t st, int env, uint32_t* d_px, int w, int h) override {
        hipLaunchKernelGGL(frame_kernel, dim3(1), dim3(kFrameThreads), 0, st, s_, atlas_, env, FrameTarget{d_px, w, h});
        return true;
    }
    // (the draw-list replay and kDebugNoPrepass take the kernel that resolves its own draws)
    bool lean() const { return !(debug_flags & (1 | kDebugNoPrepass)); }
    void launch_prepass(hipStream_t st, const uint8_t* mask) override {
        (void)mask;  // (every env: a lane's work, and the frames of the others are not drawn)
        if (lean()) hipLaunchKernelGGL(setup_kernel, dim3((s_.n * kPrepLanes + 255) / 256), dim3(256), 0, st, s_, atlas_);
    }
    // pgv_reset's frames: the complete way (new levels: their base layers are made on the way).
    void launch_render(hipStream_t st, const uint8_t* mask, StepIO io) override {
        hipLaunchKernelGGL(render_kernel<false>, dim3(s_.n), dim3(128), 0, st, s_, atlas_, mask, io, debug_flags, 0);
        if (!mask) base_valid_ = true;  // (every env has been rendered the complete way)
    }
    // A step's frames start from the base layers when every env drawn by this launch has a current one: the layers are
    // valid (not right after make / pgv_load_state), the envs reset in this step are left to the late pass (resets on their
    // own stream), the point sprite is a stamp (an opaque box in a clear rim: true of the reference's asset).
    bool from_base() const { return lean() && base_valid_ && stamp_ok_; }
    void launch_render_step(hipStream_t st, StepIO io) override {
        if (from_base())
            hipLaunchKernelGGL(render_kernel<true>, dim3(s_.n), dim3(128), 0, st, s_, atlas_, nullptr, io, debug_flags, 1);
        else
            hipLaunchKernelGGL(render_kernel<false>, dim3(s_.n), dim3(128), 0, st, s_, atlas_, nullptr, io, debug_flags, 1);
        base_valid_ = true;  // (either way every env's layer has been written by now, or is by this launch and the late pass)
    }
    static size_t up256(size_t b) { return (b + 255) & ~size_t(255); }
    size_t scratch_bytes(int n) const override {
        return up256(size_t(2) * kCells * kBlitWords * 4) + up256(size_t(n) * kMovers * kBlitWords * 4) + up256(size_t(n) * 8 * 4) +
               up256(size_t(n) * kFbWords * 4) + up256(size_t(n) * kMtN * 4) + up256(size_t(n));
    }
    void state_loaded(hipStream_t st) override {
        base_valid_ = false;
        hipMemsetAsync(s_.mt_sel, 0, size_t(s_.n), st);  // the streams that were just loaded are in mt[env]; what was made ahead is not theirs
    }
    // A snapshot takes the streams from mt[env]: the ones whose gang has moved on to the second buffer come home first.
    void prepare_save(hipStream_t st) override {
        hipLaunchKernelGGL(streams_home_kernel, dim3((s_.n + 63) / 64), dim3(64), 0, st, s_);
    }
    void bind_scratch(void* d_scratch, int n) override {
        uint8_t* p = static_cast<uint8_t*>(d_scratch);
        s_.prep.cell_blits = reinterpret_cast<uint32_t*>(p);
        p += up256(size_t(2) * kCells * kBlitWords * 4);
        s_.prep.movers = reinterpret_cast<uint32_t*>(p);
        p += up256(size_t(n) * kMovers * kBlitWords * 4);
        s_.prep.bg = reinterpret_cast<uint32_t*>(p);
        p += up256(size_t(n) * 8 * 4);
        s_.base = reinterpret_cast<uint32_t*>(p);
        p += up256(size_t(n) * kFbWords * 4);
        s_.mt_other = reinterpret_cast<uint32_t*>(p);
        p += up256(size_t(n) * kMtN * 4);
        s_.mt_sel = p;  // (the engine zeroes the scratch block: every stream is at home, nothing is made ahead)
        base_valid_ = false;
    }
    bool launch_render_late(hipStream_t st, StepIO io) override {
        const int groups = s_.n < 1024 ? s_.n : 1024;
        hipLaunchKernelGGL(render_list_kernel, dim3(groups + (s_.n + 127) / 128), dim3(128), 0, st, s_, atlas_, io, debug_flags, groups);
        return true;
    }
    // Same layout as oracle/pgo_chaser.cpp Chaser::dump_state.
    int dump_state(hipStream_t st, int env, float* out, int cap) override {
        hipStreamSynchronize(st);
        const size_t n = s_.n;
        auto rf = [&](const float* base, size_t idx) {
            float v;
            hipMemcpy(&v, base + idx, 4, hipMemcpyDeviceToHost);
            return v;
        };
        auto rb = [&](const uint8_t* base, size_t idx) {
            uint8_t v;
            hipMemcpy(&v, base + idx, 1, hipMemcpyDeviceToHost);
            return v;
        };
        auto ri = [&](int field) {
            int32_t v;
            hipMemcpy(&v, s_.i + size_t(field) * n + env, 4, hipMemcpyDeviceToHost);
            return v;
        };
        auto f = [&](int field) { return rf(s_.f, size_t(field) * n + env); };
        const int n_ent = ri(I_NENT);
        std::vector<float> v = {f(F_AX), f(F_AY), f(F_AVX), f(F_AVY), f(F_NVX), f(F_NVY), f(F_INPUT_T), f(F_ANIM_T),
                                static_cast<float>(ri(I_ANIM_I)), f(F_EAT_T), static_cast<float>(ri(I_BG)), f(F_BGSHIFT),
                                static_cast<float>(n_ent)};
        for (int e = 0; e < n_ent; e++) {
            const int info = rb(s_.eb, (size_t(env) * EB_COUNT + EB_INFO) * kMaxEnt + e);
            const int cell = rb(s_.eb, (size_t(env) * EB_COUNT + EB_CELL) * kMaxEnt + e) |
                             (kWideCells ? rb(s_.eb, (size_t(env) * EB_COUNT + EB_CELL_HI) * kMaxEnt + e) << 8 : 0);
            const int kind = info & kKindMask;
            v.push_back((info & kAlive) ? 1.0f : 0.0f);
            v.push_back(static_cast<float>(kind));
            if (kind == kEgg) {
                const int m = e - kOrbs;
                auto mf = [&](int field) { return rf(s_.mf, (size_t(field) * kMobs + m) * n + env); };
                v.push_back(mf(MF_X));
                v.push_back(mf(MF_Y));
                v.push_back(mf(MF_VX));
                v.push_back(mf(MF_VY));
                v.push_back(mf(MF_HATCH));
                v.push_back(static_cast<float>(rb(s_.mb, (size_t(0) * kMobs + m) * n + env)));
            } else {
                v.push_back(static_cast<float>(cell / H) + 0.5f);
                v.push_back(static_cast<float>(H - 1 - cell % H) + 0.5f);
                v.push_back(0.0f);
                v.push_back(0.0f);
                v.push_back(0.0f);
                v.push_back(0.0f);
            }
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
    bool base_valid_ = false;  // every env's base layer (State::base) is its current level's
    bool stamp_ok_ = false;    // the point sprite is an opaque box in a clear rim (extend_atlas)
};

}  // namespace chaser

}  // namespace PG_VARIANT_NS

std::unique_ptr<Game> PG_FACTORY(make_chaser)() { return std::make_unique<PG_VARIANT_NS::chaser::ChaserGame>(); }

}  // namespace pg
