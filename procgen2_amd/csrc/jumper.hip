// jumper on gfx950: double-jumping bunny in a cave, spikes, a carrot, and a compass HUD.
//
// Reference:
//   step   games/jumper/jumper.cpp:340-389, common_systems.cpp:57-202 (agent), :255-283 (particles), :7-24 (sprites)
//   render games/jumper/jumper.cpp:445-509 (incl. the compass), tilemap.cpp:255-281, common_systems.cpp:26-48,
//          :204-247 (agent), :285-308 (particles)
//   reset  games/jumper/jumper.cpp:511-533, tilemap.cpp:79-253, maze_generator.cpp:47-173, room_generator.cpp:4-202
// Config = the reference's compile-time default, hard_mode (40×40, pruned; jumper/tilemap.h:44-46).
//
// Machine mapping: logic one lane per env, render one wavefront per env, level generation one wavefront per env on
// LDS (pg_kruskal.h maze → noisy 3× upscale → pg_rooms.h cave/room/path → spikes and wall trimming as column bit
// masks).  No random draws during an episode, so the next level is generated ahead of time (pg_prefetch.h).
// The compass needle's angle is std::atan2(float, float): pg_atan2.h is glibc's atan2f bit for bit.
// D21 (oracle/pgo_chaser.cpp) applies to one line: `abs(velocity.x) > 0.01f` (common_systems.cpp:198) is the int abs.
#include "pg_atan2.h"
#include "../../include/procgen2_vec.h"
#include "pg_engine.h"
#include "pg_frame.h"
#include "pg_geom.h"
#include "pg_kruskal.h"
#include "pg_order.h"
#include "pg_prefetch.h"
#include "pg_prepass.h"
#include "pg_render.h"
#include "pg_rng.h"
#include "pg_defs.h"
// jumper/tilemap.cpp: world_dim by Distribution_Mode — hard_mode 40 (the reference's compile-time default, tilemap.h),
// easy_mode 20, memory_mode 45 (no pruning to the goal path, no spikes).
#if PG_VARIANT == 0
#define PG_ROOMS_DIM 40
#elif PG_VARIANT == 1
#define PG_ROOMS_DIM 20
#elif PG_VARIANT == 2
#define PG_ROOMS_DIM 45
#else
#error "jumper: unknown PG_VARIANT"
#endif
#include "pg_rooms.h"
#include "pg_tiles.h"

#ifndef PG_RENDER_WAVES
// Wavefronts per SIMD the render kernel's registers are capped for.  Five (at most 96 registers; it would take 99) lets the
// LDS, not the registers, decide how many envs a CU holds: nine instead of eight (measured: render 0.869 -> 0.845 ms).
#define PG_RENDER_WAVES 5
#endif
namespace pg {
namespace PG_VARIANT_NS {
namespace jumper {

constexpr int W = rooms::W, H = rooms::H, kCells = W * H;
constexpr int kTileStride = (kCells + 3) / 4 * 4;  // tiles are copied as 32-bit words
constexpr bool kPrune = PG_VARIANT != 2;              // tilemap.cpp:176 should_prune = mode != memory_mode
constexpr float kSpikeProb = PG_VARIANT == 2 ? 0.0f : 0.2f;  // tilemap.cpp:205
constexpr int kMaxSpikes = 126;             // entity ids: 0 carrot, 1 bunny, 2.. spikes
constexpr int kMaxSprites = kMaxSpikes + 1;  // carrot + spikes
constexpr int kPuffs = 10, kPuffSlots = 16, kSpikeSlots = 128;
static_assert(kMaxSprites + 1 <= kSpikeSlots, "draw list row");
enum Tile : uint8_t { kEmpty = 0, kWallTop = 1, kWallMid = 2, kSpike = 3 };  // tilemap.h:19-26

enum Tex {
    kTexTop = 0,  // 4 themes
    kTexMid = 4,  // 4 themes
    kTexSpike = 8,
    kTexCarrot = 9,
    kTexStand = 10,
    kTexJump = 11,
    kTexWalk1 = 12,
    kTexWalk2 = 13,
    kTexPuff = 14,
    kTexCircle = 15,
    kTexNeedle = 16,
    kTexBar = 17,
    kTexBackdrop = 18,  // 49
    kTexCount = 67
};

enum {
    F_AX, F_AY, F_AVX, F_AVY, F_APHASE, F_JUMP_T, F_CAMX, F_CAMY, F_TOGX, F_TOGY, F_GX, F_GY, F_BGSHIFT, F_PTIMER,
    F_NEEDLE_X, F_NEEDLE_Y, F_BAR_W,  // the compass as it lands on the 64×64 observation (store_compass)
    F_COUNT
};
enum { I_FLAGS, I_THEMES, I_NSPIKES, I_JUMPS, I_HASH_SPRITE, I_NEEDLE_SN, I_NEEDLE_CS, I_COUNT };
constexpr int kFlagGround = 1, kFlagForward = 2, kFlagListed = 4, kFlagPuffOn = 8;
enum { PF_X, PF_Y, PF_LIFE, PF_COUNT };

// One generated level, as the generator leaves it in LDS and as it waits in the shadow slot (pg_prefetch.h).
struct Level {
    uint8_t tiles[kTileStride];
    float ax, ay, gx, gy, bgshift;
    int32_t themes, n_spikes;
    uint16_t spike_cell[kMaxSpikes];
    uint8_t draw[kMaxSprites + 1];  // entity ids in draw order (0 = carrot, k + 2 = spike k)
};
static_assert(sizeof(Level) % 4 == 0, "Level is copied as 32-bit words");

struct GenLds {
    rooms::RoomsLds r;  // r.mt is the stream
    KruskalLds k;
    unsigned long long mid[W + 2], open[W + 2];  // per column (index x + 1): wall_mid cells / empty cells, bit = y
    float draws[64];
};

struct State {
    int n;
    Level* shadow;   // [n]  next level of each env
    int32_t* slot;   // [n]  SlotState
    uint32_t* mt;    // [n][625]  generator chain
    uint8_t* tiles;  // [n][kTileStride]  column-major y + x*H
    float* f;        // [F_COUNT][n]
    int32_t* i;      // [I_COUNT][n]
    // per-env contiguous tables: the render wavefronts' lanes index them by slot, so a table of one env is a coalesced
    // request instead of one cache line per lane
    float* pf;       // [n][PF_COUNT][kPuffSlots]
    uint16_t* spike_cell;  // [n][kSpikeSlots]
    uint8_t* draw;         // [n][kSpikeSlots]
    int float_abs;         // game_flags PGV_JUMPER_FLOAT_ABS (D21)
    // the compass ring as it lands on the observation (extend_atlas; word offsets into the atlas, 0 = not prepared)
    uint32_t hud_image, hud_list;
    uint32_t hud_under;  // 64 × 2 words: bit x of pair y = the ring's picture has an opaque texel at pixel (x, y) (pg_render.h compose_rows_from UNDER)
    uint32_t hud_cover;  // 64 words: the columns of each pixel row the ring's opaque texels overwrite (pg_prepass.h `cover`)
    PrepOut prep;  // what setup_kernel leaves for render_kernel (pg_prepass.h); not part of the state blob
    uint32_t* fat;  // [1 + n]  number of frames the pre-pass left to the complete path, then their envs (render_full_kernel)
};

PG_D float& SF(const State& s, int field, int env) { return s.f[size_t(field) * s.n + env]; }
PG_D int32_t& SI(const State& s, int field, int env) { return s.i[size_t(field) * s.n + env]; }
PG_D float& PF(const State& s, int field, int k, int env) { return s.pf[(size_t(env) * PF_COUNT + field) * kPuffSlots + k]; }
PG_D uint16_t& SPK(const State& s, int k, int env) { return s.spike_cell[size_t(env) * kSpikeSlots + k]; }
PG_D uint8_t& DRW(const State& s, int k, int env) { return s.draw[size_t(env) * kSpikeSlots + k]; }

using Win = TileWinT<W, H, kWallMid>;  // out of bounds is a wall (tilemap.h:84-89)
PG_D bool is_wall(int t) { return t == kWallMid || t == kWallTop; }
PG_D float cell_x(int cell) { return static_cast<float>(cell / H) + 0.5f; }
PG_D float cell_y(int cell) { return static_cast<float>(H - 1 - cell % H) + 0.5f; }

// ------------------------------------------------------------------------------------------------
// level generation
// ------------------------------------------------------------------------------------------------
// Maze_Generator::generate_maze_no_dead_ends, second half (maze_generator.cpp:132-173): every open cell of the padded
// grid with exactly one open neighbour gets one more opening, chosen — quirk kept — among the first `walls` entries of
// its neighbour list (−x, +x, −y, +y), not among its wall neighbours.  The scan mutates the grid it reads.
PG_D void open_dead_ends(KruskalLds& K, int dim, uint32_t* mt, int lane) {
    const int ah = dim + 2;
    for (int i = 0; i < ah * ah; i++) {
        if (K.grid[i] != 0) continue;
        const int x = i / ah, y = i % ah;
        const int nb[4] = {y + ah * (x - 1), y + ah * (x + 1), (y - 1) + ah * x, (y + 1) + ah * x};
        int spaces = 0, walls = 0;
#pragma unroll
        for (int n = 0; n < 4; n++) {
            const int g = K.grid[nb[n]];
            spaces += g == 0 ? 1 : 0;
            walls += g == 1 ? 1 : 0;
        }
        if (spaces == 1 && walls > 0) {
            const int pick = wave_rng_int(mt, 0, walls - 1, lane);
            if (lane == 0) {
                for (int n = 0; n < 4; n++) {
                    const int cell = nb[(pick + n) % walls];
                    const int cx = cell / ah, cy = cell % ah;
                    if (cx >= 1 && cy >= 1 && cx < ah - 1 && cy < ah - 1 && K.grid[cell] == 1) {
                        K.grid[cell] = 0;
                        break;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// Column masks of the tile map under construction (bit y of column x): lane = column.
PG_D void masks_from_tiles(GenLds& L, const uint8_t* tiles, int lane) {
    if (lane < W + 2) {
        unsigned long long m = 0, o = 0;
        const int x = lane - 1;
        if (x >= 0 && x < W)
            for (int y = 0; y < H; y++) {
                const int t = tiles[y + x * H];
                m |= static_cast<unsigned long long>(t == kWallMid) << y;
                o |= static_cast<unsigned long long>(t == kEmpty) << y;
            }
        L.mid[lane] = m;  // columns −1 and W: no wall_mid bit and no empty bit — at() there is wall_mid, which only
        L.open[lane] = o;  // ever matters through "is it empty" (no) and "is the cell below a wall" (handled below)
    }
    __syncthreads();
}

// is_space_on_ground for a whole column (tilemap.cpp:52-62): empty, empty above, wall below.  Row −1 is out of
// bounds = wall, row H is out of bounds = not empty.  Only wall_mid and empty exist at this point.
PG_D unsigned long long ground_mask(unsigned long long mid, unsigned long long open) {
    const unsigned long long below_is_wall = (mid << 1) | 1ull;
    return open & (open >> 1) & below_is_wall & ((1ull << H) - 1ull);
}

PG_D void generate(const State& s, int env, GenLds& L, Level& lv, bool reseed, uint32_t seed, int lane) {
    rooms::RoomsLds& R = L.r;
    uint32_t* gmt = s.mt + size_t(env) * kMtWords;
    if (reseed) {
        if (lane == 0) mt_seed(R.mt, seed);
    } else {
        for (int k = lane; k < kMtWords; k += 64) R.mt[k] = gmt[k];
    }
    __syncthreads();
    uint32_t* mt = R.mt;
    // tilemap.cpp:95-123: a (W/3)² maze (13×13 by default) without dead ends, blown up ×3 with noise
    constexpr int kScale = 3, kDim = W / kScale;
    Carver carver{L.k, 0, 0, 0, 0};
    carver.carve(kDim, mt, lane);
    open_dead_ends(L.k, kDim, mt, lane);
    wave_draws(mt, kCells, lane, [&](int i, float v) {
        const int obj = L.k.grid[((i % H) / kScale + 1) + (kDim + 2) * ((i / H) / kScale + 1)];
        R.grid[i] = v < (obj == 1 ? 0.8f : 0.2f) ? 1 : 0;
    });
    __syncthreads();
    rooms::automaton(R.grid, R.aux, lane);
    __syncthreads();
    rooms::automaton(R.aux, R.grid, lane);
    __syncthreads();
    for (int c = lane; c < kCells; c += 64) {  // border cells become wall (tilemap.cpp:125-141)
        const int x = c / H, y = c % H;
        if (x == 0 || y == 0 || x == W - 1 || y == H - 1) R.grid[c] = 1;
    }
    __syncthreads();
    const int n_room = rooms::best_room(R, lane);  // R.cells: the room in the reference's iteration order
    const int goal_cell = R.cells[wave_rng_int(mt, 0, n_room - 1, lane)];
    // tiles: everything wall_mid except the best room (tilemap.cpp:146-155); lv.tiles is the working map
    for (int c = lane; c < kTileStride; c += 64) lv.tiles[c] = kWallMid;
    __syncthreads();
    for (int k = lane; k < n_room; k += 64) lv.tiles[R.cells[k]] = kEmpty;
    __syncthreads();
    masks_from_tiles(L, lv.tiles, lane);
    // agent candidates in x-major order (tilemap.cpp:161-171): the drawn-th ground cell that is not the goal
    int agent_cell;
    {
        unsigned long long g = 0;
        if (lane < W) g = ground_mask(L.mid[lane + 1], L.open[lane + 1]);
        if (lane == goal_cell / H) g &= ~(1ull << (goal_cell % H));
        int before = __popcll(g);  // exclusive prefix over columns
        int upto = before;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(upto, off);
            if (lane >= off) upto += t;
        }
        const int total = __shfl(upto, 63);
        before = upto - before;
        const int pick = wave_rng_int(mt, 0, total - 1, lane);
        int found = -1;
        if (pick >= before && pick < upto) {
            unsigned long long v = g;
            for (int k = 0; k < pick - before; k++) v &= v - 1;
            found = __builtin_ctzll(v) + lane * H;
        }
        const unsigned long long who = __ballot(found >= 0);
        agent_cell = __shfl(found, __builtin_ctzll(who));
    }
    if (lane == 0) {
        R.agent_cell = agent_cell;
        R.goal_cell = goal_cell;
    }
    __syncthreads();
    if (kPrune) {  // find_path has no side effects, so memory_mode skips it with the pruning
        rooms::goal_path(R, agent_cell, goal_cell, lane);
        rooms::widen(R, lane);
        for (int c = lane; c < kCells; c += 64) lv.tiles[c] = R.aux[c] ? kEmpty : kWallMid;  // pruned to the wide path
        __syncthreads();
        masks_from_tiles(L, lv.tiles, lane);
    }

    // spikes (tilemap.cpp:203-211): x-major scan; a cell takes a draw when it and both horizontal neighbours are
    // ground cells — and a spike placed in the previous column makes that neighbour "not empty".  Column by column,
    // the rows of a column in one go: the k-th eligible row takes the k-th draw.
    {
        unsigned long long spikes_prev = 0;
        for (int x = 0; x < W; x++) {
            const unsigned long long here = ground_mask(L.mid[x + 1], L.open[x + 1]);
            const unsigned long long left = ground_mask(L.mid[x], L.open[x]) & ~spikes_prev;
            const unsigned long long right = ground_mask(L.mid[x + 2], L.open[x + 2]);
            const unsigned long long eligible = here & left & right;  // columns −1 and W have no ground bits
            unsigned long long placed = 0;
            const int count = __popcll(eligible);
            if (count > 0) {
                wave_draws(mt, count, lane, [&](int k, float v) { L.draws[k] = v; });
                __syncthreads();
                bool spike = false;
                if (lane < H && ((eligible >> lane) & 1ull))
                    spike = L.draws[__popcll(eligible & ((1ull << lane) - 1ull))] < kSpikeProb;
                placed = __ballot(spike);
                __syncthreads();
            }
            spikes_prev = placed;
            if (lane == 0) L.open[x + 1] &= ~placed;  // a spike is neither empty nor wall
            if (lane < H && ((placed >> lane) & 1ull)) lv.tiles[lane + x * H] = kSpike;
            __syncthreads();
        }
    }
    // no long vertical walls (tilemap.cpp:213-224): strictly sequential (every removal changes later tests), but on
    // column masks it is a few bit operations per cell; all lanes walk it together, the draws are wave-uniform.
    for (int x = 0; x < W; x++) {
        unsigned long long mid = L.mid[x + 1], open = L.open[x + 1];
        const unsigned long long open_left = L.open[x], open_right = L.open[x + 2];
        for (int y = 0; y < H; y++) {
            if ((((mid & open_right) >> y) & 7ull) == 7ull) {  // is_left_wall(x, y .. y+2)
                const int k = y + wave_rng_int(mt, 0, 2, lane);
                mid &= ~(1ull << k);
                open |= 1ull << k;
            }
            if ((((mid & open_left) >> y) & 7ull) == 7ull) {  // is_right_wall(x, y .. y+2)
                const int k = y + wave_rng_int(mt, 0, 2, lane);
                mid &= ~(1ull << k);
                open |= 1ull << k;
            }
        }
        if (lane == 0) {
            L.mid[x + 1] = mid;
            L.open[x + 1] = open;
        }
        __syncthreads();
    }
    // write the masks back, collect the spikes in index order (tilemap.cpp:236-245), cap the walls (:247-253)
    int n_spikes = 0;
    for (int c0 = 0; c0 < kCells; c0 += 64) {
        const int c = c0 + lane;
        const bool inside = c < kCells;
        const int x = inside ? c / H : 0, y = inside ? c % H : 0;
        const bool was_spike = inside && lv.tiles[c] == kSpike;
        const bool mid = (L.mid[x + 1] >> y) & 1ull;
        // is_top_wall: wall_mid with an empty cell above; the spike tiles have turned back into empty cells by then
        const bool empty_above = y + 1 < H && !((L.mid[x + 1] >> (y + 1)) & 1ull);
        const bool keep = was_spike && c != agent_cell && c != goal_cell;
        const unsigned long long m = __ballot(keep);
        if (keep) {
            const int at = n_spikes + __popcll(m & ((1ull << lane) - 1ull));
            if (at < kMaxSpikes) lv.spike_cell[at] = static_cast<uint16_t>(c);
        }
        n_spikes += __popcll(m);
        __syncthreads();
        if (inside) lv.tiles[c] = mid ? (empty_above ? kWallTop : kWallMid) : kEmpty;
    }
    if (n_spikes > kMaxSpikes) __builtin_trap();  // far beyond anything the generator produces
    __syncthreads();
    const int backdrop = wave_rng_int(mt, 0, 48, lane);
    const float shift = wave_rng_real(mt, 0.0f, 1.0f, lane);
    const int theme = wave_rng_int(mt, 0, 3, lane);
    if (lane == 0) {
        lv.ax = static_cast<float>(agent_cell / H) + 0.5f;
        lv.ay = static_cast<float>(H - 1 - (agent_cell % H));  // no +0.5 (tilemap.cpp:226)
        lv.gx = cell_x(goal_cell);
        lv.gy = cell_y(goal_cell);
        lv.bgshift = shift;
        lv.themes = backdrop | (theme << 8);
        lv.n_spikes = n_spikes;
        // draw order of the episode: the sprite set (carrot id 0, spikes ids 2..) in its iteration order, then
        // std::sort on z (all 1.0); nothing is destroyed during an episode, so it is fixed here
        int32_t packed = SI(s, I_HASH_SPRITE, env);
        HashOrder h;
        h.next = R.queue;
        h.before = reinterpret_cast<int16_t*>(R.touch);
        h.head = kNil;
        h.buckets = packed & 0xffff;
        h.next_resize = packed >> 16;
        h.count = 0;
        for (int b = 0; b < h.buckets; b++) h.before[b] = kNil;
        hash_insert(h, 0);
        for (int k = 0; k < n_spikes; k++) hash_insert(h, 2 + k);
        SI(s, I_HASH_SPRITE, env) = h.buckets | (h.next_resize << 16);
        ZItem* items = reinterpret_cast<ZItem*>(R.chain);
        int n = 0;
        for (int16_t p = static_cast<int16_t>(h.head); p != kNil; p = h.next[p]) items[n++] = {1.0f, p};
        sort_by_key(items, n);
        for (int k = 0; k < n; k++) lv.draw[k] = static_cast<uint8_t>(items[k].id);
    }
    __syncthreads();
    for (int k = lane; k < kMtWords; k += 64) gmt[k] = R.mt[k];
    __syncthreads();
}

// The level becomes the env's live state (what reset() and the component constructors initialise).
PG_D void install(const State& s, int env, const Level& lv, int lane) {
    uint32_t* tiles = reinterpret_cast<uint32_t*>(s.tiles + size_t(env) * kTileStride);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(lv.tiles);
    for (int k = lane; k < kTileStride / 4; k += 64) tiles[k] = src[k];
    const int n_spikes = lv.n_spikes;
    for (int k = lane; k < n_spikes; k += 64) SPK(s, k, env) = lv.spike_cell[k];
    for (int k = lane; k < n_spikes + 1; k += 64) DRW(s, k, env) = lv.draw[k];
    if (lane < kPuffs)
        for (int f = 0; f < PF_COUNT; f++) PF(s, f, lane, env) = 0.0f;
    if (lane == 0) {
        SF(s, F_AX, env) = lv.ax;
        SF(s, F_AY, env) = lv.ay;
        SF(s, F_AVX, env) = 0.0f;
        SF(s, F_AVY, env) = 0.0f;
        SF(s, F_APHASE, env) = 0.0f;
        SF(s, F_JUMP_T, env) = 0.0f;
        SF(s, F_GX, env) = lv.gx;
        SF(s, F_GY, env) = lv.gy;
        SF(s, F_BGSHIFT, env) = lv.bgshift;
        SF(s, F_PTIMER, env) = 0.0f;
        // on_ground = false, face_forward = true, particles enabled, draw list cleared
        SI(s, I_FLAGS, env) = kFlagForward | kFlagPuffOn;
        SI(s, I_THEMES, env) = lv.themes;
        SI(s, I_NSPIKES, env) = n_spikes;
        SI(s, I_JUMPS, env) = 2;
        // camera and System_Agent::info.to_goal keep the previous episode's values until the first update (D3)
    }
}

// What cenv_make leaves in an env besides the seeded RNG, split by owner: the generator chain (bucket counts of
// the sets that survive clear()) and the live state.  Level-seed mode (pg_engine.h LevelPlan) rebuilds every
// level from here.
PG_D void fresh_chain(const State& s, int env) {
    SI(s, I_HASH_SPRITE, env) = 1;  // empty unordered_set: one bucket, next_resize 0
}
// The compass of the observation (jumper.cpp:473-509 at the 64×64 target: game_zoom 0.3) is a function of to_goal
// alone: where the needle sits, its angle as the 16.16 sine and cosine of raster spec S6, how long the bar is.  Worked
// out here, by the lane that owns the env in the logic kernel — 64 envs per pass through atan2f / sinf / cosf —
// instead of by every lane of both render wavefronts of the env for the sake of one.  sn = 0, cs = 0 stands for an
// angle of exactly zero (drawn un-rotated, like the oracle).
constexpr float kObsZoom = 0.3f;
PG_D void store_compass(const State& s, int env, float tx, float ty) {
    SF(s, F_TOGX, env) = tx;
    SF(s, F_TOGY, env) = ty;
    const float game_zoom = kObsZoom;
    const float width = 64.0f, compass_size = 200.0f, offset_x = -32.0f, offset_y = 32.0f;
    const float angle = static_cast<float>(at_atan2f(ty, tx) * 180.0f / 3.14159265358979323846);
    const float dist = __fsqrt_rn(tx * tx + ty * ty);
    const float dist_inv = 1.0f / fmaxf(0.0001f, dist);
    const float dir_x = tx * dist_inv, dir_y = ty * dist_inv;
    const float ratio = fminf(1.0f, dist / (W * 1.414f));
    float dx = width - compass_size * 0.75f * game_zoom + offset_x * game_zoom;
    float dy = compass_size * 0.5f * game_zoom + offset_y * game_zoom;
    dx += compass_size * 0.25f * dir_x * game_zoom;
    dy += compass_size * 0.25f * dir_y * game_zoom;
    SF(s, F_NEEDLE_X, env) = dx;
    SF(s, F_NEEDLE_Y, env) = dy;
    SF(s, F_BAR_W, env) = compass_size * game_zoom * ratio;
    const double deg = static_cast<double>(angle);
    int sn = 0, cs = 0;
    if (deg != 0.0) rotation_16_16(deg, sn, cs);
    SI(s, I_NEEDLE_SN, env) = sn;
    SI(s, I_NEEDLE_CS, env) = cs;
}
PG_D void fresh_live(const State& s, int env) {
    SF(s, F_CAMX, env) = 0.0f;  // Renderer::camera_position{0} (renderer.h:18)
    SF(s, F_CAMY, env) = 0.0f;
    store_compass(s, env, 0.0f, 0.0f);  // Agent_Info::to_goal{0, 0} (common_systems.h:57-59)
}

struct Gen {  // pg_prefetch.h level_kernel<Gen>
    using State = jumper::State;
    using Level = jumper::Level;
    using GenLds = jumper::GenLds;
    PG_D static void generate(const State& s, int env, GenLds& L, Level& lv, bool reseed, uint32_t seed, int lane) {
        jumper::generate(s, env, L, lv, reseed, seed, lane);
    }
    PG_D static void install(const State& s, int env, const Level& lv, int lane) { jumper::install(s, env, lv, lane); }
    PG_D static void fresh_chain(const State& s, int env) { jumper::fresh_chain(s, env); }
    PG_D static void fresh_live(const State& s, int env) { jumper::fresh_live(s, env); }
};

// ------------------------------------------------------------------------------------------------
// step
// ------------------------------------------------------------------------------------------------
// cenv_step's four sub-steps (jumper.cpp:356-371 → System_Agent::update, System_Particles::update), lane = env.
// The reference's sub-step is physics → hazards and goal → particles, and ends the step at the first sub-step that kills or
// wins.  Neither the spikes, nor the goal, nor the particles feed back into the physics, so (round 6) the lane runs the
// physics of all four sub-steps first, keeping what each left behind; holds the four bodies against the spikes in ONE walk
// of the list (the cells sixteen at a time, two 16-byte loads in flight) and against the goal; finds the sub-step that ended
// the step, if one did; and only then lets the particles live through the sub-steps that happened, in registers.  The same
// operations on the same values in the same order within each of the three — but the kernel is a lane per env on a chain of
// memory round trips (vector ALU busy 0.14), and the step was 4 tile windows + 4 × n_spikes dependent loads + 4 passes over
// the particles' lives long; it is now one or two windows (a window serves while the body stays inside it, as in coinrun),
// n_spikes / 16 loads and one pass.  Sub-steps behind the one that ended the step are worked out and dropped.
PG_D void advance(const State& s, int env, int action, float& reward_out, bool& terminated_out) {
    const uint8_t* tiles = s.tiles + size_t(env) * kTileStride;
    const int n_spikes = SI(s, I_NSPIKES, env);
    int flags = SI(s, I_FLAGS, env), jumps = SI(s, I_JUMPS, env);
    float ax = SF(s, F_AX, env), ay = SF(s, F_AY, env), avx = SF(s, F_AVX, env), avy = SF(s, F_AVY, env);
    float phase = SF(s, F_APHASE, env), jump_t = SF(s, F_JUMP_T, env), p_timer = SF(s, F_PTIMER, env);
    const float gx = SF(s, F_GX, env), gy = SF(s, F_GY, env);
    bool ground = (flags & kFlagGround) != 0, forward = (flags & kFlagForward) != 0, puff_on = (flags & kFlagPuffOn) != 0;
    float life[kPuffs];
#pragma unroll
    for (int k = 0; k < kPuffs; k++) life[k] = PF(s, PF_LIFE, k, env);
    const uint4* spikes = reinterpret_cast<const uint4*>(&SPK(s, 0, env));  // eight cells a word of sixteen bytes
    uint4 cells_a = make_uint4(0u, 0u, 0u, 0u), cells_b = cells_a;           // the first sixteen leave with the state
    if (n_spikes > 0) cells_a = spikes[0];
    if (n_spikes > 8) cells_b = spikes[1];
    const float dt = 1.0f / 4;
    const float max_jump = 0.92f, gravity = 0.1f, max_speed = 0.5f, mix = 0.2f, air_control = 1.0f, jump_cooldown = 3.0f;
    const float movement_x = static_cast<float>((action == 6 || action == 7 || action == 8) -
                                                (action == 0 || action == 1 || action == 2));
    const bool jump = (action == 2 || action == 5 || action == 8);

    // ---- the physics of the four sub-steps: System_Agent::update (common_systems.cpp:57-202) minus hazards and goal
    float k_ax[4], k_ay[4], k_avx[4], k_avy[4], k_phase[4], k_jump_t[4];
    int k_jumps[4];
    bool k_ground[4], k_forward[4], k_puff[4];
    Win win{tiles, 0, 0, 0};
#pragma unroll
    for (int ss = 0; ss < 4; ss++) {
        const float mix_x = ground ? mix : (mix * air_control);
        avx += mix_x * (max_speed * movement_x - avx) * dt;
        if (fabsf(avx) < mix_x * max_speed * dt) avx = 0.0f;
        if (ground) jumps = 2;
        if (jump && jumps > 0 && jump_t == 0.0f) {
            avy = -max_jump;
            jumps--;
            jump_t = jump_cooldown;
        }
        if (jump_t > 0.0f) jump_t = fmaxf(0.0f, jump_t - dt);
        avy += gravity * dt;
        if (fabsf(avy) > max_jump) avy = (avy > 0.0f ? 1.0f : -1.0f) * max_jump;
        ax += avx * dt;
        ay += avy * dt;
        const Box body{ax + -0.25f, ay + -0.8f, 0.5f, 0.8f};
        {
            if (ss == 0 || !win.holds(body)) win = Win::around(tiles, body, avx, avy);
                        // (the nine-fixed-steps form, pg_tiles.h kFlat: the body is half a tile by 0.8 — 36.3 -> 29.2 µs for the kernel)
            const TileHit h = collide_plain<true>(win, body, is_wall);
            const float moved_x = h.x - body.x, moved_y = h.y - body.y;
            ground = moved_y < 0.0f && h.any;
            ax = h.x - -0.25f;
            ay = h.y - -0.8f;
            if (moved_x != 0.0f) avx = 0.0f;
            if (moved_y > 0.0f && h.any) avy = 0.0f;
            if (ground) avy = 0.0f;
        }
        phase += 0.1f * dt;
        phase = fmodf(phase, 1.0f);
        if (movement_x > 0.0f)
            forward = true;
        else if (movement_x < 0.0f)
            forward = false;
        {
            // `abs` = int abs(int) there (D21); PGV_JUMPER_FLOAT_ABS: the float overload (oracle/pgo_chaser.cpp qabs)
            const int truncated = static_cast<int>(avx);
            const float mag = s.float_abs ? fabsf(avx) : static_cast<float>(truncated < 0 ? -truncated : truncated);
            puff_on = !ground || mag > 0.01f;
        }
        k_ax[ss] = ax, k_ay[ss] = ay, k_avx[ss] = avx, k_avy[ss] = avy, k_phase[ss] = phase, k_jump_t[ss] = jump_t;
        k_jumps[ss] = jumps, k_ground[ss] = ground, k_forward[ss] = forward, k_puff[ss] = puff_on;
    }

    // ---- hazards and goal (common_systems.cpp:150-176): bit ss = the body after sub-step ss touches a spike / the carrot.
    // Any hit kills, order-free.
    int dead = 0, won = 0;
    Box body[4];
    // (what the four bodies span, in box_hit's own terms — a.x, a.x + a.w, a.y, a.y + a.h: a spike that misses the span
    // misses all four, and nearly every spike does)
    float lo_x = 1e30f, lo_y = 1e30f, hi_x = -1e30f, hi_y = -1e30f;
#pragma unroll
    for (int ss = 0; ss < 4; ss++) {
        body[ss] = Box{k_ax[ss] + -0.25f, k_ay[ss] + -0.8f, 0.5f, 0.8f};
        if (box_hit(body[ss], Box{gx + -0.5f, gy + -0.5f, 1.0f, 1.0f})) won |= 1 << ss;
        lo_x = fminf(lo_x, body[ss].x), hi_x = fmaxf(hi_x, body[ss].x + body[ss].w);
        lo_y = fminf(lo_y, body[ss].y), hi_y = fmaxf(hi_y, body[ss].y + body[ss].h);
    }
    for (int k0 = 0; k0 < n_spikes; k0 += 16) {
        const uint32_t w[8] = {cells_a.x, cells_a.y, cells_a.z, cells_a.w, cells_b.x, cells_b.y, cells_b.z, cells_b.w};
        if (k0 + 16 < n_spikes) cells_a = spikes[(k0 + 16) >> 3];  // the next sixteen, asked for before these are looked at
        if (k0 + 24 < n_spikes) cells_b = spikes[(k0 + 24) >> 3];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (k0 + j >= n_spikes) break;
            const int cell = static_cast<int>((w[j >> 1] >> (16 * (j & 1))) & 0xffffu);
            const Box spike{cell_x(cell) + -0.25f, cell_y(cell) + -0.25f, 0.5f, 0.5f};
            const bool near = lo_x < spike.x + spike.w && hi_x > spike.x && lo_y < spike.y + spike.h && hi_y > spike.y;
            if (__ballot(near) == 0) continue;  // (wave-uniform)
#pragma unroll
            for (int ss = 0; ss < 4; ss++)
                if (box_hit(body[ss], spike)) dead |= 1 << ss;
        }
    }
    const int ending = dead | won;
    const int last = ending ? __builtin_ctz(ending) : 3;  // the sub-step that ended the step, or all four took place
#pragma unroll
    for (int ss = 0; ss < 4; ss++)
        if (ss == last) {
            ax = k_ax[ss], ay = k_ay[ss], avx = k_avx[ss], avy = k_avy[ss], phase = k_phase[ss], jump_t = k_jump_t[ss];
            jumps = k_jumps[ss], ground = k_ground[ss], forward = k_forward[ss], puff_on = k_puff[ss];
        }
    const bool achieved_goal = ((won >> last) & 1) != 0, alive = ((dead >> last) & 1) == 0;

    // ---- System_Particles::update (common_systems.cpp:255-283) for the sub-steps that happened
#pragma unroll
    for (int ss = 0; ss < 4; ss++) {
        if (ss > last) break;
        const float lifespan = 5.0f, spawn_time = 0.5f;
        int dead_index = -1;
#pragma unroll
        for (int k = 0; k < kPuffs; k++) {
            life[k] = life[k] - dt;
            if (life[k] <= 0.0f) dead_index = k;
        }
        p_timer += dt;
        if (dead_index != -1 && p_timer >= spawn_time && k_puff[ss]) {
            p_timer = fmodf(p_timer, spawn_time);
#pragma unroll
            for (int k = 0; k < kPuffs; k++)
                if (k == dead_index) life[k] = lifespan;
            PF(s, PF_X, dead_index, env) = k_ax[ss] + 0.0f;
            PF(s, PF_Y, dead_index, env) = k_ay[ss] + -0.2f;
        }
    }
#pragma unroll
    for (int k = 0; k < kPuffs; k++) PF(s, PF_LIFE, k, env) = life[k];

    SF(s, F_AX, env) = ax;
    SF(s, F_AY, env) = ay;
    SF(s, F_AVX, env) = avx;
    SF(s, F_AVY, env) = avy;
    SF(s, F_APHASE, env) = phase;
    SF(s, F_JUMP_T, env) = jump_t;
    SF(s, F_PTIMER, env) = p_timer;
    SF(s, F_CAMX, env) = ax * kUnitPx;  // common_systems.cpp:179-180
    SF(s, F_CAMY, env) = (ay - 0.5f) * kUnitPx;
    store_compass(s, env, gx - ax, gy - ay);
    SI(s, I_JUMPS, env) = jumps;
    SI(s, I_FLAGS, env) = kFlagListed | (ground ? kFlagGround : 0) | (forward ? kFlagForward : 0) |
                          (puff_on ? kFlagPuffOn : 0);
    reward_out = achieved_goal * 10.0f;
    terminated_out = !alive || achieved_goal;
}

__global__ void __launch_bounds__(64) make_kernel(State s) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= s.n) return;
    fresh_chain(s, env);
    fresh_live(s, env);
}

__global__ void __launch_bounds__(64) logic_kernel(State s, const int32_t* actions, uint32_t run_seed,
                                                   uint32_t step_index, int env_offset, StepIO io, int prefetch, LevelPlan plan) {
    if (blockIdx.y == 1) {  // (block-uniform) the auto-resets whose level lies ready: a copy, beside the envs that step (pg_prefetch.h)
        __shared__ Level lv;
        install_prefetched<Gen>(s, blockIdx.x * blockDim.x, blockDim.x, prefetch, io, plan, lv, threadIdx.x, reset_served_mark(step_index), reset_due_mark(step_index));
        return;
    }
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= s.n) return;
    if (resets_in_step(io.pending[env], step_index)) return;  // this step is the env's reset (pg_prefetch.h: who serves it, and the byte)
    const int action =
        actions ? actions[env] : synthetic_action(run_seed, step_index, static_cast<uint32_t>(env_offset + env));
    float reward;
    bool terminated;
    advance(s, env, action, reward, terminated);
    io.reward[env] = reward;
    io.done[env] = terminated ? 1 : 0;
    io.pending[env] = terminated ? static_cast<uint8_t>(reset_due_mark(step_index + 1u)) : 0;  // (pg_prefetch.h: the byte)
}

// render_game(true) (jumper.cpp:445-509): one workgroup of two wavefronts per env (pg_render.h).
constexpr int kGrid = 16;  // 64 px / 4.8 px per tile → at most 16 columns/rows in view

// The complete frame of one env by its workgroup, set-up included: the frames the pre-pass marks fat, the draw-list
// replay (flags bit 0) and kDebugNoPrepass.
PG_D void render_full(const State& s, const AtlasView& atlas, const StepIO& io, int flags, int env, uint32_t* fb,
                      ComposeLds<kGrid>& L) {
    const int lane = threadIdx.x & 63, half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // two wavefronts per env (pg_render.h)
    constexpr int halves = 2;

    const float game_zoom = 0.3f;
    const Camera cam{SF(s, F_CAMX, env), SF(s, F_CAMY, env), 64.0f, 64.0f, game_zoom * 64.0f / 64.0f};
    const int themes = SI(s, I_THEMES, env), sflags = SI(s, I_FLAGS, env);
    const int backdrop = themes & 0xff, theme = (themes >> 8) & 0xff;
    const int n_draw = (sflags & kFlagListed) ? SI(s, I_NSPIKES, env) + 1 : 0;  // empty right after a reset
    const uint8_t* tiles = s.tiles + size_t(env) * kTileStride;
    const DescRegs descs = DescRegs::load(atlas, lane);
    Blit mine;

    float puff_life = 0.0f, puff_x = 0.0f, puff_y = 0.0f;
    if (lane < kPuffs) {
        puff_life = PF(s, PF_LIFE, lane, env);
        puff_x = PF(s, PF_X, lane, env);
        puff_y = PF(s, PF_Y, lane, env);
    }

    int bg_soft = 0;  // the backdrop has texels that are not opaque (descriptor .w)
    int4 bg_d;  // the background draw, jumper.cpp:459-464: texture, world position, scale — each wave resolves the axis it needs (pg_render.h BgAxis)
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
    // tile window (tilemap.cpp:255-264)
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
    const bool two = top_d.z != mid_d.z;  // the brown cap tile is 64×53 (see climber.hip)
    if (!(flags & 1) && cols <= kGrid && rows <= kGrid && top_d.y == mid_d.y && top_d.z <= mid_d.z) {
        compose_spans(fb, L, cam, x0, y0, cols, rows, mid_d.y, mid_d.z, kUnitPx / mid_d.y, lane, two ? top_d.z : 0, half, halves,
                      soft_rows_of(bg_soft, top_d.w | mid_d.w), hard_rows_of(bg_soft, mid_d.w), &bg_draw, &bga);  // (cap tiles are few: always worth the attempt)
#pragma unroll
        for (int k = half; k < kGrid * kGrid / 64; k += halves) {
            const int cell = k * 64 + lane;
            const int r = cell / kGrid, c = cell % kGrid;
            const int t = (c < cols && r < rows) ? Win::direct(tiles, x0 + c, y0 + r) : kEmpty;
            L.base[cell] = !is_wall(t) ? static_cast<int32_t>(kNoTexel)
                                       : (t == kWallTop ? (top_d.x * 4) | (two ? 1 : 0) : mid_d.x * 4);
        }
        __syncthreads();
        composed = two ? compose_rows<kGrid, true>(fb, L, atlas, bga, cols, rows, mid_d.y, lane, flags, half, halves)
                       : compose_rows<kGrid, false>(fb, L, atlas, bga, cols, rows, mid_d.y, lane, flags, half, halves);
    }
    if (!composed) {  // draw-list replay (tilemap.cpp:266-280)
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
                if (is_wall(t)) {
                    const int4 d = (t == kWallTop) ? top_d : mid_d;
                    has = resolve_draw(cam, d.y, d.z, d.x, x * kUnitPx, y * kUnitPx, kUnitPx / d.y, 1.0f, false, false,
                                       mine);
                }
            }
            wave_replay(fb, atlas, mine, __ballot(has), lane, half, halves);
        }
    }
    // The frame's draws after the tile layer, in the reference's order: particles (common_systems.cpp:285-308), the
    // positive-z sprites (carrot and spikes, :26-48), the bunny (:204-247), the compass (jumper.cpp:473-509).  They
    // differ in their parameters only, so when they fit the wave's 64 lanes they are ONE pass: one trip through
    // resolve_draw (≈ 180 vector instructions of exact float division, whatever the number of draws) instead of three.
    const int bunny_lane = kPuffs + n_draw;
    if (bunny_lane + 4 <= 64) {
        const bool is_puff = lane < kPuffs, is_draw = lane >= kPuffs && lane < bunny_lane, is_bunny = lane == bunny_lane;
        const int hud = lane - bunny_lane;  // 1, 2, 3: circle, needle, bar
        const bool is_hud = hud >= 1 && hud <= 3;
        int id = 0;
        if (is_draw) id = DRW(s, lane - kPuffs, env);
        const float avx = SF(s, F_AVX, env), phase = SF(s, F_APHASE, env);
        const bool ground = (sflags & kFlagGround) != 0;
        int bunny_tex;
        float agent_scale = 0.5f, off_x = 0.0f, off_y = 0.2f;
        if (fabsf(avx) < 0.01f && ground) {
            bunny_tex = kTexStand;
        } else if (!ground) {
            bunny_tex = kTexJump;
            agent_scale = 0.6f;
            off_x = -0.05f;
            off_y = 0.25f;
        } else if (phase > 0.5f) {
            bunny_tex = kTexWalk2;
        } else {
            bunny_tex = kTexWalk1;
        }
        int want_tex = kTexPuff;
        if (is_draw) want_tex = id == 0 ? kTexCarrot : kTexSpike;
        if (is_bunny) want_tex = bunny_tex;
        if (is_hud) want_tex = kTexCircle + (hud - 1);
        const int4 d = descs.at(want_tex);
        bool has = false, go = false, flip = false;
        float wx = 0.0f, wy = 0.0f, scale = 1.0f, alpha = 1.0f;
        if (is_puff) {
            if (puff_life > 0.0f) {
                const float lifespan = 5.0f;
                const float life_ratio = (lifespan - puff_life) / lifespan;
                alpha = 0.5f * (1.0f - life_ratio);
                const float size = 0.45f * (0.4f * life_ratio + 0.6f);
                const float offset_y = -life_ratio * 0.17f;
                wx = puff_x * kUnitPx - 0.5f * d.y * size;
                wy = (puff_y + offset_y) * kUnitPx - 0.5f * d.z * size;
                scale = size * kUnitPx / d.y;
                go = true;
            }
        } else if (is_draw) {
            float sc;
            if (id == 0) {
                sc = 1.0f * 1.0f;
                wx = (SF(s, F_GX, env) + -0.5f) * kUnitPx;
                wy = (SF(s, F_GY, env) + -0.5f) * kUnitPx;
            } else {
                const int cell = SPK(s, id - 2, env);
                sc = 1.0f * 0.4f;
                wx = (cell_x(cell) + -0.25f) * kUnitPx;
                wy = (cell_y(cell) + -0.25f) * kUnitPx;
            }
            scale = sc * kUnitPx / d.y;
            go = true;
        } else if (is_bunny) {
            const float px = SF(s, F_AX, env) - 0.25f, py = SF(s, F_AY, env) - 1.0f;
            wx = (px + off_x) * kUnitPx;
            wy = (py + off_y) * kUnitPx;
            scale = kUnitPx / d.y * agent_scale;
            flip = (sflags & kFlagForward) == 0;
            go = true;
        }
        if (go) has = resolve_draw(cam, d.y, d.z, d.x, wx, wy, scale, alpha, flip, false, mine);
        if (is_hud) {  // circle, needle and bar differ in their parameters only (store_compass worked them out)
            const float width = 64.0f, compass_size = 200.0f, offset_x = -32.0f, offset_y = 32.0f;
            float sx, sy, sw, sh;
            int sn = 0, cs = 0;
            if (hud == 1) {
                sx = width - compass_size * game_zoom + offset_x * game_zoom;
                sy = offset_y * game_zoom;
                sw = compass_size * game_zoom;
                sh = compass_size * game_zoom;
            } else if (hud == 2) {
                sx = SF(s, F_NEEDLE_X, env);
                sy = SF(s, F_NEEDLE_Y, env);
                sw = compass_size * 0.5f * game_zoom;
                sh = compass_size * 0.1f * game_zoom;
                sn = SI(s, I_NEEDLE_SN, env);
                cs = SI(s, I_NEEDLE_CS, env);
            } else {
                sx = width - compass_size * game_zoom + offset_x * game_zoom;
                sy = compass_size * game_zoom + offset_y * game_zoom;
                sw = SF(s, F_BAR_W, env);
                sh = compass_size * 0.15f * game_zoom;
            }
            has = resolve_screen_at(d.y, d.z, d.x, sx, sy, sw, sh, sn, cs, mine);
        }
        const int row_lo = half * (kObsH / halves), row_hi = (half + 1) * (kObsH / halves);
        if (s.hud_image != 0u) {
            // the ring is the same 60×60 pixels in every frame: prepared once (pg_render.h overlay_rows), in its place in
            // the draw order — after the bunny, before the needle and the bar
            wave_replay_rows(fb, atlas, mine, __ballot(has && lane <= bunny_lane), lane, row_lo, row_hi);
            if (!PG_ABL(flags, 0x10000))  // (traffic experiment, -DPG_ABLATE builds only: no compass ring)
                overlay_rows(fb, atlas.texels + s.hud_image, reinterpret_cast<const uint2*>(atlas.texels + s.hud_list), lane, row_lo);
            // (timing experiments, -DPG_ABLATE builds only: 0x20000 no needle, 0x40000 no bar)
            if (PG_ABL(flags, 0x20000)) has = has && lane != bunny_lane + 2;
            if (PG_ABL(flags, 0x40000)) has = has && lane != bunny_lane + 3;
            wave_replay_rows(fb, atlas, mine, __ballot(has && lane >= bunny_lane + 2), lane, row_lo, row_hi);
        } else {
            wave_replay_rows(fb, atlas, mine, __ballot(has), lane, row_lo, row_hi);
        }
    } else {  // (more spikes than a wave has lanes for: a pass per kind)
        {  // System_Particles::render (common_systems.cpp:285-308)
            const int4 d = descs.uniform(kTexPuff);
            bool has = false;
            if (lane < kPuffs && puff_life > 0.0f) {
                const float lifespan = 5.0f;
                const float life_ratio = (lifespan - puff_life) / lifespan;
                const float alpha = 0.5f * (1.0f - life_ratio);
                const float scale = 0.45f * (0.4f * life_ratio + 0.6f);
                const float offset_y = -life_ratio * 0.17f;
                has = resolve_draw(cam, d.y, d.z, d.x, puff_x * kUnitPx - 0.5f * d.y * scale,
                                   (puff_y + offset_y) * kUnitPx - 0.5f * d.z * scale, scale * kUnitPx / d.y, alpha, false,
                                   false, mine);
            }
            wave_replay_rows(fb, atlas, mine, __ballot(has), lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
        }
        // positive-z sprites (common_systems.cpp:26-48): carrot and spikes in draw order
        for (int first = 0; first < n_draw; first += 64) {
            const int k = first + lane;
            bool has = false;
            int id = 0;
            if (k < n_draw) {
                id = DRW(s, k, env);
                has = true;
            }
            const int4 d = descs.at(id == 0 ? kTexCarrot : kTexSpike);
            if (has) {  // carrot and spikes differ in their parameters only: pick per lane, resolve once
                float wx, wy, scale;
                if (id == 0) {
                    scale = 1.0f * 1.0f;
                    wx = (SF(s, F_GX, env) + -0.5f) * kUnitPx;
                    wy = (SF(s, F_GY, env) + -0.5f) * kUnitPx;
                } else {
                    const int cell = SPK(s, id - 2, env);
                    scale = 1.0f * 0.4f;
                    wx = (cell_x(cell) + -0.25f) * kUnitPx;
                    wy = (cell_y(cell) + -0.25f) * kUnitPx;
                }
                has = resolve_draw(cam, d.y, d.z, d.x, wx, wy, scale * kUnitPx / d.y, 1.0f, false, false, mine);
            }
            wave_replay_rows(fb, atlas, mine, __ballot(has), lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
        }
        {  // lane 0: the bunny (common_systems.cpp:204-247); lanes 1-3: the compass (jumper.cpp:473-509)
            const float avx = SF(s, F_AVX, env), phase = SF(s, F_APHASE, env);
            const bool ground = (sflags & kFlagGround) != 0;
            int want_tex;
            float agent_scale = 0.5f, off_x = 0.0f, off_y = 0.2f;
            if (fabsf(avx) < 0.01f && ground) {
                want_tex = kTexStand;
            } else if (!ground) {
                want_tex = kTexJump;
                agent_scale = 0.6f;
                off_x = -0.05f;
                off_y = 0.25f;
            } else if (phase > 0.5f) {
                want_tex = kTexWalk2;
            } else {
                want_tex = kTexWalk1;
            }
            if (lane >= 1 && lane <= 3) want_tex = kTexCircle + (lane - 1);
            const int4 d = descs.at(want_tex);
            bool has = false;
            if (lane == 0) {
                const float px = SF(s, F_AX, env) - 0.25f, py = SF(s, F_AY, env) - 1.0f;
                has = resolve_draw(cam, d.y, d.z, d.x, (px + off_x) * kUnitPx, (py + off_y) * kUnitPx,
                                   kUnitPx / d.y * agent_scale, 1.0f, (sflags & kFlagForward) == 0, false, mine);
            } else if (lane <= 3) {
                const float width = 64.0f, compass_size = 200.0f, offset_x = -32.0f, offset_y = 32.0f;
                const float tx = SF(s, F_TOGX, env), ty = SF(s, F_TOGY, env);
                const float angle = static_cast<float>(at_atan2f(ty, tx) * 180.0f / 3.14159265358979323846);
                const float dist = __fsqrt_rn(tx * tx + ty * ty);
                const float dist_inv = 1.0f / fmaxf(0.0001f, dist);
                const float dir_x = tx * dist_inv, dir_y = ty * dist_inv;
                const float ratio = fminf(1.0f, dist / (W * 1.414f));
                // circle, needle and bar differ in their parameters only: pick per lane, resolve once
                float sx, sy, sw, sh;
                double deg = 0.0;
                if (lane == 1) {
                    sx = width - compass_size * game_zoom + offset_x * game_zoom;
                    sy = offset_y * game_zoom;
                    sw = compass_size * game_zoom;
                    sh = compass_size * game_zoom;
                } else if (lane == 2) {
                    float dx = width - compass_size * 0.75f * game_zoom + offset_x * game_zoom;
                    float dy = compass_size * 0.5f * game_zoom + offset_y * game_zoom;
                    dx += compass_size * 0.25f * dir_x * game_zoom;
                    dy += compass_size * 0.25f * dir_y * game_zoom;
                    sx = dx;
                    sy = dy;
                    sw = compass_size * 0.5f * game_zoom;
                    sh = compass_size * 0.1f * game_zoom;
                    deg = static_cast<double>(angle);
                } else {
                    sx = width - compass_size * game_zoom + offset_x * game_zoom;
                    sy = compass_size * game_zoom + offset_y * game_zoom;
                    sw = compass_size * game_zoom * ratio;
                    sh = compass_size * 0.15f * game_zoom;
                }
                has = resolve_screen(d.y, d.z, d.x, sx, sy, sw, sh, deg, mine);
            }
            const int row_lo = half * (kObsH / halves), row_hi = (half + 1) * (kObsH / halves);
            if (s.hud_image != 0u) {
                // the ring is the same 60×60 pixels in every frame: prepared once (pg_render.h overlay_rows), in its place in
                // the draw order — after the bunny, before the needle and the bar
                wave_replay_rows(fb, atlas, mine, __ballot(has && lane == 0), lane, row_lo, row_hi);
                overlay_rows(fb, atlas.texels + s.hud_image, reinterpret_cast<const uint2*>(atlas.texels + s.hud_list), lane, row_lo);
                wave_replay_rows(fb, atlas, mine, __ballot(has && lane >= 2), lane, row_lo, row_hi);
            } else {
                wave_replay_rows(fb, atlas, mine, __ballot(has), lane, row_lo, row_hi);
            }
        }
    }
    // each wave stores the rows it owns (pg_render.h wave_replay_rows): no barrier
    wave_store_rows(fb, io.obs + size_t(env) * kObsBytes, lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
}

// ------------------------------------------------------------------------------------------------
// The render pre-pass (pg_prepass.h; coinrun.hip's setup_kernel is the commented model): tile spans, per-pixel
// candidates, the cell table and the resolved, culled draws of kPrepEnvs envs per workgroup.
// Reference arithmetic moved here unchanged: renderer.cpp:5-82, tilemap.cpp:255-280 (the window),
// common_systems.cpp:26-48,204-247,285-308 (sprites, bunny, particles).  The compass stays with the render wave: its
// ring is a prepared overlay, needle and bar are raw screen-space draws without a division (jumper.cpp:473-509).
// ------------------------------------------------------------------------------------------------
constexpr int kPrepEnvs = 8, kPrepThreads = 256;
enum { GW_NEEDLE_X = 0, GW_NEEDLE_Y, GW_BAR_W, GW_NEEDLE_SN, GW_NEEDLE_CS, GW_TOUCH };  // PM_GAME words (GW_TOUCH: a kept draw reaches into the compass disc's box, pg_prepass.h PrepDrawPass::touch)

struct PrepEnv {
    int32_t sflags, n_draw;
    float avx, aphase, ax, ay, gx, gy;
};
struct SetupLds {
    PrepLds<kGrid, kPrepEnvs, kMaxSpan> P;
    PrepEnv env[kPrepEnvs];
    int4 desc[kTexCount];
    uint32_t draw_ids[kPrepEnvs][kSpikeSlots / 4];    // State::draw of every env …
    uint32_t spike_cells[kPrepEnvs][kSpikeSlots / 2];  // … and State::spike_cell: fetched before anything needs them
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

    // ---- one memory round trip: descriptor table, the envs' scalars (lane = env), their draw lists and spike cells
    for (int q = tid; q < kPrepEnvs * 2 * 64; q += kPrepThreads) (&P.cover[0][0][0])[q] = 0u;
    if (tid < kTexCount) S.desc[tid] = atlas.desc[tid];
    static_assert(kTexCount <= kPrepThreads, "one descriptor per thread");
    static_assert(kPrepEnvs * kSpikeSlots / 4 == kPrepThreads, "one word of the draw lists per thread");
    {
        const int e = tid / (kSpikeSlots / 4), w = tid - e * (kSpikeSlots / 4);
        if (env0 + e < s.n) {
            S.draw_ids[e][w] = reinterpret_cast<const uint32_t*>(s.draw + size_t(env0 + e) * kSpikeSlots)[w];
            const uint32_t* cells = reinterpret_cast<const uint32_t*>(s.spike_cell + size_t(env0 + e) * kSpikeSlots);
            S.spike_cells[e][2 * w] = cells[2 * w];
            S.spike_cells[e][2 * w + 1] = cells[2 * w + 1];
        }
    }
    Camera cam{};
    int themes = 0;
    float bgshift = 0.0f;
    bool active = false;
    uint32_t game_words[5] = {0u, 0u, 0u, 0u, 0u};
    if (tid < kPrepEnvs) {
        const int e = tid, env = env0 + e;
        active = env < s.n && (!mask || mask[env]);
        if (active) {
            cam = Camera{SF(s, F_CAMX, env), SF(s, F_CAMY, env), 64.0f, 64.0f, kObsZoom * 64.0f / 64.0f};
            themes = SI(s, I_THEMES, env);
            bgshift = SF(s, F_BGSHIFT, env);
            PrepEnv pe{};
            pe.sflags = SI(s, I_FLAGS, env);
            pe.n_draw = (pe.sflags & kFlagListed) ? SI(s, I_NSPIKES, env) + 1 : 0;  // empty right after a reset
            pe.avx = SF(s, F_AVX, env);
            pe.aphase = SF(s, F_APHASE, env);
            pe.ax = SF(s, F_AX, env);
            pe.ay = SF(s, F_AY, env);
            pe.gx = SF(s, F_GX, env);
            pe.gy = SF(s, F_GY, env);
            S.env[e] = pe;
            game_words[GW_NEEDLE_X] = __float_as_uint(SF(s, F_NEEDLE_X, env));
            game_words[GW_NEEDLE_Y] = __float_as_uint(SF(s, F_NEEDLE_Y, env));
            game_words[GW_BAR_W] = __float_as_uint(SF(s, F_BAR_W, env));
            game_words[GW_NEEDLE_SN] = static_cast<uint32_t>(SI(s, I_NEEDLE_SN, env));
            game_words[GW_NEEDLE_CS] = static_cast<uint32_t>(SI(s, I_NEEDLE_CS, env));
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
            // (bit 4: traffic experiment of the -DPG_ABLATE build — every env shows backdrop 9: what the backdrops' share of
            // the render kernel's FETCH_SIZE is, and what the kernel would gain if it were not there)
            const int backdrop = PG_ABL(flags, 16) ? 9 : (themes & 0xff), theme = (themes >> 8) & 0xff;
            const int4 d = S.desc[kTexBackdrop + backdrop];
            const float aspect = static_cast<float>(d.y) / static_cast<float>(d.z);
            const float extra = aspect - 1.0f;
            v.bg = BgDraw{d, -bgshift * extra, 0.0f, 64.0f * kUnitPx / d.z};  // jumper.cpp:459-464
            const float vx = (cam.px - cam.sw * 0.5f / cam.scale) * kPxUnit;  // tilemap.cpp:255-264
            const float vy = (cam.py - cam.sh * 0.5f / cam.scale) * kPxUnit;
            const float vw = cam.sw * kPxUnit / cam.scale, vh = cam.sh * kPxUnit / cam.scale;
            v.x0 = static_cast<int>(floorf(vx));
            v.y0 = static_cast<int>(floorf(vy));
            v.cols = static_cast<int>(ceilf(vx + vw)) - v.x0 + 1;
            v.rows = static_cast<int>(ceilf(vy + vh)) - v.y0 + 1;
            const int4 top_d = S.desc[kTexTop + theme], mid_d = S.desc[kTexMid + theme];
            const bool two = top_d.z != mid_d.z;  // the brown cap tile is 64×53 (see climber.hip)
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
#pragma unroll
            for (int k = 0; k < 5; k++) P.meta[e][PM_GAME + k] = game_words[k];
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
    prep_spans<kGrid, kMaxSpan, kPrepEnvs, true>(P, tid, kPrepThreads);
    // … then sixteen kind bytes: wall_top → 0, wall_mid → 1, everything else (empty, spike) no tile
    if (cell_lane) {
        uint32_t in_rows[4];
        prep_column_rows<kGrid>(column, S.row_valid[cell_e], cell_x_ok, kWallMid, in_rows);  // out of bounds is a wall (tilemap.h:84-89)
        uint32_t kinds[4];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const uint32_t t = in_rows[w] & 0x07070707u;                  // kEmpty 0, kWallTop 1, kWallMid 2, kSpike 3
            const uint32_t lo = t & 0x01010101u, hi = (t >> 1) & 0x01010101u;
            const uint32_t wall = (lo ^ hi) * 0xffu;                      // 0xff where t is 1 or 2
            kinds[w] = (hi & wall) | ~wall;                               // kind = t - 1 for walls; 0xff: no tile
        }
        prep_column_store<kGrid>(out.cells + size_t(env0 + cell_e) * (kGrid * kGrid), cell_c, kinds);
    }
    __syncthreads();
    prep_axes<kGrid, kMaxSpan, kPrepEnvs>(P, out, env0, wave, kPrepThreads / 64, lane);

    // ---- the draws in the reference's order: particles (common_systems.cpp:285-308), the positive-z sprites (carrot and
    // spikes, :26-48), the bunny (:204-247).  Two envs per wavefront, cull first (pg_prepass.h prep_draws_pass).
    static_assert(kPrepEnvs == 2 * (kPrepThreads / 64), "two envs per wavefront");
    {
        const int ea = 2 * wave, eb = 2 * wave + 1;
        const bool on_a = P.view[ea].active != 0, on_b = P.view[eb].active != 0;
        const Camera cam_a = P.view[ea].cam, cam_b = P.view[eb].cam;
        const int cnt_a = on_a ? kPuffs + S.env[ea].n_draw + 1 : 0, cnt_b = on_b ? kPuffs + S.env[eb].n_draw + 1 : 0;
        uint32_t* const draws_a = out.draws + size_t(env0 + ea) * kPrepDraws * kBlitWords;
        uint32_t* const draws_b = out.draws + size_t(env0 + eb) * kPrepDraws * kBlitWords;
        PrepDrawPass st{0, {0, 0}, {0, 0}};
        PrepDrawQueue& Q = S.queue[wave];
        // everything drawn here lies beneath the compass ring, which is opaque over two thirds of the frame: what lands
        // wholly under it — the bunny, nearly always — is not handed to the render wave at all
        const uint32_t* const cover = (s.hud_cover != 0u && !PG_ABL(flags, 0x10000)) ? atlas.texels + s.hud_cover : nullptr;
        for (int base = 0; base < cnt_a + cnt_b; base += 64) {  // wave-uniform
            const int q = base + lane;
            const bool is_b = q >= cnt_a;
            const int e = is_b ? eb : ea, env = env0 + e;
            const int slot = is_b ? q - cnt_a : q;
            const bool valid = q < cnt_a + cnt_b;
            const PrepEnv& pe = S.env[e];
            PrepDraw p{false, false, false, kTexPuff, 0.0f, 0.0f, 1.0f, 1.0f};
            float num = kUnitPx, post = 1.0f;  // scale = num / texture width * post (one division for every kind)
            if (valid && slot < kPuffs) {
                const float life = PF(s, PF_LIFE, slot, env), px = PF(s, PF_X, slot, env), py = PF(s, PF_Y, slot, env);
                if (life > 0.0f) {
                    const int4 d = S.desc[kTexPuff];
                    const float lifespan = 5.0f;
                    const float life_ratio = (lifespan - life) / lifespan;
                    p.alpha = 0.5f * (1.0f - life_ratio);
                    const float size = 0.45f * (0.4f * life_ratio + 0.6f);
                    const float offset_y = -life_ratio * 0.17f;
                    p.wx = px * kUnitPx - 0.5f * d.y * size;
                    p.wy = (py + offset_y) * kUnitPx - 0.5f * d.z * size;
                    num = size * kUnitPx;
                    p.go = true;
                }
            } else if (valid && slot < kPuffs + pe.n_draw) {
                const int k = slot - kPuffs;
                const int id = (S.draw_ids[e][k >> 2] >> (8 * (k & 3))) & 0xffu;
                float sc;
                if (id == 0) {
                    p.tex = kTexCarrot;
                    sc = 1.0f * 1.0f;
                    p.wx = (pe.gx + -0.5f) * kUnitPx;
                    p.wy = (pe.gy + -0.5f) * kUnitPx;
                } else {
                    const int j = id - 2;
                    const int cell = (S.spike_cells[e][j >> 1] >> (16 * (j & 1))) & 0xffffu;
                    p.tex = kTexSpike;
                    sc = 1.0f * 0.4f;
                    p.wx = (cell_x(cell) + -0.25f) * kUnitPx;
                    p.wy = (cell_y(cell) + -0.25f) * kUnitPx;
                }
                num = sc * kUnitPx;
                p.go = true;
            } else if (valid) {
                const bool ground = (pe.sflags & kFlagGround) != 0;
                float agent_scale = 0.5f, off_x = 0.0f, off_y = 0.2f;
                if (fabsf(pe.avx) < 0.01f && ground) {
                    p.tex = kTexStand;
                } else if (!ground) {
                    p.tex = kTexJump;
                    agent_scale = 0.6f;
                    off_x = -0.05f;
                    off_y = 0.25f;
                } else if (pe.aphase > 0.5f) {
                    p.tex = kTexWalk2;
                } else {
                    p.tex = kTexWalk1;
                }
                const float px = pe.ax - 0.25f, py = pe.ay - 1.0f;
                p.wx = (px + off_x) * kUnitPx;
                p.wy = (py + off_y) * kUnitPx;
                post = agent_scale;  // kUnitPx / d.y * agent_scale
                p.flip_h = (pe.sflags & kFlagForward) == 0;
                p.go = true;
            }
            p.scale = num / S.desc[p.tex].y * post;
            prep_draws_pass(Q, st, S.desc, cam_a, cam_b, draws_a, draws_b, valid, is_b, p, lane, cover);
        }
        prep_draws_flush(Q, st, S.desc, cam_a, cam_b, draws_a, draws_b, lane, cover);
        if (lane == 0) {  // (the compass's needle and bar take two more lanes of the render wave)
            S.counts[ea] = st.done[0] > kPrepDraws - 3 ? kPrepDraws + 1 : st.done[0];
            S.counts[eb] = st.done[1] > kPrepDraws - 3 ? kPrepDraws + 1 : st.done[1];
            P.meta[ea][PM_GAME + GW_TOUCH] = (cover == nullptr || st.touch[0]) ? 1u : 0u;
            P.meta[eb][PM_GAME + GW_TOUCH] = (cover == nullptr || st.touch[1]) ? 1u : 0u;
        }
    }
    __syncthreads();
    prep_meta_out<kGrid, kMaxSpan, kPrepEnvs>(P, out, env0, S.counts, tid, kPrepThreads);
    if (tid < kPrepEnvs && (P.fat[tid] != 0 || S.counts[tid] > kPrepDraws) && env0 + tid < s.n)  // (prep_meta_out's own test)
#if defined(PG_FAT_WHY)
        s.fat[1 + atomicAdd(s.fat, 1u)] = static_cast<uint32_t>(env0 + tid) | (P.fat[tid] << 24) | (S.counts[tid] > kPrepDraws ? 16u << 24 : 0u);
#else
        s.fat[1 + atomicAdd(s.fat, 1u)] = static_cast<uint32_t>(env0 + tid);
#endif
}

// The complete path as a kernel of its own: every env (`listed` = 0: the draw-list replay, kDebugNoPrepass), or the few
// frames the pre-pass could not prepare (s.fat: more than 64 visible draws, a window or a span beyond the tables — none
// in a run of the default mode), a handful of workgroups walking the list.  It is not a branch of render_kernel because
// a kernel's registers and code are those of its largest path: with the complete path inside, the lean frames ran with
// 95 registers and spills instead of 81 and none (render 0.72 -> 0.65 ms without it).
constexpr int kFatBlocks = 64;
__global__ void __launch_bounds__(128, PG_RENDER_WAVES) render_full_kernel(State s, AtlasView atlas, const uint8_t* mask, StepIO io,
                                                         int flags, int listed) {
    __shared__ alignas(16) uint32_t fb[kFbWords];
    __shared__ ComposeLds<kGrid> L;
    if (!listed) {
        const int env = blockIdx.x;
        if (mask && !mask[env]) return;
        render_full(s, atlas, io, flags, env, fb, L);
        return;
    }
    const uint32_t count = s.fat[0];
#if defined(PG_FAT_WHY)
    if (blockIdx.x == 0 && threadIdx.x == 0 && count) {
        uint32_t why[5] = {0, 0, 0, 0, 0};
        for (uint32_t k = 0; k < count; k++)
            for (int b = 0; b < 5; b++) why[b] += (s.fat[1 + k] >> (24 + b)) & 1u;
        printf("fat %u: view %u list %u span %u nest %u draws %u\n", count, why[0], why[1], why[2], why[3], why[4]);
    }
    return;
#endif
    for (uint32_t k = blockIdx.x; k < count; k += gridDim.x) {  // (workgroup-uniform)
        render_full(s, atlas, io, flags, static_cast<int>(s.fat[1 + k]), fb, L);
        __syncthreads();
    }
}

// render_game(true) (jumper.cpp:445-509): one workgroup of two wavefronts per env; a frame starts from what setup_kernel
// left (coinrun.hip's render_kernel is the commented model).
__global__ void __launch_bounds__(128, PG_RENDER_WAVES) render_kernel(State s, AtlasView atlas, const uint8_t* mask, StepIO io,
                                                    int flags) {
    const int env = blockIdx.x;
    if (env == 0 && threadIdx.x == 0) s.fat[0] = 0u;  // (render_full_kernel, launched in front of this one, has read it)
    if (mask && !mask[env]) return;
    const int lane = threadIdx.x & 63, half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int halves = 2;
    __shared__ alignas(16) uint32_t fb[kFbWords];
    __shared__ ComposeLds<kGrid> L;
    PG_TL_BEGIN(8);
    PG_TL(0);
    const PrepMeta M{s.prep.meta + size_t(env) * kPrepMetaWords};
    const uint32_t colw = s.prep.axes[size_t(env) * 128 + lane], roww = s.prep.axes[size_t(env) * 128 + 64 + lane];
    const uint32_t roww2 = s.prep.axes2[size_t(env) * 64 + lane];
    const uint32_t kind_off = M.w[PM_KINDS + (lane & (kPrepKinds - 1))];
    const uint32_t two16 = reinterpret_cast<const uint16_t*>(s.prep.cells)[size_t(env) * (kGrid * kGrid / 2) + half * 64 + lane];
    const int n_draws = M.draws();
    Blit mine = prep_draw_load(s.prep.draws + (size_t(env) * kPrepDraws + lane) * kBlitWords, lane < n_draws);
    prep_cells_expand<kGrid>(L, two16, kind_off, half, lane);
    const ComposeRegs R = prep_regs<kGrid>(M, colw, roww, roww2, lane);
    __syncthreads();  // the cell table is complete
    PG_TL(1);
    if (M.fat()) return;  // (wave-uniform; render_full_kernel has drawn it)
    const int row_lo = half * (kObsH / halves), row_hi = (half + 1) * (kObsH / halves);
    // (what the compass disc is going to cover is not fetched: pg_render.h compose_rows_from UNDER)
#ifndef PG_JUMPER_UNDER
#define PG_JUMPER_UNDER 2
#endif
    uint32_t ring_rows = 0xffffffffu;  // the rows of this wave's 32 whose covered pixels do not hold the ring's texels yet
    if (PG_JUMPER_UNDER && s.hud_image != 0u && s.hud_under != 0u) {
        const unsigned long long* under = reinterpret_cast<const unsigned long long*>(atlas.texels + s.hud_under);
        uint32_t whole = 0xffffffffu;
        if (M.flags() & 2u)
            compose_rows_from<kGrid, true, false, true>(fb, L, atlas, R, lane, flags, half, halves, under, s.hud_image * 4u, &whole);
        else
            compose_rows_from<kGrid, false, false, true>(fb, L, atlas, R, lane, flags, half, halves, under, s.hud_image * 4u, &whole);
        // … and none of the frame's draws has been near them (setup_kernel, GW_TOUCH): the ring's opaque part is there already
        if (PG_JUMPER_UNDER > 1 && M.w[PM_GAME + GW_TOUCH] == 0u) ring_rows = whole;
    } else if (M.flags() & 2u) {
        compose_rows_from<kGrid, true, false>(fb, L, atlas, R, lane, flags, half, halves);
    } else {
        compose_rows_from<kGrid, false, false>(fb, L, atlas, R, lane, flags, half, halves);
    }
    PG_TL(2);
    // the compass (jumper.cpp:473-509) behind the resolved draws: lanes n_draws, + 1, + 2 = circle, needle, bar — raw
    // screen-space draws whose parameters the logic kernel worked out (store_compass)
    const int hud = lane - n_draws + 1;  // 1, 2, 3
    bool has = lane < n_draws;
    if (hud >= 1 && hud <= 3) {
        const float game_zoom = kObsZoom;
        const float width = 64.0f, compass_size = 200.0f, offset_x = -32.0f, offset_y = 32.0f;
        const int4 d = atlas.desc[kTexCircle + (hud - 1)];
        float sx, sy, sw, sh;
        int sn = 0, cs = 0;
        if (hud == 1) {
            sx = width - compass_size * game_zoom + offset_x * game_zoom;
            sy = offset_y * game_zoom;
            sw = compass_size * game_zoom;
            sh = compass_size * game_zoom;
        } else if (hud == 2) {
            sx = __uint_as_float(M.w[PM_GAME + GW_NEEDLE_X]);
            sy = __uint_as_float(M.w[PM_GAME + GW_NEEDLE_Y]);
            sw = compass_size * 0.5f * game_zoom;
            sh = compass_size * 0.1f * game_zoom;
            sn = static_cast<int>(M.w[PM_GAME + GW_NEEDLE_SN]);
            cs = static_cast<int>(M.w[PM_GAME + GW_NEEDLE_CS]);
        } else {
            sx = width - compass_size * game_zoom + offset_x * game_zoom;
            sy = compass_size * game_zoom + offset_y * game_zoom;
            sw = __uint_as_float(M.w[PM_GAME + GW_BAR_W]);
            sh = compass_size * 0.15f * game_zoom;
        }
        has = resolve_screen_at(d.y, d.z, d.x, sx, sy, sw, sh, sn, cs, mine);
        // (the bar's rectangle starts at y = 69 of the 64-pixel observation, always: a draw that is not rotated and lies
        // wholly beyond the target reaches no pixel — raster rule S5 — and need not be replayed to find that out)
        if (!(mine.flip_mod & kRotated) && (mine.dy >= kObsH || mine.dx >= kObsW || mine.dy + mine.dh <= 0 || mine.dx + mine.dw <= 0)) has = false;
    }
    PG_TL(3);
    if (s.hud_image != 0u) {
        // the ring is the same 60×60 pixels in every frame: prepared once (pg_render.h overlay_rows), in its place in the
        // draw order — after the bunny, before the needle and the bar
        wave_replay_rows(fb, atlas, mine, __ballot(has && lane < n_draws), lane, row_lo, row_hi);
        PG_TL(4);
        overlay_rows(fb, atlas.texels + s.hud_image, reinterpret_cast<const uint2*>(atlas.texels + s.hud_list), lane, row_lo, ring_rows);
        PG_TL(5);
        // The needle (82 of this kernel's 514 µs, 74 of them its scan; five or more texels a lane in flight instead of four
        // change nothing or spill) lies in the lower half of the frame nearly always, and a frame is done when its slower
        // wavefront is: so it is the one draw the two wavefronts SHARE — every other 64 pixels of its scan each, on either
        // half's rows, between two barriers (the ring under it complete, the bar over it not begun).
        {
            const unsigned long long drawn = __ballot(has);
            __syncthreads();
            if ((drawn >> (n_draws + 1)) & 1ull) {  // (workgroup-uniform: both wavefronts resolved the same draws)
                const Blit needle = blit_from_lane(blit_pack(mine), mine, n_draws + 1);
                if (needle.flip_mod & kRotated)
                    wave_blit_rotated(fb, atlas, needle, rot_box(needle), lane + 64 * half, 64 * halves);
                else
                    wave_blit(fb, atlas, needle, lane, half, halves);
            }
            __syncthreads();
            if (drawn & (1ull << (n_draws + 2))) wave_replay_rows(fb, atlas, mine, drawn & (1ull << (n_draws + 2)), lane, row_lo, row_hi);
        }
        PG_TL(6);
    } else {
        wave_replay_rows(fb, atlas, mine, __ballot(has), lane, row_lo, row_hi);
    }
    wave_store_rows(fb, io.obs + size_t(env) * kObsBytes, lane, row_lo, row_hi);
    PG_TL_END(8, s.hud_image != 0u, io.obs + size_t(env) * kObsBytes + half * (kObsBytes / 2));
}

// cenv_render's frame (render_game(false)) for one env: pg_frame.h; the draw list of render_kernel, one draw at a time.
__global__ void __launch_bounds__(kFrameThreads) frame_kernel(State s, AtlasView atlas, int env, FrameTarget t) {
    const float fw = static_cast<float>(t.w), fh = static_cast<float>(t.h);
    const float game_zoom = 0.3f;
    FramePainter P{t, atlas, Camera{SF(s, F_CAMX, env), SF(s, F_CAMY, env), fw, fh, game_zoom * fw / 64.0f},
                   static_cast<int>(threadIdx.x), kFrameThreads};
    const int themes = SI(s, I_THEMES, env), sflags = SI(s, I_FLAGS, env);
    const int backdrop = themes & 0xff, theme = (themes >> 8) & 0xff;
    const int n_draw = (sflags & kFlagListed) ? SI(s, I_NSPIKES, env) + 1 : 0;
    const uint8_t* tiles = s.tiles + size_t(env) * kTileStride;
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
            if (!is_wall(tile)) continue;
            const int tex = (tile == kWallTop ? kTexTop : kTexMid) + theme;
            P.draw(tex, x * kUnitPx, y * kUnitPx, kUnitPx / P.desc(tex).y);
        }
    for (int k = 0; k < kPuffs; k++) {
        const float life = PF(s, PF_LIFE, k, env);
        if (life <= 0.0f) continue;
        const int4 d = P.desc(kTexPuff);
        const float lifespan = 5.0f;
        const float life_ratio = (lifespan - life) / lifespan;
        const float alpha = 0.5f * (1.0f - life_ratio);
        const float scale = 0.45f * (0.4f * life_ratio + 0.6f);
        const float offset_y = -life_ratio * 0.17f;
        P.draw(kTexPuff, PF(s, PF_X, k, env) * kUnitPx - 0.5f * d.y * scale,
               (PF(s, PF_Y, k, env) + offset_y) * kUnitPx - 0.5f * d.z * scale, scale * kUnitPx / d.y, alpha);
    }
    for (int k = 0; k < n_draw; k++) {
        const int id = DRW(s, k, env);
        if (id == 0) {
            const float scale = 1.0f * 1.0f;
            P.draw(kTexCarrot, (SF(s, F_GX, env) + -0.5f) * kUnitPx, (SF(s, F_GY, env) + -0.5f) * kUnitPx,
                   scale * kUnitPx / P.desc(kTexCarrot).y);
        } else {
            const int cell = SPK(s, id - 2, env);
            const float scale = 1.0f * 0.4f;
            P.draw(kTexSpike, (cell_x(cell) + -0.25f) * kUnitPx, (cell_y(cell) + -0.25f) * kUnitPx,
                   scale * kUnitPx / P.desc(kTexSpike).y);
        }
    }
    {
        const float avx = SF(s, F_AVX, env), phase = SF(s, F_APHASE, env);
        const bool ground = (sflags & kFlagGround) != 0;
        int tex;
        float agent_scale = 0.5f, off_x = 0.0f, off_y = 0.2f;
        if (fabsf(avx) < 0.01f && ground) {
            tex = kTexStand;
        } else if (!ground) {
            tex = kTexJump;
            agent_scale = 0.6f;
            off_x = -0.05f;
            off_y = 0.25f;
        } else if (phase > 0.5f) {
            tex = kTexWalk2;
        } else {
            tex = kTexWalk1;
        }
        const float px = SF(s, F_AX, env) - 0.25f, py = SF(s, F_AY, env) - 1.0f;
        P.draw(tex, (px + off_x) * kUnitPx, (py + off_y) * kUnitPx, kUnitPx / P.desc(tex).y * agent_scale, 1.0f,
               (sflags & kFlagForward) == 0);
    }
    {   // compass (jumper.cpp:473-509): sized by the base zoom, not by the window
        const float width = fw, compass_size = 200.0f, offset_x = -32.0f, offset_y = 32.0f;
        const float tx = SF(s, F_TOGX, env), ty = SF(s, F_TOGY, env);
        const float angle = static_cast<float>(at_atan2f(ty, tx) * 180.0f / 3.14159265358979323846);
        const float dist = __fsqrt_rn(tx * tx + ty * ty);
        const float dist_inv = 1.0f / fmaxf(0.0001f, dist);
        const float dir_x = tx * dist_inv, dir_y = ty * dist_inv;
        const float ratio = fminf(1.0f, dist / (W * 1.414f));
        P.screen(kTexCircle, width - compass_size * game_zoom + offset_x * game_zoom, offset_y * game_zoom,
                 compass_size * game_zoom, compass_size * game_zoom, 0.0);
        float dx = width - compass_size * 0.75f * game_zoom + offset_x * game_zoom;
        float dy = compass_size * 0.5f * game_zoom + offset_y * game_zoom;
        dx += compass_size * 0.25f * dir_x * game_zoom;
        dy += compass_size * 0.25f * dir_y * game_zoom;
        P.screen(kTexNeedle, dx, dy, compass_size * 0.5f * game_zoom, compass_size * 0.1f * game_zoom,
                 static_cast<double>(angle));
        P.screen(kTexBar, width - compass_size * game_zoom + offset_x * game_zoom,
                 compass_size * game_zoom + offset_y * game_zoom, compass_size * game_zoom * ratio,
                 compass_size * 0.15f * game_zoom, 0.0);
    }
}

class JumperGame final : public Game {
   public:
    const char* name() const override { return "jumper"; }
    std::vector<std::string> texture_names() const override {
        std::vector<std::string> v;
        for (const char* t : {"tileBlue_05", "tileGreen_05", "tileYellow_06", "tileBrown_06", "tileBlue_08",
                              "tileGreen_08", "tileYellow_09", "tileBrown_09"})
            v.push_back(std::string("platformer/") + t + ".png");
        for (const char* t : {"spikeMan_stand", "carrot", "bunny2_ready", "bunny2_jump", "bunny2_walk1", "bunny2_walk2",
                              "iconCircle_white"})
            v.push_back(std::string("misc_assets/") + t + ".png");
        for (const char* t : {"jumper_compass_circle", "jumper_compass_needle", "jumper_compass_bar"})
            v.push_back(std::string("custom/") + t + ".png");
        for (const char* b :
             {"alien_bg", "another_world_bg", "back_cave", "caverns", "cyberpunk_bg", "parallax_forest", "scifi_bg",
              "scifi2_bg", "living_tissue_bg", "airadventurelevel1", "airadventurelevel2", "airadventurelevel3",
              "airadventurelevel4", "cave_background", "blue_desert", "blue_grass", "blue_land", "blue_shroom",
              "colored_desert", "colored_grass", "colored_land", "colored_shroom", "landscape1", "landscape2",
              "landscape3", "landscape4", "battleback1", "battleback2", "battleback3", "battleback4", "battleback5",
              "battleback6", "battleback7", "battleback8", "battleback9", "battleback10", "sunrise"})
            v.push_back(std::string("platform_backgrounds/") + b + ".png");
        for (const char* b : {"beach1", "beach2", "beach3", "beach4", "fantasy1", "fantasy2", "fantasy3", "fantasy4",
                              "candy1", "candy2", "candy3", "candy4"})
            v.push_back(std::string("platform_backgrounds_2/") + b + ".png");
        return v;
    }
    std::string check_atlas(const std::vector<std::pair<int, int>>& sizes) const override {
        return static_cast<int>(sizes.size()) == kTexCount ? "" : "jumper: unexpected texture count";
    }
    static size_t align256(size_t x) { return (x + 255) & ~size_t(255); }
    struct Layout {
        size_t shadow, slot, mt, tiles, f, i, pf, spike, draw, total;
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
        l.f = take(size_t(F_COUNT) * n * 4);
        l.i = take(size_t(I_COUNT) * n * 4);
        l.pf = take(size_t(PF_COUNT) * kPuffSlots * n * 4);
        l.spike = take(size_t(kSpikeSlots) * n * 2);
        l.draw = take(size_t(kSpikeSlots) * n);
        l.total = off;
        return l;
    }
    size_t state_bytes(int n) const override { return layout(n).total; }
    // The compass ring (jumper.cpp:485-489: SDL_RenderTextureRotated(circle, NULL, &dst, 0°) at a fixed screen rectangle,
    // drawn for the observation too, D17) lands on the same pixels with the same texels in every frame: sample it once
    // here, under raster spec S1-S3, for the 64×64 observation (the human-size frame keeps the general path).
    void extend_atlas(Atlas& atlas) override {
        const int4 d = atlas.desc_host(kTexCircle);
        const uint32_t* tex = atlas.texels_host(kTexCircle);
        const float game_zoom = 0.3f, width = 64.0f, compass_size = 200.0f, offset_x = -32.0f, offset_y = 32.0f;
        const float fx = width - compass_size * game_zoom + offset_x * game_zoom, fy = offset_y * game_zoom;
        const float fw = compass_size * game_zoom, fh = compass_size * game_zoom;
        // Whenever the ring cannot be prepared the frames stay right — the kernels draw it as the 60 × 60 blit it is — but a
        // quarter slower, and the draws it hides are no longer dropped: say so once instead of falling back in silence.
        auto unprepared = [](const char* why) {
            std::fprintf(stderr, "procgen2_amd: jumper: compass ring not prepared (%s): drawn as a plain blit, ≈ 25 %% slower\n", why);
        };
        if (!(fw >= 1.0f && fh >= 1.0f && fw < 32768.0f && fh < 32768.0f)) return unprepared("degenerate size");  // resolve_screen's S1
        if (!(fx > -32768.0f && fx < 32768.0f && fy > -32768.0f && fy < 32768.0f)) return unprepared("degenerate place");
        const int dx = static_cast<int>(fx), dy = static_cast<int>(fy), dw = static_cast<int>(fw), dh = static_cast<int>(fh);
        std::vector<uint32_t> image(size_t(kObsW) * kObsH, 0u), list;
        const size_t half_words = size_t(kOverlayPerLane) * 64 * 2;  // (pg_render.h overlay_rows: a half's share of the list)
        for (int half = 0; half < 2; half++) {
            for (int y = half * (kObsH / 2); y < (half + 1) * (kObsH / 2); y++)
                for (int x = 0; x < kObsW; x++) {
                    const int i = x - dx, j = y - dy;
                    if (i < 0 || j < 0 || i >= dw || j >= dh) continue;
                    const uint32_t t = tex[sample_index(0, d.z, j, dh) * d.y + sample_index(0, d.y, i, dw)];
                    const uint32_t a = t >> 24;
                    if (a == 255u) {
                        image[size_t(y) * kObsW + x] = t;
                    } else if (a != 0u) {
                        list.push_back(static_cast<uint32_t>(y * kObsW + x));
                        list.push_back(t);
                    }
                }
            if (list.size() > (half + 1) * half_words)  // more translucent texels than the overlay's lanes take: the blit stays
                return unprepared("more translucent texels in a half frame than pg_render.h kOverlayPerLane allows");
            while (list.size() < (half + 1) * half_words) {
                list.push_back(0xffffffffu);
                list.push_back(0u);
            }
        }
        // the columns each row's OPAQUE texels cover, for the pre-pass to drop what lies wholly beneath (pg_prepass.h
        // `cover`): one run per row, starts valley-shaped and ends hill-shaped over the rows — else no table
        std::vector<uint32_t> cover(kObsH, 0x00ffu);
        bool shaped = true;
        int falling_lo = 1, rising_hi = 1, prev_lo = 256, prev_hi = -1, seen = 0, ended = 0;
        for (int y = 0; y < kObsH; y++) {
            int lo = -1, hi = -1, runs = 0;
            for (int x = 0; x < kObsW; x++) {
                const bool op = image[size_t(y) * kObsW + x] != 0u;
                if (op && (x == 0 || image[size_t(y) * kObsW + x - 1] == 0u)) runs++, lo = lo < 0 ? x : lo;
                if (op) hi = x;
            }
            if (runs == 0) {
                ended = seen;
                continue;
            }
            if (runs > 1 || ended) shaped = false;
            if (seen) {
                if (lo > prev_lo) falling_lo = 0;
                else if (lo < prev_lo && !falling_lo) shaped = false;
                if (hi < prev_hi) rising_hi = 0;
                else if (hi > prev_hi && !rising_hi) shaped = false;
            }
            seen = 1, prev_lo = lo, prev_hi = hi;
            cover[y] = static_cast<uint32_t>(lo) | (static_cast<uint32_t>(hi) << 8);
        }
        {   // word kObsH: the bounding box of everything opaque (pg_prepass.h PrepDrawPass::touch)
            int bx0 = 255, bx1 = 0, by0 = 255, by1 = 0;
            for (int y = 0; y < kObsH; y++)
                for (int x = 0; x < kObsW; x++)
                    if (image[size_t(y) * kObsW + x] != 0u) {
                        bx0 = x < bx0 ? x : bx0, bx1 = x > bx1 ? x : bx1;
                        by0 = y < by0 ? y : by0, by1 = y > by1 ? y : by1;
                    }
            cover.push_back(static_cast<uint32_t>(bx0) | static_cast<uint32_t>(bx1) << 8 | static_cast<uint32_t>(by0) << 16 | static_cast<uint32_t>(by1) << 24);
        }
        if ((atlas.texel_bytes() / 4) % 2 != 0) atlas.append_words({0u});  // the list is read as 8-byte pairs
        s_.hud_cover = shaped && seen ? atlas.append_words(cover) : 0u;
        if ((atlas.texel_bytes() / 4) % 2 != 0) atlas.append_words({0u});
        s_.hud_image = atlas.append_words(image);
        s_.hud_list = atlas.append_words(list);
        // … and the pixels its opaque texels cover, a 64-bit mask per pixel row (pg_render.h compose_rows_from UNDER)
        std::vector<uint32_t> under(size_t(kObsH) * 2, 0u);
        for (int y = 0; y < kObsH; y++)
            for (int x = 0; x < kObsW; x++)
                if ((image[size_t(y) * kObsW + x] >> 24) == 255u) under[size_t(y) * 2 + (x >> 5)] |= 1u << (x & 31);
        if ((atlas.texel_bytes() / 4) % 2 != 0) atlas.append_words({0u});
        s_.hud_under = atlas.append_words(under);
    }
    bool set_game_flags(uint32_t flags) override {  // include/procgen2_vec.h PGV_JUMPER_FLOAT_ABS
        s_.float_abs = (flags & PGV_JUMPER_FLOAT_ABS) ? 1 : 0;
        return (flags & ~PGV_JUMPER_FLOAT_ABS) == 0;
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
        s_.pf = reinterpret_cast<float*>(p + l.pf);
        s_.spike_cell = reinterpret_cast<uint16_t*>(p + l.spike);
        s_.draw = p + l.draw;
        atlas_ = atlas;
    }
    int blocks() const { return (s_.n + 63) / 64; }
    int prefetch() const { return (debug_flags & kDebugNoPrefetch) ? 0 : 1; }
    void launch_make(hipStream_t st, uint32_t seed_base, int env_offset) override {
        hipLaunchKernelGGL(make_kernel, dim3(blocks()), dim3(64), 0, st, s_);
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
    void launch_logic(hipStream_t st, const int32_t* actions, uint32_t run_seed, uint32_t step_index, int env_offset,
                      StepIO io) override {
        // prefetched levels are installed beside the logic (its second row of blocks); the level kernel behind it
        // generates, synchronously, the levels that were not ready — none in steady state (pg_prefetch.h install_prefetched)
        const bool fused = prefetch() != 0;
        if (!fused) LevelLaunch<Gen>::auto_reset(st, s_, prefetch(), io, plan, PG_RESET_SPAN, reset_served_mark(step_index), reset_due_mark(step_index));
        hipLaunchKernelGGL(logic_kernel, dim3((s_.n + 63) / 64, fused ? 2 : 1), dim3(64), 0, st, s_, actions, run_seed, step_index,
                           env_offset, io, prefetch(), plan);
        if (fused) LevelLaunch<Gen>::auto_reset(st, s_, prefetch(), io, plan, PG_RESET_SPAN, reset_served_mark(step_index), reset_due_mark(step_index));
    }
    bool launch_frame(hipStream_t st, int env, uint32_t* d_px, int w, int h) override {
        hipLaunchKernelGGL(frame_kernel, dim3(1), dim3(kFrameThreads), 0, st, s_, atlas_, env, FrameTarget{d_px, w, h});
        return true;
    }
    size_t scratch_bytes(int n) const override { return prep_bytes(n, kGrid, kBlitWords, true) + size_t(n + 1) * 4; }
    void bind_scratch(void* d_scratch, int n) override {
        s_.prep = prep_bind(d_scratch, n, kGrid, kBlitWords, true);
        s_.fat = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(d_scratch) + prep_bytes(n, kGrid, kBlitWords, true));  // (zeroed by the engine)
    }
    bool lean() const { return !(debug_flags & (1 | kDebugNoPrepass)); }
    void launch_prepass(hipStream_t st, const uint8_t* mask) override {
        if (lean()) hipLaunchKernelGGL(setup_kernel, dim3((s_.n + kPrepEnvs - 1) / kPrepEnvs), dim3(kPrepThreads), 0, st, s_, atlas_, mask, debug_flags);
    }
    void launch_render(hipStream_t st, const uint8_t* mask, StepIO io) override {
        if (lean()) {
            hipLaunchKernelGGL(render_full_kernel, dim3(kFatBlocks), dim3(128), 0, st, s_, atlas_, mask, io, debug_flags, 1);
            hipLaunchKernelGGL(render_kernel, dim3(s_.n), dim3(128), 0, st, s_, atlas_, mask, io, debug_flags);
        } else {
            hipLaunchKernelGGL(render_full_kernel, dim3(s_.n), dim3(128), 0, st, s_, atlas_, mask, io, debug_flags, 0);
        }
    }
    // Same layout as oracle/pgo_jumper.cpp Jumper::dump_state.
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
        const int flags = ri(I_FLAGS), themes = ri(I_THEMES), n_spikes = ri(I_NSPIKES);
        std::vector<float> v = {f(F_AX), f(F_AY), f(F_AVX), f(F_AVY), (flags & kFlagGround) ? 1.0f : 0.0f,
                                (flags & kFlagForward) ? 1.0f : 0.0f, f(F_APHASE), f(F_JUMP_T),
                                static_cast<float>(ri(I_JUMPS)), f(F_CAMX), f(F_CAMY), f(F_TOGX), f(F_TOGY),
                                static_cast<float>(themes & 0xff), f(F_BGSHIFT), static_cast<float>((themes >> 8) & 0xff),
                                f(F_PTIMER), (flags & kFlagPuffOn) ? 1.0f : 0.0f, f(F_GX), f(F_GY),
                                static_cast<float>(n_spikes)};
        for (int k = 0; k < kPuffs; k++)
            for (int fld : {PF_X, PF_Y, PF_LIFE}) v.push_back(rf(s_.pf, (size_t(env) * PF_COUNT + fld) * kPuffSlots + k));
        for (int k = 0; k < n_spikes; k++) {
            uint16_t cell;
            hipMemcpy(&cell, s_.spike_cell + size_t(env) * kSpikeSlots + k, 2, hipMemcpyDeviceToHost);
            v.push_back(static_cast<float>(cell / H) + 0.5f);
            v.push_back(static_cast<float>(H - 1 - cell % H) + 0.5f);
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

}  // namespace jumper

}  // namespace PG_VARIANT_NS

std::unique_ptr<Game> PG_FACTORY(make_jumper)() { return std::make_unique<PG_VARIANT_NS::jumper::JumperGame>(); }

}  // namespace pg
