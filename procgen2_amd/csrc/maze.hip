// maze on gfx950: integer grid physics, randomised-Kruskal level generation.
//
// Reference:
//   step   games/maze/maze.cpp:279-330, common_systems.cpp:69-136
//   render games/maze/maze.cpp:386-414, tilemap.cpp:111-133, common_systems.cpp:41-63,138-150
//   reset  games/maze/maze.cpp:416-438, tilemap.cpp:31-109, maze_generator.cpp:55-139,183-195
// Config: one compiled variant per distribution mode (below); variant 0 = the reference's compile-time default,
// hard_mode: 25×25 world, all visible, fixed camera
// (maze/tilemap.h:40-42, tilemap.cpp:35-38).
// Machine mapping: logic one lane per env (SoA across envs), render two waves per env, level generation one wave per
// env on LDS.  The game draws no random numbers during an episode and every episode ends within 500 steps, so
// resets are frequent and bursty; the next maze of every env is carved ahead of time on a side stream and copied
// in at reset (pg_prefetch.h).
#include "pg_engine.h"
#include "pg_frame.h"
#include "pg_geom.h"
#include "pg_kruskal.h"
#include "pg_prefetch.h"
#include "pg_render.h"
#include "pg_prepass.h"
#include "pg_rng.h"

namespace pg {
namespace PG_VARIANT_NS {
namespace maze {

// tilemap.cpp:31-48: world side, visible side (the zoom, maze.cpp:397), camera on the agent or on the world centre
#if PG_VARIANT == 0  // hard_mode — the reference's compile-time default (tilemap.h:41)
constexpr int W = 25, kVisible = 25;
constexpr bool kCentred = false;
#elif PG_VARIANT == 1  // easy_mode
constexpr int W = 15, kVisible = 15;
constexpr bool kCentred = false;
#elif PG_VARIANT == 2  // memory_mode
constexpr int W = 31, kVisible = 8;
constexpr bool kCentred = true;
#else
#error "maze: unknown PG_VARIANT"
#endif
constexpr int H = W, kCells = W * H;
constexpr int kTileStride = (kCells + 127) / 128 * 128;  // 625 → 640
constexpr int kTimeout = 500;     // maze.cpp:49
enum Cell : uint8_t { kOpen = 0, kWall = 1 };

enum Tex { kTexWall = 0, kTexCheese = 1, kTexMouse = 2, kTexFloor = 3, kTexCount = 12 };

enum { F_AX, F_AY, F_GX, F_GY, F_BGSHIFT, F_CAMX, F_CAMY, F_COUNT };
enum { I_FLAGS, I_STEPS, I_BG, I_COUNT };
constexpr int kFlagForward = 1, kFlagListed = 2;

// One generated level, as the generator leaves it in LDS and as it waits in the shadow slot.
struct Level {
    uint8_t tiles[kTileStride];
    float ax, ay, gx, gy, bgshift;
    int32_t bg;
};

struct State {
    int n;
    Level* shadow;   // [n]  next level of each env (pg_prefetch.h)
    int32_t* slot;   // [n]  SlotState
    uint32_t* mt;    // [n][625]  generator chain: the stream position after the newest generated level
    uint8_t* tiles;  // [n][kTileStride], column-major y + x*H
    // !kCentred: the camera rests on the world centre and shows all of it, so the row composer's span and hand-over
    // tables are the same in every frame: worked out once (prepare_kernel), read from here (pg_render.h compose_prepare)
    ComposeHand* prepared;
    float* f;        // [F_COUNT][n]
    int32_t* i;      // [I_COUNT][n]
    // The render pre-pass's hand-over (scratch memory, not state): the frame's two draws and the background's two axes, worked
    // out by four lanes an env of a dense kernel (setup_kernel) instead of by 2 + 1 lanes of each of the env's two render
    // wavefronts — a quarter of what those executed.
    struct Prep {
        uint32_t* draws;  // [n][2][kBlitWords]  cheese, mouse (pg_render.h BlitWords; word 1 = 0: not drawn)
        uint32_t* bg;     // [n][8]              background, x axis then y axis: d0 | dn << 16, s0 | sn << 16, first texel, width
    } prep;
};

PG_D float& SF(const State& s, int field, int env) { return s.f[size_t(field) * s.n + env]; }
PG_D int32_t& SI(const State& s, int field, int env) { return s.i[size_t(field) * s.n + env]; }

PG_D int tile_at(const uint8_t* t, int x, int y) {
    if (x < 0 || y < 0 || x >= W || y >= H) return kWall;  // maze/tilemap.h:79-84
    return t[y + x * H];
}

using MazeLds = KruskalLdsT<(W > 25 ? W : 25)>;
struct GenLds {
    uint32_t mt[kMtWords];
    MazeLds k;
};

// reset() (maze.cpp:416-438 + tilemap.cpp:31-109) for one env by one wavefront: advances the env's generator chain
// (s.mt) and leaves the level in `lv` (LDS).
PG_D void generate(const State& s, int env, GenLds& L, Level& lv, bool reseed, uint32_t seed, int lane) {
    uint32_t* gmt = s.mt + size_t(env) * kMtWords;
    if (reseed) {
        if (lane == 0) mt_seed(L.mt, seed);
    } else {
        for (int k = lane; k < kMtWords; k += 64) L.mt[k] = gmt[k];
    }
    for (int k = lane; k < kTileStride; k += 64) lv.tiles[k] = kWall;
    __syncthreads();
    uint32_t* mt = L.mt;
    const int dim = wave_rng_int(mt, 0, (W - 1) / 2 - 1, lane) * 2 + 3;
    const int margin = (W - dim) / 2;
    CarverT<MazeLds> carver{L.k, 0, 0, 0, 0};
    carver.carve(dim, mt, lane);
    carver.drop(2, mt, lane);
    for (int c = lane; c < dim * dim; c += 64) {
        const int a = c / dim, b = c % dim;
        const int t = carver.get(a + 1, b + 1);
        lv.tiles[(b + margin) + (a + margin) * H] = (t == 1) ? kWall : kOpen;
        if (t == 2) {
            lv.gx = static_cast<float>(a + margin) + 0.5f;
            lv.gy = static_cast<float>(H - 1 - (b + margin)) + 0.5f;
        }
    }
    const int bg = wave_rng_int(mt, 0, 8, lane);
    const float shift = wave_rng_real(mt, 0.0f, 1.0f, lane);
    if (lane == 0) {
        lv.ax = static_cast<float>(margin) + 0.5f;
        lv.ay = static_cast<float>(H - 1 - margin) + 0.5f;
        lv.bg = bg;
        lv.bgshift = shift;
    }
    __syncthreads();
    for (int k = lane; k < kMtWords; k += 64) gmt[k] = L.mt[k];
    __syncthreads();
}

// The level becomes the env's live state.
PG_D void install(const State& s, int env, const Level& lv, int lane) {
    uint32_t* tiles = reinterpret_cast<uint32_t*>(s.tiles + size_t(env) * kTileStride);
    const uint32_t* src = reinterpret_cast<const uint32_t*>(lv.tiles);
    for (int k = lane; k < kTileStride / 4; k += 64) tiles[k] = src[k];
    if (lane == 0) {
        SF(s, F_GX, env) = lv.gx;
        SF(s, F_GY, env) = lv.gy;
        SF(s, F_AX, env) = lv.ax;
        SF(s, F_AY, env) = lv.ay;
        SI(s, I_STEPS, env) = 0;
        SI(s, I_BG, env) = lv.bg;
        SF(s, F_BGSHIFT, env) = lv.bgshift;
        SI(s, I_FLAGS, env) = kFlagForward;  // face_forward = true; draw list cleared (D2)
        SF(s, F_CAMX, env) = W * 0.5f * kUnitPx;  // maze.cpp:436-437
        SF(s, F_CAMY, env) = H * 0.5f * kUnitPx;
    }
}

// What cenv_make leaves in an env besides the seeded RNG, split by owner: the generator chain (bucket counts of
// the sets that survive clear()) and the live state.  Level-seed mode (pg_engine.h LevelPlan) rebuilds every
// level from here.
PG_D void fresh_chain(const State& s, int env) {
}
PG_D void fresh_live(const State& s, int env) {
}

struct Gen {  // pg_prefetch.h level_kernel<Gen>
    using State = maze::State;
    using Level = maze::Level;
    using GenLds = maze::GenLds;
    PG_D static void generate(const State& s, int env, GenLds& L, Level& lv, bool reseed, uint32_t seed, int lane) {
        maze::generate(s, env, L, lv, reseed, seed, lane);
    }
    PG_D static void install(const State& s, int env, const Level& lv, int lane) { maze::install(s, env, lv, lane); }
    PG_D static void fresh_chain(const State& s, int env) { maze::fresh_chain(s, env); }
    PG_D static void fresh_live(const State& s, int env) { maze::fresh_live(s, env); }
};

PG_D void advance(const State& s, int env, int action, float& reward_out, bool& terminated_out) {
    const uint8_t* tiles = s.tiles + size_t(env) * kTileStride;
    float ax = SF(s, F_AX, env), ay = SF(s, F_AY, env);
    int flags = SI(s, I_FLAGS, env);
    const int mx = action / 3 - 1;  // actions 9..14 give 2/3-cell jumps (D6)
    const int my = mx ? 0 : -(action % 3 - 1);
    if (mx) {
        if (tile_at(tiles, static_cast<int>(ax + mx), H - 1 - static_cast<int>(ay)) == kOpen)
            ax = static_cast<int>(ax + mx) + 0.5f;
    } else if (my) {
        if (tile_at(tiles, static_cast<int>(ax), H - 1 - static_cast<int>(ay + my)) == kOpen)
            ay = static_cast<int>(ay + my) + 0.5f;
    }
    const Box body{ax + -0.5f, ay + -0.5f, 1.0f, 1.0f};
    const Box goal{SF(s, F_GX, env) + -0.5f, SF(s, F_GY, env) + -0.5f, 1.0f, 1.0f};
    const bool reached = box_hit(body, goal);
    if (mx > 0)
        flags |= kFlagForward;
    else if (mx < 0)
        flags &= ~kFlagForward;
    flags |= kFlagListed;
    const int steps = SI(s, I_STEPS, env) + 1;
    SF(s, F_AX, env) = ax;
    SF(s, F_AY, env) = ay;
    if (kCentred) {  // common_systems.cpp:119-123: the camera follows the agent
        SF(s, F_CAMX, env) = ax * kUnitPx;
        SF(s, F_CAMY, env) = ay * kUnitPx;
    }
    SI(s, I_FLAGS, env) = flags;
    SI(s, I_STEPS, env) = steps;
    reward_out = reached * 10.0f;
    terminated_out = reached || steps >= kTimeout;  // maze.cpp:302-310: the cap sets `terminated` (D5)
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

constexpr int kGrid = kVisible + 3 <= 20 ? 20 : 28;  // visible tiles + the border cells of the inclusive window; as small as it
// may be: the composer tables are LDS, and LDS decides how many envs a CU holds (28: 7 per CU, 32: 6)
constexpr int kSpan = 64 / kVisible + 2 <= kMaxSpan ? kMaxSpan : 16;  // pixels a tile covers (+ seam padding)

// the wall window of a camera (tilemap.cpp:111-121)
struct Window {
    int x0, y0, cols, rows;
};
PG_D Window window_of(const Camera& cam) {
    const float vx = (cam.px - cam.sw * 0.5f / cam.scale) * kPxUnit;
    const float vy = (cam.py - cam.sh * 0.5f / cam.scale) * kPxUnit;
    const float vw = cam.sw * kPxUnit / cam.scale, vh = cam.sh * kPxUnit / cam.scale;
    const int x0 = static_cast<int>(floorf(vx)), y0 = static_cast<int>(floorf(vy));
    const int x1 = static_cast<int>(ceilf(vx + vw)), y1 = static_cast<int>(ceilf(vy + vh));
    return Window{x0, y0, x1 - x0 + 1, y1 - y0 + 1};
}
PG_D float zoom_of_obs() { return 64.0f / (kUnitPx * static_cast<float>(kVisible)); }  // maze.cpp:397-400

// Once per engine (!kCentred): the composer's tables for the camera on the world centre (State::prepared).
__global__ void __launch_bounds__(128) prepare_kernel(State s, AtlasView atlas) {
    __shared__ alignas(16) uint32_t fb[kFbWords];
    __shared__ ComposeLds<kGrid> L;
    const int lane = threadIdx.x & 63, half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const Camera cam{W * 0.5f * kUnitPx, H * 0.5f * kUnitPx, 64.0f, 64.0f, zoom_of_obs()};  // maze.cpp:436-437
    const Window w = window_of(cam);
    const int4 wall = atlas.desc[kTexWall];
    compose_spans<kGrid, kSpan>(fb, L, cam, w.x0, w.y0, w.cols, w.rows, wall.y, wall.z, kUnitPx / wall.y, lane, 0, half, 2,
                                soft_rows_of(0, wall.w), hard_rows_of(0, wall.w));
    __syncthreads();
    compose_hand_build<kGrid, false, false>(fb, L, BgAxis{}, wall.y, lane, 0, half, make_int4(0, -1, 0, -1));
    compose_prepare<kGrid>(fb, L, s.prepared, lane, half);
}

// flags bit 0: force the draw-list replay for background + walls (fallback path).
// The render pre-pass: four lanes an env — cheese, mouse (tilemap.cpp:88, common_systems.cpp:138-150 as render_kernel states
// them), the background's x and y axis (maze.cpp:402-408; pg_render.h bg_axis).
__global__ void __launch_bounds__(256) setup_kernel(State s, AtlasView atlas) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int env = gid >> 2, item = gid & 3;
    if (env >= s.n) return;
    const Camera cam{kCentred ? SF(s, F_CAMX, env) : W * 0.5f * kUnitPx, kCentred ? SF(s, F_CAMY, env) : H * 0.5f * kUnitPx,
                     64.0f, 64.0f, zoom_of_obs()};
    if (item < 2) {
        const bool cheese = item == 0;
        const int sflags = SI(s, I_FLAGS, env);
        const int4 d = atlas.desc[cheese ? kTexCheese : kTexMouse];
        const float wx = cheese ? (SF(s, F_GX, env) + -0.48f) * kUnitPx : (SF(s, F_AX, env) + -0.5f) * kUnitPx;
        const float wy = cheese ? (SF(s, F_GY, env) + -0.5f) * kUnitPx : (SF(s, F_AY, env) + -0.5f) * kUnitPx;
        const float scale = cheese ? (1.0f * 0.95f) * kUnitPx / d.y : kUnitPx / d.y * 1.0f;
        bool has = cheese ? (sflags & kFlagListed) != 0 : true;
        Blit b;
        if (has) has = resolve_draw(cam, d.y, d.z, d.x, wx, wy, scale, 1.0f, !cheese && (sflags & kFlagForward) != 0, false, b);
        BlitWords w{};
        if (has) w = blit_pack(b);
        uint2* at = reinterpret_cast<uint2*>(s.prep.draws + (size_t(env) * 2 + item) * kBlitWords);
        at[0] = make_uint2(w.w[0], w.w[1]);
        at[1] = make_uint2(w.w[2], w.w[3]);
        at[2] = make_uint2(w.w[4], w.w[5]);
    } else {
        const int axis = item - 2;
        const int4 d = atlas.desc[kTexFloor + SI(s, I_BG, env)];
        const float aspect = static_cast<float>(d.y) / static_cast<float>(d.z);
        const float extra = aspect - 1.0f;
        const BgAxis a = bg_axis(cam, d, -SF(s, F_BGSHIFT, env) * extra, 0.0f, 64.0f * kUnitPx / d.z, axis);
        uint4* at = reinterpret_cast<uint4*>(s.prep.bg + size_t(env) * 8 + axis * 4);
        *at = make_uint4(pack_halves(a.d0, a.dn), pack_halves(a.s0, a.sn), static_cast<uint32_t>(a.tex_off), static_cast<uint32_t>(a.tex_w));
    }
}

// kPrepped: the two draws and (camera at rest) the background's axes come from the pre-pass; false: the debug paths.
template <bool kPrepped>
__global__ void __launch_bounds__(128, 4) render_kernel(State s, AtlasView atlas, const uint8_t* mask, StepIO io,
                                                    int flags) {
    const int env = blockIdx.x;
    if (mask && !mask[env]) return;
    const int lane = threadIdx.x & 63, half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // two wavefronts per env (pg_render.h)
    constexpr int halves = 2;
    __shared__ alignas(16) uint32_t fb[kFbWords];
    __shared__ ComposeLds<kGrid> L;

    // maze.cpp:397-400, 436-437: zoom = 64 / (16 * visible_width); camera at the world centre, or on the agent
    const float zoom = zoom_of_obs();
    const Camera cam{kCentred ? SF(s, F_CAMX, env) : W * 0.5f * kUnitPx, kCentred ? SF(s, F_CAMY, env) : H * 0.5f * kUnitPx,
                     64.0f, 64.0f, zoom};
    const int sflags = SI(s, I_FLAGS, env);
    const uint8_t* tiles = s.tiles + size_t(env) * kTileStride;
    Blit mine;

    int bg_soft = 0;  // the backdrop has texels that are not opaque (descriptor .w)
    int4 bg_d;  // the background draw, background (maze.cpp:402-408): texture, world position, scale — each wave resolves the axis it needs (pg_render.h BgAxis)
    float bg_px, bg_py, bg_sc;
    {
        const int4 d = atlas.desc[kTexFloor + SI(s, I_BG, env)];
        bg_soft = d.w;
        const float aspect = static_cast<float>(d.y) / static_cast<float>(d.z);
        const float extra = aspect - 1.0f;
        bg_d = d;
        bg_px = -SF(s, F_BGSHIFT, env) * extra;
        bg_py = 0.0f;
        bg_sc = 64.0f * kUnitPx / d.z;
    }
    const Window win = window_of(cam);
    const int x0 = win.x0, y0 = win.y0, cols = win.cols, rows = win.rows, cells = cols * rows;
    const int4 wall = atlas.desc[kTexWall];

    const BgDraw bg_draw{bg_d, bg_px, bg_py, bg_sc};
    BgAxis bga{};  // this wave's axis of it (wave 0: x, wave 1: y), resolved along with the tile spans
    bool composed = false;
    if (!(flags & 1) && cols <= kGrid && rows <= kGrid) {
        if (kCentred) {
            compose_spans<kGrid, kSpan>(fb, L, cam, x0, y0, cols, rows, wall.y, wall.z, kUnitPx / wall.y, lane, 0, half, halves,
                                        soft_rows_of(bg_soft, wall.w), hard_rows_of(bg_soft, wall.w), &bg_draw, &bga);
        } else {  // the tile spans are the prepared ones: only this wave's axis of the background is resolved
            if (kPrepped) {
                const uint32_t* w = s.prep.bg + size_t(env) * 8 + half * 4;
                bga = BgAxis{static_cast<int32_t>(w[0] << 16) >> 16, static_cast<int32_t>(w[0]) >> 16, static_cast<int32_t>(w[1] & 0xffffu),
                             static_cast<int32_t>(w[1] >> 16), static_cast<int32_t>(w[2]), static_cast<int32_t>(w[3])};
            } else {
                bga = bg_axis(cam, bg_d, bg_px, bg_py, bg_sc, half);
            }
            if (half == 0 && lane < 2) L.base[kGrid * kGrid + lane] = static_cast<int32_t>(kNoTexel);
        }
        for (int cell = lane + 64 * half; cell < cells; cell += 64 * halves) {
            const int r = cell / cols, c = cell - r * cols;
            L.base[r * kGrid + c] =
                tile_at(tiles, x0 + c, H - 1 - (y0 + r)) == kOpen ? static_cast<int32_t>(kNoTexel) : wall.x * 4;
        }
        __syncthreads();
        composed = kCentred ? compose_rows<kGrid>(fb, L, atlas, bga, cols, rows, wall.y, lane, flags, half, halves)
                            : compose_rows<kGrid, false, false, true>(fb, L, atlas, bga, cols, rows, wall.y, lane, flags, half, halves,
                                                                      make_int4(0, -1, 0, -1), s.prepared, bg_soft);
    }
    if (!composed) {  // draw-list replay (tilemap.cpp:111-133)
        wave_clear(fb, lane, half, halves);
        const bool has_bg = resolve_draw(cam, bg_d.y, bg_d.z, bg_d.x, bg_px, bg_py, bg_sc, 1.0f, false, false, mine);
        wave_replay(fb, atlas, mine, has_bg ? 1ull : 0ull, lane, half, halves);
        for (int base = 0; base < cells; base += 64) {
            const int cell = base + lane;
            bool has = false;
            if (cell < cells) {
                const int row = cell / cols;
                const int x = x0 + (cell - row * cols), y = y0 + row;
                if (tile_at(tiles, x, H - 1 - y) != kOpen)
                    has = resolve_draw(cam, wall.y, wall.z, wall.x, x * kUnitPx, y * kUnitPx, kUnitPx / wall.y, 1.0f,
                                       false, false, mine);
            }
            wave_replay(fb, atlas, mine, __ballot(has), lane, half, halves);
        }
    }
    if (kPrepped) {  // lane 0: the cheese, lane 1: the mouse, as the pre-pass resolved them
        mine = prep_draw_load(s.prep.draws + (size_t(env) * 2 + (lane & 1)) * kBlitWords, lane < 2);
        wave_replay_rows(fb, atlas, mine, __ballot(lane < 2 && mine.dw > 0), lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
    } else {
        // lane 0: the cheese sprite (tilemap.cpp:88): offset (-0.48,-0.5), scale 0.95, z = 1, once it is listed;
        // lane 1: the mouse (common_systems.cpp:138-150); flip = face_forward.  The two differ in their parameters
        // only: one pass through resolve_draw, one replay, in this order.
        const bool cheese = lane == 0;
        const int4 d = atlas.desc[cheese ? kTexCheese : kTexMouse];
        const float wx = cheese ? (SF(s, F_GX, env) + -0.48f) * kUnitPx : (SF(s, F_AX, env) + -0.5f) * kUnitPx;
        const float wy = cheese ? (SF(s, F_GY, env) + -0.5f) * kUnitPx : (SF(s, F_AY, env) + -0.5f) * kUnitPx;
        const float scale = cheese ? (1.0f * 0.95f) * kUnitPx / d.y : kUnitPx / d.y * 1.0f;
        bool has = cheese ? (sflags & kFlagListed) != 0 : lane == 1;
        if (has)
            has = resolve_draw(cam, d.y, d.z, d.x, wx, wy, scale, 1.0f, !cheese && (sflags & kFlagForward) != 0, false, mine);
        wave_replay_rows(fb, atlas, mine, __ballot(has), lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
    }
    // each wave stores the rows it owns (pg_render.h wave_replay_rows): no barrier
    wave_store_rows(fb, io.obs + size_t(env) * kObsBytes, lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
}

// cenv_render's frame (render_game(false)) for one env: pg_frame.h; the draw list of render_kernel, one draw at a time.
__global__ void __launch_bounds__(kFrameThreads) frame_kernel(State s, AtlasView atlas, int env, FrameTarget t) {
    const float fw = static_cast<float>(t.w), fh = static_cast<float>(t.h);
    const float zoom = fw / (kUnitPx * static_cast<float>(kVisible));  // maze.cpp:397-400
    FramePainter P{t, atlas, Camera{SF(s, F_CAMX, env), SF(s, F_CAMY, env), fw, fh, zoom}, static_cast<int>(threadIdx.x),
                   kFrameThreads};
    const int sflags = SI(s, I_FLAGS, env);
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
            if (tile_at(tiles, x, H - 1 - y) != kOpen) P.draw(kTexWall, x * kUnitPx, y * kUnitPx, kUnitPx / P.desc(kTexWall).y);
    if (sflags & kFlagListed) {
        const float scale = 1.0f * 0.95f;
        P.draw(kTexCheese, (SF(s, F_GX, env) + -0.48f) * kUnitPx, (SF(s, F_GY, env) + -0.5f) * kUnitPx,
               scale * kUnitPx / P.desc(kTexCheese).y);
    }
    P.draw(kTexMouse, (SF(s, F_AX, env) + -0.5f) * kUnitPx, (SF(s, F_AY, env) + -0.5f) * kUnitPx,
           kUnitPx / P.desc(kTexMouse).y * 1.0f, 1.0f, (sflags & kFlagForward) != 0);
}

class MazeGame final : public Game {
   public:
    const char* name() const override { return "maze"; }
    std::vector<std::string> texture_names() const override {
        std::vector<std::string> v = {"kenney/Ground/Sand/sandCenter.png", "misc_assets/cheese.png",
                                      "kenney/Enemies/mouse_move.png", "topdown_backgrounds/floortiles.png"};
        for (int k = 1; k <= 8; k++) v.push_back("topdown_backgrounds/backgrounddetailed" + std::to_string(k) + ".png");
        return v;
    }
    static size_t align256(size_t x) { return (x + 255) & ~size_t(255); }
    size_t state_bytes(int n) const override {
        return align256(size_t(n) * sizeof(Level)) + align256(size_t(n) * 4) + align256(size_t(n) * kMtWords * 4) +
               align256(size_t(n) * kTileStride) +
               align256(size_t(F_COUNT) * n * 4) + align256(size_t(I_COUNT) * n * 4) + align256(sizeof(ComposeHand));
    }
    void bind(void* d_state, int n, AtlasView atlas) override {
        uint8_t* p = static_cast<uint8_t*>(d_state);
        auto take = [&](size_t bytes) {
            uint8_t* q = p;
            p += align256(bytes);
            return q;
        };
        s_.n = n;
        s_.shadow = reinterpret_cast<Level*>(take(size_t(n) * sizeof(Level)));
        s_.slot = reinterpret_cast<int32_t*>(take(size_t(n) * 4));
        s_.mt = reinterpret_cast<uint32_t*>(take(size_t(n) * kMtWords * 4));
        s_.tiles = take(size_t(n) * kTileStride);
        s_.f = reinterpret_cast<float*>(take(size_t(F_COUNT) * n * 4));
        s_.i = reinterpret_cast<int32_t*>(take(size_t(I_COUNT) * n * 4));
        s_.prepared = reinterpret_cast<ComposeHand*>(take(sizeof(ComposeHand)));
        atlas_ = atlas;
    }
    int blocks() const { return (s_.n + 63) / 64; }
    void launch_make(hipStream_t st, uint32_t seed_base, int env_offset) override {
        LevelLaunch<Gen>::make(st, s_, prefetch(), seed_base, env_offset, plan);
        if (!kCentred) hipLaunchKernelGGL(prepare_kernel, dim3(1), dim3(128), 0, st, s_, atlas_);
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
        hipLaunchKernelGGL(logic_kernel, dim3((s_.n + 63) / 64, fused ? 2 : 1), dim3(64), 0, st, s_, actions, run_seed, step_index,
                           env_offset, io, prefetch(), plan);
        if (fused) LevelLaunch<Gen>::auto_reset(st, s_, prefetch(), io, plan, PG_RESET_SPAN, reset_served_mark(step_index), reset_due_mark(step_index));
    }
    bool launch_frame(hipStream_t st, int env, uint32_t* d_px, int w, int h) override {
        hipLaunchKernelGGL(frame_kernel, dim3(1), dim3(kFrameThreads), 0, st, s_, atlas_, env, FrameTarget{d_px, w, h});
        return true;
    }
    // (the draw-list replay and kDebugNoPrepass take the kernel that resolves its own draws)
    bool lean() const { return !(debug_flags & (1 | kDebugNoPrepass)); }
    void launch_prepass(hipStream_t st, const uint8_t* mask) override {
        (void)mask;  // (every env: a lane's work, and the frames of the others are not drawn)
        if (lean()) hipLaunchKernelGGL(setup_kernel, dim3((s_.n * 4 + 255) / 256), dim3(256), 0, st, s_, atlas_);
    }
    void launch_render(hipStream_t st, const uint8_t* mask, StepIO io) override {
        if (lean())
            hipLaunchKernelGGL(render_kernel<true>, dim3(s_.n), dim3(128), 0, st, s_, atlas_, mask, io, debug_flags);
        else
            hipLaunchKernelGGL(render_kernel<false>, dim3(s_.n), dim3(128), 0, st, s_, atlas_, mask, io, debug_flags);
    }
    static size_t up256(size_t b) { return (b + 255) & ~size_t(255); }
    size_t scratch_bytes(int n) const override { return up256(size_t(n) * 2 * kBlitWords * 4) + up256(size_t(n) * 8 * 4); }
    void bind_scratch(void* d_scratch, int n) override {
        uint8_t* p = static_cast<uint8_t*>(d_scratch);
        s_.prep.draws = reinterpret_cast<uint32_t*>(p);
        s_.prep.bg = reinterpret_cast<uint32_t*>(p + up256(size_t(n) * 2 * kBlitWords * 4));
    }
    // Same layout as oracle/pgo_maze.cpp Maze::dump_state.
    int dump_state(hipStream_t st, int env, float* out, int cap) override {
        hipStreamSynchronize(st);
        auto f = [&](int field) {
            float v;
            hipMemcpy(&v, s_.f + size_t(field) * s_.n + env, 4, hipMemcpyDeviceToHost);
            return v;
        };
        auto iv = [&](int field) {
            int32_t v;
            hipMemcpy(&v, s_.i + size_t(field) * s_.n + env, 4, hipMemcpyDeviceToHost);
            return v;
        };
        const float v[10] = {f(F_AX), f(F_AY), (iv(I_FLAGS) & kFlagForward) ? 1.0f : 0.0f, f(F_GX), f(F_GY),
                             static_cast<float>(iv(I_STEPS)), static_cast<float>(iv(I_BG)), f(F_BGSHIFT), f(F_CAMX),
                             f(F_CAMY)};
        for (int k = 0; k < 10 && k < cap; k++) out[k] = v[k];
        return 10;
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

}  // namespace maze

}  // namespace PG_VARIANT_NS

std::unique_ptr<Game> PG_FACTORY(make_maze)() { return std::make_unique<PG_VARIANT_NS::maze::MazeGame>(); }

}  // namespace pg
