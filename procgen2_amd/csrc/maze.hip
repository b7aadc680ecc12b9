// maze on gfx950: integer grid physics, randomised-Kruskal level generation.
//
// Reference:
//   step   games/maze/maze.cpp:279-330, common_systems.cpp:69-136
//   render games/maze/maze.cpp:386-414, tilemap.cpp:111-133, common_systems.cpp:41-63,138-150
//   reset  games/maze/maze.cpp:416-438, tilemap.cpp:31-109, maze_generator.cpp:55-139,183-195
// Config = the reference's compile-time default, hard_mode: 25×25 world, all visible, fixed camera
// (maze/tilemap.h:40-42, tilemap.cpp:35-38).
// Same machine mapping as coinrun.hip: logic one lane per env (SoA across envs), render one wave per env.
#include "pg_engine.h"
#include "pg_geom.h"
#include "pg_render.h"
#include "pg_rng.h"

namespace pg {
namespace maze {

constexpr int W = 25, H = 25, kCells = W * H;
constexpr int kTileStride = 640;  // 625 padded to a 128-byte multiple
constexpr int kTimeout = 500;     // maze.cpp:49
enum Cell : uint8_t { kOpen = 0, kWall = 1 };

enum Tex { kTexWall = 0, kTexCheese = 1, kTexMouse = 2, kTexFloor = 3, kTexCount = 12 };

enum { F_AX, F_AY, F_GX, F_GY, F_BGSHIFT, F_COUNT };
enum { I_FLAGS, I_STEPS, I_BG, I_COUNT };
constexpr int kFlagForward = 1, kFlagListed = 2;

struct State {
    int n;
    uint32_t* mt;    // [n][625]
    uint8_t* tiles;  // [n][640], column-major y + x*H
    float* f;        // [F_COUNT][n]
    int32_t* i;      // [I_COUNT][n]
};

PG_D float& SF(const State& s, int field, int env) { return s.f[size_t(field) * s.n + env]; }
PG_D int32_t& SI(const State& s, int field, int env) { return s.i[size_t(field) * s.n + env]; }

PG_D int tile_at(const uint8_t* t, int x, int y) {
    if (x < 0 || y < 0 || x >= W || y >= H) return kWall;  // maze/tilemap.h:79-84
    return t[y + x * H];
}

// Randomised Kruskal with union by rank + path halving over a 1-cell padded grid
// (maze_generator.cpp:55-139).  Scratch lives in private memory; types are narrowed to keep it small.
struct Carver {
    static constexpr int kMaxDim = 25, kPadDim = kMaxDim + 2;
    int mw, mh, aw, ah;
    uint8_t grid[kPadDim * kPadDim];
    uint8_t rank[kMaxDim * kMaxDim];
    int16_t parent[kMaxDim * kMaxDim];
    int16_t open_cells[kPadDim * kPadDim];
    uint8_t seen[(kMaxDim * kMaxDim + 7) / 8];
    uint8_t segs[312][4];  // walls between cells, 2 * 12 * 13 for a 25×25 maze
    int n_open, n_segs;

    PG_D int idx(int x, int y) const { return y + ah * x; }
    PG_D int get(int x, int y) const {
        if (x < 0 || y < 0 || x >= aw || y >= ah) return 1;
        return grid[idx(x, y)];
    }
    PG_D int root(int c) {
        int cur = c;
        while (parent[cur] != cur) {
            parent[cur] = parent[parent[cur]];
            cur = parent[cur];
        }
        return cur;
    }
    PG_D void open(int x, int y) {  // maze_generator.cpp:34-45
        grid[idx(x + 1, y + 1)] = 0;
        const int cell = y + mh * x;
        if (!(seen[cell >> 3] & (1 << (cell & 7)))) {
            open_cells[n_open++] = static_cast<int16_t>(cell);
            seen[cell >> 3] |= static_cast<uint8_t>(1 << (cell & 7));
        }
    }
    PG_D void carve(int dim, uint32_t* mt) {
        mw = mh = dim;
        aw = ah = dim + 2;
        for (int k = 0; k < aw * ah; k++) {
            grid[k] = 1;
            open_cells[k] = 0;
        }
        grid[idx(1, 1)] = 0;
        for (int k = 0; k < mw * mh; k++) {
            parent[k] = static_cast<int16_t>(k);
            rank[k] = 0;
        }
        for (int k = 0; k < static_cast<int>(sizeof(seen)); k++) seen[k] = 0;
        n_open = 0;
        n_segs = 0;
        for (int a = 1; a < mw; a += 2)
            for (int b = 0; b < mh; b += 2)
                if (a > 0 && a < mw - 1) {
                    segs[n_segs][0] = static_cast<uint8_t>(a - 1);
                    segs[n_segs][1] = static_cast<uint8_t>(b);
                    segs[n_segs][2] = static_cast<uint8_t>(a + 1);
                    segs[n_segs][3] = static_cast<uint8_t>(b);
                    n_segs++;
                }
        for (int a = 0; a < mw; a += 2)
            for (int b = 1; b < mh; b += 2)
                if (b > 0 && b < mh - 1) {
                    segs[n_segs][0] = static_cast<uint8_t>(a);
                    segs[n_segs][1] = static_cast<uint8_t>(b - 1);
                    segs[n_segs][2] = static_cast<uint8_t>(a);
                    segs[n_segs][3] = static_cast<uint8_t>(b + 1);
                    n_segs++;
                }
        while (n_segs > 0) {
            const int pick = rng_int(mt, 0, n_segs - 1);
            const int x1 = segs[pick][0], y1 = segs[pick][1], x2 = segs[pick][2], y2 = segs[pick][3];
            const int r0 = root(y1 + mh * x1);
            const int r1 = root(y2 + mh * x2);
            const int mx = (x1 + x2) / 2, my = (y1 + y2) / 2;
            const int centre = my + mh * mx;
            if (get(mx + 1, my + 1) == 1 && r0 != r1) {
                open(x1, y1);
                open(mx, my);
                open(x2, y2);
                if (rank[r0] > rank[r1]) {
                    parent[r1] = static_cast<int16_t>(r0);
                    parent[centre] = static_cast<int16_t>(r0);
                } else {
                    parent[r0] = static_cast<int16_t>(r1);
                    parent[centre] = static_cast<int16_t>(r1);
                    if (rank[r0] == rank[r1]) rank[r1]++;
                }
            }
            for (int k = pick; k < n_segs - 1; k++) {  // walls.erase(walls.begin() + n)
                segs[k][0] = segs[k + 1][0];
                segs[k][1] = segs[k + 1][1];
                segs[k][2] = segs[k + 1][2];
                segs[k][3] = segs[k + 1][3];
            }
            n_segs--;
        }
    }
    // maze_generator.cpp:183-195; START_CELL = 10 is compared with the cell index (D7).
    PG_D void drop(int kind, uint32_t* mt) {
        int k = rng_int(mt, 0, n_open - 1);
        while (open_cells[k] == -1 || open_cells[k] == 10) k = rng_int(mt, 0, n_open - 1);
        const int cell = open_cells[k];
        open_cells[k] = -1;
        grid[idx(cell / mh + 1, cell % mh + 1)] = static_cast<uint8_t>(kind);
    }
};

PG_D void new_level(const State& s, int env) {  // maze.cpp:416-438 + tilemap.cpp:31-109
    uint32_t* mt = s.mt + size_t(env) * kMtWords;
    uint8_t* tiles = s.tiles + size_t(env) * kTileStride;
    for (int k = 0; k < kCells; k++) tiles[k] = kWall;
    const int dim = rng_int(mt, 0, (W - 1) / 2 - 1) * 2 + 3;
    const int margin = (W - dim) / 2;
    Carver carver;
    carver.carve(dim, mt);
    carver.drop(2, mt);
    int gx = 0, gy = 0;
    for (int a = 0; a < dim; a++)
        for (int b = 0; b < dim; b++) {
            const int t = carver.get(a + 1, b + 1);
            tiles[(b + margin) + (a + margin) * H] = (t == 1) ? kWall : kOpen;
            if (t == 2) {
                gx = a + margin;
                gy = b + margin;
            }
        }
    SF(s, F_GX, env) = static_cast<float>(gx) + 0.5f;
    SF(s, F_GY, env) = static_cast<float>(H - 1 - gy) + 0.5f;
    SF(s, F_AX, env) = static_cast<float>(margin) + 0.5f;
    SF(s, F_AY, env) = static_cast<float>(H - 1 - margin) + 0.5f;
    SI(s, I_STEPS, env) = 0;
    SI(s, I_BG, env) = rng_int(mt, 0, 8);
    SF(s, F_BGSHIFT, env) = rng_real(mt, 0.0f, 1.0f);
    SI(s, I_FLAGS, env) = kFlagForward;  // face_forward = true; draw list cleared (D2)
}

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
    SI(s, I_FLAGS, env) = flags;
    SI(s, I_STEPS, env) = steps;
    reward_out = reached * 10.0f;
    terminated_out = reached || steps >= kTimeout;  // maze.cpp:302-310: the cap sets `terminated` (D5)
}

__global__ void __launch_bounds__(64) make_kernel(State s, uint32_t seed_base, int env_offset) {
    const int env = blockIdx.x * 64 + threadIdx.x;
    if (env >= s.n) return;
    mt_seed(s.mt + size_t(env) * kMtWords, seed_base + static_cast<uint32_t>(env_offset + env));
    new_level(s, env);
}

__global__ void __launch_bounds__(64) reset_kernel(State s, const uint8_t* mask, const int32_t* seeds, StepIO io) {
    const int env = blockIdx.x * 64 + threadIdx.x;
    if (env >= s.n) return;
    if (mask && !mask[env]) return;
    if (seeds) mt_seed(s.mt + size_t(env) * kMtWords, static_cast<uint32_t>(seeds[env]));
    new_level(s, env);
    io.reward[env] = 0.0f;
    io.done[env] = 0;
    io.pending[env] = 0;
}

__global__ void __launch_bounds__(64) logic_kernel(State s, const int32_t* actions, uint32_t run_seed,
                                                   uint32_t step_index, int env_offset, StepIO io) {
    const int env = blockIdx.x * 64 + threadIdx.x;
    if (env >= s.n) return;
    if (io.pending[env]) {
        new_level(s, env);
        io.reward[env] = 0.0f;
        io.done[env] = 0;
        io.pending[env] = 0;
        return;
    }
    const int action =
        actions ? actions[env] : synthetic_action(run_seed, step_index, static_cast<uint32_t>(env_offset + env));
    float reward;
    bool terminated;
    advance(s, env, action, reward, terminated);
    io.reward[env] = reward;
    io.done[env] = terminated ? 1 : 0;
    io.pending[env] = terminated ? 1 : 0;
}

// flags bit 0: force the draw-list replay for background + walls (fallback path).
__global__ void __launch_bounds__(64) render_kernel(State s, AtlasView atlas, const uint8_t* mask, StepIO io,
                                                    int flags) {
    const int env = blockIdx.x;
    if (mask && !mask[env]) return;
    const int lane = threadIdx.x;
    __shared__ uint32_t fb[kFbWords];
    constexpr int kGrid = 32;  // 25 visible tiles + the border cells of the inclusive window
    __shared__ ComposeLds<kGrid> L;

    // maze.cpp:397-400, 436-437: zoom = 64 / (16 * visible_width), camera at the world centre.
    const float zoom = 64.0f / (kUnitPx * 25.0f);
    const Camera cam{W * 0.5f * kUnitPx, H * 0.5f * kUnitPx, 64.0f, 64.0f, zoom};
    const int sflags = SI(s, I_FLAGS, env);
    const uint8_t* tiles = s.tiles + size_t(env) * kTileStride;
    Blit mine;

    Blit bg;  // background (maze.cpp:402-408)
    bool has_bg;
    {
        const int4 d = atlas.desc[kTexFloor + SI(s, I_BG, env)];
        const float aspect = static_cast<float>(d.y) / static_cast<float>(d.z);
        const float extra = aspect - 1.0f;
        has_bg = resolve_draw(cam, d.y, d.z, d.x, -SF(s, F_BGSHIFT, env) * extra, 0.0f, 64.0f * kUnitPx / d.z, 1.0f,
                              false, false, bg);
    }
    // wall window (tilemap.cpp:111-121)
    const float vx = (cam.px - cam.sw * 0.5f / cam.scale) * kPxUnit;
    const float vy = (cam.py - cam.sh * 0.5f / cam.scale) * kPxUnit;
    const float vw = cam.sw * kPxUnit / cam.scale, vh = cam.sh * kPxUnit / cam.scale;
    const int x0 = static_cast<int>(floorf(vx)), y0 = static_cast<int>(floorf(vy));
    const int x1 = static_cast<int>(ceilf(vx + vw)), y1 = static_cast<int>(ceilf(vy + vh));
    const int cols = x1 - x0 + 1, rows = y1 - y0 + 1, cells = cols * rows;
    const int4 wall = atlas.desc[kTexWall];

    bool composed = false;
    if (!(flags & 1) && cols <= kGrid && rows <= kGrid) {
        compose_spans(L, cam, x0, y0, cols, rows, wall.y, wall.z, kUnitPx / wall.y, lane);
        for (int cell = lane; cell < cells; cell += 64) {
            const int r = cell / cols, c = cell - r * cols;
            L.base[r * kGrid + c] =
                tile_at(tiles, x0 + c, H - 1 - (y0 + r)) == kOpen ? static_cast<int32_t>(kNoTexel) : wall.x * 4;
        }
        __syncthreads();
        composed = compose_rows(fb, L, atlas, bg, has_bg, cols, rows, wall.y, lane, flags);
    }
    if (!composed) {  // draw-list replay (tilemap.cpp:111-133)
        wave_clear(fb, lane);
        mine = bg;
        wave_replay(fb, atlas, mine, has_bg ? 1ull : 0ull, lane);
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
            wave_replay(fb, atlas, mine, __ballot(has), lane);
        }
    }
    if (sflags & kFlagListed) {  // the cheese sprite (tilemap.cpp:88): offset (-0.48,-0.5), scale 0.95, z = 1
        const int4 d = atlas.desc[kTexCheese];
        const float scale = 1.0f * 0.95f;
        const bool ok = resolve_draw(cam, d.y, d.z, d.x, (SF(s, F_GX, env) + -0.48f) * kUnitPx,
                                     (SF(s, F_GY, env) + -0.5f) * kUnitPx, scale * kUnitPx / d.y, 1.0f, false, false,
                                     mine);
        wave_replay(fb, atlas, mine, ok ? 1ull : 0ull, lane);
    }
    {  // the mouse (common_systems.cpp:138-150); flip = face_forward
        const int4 d = atlas.desc[kTexMouse];
        const bool ok = resolve_draw(cam, d.y, d.z, d.x, (SF(s, F_AX, env) + -0.5f) * kUnitPx,
                                     (SF(s, F_AY, env) + -0.5f) * kUnitPx, kUnitPx / d.y * 1.0f, 1.0f,
                                     (sflags & kFlagForward) != 0, false, mine);
        wave_replay(fb, atlas, mine, ok ? 1ull : 0ull, lane);
    }
    wave_store_obs(fb, io.obs + size_t(env) * kObsBytes, lane);
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
        return align256(size_t(n) * kMtWords * 4) + align256(size_t(n) * kTileStride) +
               align256(size_t(F_COUNT) * n * 4) + align256(size_t(I_COUNT) * n * 4);
    }
    void bind(void* d_state, int n, AtlasView atlas) override {
        uint8_t* p = static_cast<uint8_t*>(d_state);
        auto take = [&](size_t bytes) {
            uint8_t* q = p;
            p += align256(bytes);
            return q;
        };
        s_.n = n;
        s_.mt = reinterpret_cast<uint32_t*>(take(size_t(n) * kMtWords * 4));
        s_.tiles = take(size_t(n) * kTileStride);
        s_.f = reinterpret_cast<float*>(take(size_t(F_COUNT) * n * 4));
        s_.i = reinterpret_cast<int32_t*>(take(size_t(I_COUNT) * n * 4));
        atlas_ = atlas;
    }
    int blocks() const { return (s_.n + 63) / 64; }
    void launch_make(hipStream_t st, uint32_t seed_base, int env_offset) override {
        hipLaunchKernelGGL(make_kernel, dim3(blocks()), dim3(64), 0, st, s_, seed_base, env_offset);
    }
    void launch_reset(hipStream_t st, const uint8_t* mask, const int32_t* seeds, StepIO io) override {
        hipLaunchKernelGGL(reset_kernel, dim3(blocks()), dim3(64), 0, st, s_, mask, seeds, io);
    }
    void launch_logic(hipStream_t st, const int32_t* actions, uint32_t run_seed, uint32_t step_index, int env_offset,
                      StepIO io) override {
        hipLaunchKernelGGL(logic_kernel, dim3(blocks()), dim3(64), 0, st, s_, actions, run_seed, step_index,
                           env_offset, io);
    }
    void launch_render(hipStream_t st, const uint8_t* mask, StepIO io) override {
        hipLaunchKernelGGL(render_kernel, dim3(s_.n), dim3(64), 0, st, s_, atlas_, mask, io, debug_flags);
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
        const float v[8] = {f(F_AX), f(F_AY), (iv(I_FLAGS) & kFlagForward) ? 1.0f : 0.0f, f(F_GX), f(F_GY),
                            static_cast<float>(iv(I_STEPS)), static_cast<float>(iv(I_BG)), f(F_BGSHIFT)};
        for (int k = 0; k < 8 && k < cap; k++) out[k] = v[k];
        return 8;
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

std::unique_ptr<Game> make_maze() { return std::make_unique<maze::MazeGame>(); }

}  // namespace pg
