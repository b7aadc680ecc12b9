// Vector env engine + the two C ABIs (include/procgen2_vec.h, include/procgen2_cenv.h).
#include <dirent.h>
#include <dlfcn.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <mutex>

#include "../../include/procgen2_cenv.h"
#include "../../include/procgen2_vec.h"
#include "pg_engine.h"
#include "pg_order.h"
#include "png_decode.h"

#ifndef PG_DEFAULT_GAME
#define PG_DEFAULT_GAME 0
#endif

namespace pg {

static thread_local std::string g_error;

static int fail(const std::string& msg) {
    g_error = msg;
    return 1;
}

#define PG_HIP(call)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) return fail(std::string(#call) + ": " + hipGetErrorString(e_));    \
    } while (0)

// ------------------------------------------------------------------------------------------------
// Atlas
// ------------------------------------------------------------------------------------------------
Atlas::~Atlas() {
    if (d_texels_) hipFree(d_texels_);
    if (d_desc_) hipFree(d_desc_);
    if (d_ranks_) hipFree(d_ranks_);
}

bool Atlas::load(const std::string& root, const std::vector<std::string>& names, std::string& err) {
    texels_.clear();
    desc_.clear();
    for (const auto& name : names) {
        Image img;
        if (!decode_png_file(root + "/" + name, img, err)) return false;
        int4 d;
        d.x = static_cast<int>(texels_.size());
        d.y = img.w;
        d.z = img.h;
        // .w bit 0: some texel is not opaque (alpha < 255) — under this texture something else can show, and the row
        // composer may have to blend; bit 1: more than 3 % of them are not — a frame that shows this texture is unlikely
        // to get away with its one-texel-per-pixel attempt (pg_render.h compose_rows), so it does not try.
        d.w = 0;
        size_t not_opaque = 0;
        for (size_t k = 3; k < img.rgba.size(); k += 4) not_opaque += img.rgba[k] != 255;
        if (not_opaque) d.w |= 1;
        if (not_opaque * 100 > size_t(img.w) * img.h * 3) d.w |= 2;
        // bit 2: some texel is translucent (alpha neither 0 nor 255); bit 3: every texel outside the central half of
        // the texture (rows and columns [size/4, 3·size/4)) is fully transparent — what chaser.hip asks of its point
        // sprite before it lets the points join the tile layer.
        bool clear_rim = true;
        for (int y = 0; y < img.h; y++)
            for (int x = 0; x < img.w; x++) {
                const uint8_t a = img.rgba[(size_t(y) * img.w + x) * 4 + 3];
                if (a != 0 && a != 255) d.w |= 4;
                const bool inner = y >= img.h / 4 && y < img.h - img.h / 4 && x >= img.w / 4 && x < img.w - img.w / 4;
                if (!inner && a != 0) clear_rim = false;
            }
        if (clear_rim) d.w |= 8;
        desc_.push_back(d);
        size_t base = texels_.size();
        texels_.resize(base + size_t(img.w) * img.h);
        std::memcpy(&texels_[base], img.rgba.data(), size_t(img.w) * img.h * 4);
        // A texel with alpha 0 never reaches a pixel (raster spec S4), whatever colour the PNG left in it: store it as
        // the word 0, so that "nothing here" is one value (pg_render.h compose_rows picks the last drawn candidate
        // with an unsigned maximum).
        for (size_t k = base; k < texels_.size(); k++)
            if ((texels_[k] >> 24) == 0) texels_[k] = 0;
    }
    return true;
}

bool Atlas::upload(std::string& err) {
    if (texels_.size() * 4 >= 0x10000000ull) {  // byte offsets share their word with two rank bits (pg_render.h kRank)
        err = "atlas exceeds 256 MiB (pg_render.h kRank, kNoTexel)";
        return false;
    }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_texels_), texels_.size() * 4);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_desc_), desc_.size() * sizeof(int4));
    if (e == hipSuccess) e = hipMemcpy(d_texels_, texels_.data(), texels_.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_desc_, desc_.data(), desc_.size() * sizeof(int4), hipMemcpyHostToDevice);
    std::vector<uint8_t> ranks(kRankTableBytes);
    build_equal_key_ranks(ranks.data());
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_ranks_), ranks.size());
    if (e == hipSuccess) e = hipMemcpy(d_ranks_, ranks.data(), ranks.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        err = std::string("atlas upload: ") + hipGetErrorString(e);
        return false;
    }
    return true;
}

// Several engines on one GPU want more hardware queues than the HIP runtime's default of four (streams that share a
// queue run their kernels one after the other: the mixed seven-game workload, fourteen streams, gains 9.5 % with twelve
// or more).  The runtime reads GPU_MAX_HW_QUEUES when it initialises, at the first HIP call of the process; loading this
// library is usually earlier.  A value the caller has set is left alone.
// When the runtime is up already (the embedding process made a HIP call first, or rocprofv3's preloaded tool did) the
// default comes too late and is silently without effect: that is noticed here — an initialised ROCm runtime holds
// /dev/kfd open — and said once, on stderr, by the first pgv_make that would have profited (the third engine on a
// device: a single game's two streams are served by the runtime's own four queues).
static bool g_queue_default_too_late = false;
static bool rocm_runtime_is_up() {
    DIR* d = opendir("/proc/self/fd");
    if (!d) return false;
    bool up = false;
    while (dirent* ent = readdir(d)) {
        char link[64], target[64];
        std::snprintf(link, sizeof link, "/proc/self/fd/%s", ent->d_name);
        const ssize_t n = readlink(link, target, sizeof target - 1);
        if (n > 0) {
            target[n] = 0;
            if (!std::strcmp(target, "/dev/kfd")) up = true;
        }
    }
    closedir(d);
    return up;
}
__attribute__((constructor)) static void runtime_defaults() {
    if (std::getenv("GPU_MAX_HW_QUEUES")) return;  // the caller's choice
    g_queue_default_too_late = rocm_runtime_is_up();
    setenv("GPU_MAX_HW_QUEUES", "16", /*overwrite=*/0);
}
static std::atomic<int> g_live_engines[16];
static void note_engine_made(int device) {
    if (device < 0 || device >= 16) return;
    static std::atomic<bool> said{false};
    if (++g_live_engines[device] >= 3 && g_queue_default_too_late && !said.exchange(true))
        std::fprintf(stderr,
                     "procgen2_hip: %d engines on device %d, but the HIP runtime was initialised before this library was "
                     "loaded, so its default GPU_MAX_HW_QUEUES=16 came too late and the streams of these engines share the "
                     "runtime's default hardware queues (kernels of different engines run in turn; measured: -9 %% on the "
                     "seven-game workload).  Export GPU_MAX_HW_QUEUES=16 before the process starts.\n",
                     g_live_engines[device].load(), device);
}

static std::string asset_root() {
    if (const char* env = std::getenv("PROCGEN2_ASSETS")) return env;
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(&asset_root), &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        size_t slash = p.rfind('/');
        std::string dir = slash == std::string::npos ? "." : p.substr(0, slash);
        return dir + "/../assets";
    }
    return "assets";
}

static const char* const kGameNames[kNumGames] = {"coinrun", "maze", "bossfight", "climber", "caveflyer", "chaser", "jumper"};

// (game, distribution mode) → compiled variant.  The first row of a game is its PGV_MODE_DEFAULT, i.e. the reference's
// compile-time `Config` (SURVEY.md §5 "config / flags").  coinrun: `easy_mode` only feeds `allow_monsters`, which
// nothing reads (coinrun/tilemap.cpp:148), so both modes are the same game.
struct Variant {
    int game, mode;
    std::unique_ptr<Game> (*make)();
};
static const Variant kVariants[] = {
    {kGameCoinrun, PGV_MODE_HARD, make_coinrun_v0},     {kGameCoinrun, PGV_MODE_EASY, make_coinrun_v0},
    {kGameMaze, PGV_MODE_HARD, make_maze_v0},           {kGameMaze, PGV_MODE_EASY, make_maze_v1},
    {kGameMaze, PGV_MODE_MEMORY, make_maze_v2},
    {kGameBossfight, PGV_MODE_HARD, make_bossfight_v0}, {kGameBossfight, PGV_MODE_EASY, make_bossfight_v1},
    {kGameClimber, PGV_MODE_HARD, make_climber_v0},     {kGameClimber, PGV_MODE_EASY, make_climber_v1},
    {kGameCaveflyer, PGV_MODE_HARD, make_caveflyer_v0}, {kGameCaveflyer, PGV_MODE_EASY, make_caveflyer_v1},
    {kGameCaveflyer, PGV_MODE_MEMORY, make_caveflyer_v2},
    {kGameChaser, PGV_MODE_EASY, make_chaser_v0},       {kGameChaser, PGV_MODE_HARD, make_chaser_v1},
    {kGameChaser, PGV_MODE_EXTREME, make_chaser_v2},
    {kGameJumper, PGV_MODE_HARD, make_jumper_v0},       {kGameJumper, PGV_MODE_EASY, make_jumper_v1},
    {kGameJumper, PGV_MODE_MEMORY, make_jumper_v2},
};

static const Variant* find_variant(int game, int mode) {
    for (const Variant& v : kVariants)
        if (v.game == game && (mode == PGV_MODE_DEFAULT || v.mode == mode)) return &v;
    return nullptr;
}

}  // namespace pg

// ------------------------------------------------------------------------------------------------
// The vector env object
// ------------------------------------------------------------------------------------------------
struct pgv_env {
    int n = 0, device = 0, env_offset = 0, mode = 0;
    uint32_t game_flags = 0;
    uint32_t step_index = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::unique_ptr<pg::Game> game;
    pg::Atlas atlas;
    void* d_state = nullptr;
    bool counted = false;       // among g_live_engines (note_engine_made)
    void* d_scratch = nullptr;  // Game::scratch_bytes: per-frame hand-over between a game's kernels, not part of any snapshot
    uint8_t* d_obs = nullptr;
    float* d_reward = nullptr;
    uint8_t* d_done = nullptr;
    uint8_t* d_pending = nullptr;
    int32_t* d_host_i32 = nullptr;  // staging for the *_host entry points
    uint8_t* d_host_u8 = nullptr;
    bool own_obs = false, own_reward = false, own_done = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // level prefetch (pg_prefetch.h): generator launches on a side stream, ordered behind the main stream by events
    hipStream_t side = nullptr;
    hipEvent_t side_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int side_ev_next = 0, since_pregen = 0;
    int64_t generator_launches = 0;  // pgv_generator_launches
    // auto-resets beside the logic kernels (pg_engine.h Game::reset_stream)
    hipStream_t reset_stream = nullptr;
    hipEvent_t reset_fork = nullptr, reset_join = nullptr;

    pg::StepIO io() const { return {d_obs, d_reward, d_done, d_pending}; }
};

// Queue one generator launch behind everything the main stream holds so far.  In the step loop: every
// pregen_every()-th step (pg_engine.h: the game says how long a queued slot may wait) — one launch serves every slot
// queued by then, and a launch costs its single-env latency (ms) even for one env, so launching per step would only pile
// launches up.  The cadence is a count of steps and nothing else: until round 5 a launch was also made whenever the
// HOST saw the side stream idle (hipStreamQuery), which a benchmark loop that runs hundreds of steps ahead of the device
// never does and a caller that synchronises every step always does — two callers, two cadences, and the second one
// the slower for coinrun, climber and jumper (tests/test_levels.py::test_generator_cadence_…).
static void pregen(pgv_env* e, bool bulk, bool force) {
    if (!e->side) return;
    e->since_pregen++;
    if (!force && e->since_pregen < e->game->pregen_every()) return;
    hipEvent_t ev = e->side_ev[e->side_ev_next];
    e->side_ev_next = (e->side_ev_next + 1) % 8;
    if (hipEventRecord(ev, e->stream) != hipSuccess || hipStreamWaitEvent(e->side, ev, 0) != hipSuccess) return;
    if (!e->game->launch_pregen(e->side, bulk)) {
        hipStreamDestroy(e->side);  // this game generates its levels inside the step
        e->side = nullptr;
        return;
    }
    e->generator_launches++;
    e->since_pregen = 0;
}

using pg::fail;

extern "C" {

const char* pgv_last_error(void) { return pg::g_error.c_str(); }

const char* pgv_game_name(int32_t id) { return (id >= 0 && id < pg::kNumGames) ? pg::kGameNames[id] : nullptr; }

int32_t pgv_game_id(const char* name) {
    for (int i = 0; i < pg::kNumGames; i++)
        if (name && !std::strcmp(name, pg::kGameNames[i])) return i;
    return -1;
}

int32_t pgv_synthetic_action(uint32_t run_seed, uint32_t step_index, uint32_t global_env) {
    return pg::synthetic_action(run_seed, step_index, global_env);
}

void pgv_close(pgv_env* e) {
    if (!e) return;
    if (e->counted && e->device >= 0 && e->device < 16) --pg::g_live_engines[e->device];
    hipSetDevice(e->device);
    if (e->stream) hipStreamSynchronize(e->stream);
    if (e->side) {
        hipStreamSynchronize(e->side);
        hipStreamDestroy(e->side);
    }
    for (auto& ev : e->side_ev)
        if (ev) hipEventDestroy(ev);
    for (auto& ev : e->ev)
        if (ev) hipEventDestroy(ev);
    if (e->reset_stream) {
        hipStreamSynchronize(e->reset_stream);
        hipStreamDestroy(e->reset_stream);
    }
    if (e->reset_fork) hipEventDestroy(e->reset_fork);
    if (e->reset_join) hipEventDestroy(e->reset_join);
    if (e->d_state) hipFree(e->d_state);
    if (e->d_scratch) hipFree(e->d_scratch);
    if (e->own_obs && e->d_obs) hipFree(e->d_obs);
    if (e->own_reward && e->d_reward) hipFree(e->d_reward);
    if (e->own_done && e->d_done) hipFree(e->d_done);
    if (e->d_pending) hipFree(e->d_pending);
    if (e->d_host_i32) hipFree(e->d_host_i32);
    if (e->d_host_u8) hipFree(e->d_host_u8);
    if (e->own_stream && e->stream) hipStreamDestroy(e->stream);
    delete e;
}

// The device state blob: the game's SoA state followed by the level plan's two per-env words (pg_engine.h
// LevelPlan), so that a snapshot of the blob carries them.
static size_t game_state_bytes(const pgv_env* e) { return (e->game->state_bytes(e->n) + 255) / 256 * 256; }
static size_t state_blob_bytes(const pgv_env* e) { return game_state_bytes(e) + size_t(e->n) * 8; }

int32_t pgv_make(const char* game, int32_t num_envs, int32_t device, uint32_t seed_base, int32_t env_offset,
                 void* stream, pgv_env** out) {
    return pgv_make_levels(game, num_envs, device, seed_base, env_offset, stream, 0, 0, out);
}

int32_t pgv_make_levels(const char* game, int32_t num_envs, int32_t device, uint32_t seed_base, int32_t env_offset,
                        void* stream, int32_t num_levels, int32_t start_level, pgv_env** out) {
    pgv_config cfg{};
    cfg.struct_size = sizeof(pgv_config);
    cfg.game = game;
    cfg.num_envs = num_envs;
    cfg.device = device;
    cfg.seed_base = seed_base;
    cfg.env_offset = env_offset;
    cfg.stream = stream;
    cfg.num_levels = num_levels;
    cfg.start_level = start_level;
    cfg.mode = PGV_MODE_DEFAULT;
    return pgv_make_config(&cfg, out);
}

uint32_t pgv_game_modes(int32_t game_id) {
    uint32_t bits = 0;
    for (const pg::Variant& v : pg::kVariants)
        if (v.game == game_id) bits |= 1u << v.mode;
    return bits;
}

int32_t pgv_mode(pgv_env* e) { return e ? e->mode : -1; }

int32_t pgv_make_config(const pgv_config* cfg, pgv_env** out) {
    if (!out) return fail("pgv_make: out is NULL");
    *out = nullptr;
    if (!cfg || cfg->struct_size < sizeof(pgv_config)) return fail("pgv_make_config: config is NULL or struct_size too small");
    const char* game = cfg->game;
    const int32_t num_envs = cfg->num_envs, device = cfg->device, env_offset = cfg->env_offset;
    const uint32_t seed_base = cfg->seed_base;
    void* stream = cfg->stream;
    const int32_t num_levels = cfg->num_levels, start_level = cfg->start_level;
    if (num_levels < 0) return fail("pgv_make: num_levels must be >= 0 (0 = every level is new)");
    const int gid = pgv_game_id(game);
    if (gid < 0) return fail(std::string("pgv_make: unknown game '") + (game ? game : "(null)") + "'");
    const pg::Variant* variant = pg::find_variant(gid, cfg->mode);
    if (!variant)
        return fail(std::string("pgv_make: game '") + game + "' has no distribution mode " + std::to_string(cfg->mode) +
                    " (see pgv_game_modes)");
    if (num_envs <= 0) return fail("pgv_make: num_envs must be positive");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail("pgv_make: no HIP device available (the engine has no CPU path)");
    if (device < 0 || device >= count) return fail("pgv_make: device index out of range");
    PG_HIP(hipSetDevice(device));

    // every failure path below releases what was created so far (streams, events, atlas, device memory)
    std::unique_ptr<pgv_env, void (*)(pgv_env*)> e(new pgv_env(), pgv_close);
    e->n = num_envs;
    e->device = device;
    e->env_offset = env_offset;
    e->game = variant->make();
    e->mode = variant->mode;
    e->game_flags = cfg->game_flags;
    if (!e->game->set_game_flags(cfg->game_flags))
        return fail(std::string("pgv_make: game '") + game + "' does not take game_flags " + std::to_string(cfg->game_flags));
    if (stream) {
        e->stream = static_cast<hipStream_t>(stream);
    } else {
        PG_HIP(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        e->own_stream = true;
    }
    for (auto& ev : e->ev) PG_HIP(hipEventCreate(&ev));
    PG_HIP(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));  // the level generator's stream
    for (auto& ev : e->side_ev) PG_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    if (e->game->resets_beside_logic()) {
        {   // the few long wavefronts of the in-step level kernel go first; the logic kernel's many short ones fill in around them
            int least = 0, greatest = 0;
            PG_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
            PG_HIP(hipStreamCreateWithPriority(&e->reset_stream, hipStreamNonBlocking, greatest));
        }
        PG_HIP(hipEventCreateWithFlags(&e->reset_fork, hipEventDisableTiming));
        PG_HIP(hipEventCreateWithFlags(&e->reset_join, hipEventDisableTiming));
        e->game->reset_stream = e->reset_stream;
    }

    std::string err;
    if (!e->atlas.load(pg::asset_root(), e->game->texture_names(), err)) return fail("pgv_make: " + err);
    err = e->game->check_atlas(e->atlas.sizes());
    if (!err.empty()) return fail("pgv_make: " + err);
    e->game->extend_atlas(e->atlas);
    if (!e->atlas.upload(err)) return fail("pgv_make: " + err);
    const size_t sb = state_blob_bytes(e.get());
    PG_HIP(hipMalloc(&e->d_state, sb));
    PG_HIP(hipMemsetAsync(e->d_state, 0, sb, e->stream));
    PG_HIP(hipMalloc(reinterpret_cast<void**>(&e->d_obs), size_t(num_envs) * pg::kObsBytes));
    PG_HIP(hipMalloc(reinterpret_cast<void**>(&e->d_reward), size_t(num_envs) * 4));
    PG_HIP(hipMalloc(reinterpret_cast<void**>(&e->d_done), size_t(num_envs)));
    PG_HIP(hipMalloc(reinterpret_cast<void**>(&e->d_pending), size_t(num_envs)));
    e->own_obs = e->own_reward = e->own_done = true;
    PG_HIP(hipMemsetAsync(e->d_obs, 0, size_t(num_envs) * pg::kObsBytes, e->stream));
    PG_HIP(hipMemsetAsync(e->d_reward, 0, size_t(num_envs) * 4, e->stream));
    PG_HIP(hipMemsetAsync(e->d_done, 0, size_t(num_envs), e->stream));
    PG_HIP(hipMemsetAsync(e->d_pending, 0, size_t(num_envs), e->stream));

    e->game->bind(e->d_state, num_envs, e->atlas.view());
    if (const size_t scratch = e->game->scratch_bytes(num_envs)) {
        PG_HIP(hipMalloc(&e->d_scratch, scratch));
        PG_HIP(hipMemsetAsync(e->d_scratch, 0, scratch, e->stream));
        e->game->bind_scratch(e->d_scratch, num_envs);
    }
    {
        uint32_t* words = reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(e->d_state) + game_state_bytes(e.get()));
        e->game->plan = pg::LevelPlan{num_levels, start_level, words, words + num_envs};
    }
    e->game->launch_make(e->stream, seed_base, env_offset);
    PG_HIP(hipGetLastError());
    PG_HIP(hipStreamSynchronize(e->stream));
    pregen(e.get(), true, true);
    pg::note_engine_made(device);
    e->counted = true;
    *out = e.release();
    return 0;
}

int32_t pgv_reset(pgv_env* e, const uint8_t* d_mask, const int32_t* d_seeds) {
    if (!e) return fail("pgv_reset: env is NULL");
    PG_HIP(hipSetDevice(e->device));
    e->game->launch_reset(e->stream, d_mask, d_seeds, e->io());
    e->game->launch_prepass(e->stream, d_mask);
    e->game->launch_render(e->stream, d_mask, e->io());
    pregen(e, true, true);
    PG_HIP(hipGetLastError());
    return 0;
}

// One step's launches.  The launch status is read after each group of launches: hipGetLastError reports (and clears)
// only the most recent error.
// `after_logic` / `before_render` / `after_render`: optional events (measurement only) — behind the logic kernels (and the
// level generator's launch on the side stream), behind the render pre-pass = in front of the render kernel, behind the
// render launch (jumper: its list kernel included).  What follows after_render inside a step is the late pass of a game
// that resets beside its render kernel (chaser).
static int32_t step_impl(pgv_env* e, const int32_t* d_actions, uint32_t run_seed, hipEvent_t before_render = nullptr,
                         hipEvent_t after_render = nullptr, hipEvent_t after_logic = nullptr) {
    // hipGetLastError is sticky and per thread: whatever the embedding application left behind (a stream query's
    // NotReady, say) is not this step's; start clean.
    (void)hipGetLastError();
    const bool forked = e->reset_stream != nullptr;
    if (forked) {  // fork: the auto-resets of this step go beside its logic and render kernels
        PG_HIP(hipEventRecord(e->reset_fork, e->stream));
        PG_HIP(hipStreamWaitEvent(e->reset_stream, e->reset_fork, 0));
    }
    hipError_t status = hipSuccess;
    auto launched = [&]() {  // the status of the launches since the last call
        const hipError_t now = hipGetLastError();
        if (status == hipSuccess && now != hipSuccess) status = now;
    };
    // The step counter moves as soon as the logic launch has been issued, whatever becomes of the rest: the logic kernels
    // write "reset due in step index + 1" into the pending bytes (pg_prefetch.h reset_due_mark), and a counter that stayed
    // behind after a failed later launch would have the next call read those marks as another step's.
    const uint32_t step_index = e->step_index++;
    e->game->launch_logic(e->stream, d_actions, run_seed, step_index, e->env_offset, e->io());
    launched();
    pregen(e, false, false);  // before the render launch: the generator overlaps it
    launched();
    if (after_logic && status == hipSuccess) status = hipEventRecord(after_logic, e->stream);
    if (status == hipSuccess) {
        e->game->launch_prepass(e->stream, nullptr);
        launched();
    }
    if (before_render && status == hipSuccess) status = hipEventRecord(before_render, e->stream);
    if (status == hipSuccess) {
        e->game->launch_render_step(e->stream, e->io());
        launched();
    }
    if (after_render && status == hipSuccess) status = hipEventRecord(after_render, e->stream);
    if (forked) {  // join — on every way out, so the two streams stay consistent — and the late frames of the reset envs
        hipError_t j = hipEventRecord(e->reset_join, e->reset_stream);
        if (j == hipSuccess) j = hipStreamWaitEvent(e->stream, e->reset_join, 0);
        if (status == hipSuccess) status = j;
        if (status == hipSuccess) {
            e->game->launch_render_late(e->stream, e->io());
            launched();
        }
    }
    if (status != hipSuccess) return fail(std::string("step: ") + hipGetErrorString(status));
    return 0;
}

int32_t pgv_step(pgv_env* e, const int32_t* d_actions) {
    if (!e) return fail("pgv_step: env is NULL");
    if (!d_actions) return fail("pgv_step: actions is NULL");
    PG_HIP(hipSetDevice(e->device));
    return step_impl(e, d_actions, 0);
}

int32_t pgv_step_synthetic(pgv_env* e, uint32_t run_seed) {
    if (!e) return fail("pgv_step_synthetic: env is NULL");
    PG_HIP(hipSetDevice(e->device));
    return step_impl(e, nullptr, run_seed);
}

int32_t pgv_step_synthetic_many(pgv_env* const* envs, int32_t count, int32_t steps, uint32_t run_seed) {
    if (!envs || count < 1 || steps < 1) return fail("pgv_step_synthetic_many: bad arguments");
    for (int32_t k = 0; k < count; k++)
        if (!envs[k]) return fail("pgv_step_synthetic_many: env is NULL");
    for (int32_t s = 0; s < steps; s++)
        for (int32_t k = 0; k < count; k++) {
            PG_HIP(hipSetDevice(envs[k]->device));
            if (step_impl(envs[k], nullptr, run_seed))  // (the envs before k have taken step s, those from k on have not)
                return fail(pg::g_error + " (pgv_step_synthetic_many: env " + std::to_string(k) + " of " + std::to_string(count) +
                            ", step " + std::to_string(s) + " of " + std::to_string(steps) + ")");
        }
    return 0;
}

static int32_t ensure_staging(pgv_env* e) {
    if (!e->d_host_i32) PG_HIP(hipMalloc(reinterpret_cast<void**>(&e->d_host_i32), size_t(e->n) * 4));
    if (!e->d_host_u8) PG_HIP(hipMalloc(reinterpret_cast<void**>(&e->d_host_u8), size_t(e->n)));
    return 0;
}

int32_t pgv_step_host(pgv_env* e, const int32_t* h_actions) {
    if (!e) return fail("pgv_step_host: env is NULL");
    if (!h_actions) return fail("pgv_step_host: actions is NULL");
    PG_HIP(hipSetDevice(e->device));
    if (ensure_staging(e)) return 1;
    PG_HIP(hipMemcpyAsync(e->d_host_i32, h_actions, size_t(e->n) * 4, hipMemcpyHostToDevice, e->stream));
    PG_HIP(hipStreamSynchronize(e->stream));  // the host buffer is the caller's: do not keep reading it
    return pgv_step(e, e->d_host_i32);
}

int32_t pgv_reset_host(pgv_env* e, const uint8_t* h_mask, const int32_t* h_seeds) {
    if (!e) return fail("pgv_reset_host: env is NULL");
    PG_HIP(hipSetDevice(e->device));
    if (ensure_staging(e)) return 1;
    if (h_mask) PG_HIP(hipMemcpyAsync(e->d_host_u8, h_mask, size_t(e->n), hipMemcpyHostToDevice, e->stream));
    if (h_seeds) PG_HIP(hipMemcpyAsync(e->d_host_i32, h_seeds, size_t(e->n) * 4, hipMemcpyHostToDevice, e->stream));
    PG_HIP(hipStreamSynchronize(e->stream));
    int rc = pgv_reset(e, h_mask ? e->d_host_u8 : nullptr, h_seeds ? e->d_host_i32 : nullptr);
    if (rc) return rc;
    PG_HIP(hipStreamSynchronize(e->stream));  // staging buffers are reused by the next *_host call
    return 0;
}

int32_t pgv_decode_png(const char* path, int32_t* w, int32_t* h, uint8_t* h_rgba, int64_t cap) {
    pg::Image img;
    std::string err;
    if (!path || !pg::decode_png_file(path, img, err)) return fail("pgv_decode_png: " + err);
    if (w) *w = img.w;
    if (h) *h = img.h;
    if (h_rgba && cap > 0) {
        size_t nbytes = img.rgba.size() < static_cast<size_t>(cap) ? img.rgba.size() : static_cast<size_t>(cap);
        std::memcpy(h_rgba, img.rgba.data(), nbytes);
    }
    return 0;
}

int64_t pgv_generator_launches(pgv_env* e) { return e ? e->generator_launches : -1; }

int32_t pgv_sync(pgv_env* e) {
    if (!e) return fail("pgv_sync: env is NULL");
    PG_HIP(hipSetDevice(e->device));
    PG_HIP(hipStreamSynchronize(e->stream));
    return 0;
}

// Whole-batch snapshot / restore (SURVEY.md §8f-4, §5 checkpoint/resume): everything a rollout depends on — the game
// state blob (live state, generator chains, prefetched levels and their slot words), reward / done / pending, the
// step counter and the observation slab — as one host buffer.  Both streams are drained first, so no slot is busy.
struct SnapshotHeader {
    uint32_t magic, game;
    int32_t n, env_offset;
    uint64_t state_bytes;
    uint32_t step_index;
    int32_t num_levels, start_level, mode;
    uint32_t game_flags, reserved;
};
// "PGN" + the version of the state-blob layouts: bump it whenever any game's State / Level / field enums change — equal
// state_bytes does not mean equal layout (sizes are rounded to 256 bytes), and an old blob would load silently.
// 2: round 2.  3: round 3 (per-env contiguous rings and entity tables in bossfight, caveflyer, chaser, climber; caveflyer's
// wall-bit columns and hazard places).  4: round 5/6 (the pending byte carries the parity of the step it is about —
// pg_prefetch.h reset_due_mark / reset_served_mark — where it used to be 0 / 1; coinrun's hazard hand-off left the blob).
static constexpr uint32_t kSnapshotMagic = 0x50474e34u;

static size_t snapshot_bytes(const pgv_env* e) {
    return sizeof(SnapshotHeader) + state_blob_bytes(e) + size_t(e->n) * (4 + 1 + 1) +
           size_t(e->n) * pg::kObsBytes;
}

int64_t pgv_snapshot_bytes(pgv_env* e) { return e ? static_cast<int64_t>(snapshot_bytes(e)) : -1; }

int32_t pgv_save_state(pgv_env* e, void* h_buffer, int64_t capacity) {
    if (!e || !h_buffer) return fail("pgv_save_state: NULL argument");
    if (capacity < static_cast<int64_t>(snapshot_bytes(e))) return fail("pgv_save_state: buffer too small");
    PG_HIP(hipSetDevice(e->device));
    e->game->prepare_save(e->stream);
    PG_HIP(hipGetLastError());
    PG_HIP(hipStreamSynchronize(e->stream));
    if (e->side) PG_HIP(hipStreamSynchronize(e->side));
    uint8_t* out = static_cast<uint8_t*>(h_buffer);
    SnapshotHeader hd{kSnapshotMagic, static_cast<uint32_t>(pgv_game_id(e->game->name())), e->n, e->env_offset,
                      state_blob_bytes(e), e->step_index, e->game->plan.num_levels, e->game->plan.start_level, e->mode, e->game_flags, 0};
    std::memcpy(out, &hd, sizeof(hd));
    out += sizeof(hd);
    PG_HIP(hipMemcpy(out, e->d_state, hd.state_bytes, hipMemcpyDeviceToHost));
    out += hd.state_bytes;
    PG_HIP(hipMemcpy(out, e->d_reward, size_t(e->n) * 4, hipMemcpyDeviceToHost));
    out += size_t(e->n) * 4;
    PG_HIP(hipMemcpy(out, e->d_done, size_t(e->n), hipMemcpyDeviceToHost));
    out += size_t(e->n);
    PG_HIP(hipMemcpy(out, e->d_pending, size_t(e->n), hipMemcpyDeviceToHost));
    out += size_t(e->n);
    PG_HIP(hipMemcpy(out, e->d_obs, size_t(e->n) * pg::kObsBytes, hipMemcpyDeviceToHost));
    return 0;
}

int32_t pgv_load_state(pgv_env* e, const void* h_buffer, int64_t size) {
    if (!e || !h_buffer) return fail("pgv_load_state: NULL argument");
    if (size < static_cast<int64_t>(snapshot_bytes(e))) return fail("pgv_load_state: buffer too small for this env");
    const uint8_t* in = static_cast<const uint8_t*>(h_buffer);
    SnapshotHeader hd;
    std::memcpy(&hd, in, sizeof(hd));
    if (hd.magic != kSnapshotMagic || hd.game != static_cast<uint32_t>(pgv_game_id(e->game->name())) || hd.n != e->n ||
        hd.env_offset != e->env_offset || hd.state_bytes != state_blob_bytes(e) ||
        hd.num_levels != e->game->plan.num_levels || hd.start_level != e->game->plan.start_level || hd.mode != e->mode ||
        hd.game_flags != e->game_flags)
        return fail("pgv_load_state: snapshot of a different env (game, mode, size, shard or level set)");
    PG_HIP(hipSetDevice(e->device));
    PG_HIP(hipStreamSynchronize(e->stream));
    if (e->side) PG_HIP(hipStreamSynchronize(e->side));
    in += sizeof(hd);
    PG_HIP(hipMemcpy(e->d_state, in, hd.state_bytes, hipMemcpyHostToDevice));
    in += hd.state_bytes;
    PG_HIP(hipMemcpy(e->d_reward, in, size_t(e->n) * 4, hipMemcpyHostToDevice));
    in += size_t(e->n) * 4;
    PG_HIP(hipMemcpy(e->d_done, in, size_t(e->n), hipMemcpyHostToDevice));
    in += size_t(e->n);
    PG_HIP(hipMemcpy(e->d_pending, in, size_t(e->n), hipMemcpyHostToDevice));
    in += size_t(e->n);
    PG_HIP(hipMemcpy(e->d_obs, in, size_t(e->n) * pg::kObsBytes, hipMemcpyHostToDevice));
    e->step_index = hd.step_index;
    e->game->state_loaded(e->stream);  // (on the env's stream: ordered in front of the next step)
    pregen(e, true, true);  // queued shadow slots of the snapshot get their generator launch
    return 0;
}

uint8_t* pgv_obs(pgv_env* e) { return e ? e->d_obs : nullptr; }
float* pgv_reward(pgv_env* e) { return e ? e->d_reward : nullptr; }
uint8_t* pgv_done(pgv_env* e) { return e ? e->d_done : nullptr; }
int32_t pgv_num_envs(pgv_env* e) { return e ? e->n : 0; }
int32_t pgv_device(pgv_env* e) { return e ? e->device : -1; }
void* pgv_stream(pgv_env* e) { return e ? static_cast<void*>(e->stream) : nullptr; }

int32_t pgv_bind_outputs(pgv_env* e, uint8_t* d_obs, float* d_reward, uint8_t* d_done) {
    if (!e) return fail("pgv_bind_outputs: env is NULL");
    PG_HIP(hipSetDevice(e->device));
    PG_HIP(hipStreamSynchronize(e->stream));
    if (d_obs) {
        PG_HIP(hipMemcpy(d_obs, e->d_obs, size_t(e->n) * pg::kObsBytes, hipMemcpyDeviceToDevice));
        if (e->own_obs) hipFree(e->d_obs);
        e->d_obs = d_obs;
        e->own_obs = false;
    }
    if (d_reward) {
        PG_HIP(hipMemcpy(d_reward, e->d_reward, size_t(e->n) * 4, hipMemcpyDeviceToDevice));
        if (e->own_reward) hipFree(e->d_reward);
        e->d_reward = d_reward;
        e->own_reward = false;
    }
    if (d_done) {
        PG_HIP(hipMemcpy(d_done, e->d_done, size_t(e->n), hipMemcpyDeviceToDevice));
        if (e->own_done) hipFree(e->d_done);
        e->d_done = d_done;
        e->own_done = false;
    }
    return 0;
}

int32_t pgv_copy_out(pgv_env* e, uint8_t* h_obs, float* h_reward, uint8_t* h_done) {
    if (!e) return fail("pgv_copy_out: env is NULL");
    PG_HIP(hipSetDevice(e->device));
    PG_HIP(hipStreamSynchronize(e->stream));
    if (h_obs) PG_HIP(hipMemcpy(h_obs, e->d_obs, size_t(e->n) * pg::kObsBytes, hipMemcpyDeviceToHost));
    if (h_reward) PG_HIP(hipMemcpy(h_reward, e->d_reward, size_t(e->n) * 4, hipMemcpyDeviceToHost));
    if (h_done) PG_HIP(hipMemcpy(h_done, e->d_done, size_t(e->n), hipMemcpyDeviceToHost));
    return 0;
}

// Per-step events of a run of synthetic steps, created before the region starts and destroyed on every way out; they are
// read back after the region so the host never stalls the stream inside it.
namespace {
struct StepEvents {
    std::vector<hipEvent_t> v;
    ~StepEvents() {
        for (hipEvent_t p : v)
            if (p) hipEventDestroy(p);
    }
};
}  // namespace

namespace pg {
// `steps` synthetic steps of `count` envs side by side (step s of every env enqueued before step s + 1 of any, each on
// its own stream, as pgv_step_synthetic_many), every step cut into its phases by events on the env's stream.
// out[5 * k + j]: `steps` floats of env k — j = 0 the whole step, 1 logic, 2 pre-pass, 3 render, 4 late — or NULL.
static int32_t step_phases_many(pgv_env* const* envs, int count, int steps, uint32_t run_seed, float* const* out) {
    if (steps < 1 || steps > (1 << 20)) return fail("pgv_step_phases: steps must be in 1..1048576");
    std::vector<StepEvents> ev(static_cast<size_t>(count));  // per step: start, after logic, before render, after render; one more at the end
    for (int k = 0; k < count; k++) {
        PG_HIP(hipSetDevice(envs[k]->device));
        ev[k].v.assign(size_t(steps) * 4 + 1, nullptr);
        for (auto& p : ev[k].v) PG_HIP(hipEventCreate(&p));
    }
    for (int s = 0; s < steps; s++)
        for (int k = 0; k < count; k++) {
            pgv_env* e = envs[k];
            PG_HIP(hipSetDevice(e->device));
            PG_HIP(hipEventRecord(ev[k].v[4 * s], e->stream));
            if (step_impl(e, nullptr, run_seed, ev[k].v[4 * s + 2], ev[k].v[4 * s + 3], ev[k].v[4 * s + 1])) return 1;
        }
    for (int k = 0; k < count; k++) {
        PG_HIP(hipSetDevice(envs[k]->device));
        PG_HIP(hipEventRecord(ev[k].v[size_t(steps) * 4], envs[k]->stream));
    }
    static const int kFrom[5] = {0, 0, 1, 2, 3}, kTo[5] = {4, 1, 2, 3, 4};
    for (int k = 0; k < count; k++) {
        PG_HIP(hipSetDevice(envs[k]->device));
        PG_HIP(hipEventSynchronize(ev[k].v[size_t(steps) * 4]));
        for (int j = 0; j < 5; j++) {
            float* to = out[5 * k + j];
            if (!to) continue;
            for (int s = 0; s < steps; s++) PG_HIP(hipEventElapsedTime(&to[s], ev[k].v[4 * s + kFrom[j]], ev[k].v[4 * s + kTo[j]]));
        }
    }
    return 0;
}
}  // namespace pg

int32_t pgv_timed_steps(pgv_env* e, int32_t steps, uint32_t run_seed, double* total_ms, double* render_kernel_ms) {
    if (!e) return fail("pgv_timed_steps: env is NULL");
    if (steps < 1 || steps > (1 << 20)) return fail("pgv_timed_steps: steps must be in 1..1048576");
    PG_HIP(hipSetDevice(e->device));
    // render_kernel_ms == NULL: nothing but the steps between the region's two events (the form bench.py's `value` uses).
    StepEvents pairs;
    if (render_kernel_ms) {
        pairs.v.assign(size_t(steps) * 2, nullptr);
        for (auto& p : pairs.v) PG_HIP(hipEventCreate(&p));
    }
    PG_HIP(hipEventRecord(e->ev[0], e->stream));  // whole region on the env's stream
    for (int s = 0; s < steps; s++)
        if (step_impl(e, nullptr, run_seed, render_kernel_ms ? pairs.v[2 * s] : nullptr,
                      render_kernel_ms ? pairs.v[2 * s + 1] : nullptr))
            return 1;
    PG_HIP(hipEventRecord(e->ev[1], e->stream));
    PG_HIP(hipEventSynchronize(e->ev[1]));
    float ms = 0.0f;
    PG_HIP(hipEventElapsedTime(&ms, e->ev[0], e->ev[1]));
    if (total_ms) *total_ms = ms;
    if (render_kernel_ms) {
        double render_sum = 0.0;
        for (int s = 0; s < steps; s++) {
            float k = 0.0f;
            PG_HIP(hipEventElapsedTime(&k, pairs.v[2 * s], pairs.v[2 * s + 1]));
            render_sum += k;
        }
        *render_kernel_ms = render_sum;
    }
    return 0;
}

int32_t pgv_step_times(pgv_env* e, int32_t steps, uint32_t run_seed, float* h_step_ms, float* h_render_ms) {
    return pgv_step_phases(e, steps, run_seed, h_step_ms, nullptr, nullptr, h_render_ms, nullptr);
}

int32_t pgv_step_phases(pgv_env* e, int32_t steps, uint32_t run_seed, float* h_step_ms, float* h_logic_ms, float* h_prepass_ms,
                        float* h_render_ms, float* h_late_ms) {
    if (!e) return fail("pgv_step_phases: env is NULL");
    float* out[5] = {h_step_ms, h_logic_ms, h_prepass_ms, h_render_ms, h_late_ms};
    return pg::step_phases_many(&e, 1, steps, run_seed, out);
}

int32_t pgv_step_phases_many(pgv_env* const* envs, int32_t count, int32_t steps, uint32_t run_seed, float* h_ms) {
    if (!envs || count < 1 || !h_ms) return fail("pgv_step_phases_many: bad arguments");
    for (int32_t k = 0; k < count; k++)
        if (!envs[k]) return fail("pgv_step_phases_many: env is NULL");
    std::vector<float*> out(size_t(count) * 5);
    for (size_t k = 0; k < out.size(); k++) out[k] = h_ms + k * size_t(steps > 0 ? steps : 0);
    return pg::step_phases_many(envs, count, steps, run_seed, out.data());
}

int32_t pgv_render_frame(pgv_env* e, int32_t index, int32_t width, int32_t height, uint8_t* h_rgb) {
    if (!e) return fail("pgv_render_frame: env is NULL");
    if (index < 0 || index >= e->n) return fail("pgv_render_frame: env index out of range");
    if (width < 1 || height < 1 || width > 4096 || height > 4096 || !h_rgb)
        return fail("pgv_render_frame: bad frame size or NULL buffer");
    PG_HIP(hipSetDevice(e->device));
    const size_t px = size_t(width) * height;
    uint32_t* d_px = nullptr;
    PG_HIP(hipMalloc(reinterpret_cast<void**>(&d_px), px * 4));
    if (!e->game->launch_frame(e->stream, index, d_px, width, height)) {
        hipFree(d_px);
        return fail(std::string("pgv_render_frame: not implemented for ") + e->game->name());
    }
    std::vector<uint32_t> host(px);
    hipError_t err = hipGetLastError();
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    if (err == hipSuccess) err = hipMemcpy(host.data(), d_px, px * 4, hipMemcpyDeviceToHost);
    hipFree(d_px);
    if (err != hipSuccess) return fail(std::string("pgv_render_frame: ") + hipGetErrorString(err));
    for (size_t k = 0; k < px; k++) {  // RGBA → RGB, row-major (coinrun.cpp:401-407)
        h_rgb[3 * k + 0] = static_cast<uint8_t>(host[k]);
        h_rgb[3 * k + 1] = static_cast<uint8_t>(host[k] >> 8);
        h_rgb[3 * k + 2] = static_cast<uint8_t>(host[k] >> 16);
    }
    return 0;
}

int32_t pgv_set_debug(pgv_env* e, int32_t flags) {
    if (!e) return fail("pgv_set_debug: env is NULL");
#ifndef PG_ABLATE
    if (flags & ~(1 | pg::kDebugNoPrefetch | pg::kDebugNoPrepass | pg::kDebugFatThirds | pg::kDebugCoinrunNoReach | pg::kDebugChaserSerialMobs))
        return fail("pgv_set_debug: only bit 0 (draw-list replay), bit 8 (no level prefetch), bit 21 (no render pre-pass), bit 23 (every third frame by the complete path), bit 24 (coinrun: hazards the long way) and bit 25 (chaser: enemies the long way) exist in this build");
#endif
    if (e->side) hipStreamSynchronize(e->side);
    e->game->debug_flags = flags;
    return 0;
}

int32_t pgv_dump_state(pgv_env* e, int32_t index, float* h_out, int32_t cap) {
    if (!e || index < 0 || index >= e->n) return -1;
    hipSetDevice(e->device);
    return e->game->dump_state(e->stream, index, h_out, cap);
}

int32_t pgv_dump_tiles(pgv_env* e, int32_t index, uint8_t* h_out, int32_t cap) {
    if (!e || index < 0 || index >= e->n) return -1;
    hipSetDevice(e->device);
    return e->game->dump_tiles(e->stream, index, h_out, cap);
}

// ------------------------------------------------------------------------------------------------
// cenv ABI (include/procgen2_cenv.h) on top of one process-global vector env.
// ------------------------------------------------------------------------------------------------
cenv_make_data make_data;
cenv_reset_data reset_data;
cenv_step_data step_data;
cenv_render_data render_data;

}  // extern "C"

namespace {

struct CenvGlobal {
    pgv_env* env = nullptr;
    int n = 0;
    int window_w = 512, window_h = 512;  // coinrun.cpp:29-30
    cenv_key_value obs_space{}, act_space{};
    cenv_key_value observations[3]{};
    float box_bounds[2] = {0.0f, 255.0f};
    int32_t nvec[1] = {pg::kNumActions};
    std::vector<uint8_t> h_obs, h_done, h_frame;
    std::vector<float> h_reward;
    int32_t* d_actions = nullptr;
    int32_t* d_seeds = nullptr;
    std::vector<int32_t> h_actions;
};
CenvGlobal g;

const int kVersion = 100;  // coinrun.cpp:9

int opt_int(const cenv_option& o, int* out) {
    if (o.value_type == CENV_VALUE_TYPE_INT) {
        *out = o.value.i;
        return 0;
    }
    if (o.value_type == CENV_VALUE_TYPE_DOUBLE) {  // python float → DOUBLE (cenv.py:39-42)
        *out = static_cast<int>(o.value.d);
        return 0;
    }
    return 1;
}

void publish_results(bool from_step) {
    pgv_copy_out(g.env, g.h_obs.data(), g.h_reward.data(), g.h_done.data());
    if (from_step) {
        double sum = 0.0;
        bool all = true;
        for (int i = 0; i < g.n; i++) {
            sum += g.h_reward[i];
            all = all && g.h_done[i];
        }
        step_data.reward.f = static_cast<float>(g.n == 1 ? g.h_reward[0] : sum / g.n);
        step_data.terminated = g.n == 1 ? (g.h_done[0] != 0) : all;
        step_data.truncated = false;
    }
}

}  // namespace

extern "C" {

int32_t cenv_get_env_version(void) { return kVersion; }

int32_t cenv_make(const char* render_mode, cenv_option* options, int32_t options_size) {
    (void)render_mode;
    if (g.env) cenv_close();
    g.window_w = g.window_h = 512;               // coinrun.cpp:29-30; a previous make's size does not carry over
    int seed = static_cast<int>(time(nullptr));  // coinrun.cpp:130
    int num_envs = 1, game = PG_DEFAULT_GAME, device = 0, env_offset = 0, num_levels = 0, start_level = 0, mode = 0, game_flags = 0;
    for (int i = 0; i < options_size; i++) {
        const std::string name(options[i].name ? options[i].name : "");
        int v = 0;
        if (name == "seed" || name == "width" || name == "height" || name == "num_envs" || name == "game" ||
            name == "device" || name == "env_offset" || name == "num_levels" || name == "start_level" ||
            name == "distribution_mode" || name == "game_flags") {
            if (opt_int(options[i], &v)) return fail("cenv_make: option '" + name + "' must be INT");
        }
        if (name == "seed")
            seed = v;
        else if (name == "width")
            g.window_w = v;
        else if (name == "height")
            g.window_h = v;
        else if (name == "num_envs")
            num_envs = v;
        else if (name == "game")
            game = v;
        else if (name == "device")
            device = v;
        else if (name == "env_offset")
            env_offset = v;
        else if (name == "num_levels")
            num_levels = v;
        else if (name == "start_level")
            start_level = v;
        else if (name == "distribution_mode")
            mode = v;
        else if (name == "game_flags")
            game_flags = v;
    }
    const char* gname = pgv_game_name(game);
    if (!gname) return fail("cenv_make: unknown game id");
    pgv_config cfg{};
    cfg.struct_size = sizeof(pgv_config);
    cfg.game = gname;
    cfg.num_envs = num_envs;
    cfg.device = device;
    cfg.seed_base = static_cast<uint32_t>(seed);
    cfg.env_offset = env_offset;
    cfg.num_levels = num_levels;
    cfg.start_level = start_level;
    cfg.mode = mode;
    cfg.game_flags = static_cast<uint32_t>(game_flags);
    int rc = pgv_make_config(&cfg, &g.env);
    if (rc) return rc;
    g.n = num_envs;
    g.h_obs.assign(size_t(num_envs) * pg::kObsBytes, 0);
    g.h_reward.assign(num_envs, 0.0f);
    g.h_done.assign(num_envs, 0);
    g.h_actions.assign(num_envs, 0);
    g.h_frame.assign(size_t(g.window_w) * g.window_h * 3, 0);
    if (hipMalloc(reinterpret_cast<void**>(&g.d_actions), size_t(num_envs) * 4) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&g.d_seeds), size_t(num_envs) * 4) != hipSuccess) {
        cenv_close();
        return fail("cenv_make: hipMalloc failed");
    }

    // Spaces (coinrun.cpp:154-172): "screen" Box(0,255), "action" MultiDiscrete([15]).
    g.obs_space.key = "screen";
    g.obs_space.value_type = CENV_SPACE_TYPE_BOX;
    g.obs_space.value_buffer_size = 2;
    g.obs_space.value_buffer.f = g.box_bounds;
    g.act_space.key = "action";
    g.act_space.value_type = CENV_SPACE_TYPE_MULTI_DISCRETE;
    g.act_space.value_buffer_size = 1;
    g.act_space.value_buffer.i = g.nvec;
    make_data.observation_spaces_size = 1;
    make_data.observation_spaces = &g.obs_space;
    make_data.action_spaces_size = 1;
    make_data.action_spaces = &g.act_space;

    g.observations[0].key = "screen";
    g.observations[0].value_type = CENV_VALUE_TYPE_BYTE;
    g.observations[0].value_buffer_size = num_envs * pg::kObsBytes;
    g.observations[0].value_buffer.b = g.h_obs.data();
    g.observations[1].key = "reward";
    g.observations[1].value_type = CENV_VALUE_TYPE_FLOAT;
    g.observations[1].value_buffer_size = num_envs;
    g.observations[1].value_buffer.f = g.h_reward.data();
    g.observations[2].key = "terminated";
    g.observations[2].value_type = CENV_VALUE_TYPE_BYTE;
    g.observations[2].value_buffer_size = num_envs;
    g.observations[2].value_buffer.b = g.h_done.data();
    const int nobs = num_envs == 1 ? 1 : 3;
    reset_data.observations_size = nobs;
    reset_data.observations = g.observations;
    reset_data.infos_size = 0;
    reset_data.infos = nullptr;
    step_data.observations_size = nobs;
    step_data.observations = g.observations;
    step_data.reward.f = 0.0f;
    step_data.terminated = false;
    step_data.truncated = false;
    step_data.infos_size = 0;
    step_data.infos = nullptr;
    render_data.value_type = CENV_VALUE_TYPE_BYTE;
    render_data.value_buffer_width = g.window_w;
    render_data.value_buffer_height = g.window_h;
    render_data.value_buffer_channels = 3;
    render_data.value_buffer.b = g.h_frame.data();
    return 0;
}

int32_t cenv_reset(cenv_option* options, int32_t options_size) {
    if (!g.env) return fail("cenv_reset: cenv_make has not been called");
    const int32_t* seeds = nullptr;
    for (int i = 0; i < options_size; i++) {
        const std::string name(options[i].name ? options[i].name : "");
        if (name == "seed") {
            int v = 0;
            if (opt_int(options[i], &v)) return fail("cenv_reset: option 'seed' must be INT");
            for (int k = 0; k < g.n; k++) g.h_actions[k] = v + k;
            if (hipMemcpy(g.d_seeds, g.h_actions.data(), size_t(g.n) * 4, hipMemcpyHostToDevice) != hipSuccess)
                return fail("cenv_reset: seed upload failed");
            seeds = g.d_seeds;
        }
    }
    int rc = pgv_reset(g.env, nullptr, seeds);
    if (rc) return rc;
    publish_results(false);
    return 0;
}

int32_t cenv_step(cenv_key_value* actions, int32_t actions_size) {
    if (!g.env) return fail("cenv_step: cenv_make has not been called");
    for (int k = 0; k < g.n; k++) g.h_actions[k] = 0;  // `int action = 0;` (coinrun.cpp:342)
    for (int i = 0; i < actions_size; i++) {
        if (!actions[i].key || std::strcmp(actions[i].key, "action")) continue;
        if (actions[i].value_type != CENV_VALUE_TYPE_INT) return fail("cenv_step: 'action' must be INT");
        const int m = actions[i].value_buffer_size < g.n ? actions[i].value_buffer_size : g.n;
        for (int k = 0; k < m; k++) g.h_actions[k] = actions[i].value_buffer.i[k];
    }
    if (hipMemcpy(g.d_actions, g.h_actions.data(), size_t(g.n) * 4, hipMemcpyHostToDevice) != hipSuccess)
        return fail("cenv_step: action upload failed");
    if (g.n == 1) {
        // Reference semantics: no auto-reset — stepping a terminated env keeps stepping it.  The vector
        // engine's pending flag is cleared so the single-env path never resets on its own.
        uint8_t zero = 0;
        hipMemcpy(g.env->d_pending, &zero, 1, hipMemcpyHostToDevice);
    }
    int rc = pgv_step(g.env, g.d_actions);
    if (rc) return rc;
    publish_results(true);
    return 0;
}

int32_t cenv_render(void) {
    // Human-size frame (coinrun.cpp:393-411): render_game(false) of env 0 on the GPU (pg_frame.h).
    if (!g.env) return fail("cenv_render: cenv_make has not been called");
    if (pgv_render_frame(g.env, 0, g.window_w, g.window_h, g.h_frame.data()) == 0) return 0;
    // a game without a frame kernel: the 64×64 observation of env 0, nearest-neighbour enlarged
    for (int y = 0; y < g.window_h; y++)
        for (int x = 0; x < g.window_w; x++) {
            const int sx = x * pg::kObsW / g.window_w, sy = y * pg::kObsH / g.window_h;
            for (int c = 0; c < 3; c++) g.h_frame[c + 3 * (x + g.window_w * y)] = g.h_obs[c + 3 * (sx + pg::kObsW * sy)];
        }
    return 0;
}

void cenv_close(void) {
    if (!g.env) return;
    if (g.d_actions) hipFree(g.d_actions);
    if (g.d_seeds) hipFree(g.d_seeds);
    g.d_actions = g.d_seeds = nullptr;
    pgv_close(g.env);
    g.env = nullptr;
}

}  // extern "C"
