// Engine-internal interfaces: device sprite atlas, per-game kernel launchers, and the vector env
// object behind the C ABI of include/procgen2_vec.h.  Nothing here is exported.
#pragma once

#include <hip/hip_runtime.h>

#include <memory>
#include <string>
#include <vector>

#include "pg_defs.h"

namespace pg {

// Device view of the atlas: all textures of one game packed into one RGBA8 array in HBM.
// desc[t] = {texel offset, width, height, 0}.  Replaces Asset_Texture / manager_texture
// (games/*/common_assets.h, asset_manager.h) — SURVEY.md rows A1, E4.
struct AtlasView {
    const uint32_t* texels;
    const int4* desc;
    int count;
    uint32_t texel_bytes;  // size of `texels` in bytes (< 256 MiB: pg_render.h kRank, kNoTexel)
    const uint8_t* sort_ranks;  // pg_order.h equal_key_ranks table (kRankTableBytes), uploaded with the atlas
};

class Atlas {
   public:
    ~Atlas();
    // names are reference asset paths relative to the asset root ("kenney/Items/coinGold.png").
    bool load(const std::string& root, const std::vector<std::string>& names, std::string& err);
    bool upload(std::string& err);
    AtlasView view() const {
        return {d_texels_, d_desc_, static_cast<int>(desc_.size()), static_cast<uint32_t>(texels_.size() * 4), d_ranks_};
    }
    size_t texel_bytes() const { return texels_.size() * 4; }
    // Host-side access for Game::extend_atlas: the decoded textures, and room for derived data behind them (returns
    // the word offset of the appended block; the atlas is one flat array of 32-bit words on the device).
    const uint32_t* texels_host(int tex) const { return texels_.data() + desc_[tex].x; }
    int4 desc_host(int tex) const { return desc_[tex]; }
    uint32_t append_words(const std::vector<uint32_t>& words) {
        const uint32_t at = static_cast<uint32_t>(texels_.size());
        texels_.insert(texels_.end(), words.begin(), words.end());
        return at;
    }
    std::vector<std::pair<int, int>> sizes() const {
        std::vector<std::pair<int, int>> v;
        for (const auto& d : desc_) v.emplace_back(d.y, d.z);
        return v;
    }

   private:
    std::vector<uint32_t> texels_;
    std::vector<int4> desc_;
    uint32_t* d_texels_ = nullptr;
    int4* d_desc_ = nullptr;
    uint8_t* d_ranks_ = nullptr;
};

// Buffers every game writes: the contiguous observation slab and the per-env scalars.
struct StepIO {
    uint8_t* obs;      // [n][64][64][3]
    float* reward;     // [n]
    uint8_t* done;     // [n]  terminated (truncated is always false in the reference: coinrun.cpp:367)
    // [n]  1: the env terminated last step → the next step performs the reset instead: the level kernel serves it and
    // leaves 2, the logic kernel skips it and puts 0 back.  A game whose logic kernel runs BESIDE the level kernel
    // (Game::resets_beside_logic) cannot hand the flag over like that: there the logic kernel skips every env that is
    // not 0 and writes 3 — not 1, which the level kernel running next to it would take for last step's — for an env that
    // terminates now, and the step's render kernel, which runs after both, turns 2 into 0 and 3 into 1.
    // Between steps: 0 or 1 either way.  The games that install prefetched levels inside their one logic launch
    // (pg_prefetch.h install_prefetched: maze, jumper, climber, caveflyer) keep the step's parity in the byte instead:
    // 4 | t mod 2 = reset due in step t, 2 | t mod 2 = reset served in step t, anything else = step.
    uint8_t* pending;
};

// Level-seed mode (absent from the reference, modelled on the original procgen's
// num_levels / start_level).  num_levels = 0: the reference's behaviour — one RNG stream per env, every level is new.
// num_levels > 0: the k-th level an env builds since it was (re)seeded is level number
//     start_level + mix32(mix32(chain_seed) + k) % num_levels,
// and "level number L" means exactly what a fresh `cenv_make(seed = L)` builds as its level 0 — fresh containers,
// fresh camera, rng.seed(L) — so the same number always gives the same level, whatever the env played before.
struct LevelPlan {
    int32_t num_levels;    // 0 = off
    int32_t start_level;
    uint32_t* chain_seed;  // [n]  the seed the env was made / last reseeded with
    uint32_t* drawn;       // [n]  k: levels built since then
};

class Game {
   public:
    virtual ~Game() = default;
    virtual const char* name() const = 0;
    virtual std::vector<std::string> texture_names() const = 0;
    virtual size_t state_bytes(int n) const = 0;
    virtual void bind(void* d_state, int n, AtlasView atlas) = 0;
    // cenv_make: seed rng with seed_base + env_offset + i, build and discard level 0 (D1).
    virtual void launch_make(hipStream_t s, uint32_t seed_base, int env_offset) = 0;
    // cenv_reset for envs where mask != 0 (nullptr = all); seeds nullptr = keep the stream (no reseed).
    virtual void launch_reset(hipStream_t s, const uint8_t* mask, const int32_t* seeds, StepIO io) = 0;
    // cenv_step with next-step auto-reset; actions nullptr = synthetic hash(run_seed, step, env).
    virtual void launch_logic(hipStream_t s, const int32_t* actions, uint32_t run_seed, uint32_t step_index,
                              int env_offset, StepIO io) = 0;
    // render_game(true) + RGB pack into io.obs for envs where mask != 0 (nullptr = all).  A game with a render pre-pass
    // (pg_prepass.h) has it launched right before, by the engine: launch_prepass, same mask — a launch of its own so that
    // the engine's render events (pgv_step_times, bench.py's roofline leg) bracket the render kernel alone.
    virtual void launch_render(hipStream_t s, const uint8_t* mask, StepIO io) = 0;
    virtual void launch_prepass(hipStream_t s, const uint8_t* mask) { (void)s; (void)mask; }
    // Level prefetch (pg_prefetch.h): launch the generator that fills queued shadow slots on the side stream.
    // bulk = most envs are expected to be queued (after make / a full reset).  False = game has no prefetch.
    virtual bool launch_pregen(hipStream_t side, bool bulk) { return false; }
    // While the side stream is busy the engine launches the generator again only every this-many steps: an env whose
    // episode ends before its slot is refilled generates its level inside the step, on the main stream, and one such env
    // costs the step a whole level's latency.  Four suits the games whose episodes last hundreds of steps; maze's
    // smallest mazes are solved in a few (two: 160 -> 195 M env-steps/s; one: 191), caveflyer gains 4 % at two,
    // coinrun, climber and jumper lose 1-2 %.
    virtual int pregen_every() const { return 4; }
    // cenv_render's human-size frame (render_game(false)) of one env into a w×h target of 0x00BBGGRR words in device
    // memory (pg_frame.h).  False = not implemented for this game.
    virtual bool launch_frame(hipStream_t s, int env, uint32_t* d_px, int w, int h) { return false; }
    // Debug tap used by the parity tests: game-defined float dump of one env (host pointer).
    virtual int dump_state(hipStream_t s, int env, float* out, int cap) = 0;
    virtual int dump_tiles(hipStream_t s, int env, uint8_t* out, int cap) = 0;
    // pgv_config.game_flags (include/procgen2_vec.h); false = this game does not know these switches.
    virtual bool set_game_flags(uint32_t flags) { return flags == 0; }
    // Called once between loading and uploading the atlas: a game may append data derived from its textures (jumper:
    // its compass ring as it lands on the 64×64 observation — the same pixels in every frame of every env).
    virtual void extend_atlas(Atlas& atlas) { (void)atlas; }
    // Called by pgv_save_state before it copies the state blob: a game that keeps part of what a rollout depends on
    // outside the blob for speed (chaser: the random streams' second buffers) puts it back first.  The engine
    // synchronises the stream afterwards.
    virtual void prepare_save(hipStream_t st) { (void)st; }
    // Host-side sanity check of the loaded atlas (sizes[i] = {w, h} of texture i); empty string = fine.
    virtual std::string check_atlas(const std::vector<std::pair<int, int>>& sizes) const { return ""; }

    // A game that generates its levels inside the step (chaser: 0.2 ms a step) can have the level kernel's auto-reset
    // (pg_prefetch.h mode 2) run beside its logic kernel — the envs being reset sit the logic out — on `reset_stream`,
    // which the engine forks off its main stream before launch_logic and joins before the render launch.  The hand-over
    // between the streams costs about 25 µs, so games whose auto-reset only installs a prefetched level do not ask.
    virtual bool resets_beside_logic() const { return false; }
    hipStream_t reset_stream = nullptr;
    // Such a game may go further and keep the reset stream apart until AFTER the step's render launch: the envs being
    // reset are a per cent of the batch, their levels one long chain per wavefront — beside the render kernel that
    // chain is hidden altogether.  launch_render_step() then renders every env that is not being reset,
    // launch_render_late() — called once the reset stream has joined — the few that were.  Default: one launch.
    virtual void launch_render_step(hipStream_t s, StepIO io) { launch_render(s, nullptr, io); }
    virtual bool launch_render_late(hipStream_t s, StepIO io) { return false; }

    // Called after pgv_load_state has replaced the state blob: whatever a game derives from its state and keeps OUTSIDE the
    // blob (chaser: the base layer of every env's frame, pg chaser.hip) is stale from here on.
    virtual void state_loaded(hipStream_t st) { (void)st; }

    // Device memory a game's kernels hand results to each other through within one frame (the render pre-pass,
    // pg_prepass.h): allocated by the engine beside the state, never part of a snapshot.
    virtual size_t scratch_bytes(int n) const { (void)n; return 0; }
    virtual void bind_scratch(void* d_scratch, int n) { (void)d_scratch; (void)n; }

    // Bit 0: render background + tiles by draw-list replay instead of the row composer (fallback path).
    // Bit 8: no level prefetch — every reset generates its level synchronously inside the step.
    // Bit 21: no render pre-pass — every frame's workgroup does its own set-up (the complete path; pg_prepass.h).
    // Bit 23: the pre-pass leaves every third env's frame to the complete path (`fat`), as it does by itself for the frames
    //         its tables do not hold — which some games never have in a normal run: the tests' way to that path.
    // Bit 24: coinrun — the entity lanes' hazard pre-selection assumes an agent that does not move (coinrun.hip hazard_near):
    //         the agent's own check of that assumption fails and resolve_kernel works the hazards out the long way.
    int debug_flags = 0;
    LevelPlan plan{0, 0, nullptr, nullptr};  // set by the engine after bind()
};

constexpr int kDebugNoPrefetch = 1 << 8;
constexpr int kDebugNoPrepass = 1 << 21;  // (clear of the -DPG_ABLATE experiment bits the games use)
constexpr int kDebugFatThirds = 1 << 23;
constexpr int kDebugCoinrunNoReach = 1 << 24;
constexpr int kDebugChaserSerialMobs = 1 << 25;  // chaser: the enemies one after the other on the stream itself (chaser.hip advance)

// Factories, one per compiled variant of a game (pg_defs.h PG_VARIANT; v0 = the reference's compile-time default).
std::unique_ptr<Game> make_coinrun_v0();
std::unique_ptr<Game> make_maze_v0();
std::unique_ptr<Game> make_maze_v1();
std::unique_ptr<Game> make_maze_v2();
std::unique_ptr<Game> make_bossfight_v0();
std::unique_ptr<Game> make_bossfight_v1();
std::unique_ptr<Game> make_climber_v0();
std::unique_ptr<Game> make_climber_v1();
std::unique_ptr<Game> make_caveflyer_v0();
std::unique_ptr<Game> make_caveflyer_v1();
std::unique_ptr<Game> make_caveflyer_v2();
std::unique_ptr<Game> make_chaser_v0();
std::unique_ptr<Game> make_chaser_v1();
std::unique_ptr<Game> make_chaser_v2();
std::unique_ptr<Game> make_jumper_v0();
std::unique_ptr<Game> make_jumper_v1();
std::unique_ptr<Game> make_jumper_v2();

// Counter-based synthetic action shared with the oracle (oracle/pgo_api.cpp pgo_synthetic_action).
PG_HD uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
PG_HD uint32_t level_number(int32_t num_levels, int32_t start_level, uint32_t chain_seed, uint32_t k) {
    return static_cast<uint32_t>(start_level) + mix32(mix32(chain_seed) + k) % static_cast<uint32_t>(num_levels);
}
PG_HD int synthetic_action(uint32_t run_seed, uint32_t step, uint32_t env) {
    uint32_t h = mix32(mix32(step * 0x9E3779B9u + run_seed) ^ (env * 0x85EBCA6Bu + 0xC2B2AE35u));
    return static_cast<int>((static_cast<uint64_t>(h) * 15u) >> 32);
}

}  // namespace pg
