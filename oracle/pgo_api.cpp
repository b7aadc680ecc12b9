// ORACLE — TEST INFRASTRUCTURE ONLY (see pgo_common.h header).
//
// Plain-C entry points for ctypes (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
#include <chrono>
#include <cstdio>
#include <thread>

#include "pgo_env.h"

using pgo::Env;

extern "C" {

// Registers a decoded RGBA8 texture under the reference's asset path.
void pgo_put_texture(const char* name, int w, int h, const uint8_t* rgba) {
    pgo::TextureBank::global().put(name, w, h, rgba);
}

int pgo_texture_count() { return static_cast<int>(pgo::TextureBank::global().size()); }

// (game, PGV_MODE_*) → the mode the env runs in, -1 if the game has no such mode.  0 = the reference's compile-time
// default (SURVEY.md §5 "config / flags").
int pgo_resolve_mode(const char* game, int mode) {
    const std::string g(game);
    const bool plain = g == "coinrun" || g == "climber" || g == "bossfight";  // easy | hard
    const bool mem = g == "maze" || g == "jumper" || g == "caveflyer";      // + memory
    if (mode == 0) return g == "chaser" ? Env::kEasy : Env::kHard;
    if (mode == Env::kEasy || mode == Env::kHard) return (plain || mem || g == "chaser") ? mode : -1;
    if (mode == Env::kMemory) return mem ? mode : -1;
    if (mode == Env::kExtreme) return g == "chaser" ? mode : -1;
    return -1;
}

void* pgo_make_mode(const char* game, uint32_t seed, int render_enabled, int mode);
void* pgo_make_config(const char* game, uint32_t seed, int render_enabled, int mode, uint32_t flags);

// game: one of the seven names.  Returns nullptr for unknown games.
void* pgo_make(const char* game, uint32_t seed, int render_enabled) { return pgo_make_mode(game, seed, render_enabled, 0); }

void* pgo_make_mode(const char* game, uint32_t seed, int render_enabled, int mode) {
    return pgo_make_config(game, seed, render_enabled, mode, 0);
}

void* pgo_make_config(const char* game, uint32_t seed, int render_enabled, int mode, uint32_t flags) {
    std::string g(game);
    // include/procgen2_vec.h: coinrun PGV_COINRUN_NO_* (bits 0-3), chaser / jumper PGV_*_FLOAT_ABS (bit 0)
    const uint32_t known = g == "coinrun" ? 15u : (g == "chaser" || g == "jumper") ? 1u : 0u;
    if (flags & ~known) return nullptr;
    const int resolved = pgo_resolve_mode(game, mode);
    if (resolved < 0) return nullptr;
    Env* e = nullptr;
    if (g == "coinrun")
        e = pgo::new_coinrun();
    else if (g == "maze")
        e = pgo::new_maze();
    else if (g == "bossfight")
        e = pgo::new_bossfight();
    else if (g == "climber")
        e = pgo::new_climber();
    else if (g == "caveflyer")
        e = pgo::new_caveflyer();
    else if (g == "chaser")
        e = pgo::new_chaser();
    else if (g == "jumper")
        e = pgo::new_jumper();
    if (!e) return nullptr;
    e->set_render_enabled(render_enabled != 0);
    e->set_mode(resolved);
    e->set_flags(flags);
    e->make(seed);
    return e;
}

void pgo_close(void* h) { delete static_cast<Env*>(h); }

void pgo_render_frame(void* h, int width, int height, uint8_t* out_rgb) {
    static_cast<Env*>(h)->render_frame(width, height, out_rgb);
}
void pgo_present(void* h) { static_cast<Env*>(h)->present(); }
void pgo_reset(void* h, int reseed, int32_t seed) { static_cast<Env*>(h)->reset(reseed != 0, seed); }

void pgo_step(void* h, int action) { static_cast<Env*>(h)->step(action); }

float pgo_reward(void* h) { return static_cast<Env*>(h)->reward; }
int pgo_terminated(void* h) { return static_cast<Env*>(h)->terminated ? 1 : 0; }
int pgo_truncated(void* h) { return static_cast<Env*>(h)->truncated ? 1 : 0; }
const uint8_t* pgo_obs(void* h) { return static_cast<Env*>(h)->obs; }
int pgo_dump_state(void* h, float* out, int cap) { return static_cast<Env*>(h)->dump_state(out, cap); }
int pgo_dump_tiles(void* h, uint8_t* out, int cap) { return static_cast<Env*>(h)->dump_tiles(out, cap); }
uint32_t pgo_rng_peek(void* h) { return static_cast<Env*>(h)->rng_peek(); }

// SURVEY.md Appendix C driver: make(seed) → reset → `steps` LCG actions, reset on terminated.
// Fills crc of the (reward f32 LE, terminated u8) stream, episode count, reward sum and the
// first `cap` episode lengths.  Rendering is off (the traces are raster-independent).
int pgo_trace_flags(const char* game, uint32_t seed, int steps, uint32_t flags, uint32_t* crc_out, int* episodes_out,
                    double* reward_sum_out, int* lengths_out, int cap);
int pgo_trace(const char* game, uint32_t seed, int steps, uint32_t* crc_out, int* episodes_out, double* reward_sum_out,
              int* lengths_out, int cap) {
    return pgo_trace_flags(game, seed, steps, 0, crc_out, episodes_out, reward_sum_out, lengths_out, cap);
}
int pgo_trace_flags(const char* game, uint32_t seed, int steps, uint32_t flags, uint32_t* crc_out, int* episodes_out,
                    double* reward_sum_out, int* lengths_out, int cap) {
    Env* e = static_cast<Env*>(pgo_make_config(game, seed, 0, 0, flags));
    if (!e) return -1;
    e->reset(false, 0);
    pgo::Crc32 crc;
    uint32_t s = 1;
    int episodes = 0, len = 0;
    double total = 0.0;
    for (int i = 0; i < steps; i++) {
        s = s * 1664525u + 1013904223u;
        int action = static_cast<int>((s >> 16) % 15);
        e->step(action);
        len++;
        float r = e->reward;
        uint8_t t = e->terminated ? 1 : 0;
        crc.feed(&r, 4);
        crc.feed(&t, 1);
        total += r;
        if (t) {
            if (episodes < cap) lengths_out[episodes] = len;
            episodes++;
            len = 0;
            e->reset(false, 0);
        }
    }
    *crc_out = crc.value();
    *episodes_out = episodes;
    *reward_sum_out = total;
    delete e;
    return 0;
}

// Vector rollout used for parity checks and the CPU baseline: N independent envs with seeds
// seed_base + i (make, then one reset — exactly CEnv(lib, options={"seed": s}).reset()),
// actions from the same counter hash the HIP engine uses (include/procgen2_vec.h), auto-reset
// on the step after `terminated` (the reset replaces that step; reward 0, done 0).
// obs_out: nullptr or [steps][n][12288]; rew_out [steps][n]; done_out [steps][n].
struct pgo_vec;

static inline uint32_t mix32(uint32_t x) {  // the action hash of include/procgen2_vec.h
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

int pgo_synthetic_action(uint32_t run_seed, uint32_t step, uint32_t env) {
    uint32_t h = mix32(mix32(step * 0x9E3779B9u + run_seed) ^ (env * 0x85EBCA6Bu + 0xC2B2AE35u));
    return static_cast<int>((static_cast<uint64_t>(h) * 15u) >> 32);
}

struct VecState {
    std::vector<Env*> envs;
    std::vector<uint8_t> pending_reset;
    uint32_t step_counter = 0;
    // level-seed mode (include/procgen2_vec.h pgv_make_levels); num_levels 0 = off
    std::string game;
    int render = 0, num_levels = 0, start_level = 0, mode = 0;
    uint32_t flags = 0;
    std::vector<uint32_t> chain_seed, drawn;
};

// One new level for env i.  Level-seed mode: level number L = what a fresh cenv_make(seed = L) builds as its
// level 0 — so the env object is literally made anew.
static void vec_new_level(VecState* v, int i, bool restart, uint32_t seed) {
    if (restart) {
        v->chain_seed[i] = seed;
        v->drawn[i] = 0;
    }
    if (v->num_levels > 0) {
        const uint32_t k = v->drawn[i]++;
        const uint32_t number = static_cast<uint32_t>(v->start_level) +
                                mix32(mix32(v->chain_seed[i]) + k) % static_cast<uint32_t>(v->num_levels);
        delete v->envs[i];
        v->envs[i] = static_cast<Env*>(pgo_make_config(v->game.c_str(), number, v->render, v->mode, v->flags));
        v->envs[i]->present();
    } else {
        v->envs[i]->reset(restart, static_cast<int32_t>(seed));
    }
}

void* pgo_vec_make_flags(const char* game, int n, uint32_t seed_base, int env_offset, int render_enabled,
                         int num_levels, int start_level, int mode, uint32_t flags) {
    auto* v = new VecState();
    v->game = game;
    v->mode = mode;
    v->flags = flags;
    v->render = render_enabled;
    v->num_levels = num_levels;
    v->start_level = start_level;
    v->chain_seed.assign(n, 0);
    v->drawn.assign(n, 0);
    for (int i = 0; i < n; i++) {
        const uint32_t seed = seed_base + static_cast<uint32_t>(env_offset + i);
        Env* e = static_cast<Env*>(pgo_make_config(game, seed, render_enabled, mode, flags));  // level 0, never observed (D1)
        if (!e) {
            delete v;
            return nullptr;
        }
        v->envs.push_back(e);
        v->chain_seed[i] = seed;
        v->drawn[i] = num_levels > 0 ? 1 : 0;  // the engine's make draws a level number for its hidden level too
        vec_new_level(v, i, false, 0);
    }
    v->pending_reset.assign(n, 0);
    return v;
}

// cenv_make alone — level 0 built and never observed (D1), NO first reset: what the cenv shim (pgo_cenv.cpp) needs, whose
// caller's cenv_reset comes as a call of its own.  (pgo_vec_make = this + one reset of every env.)
void* pgo_vec_make_only(const char* game, int n, uint32_t seed_base, int env_offset, int render_enabled) {
    auto* v = new VecState();
    v->game = game;
    v->mode = 0;
    v->flags = 0;
    v->render = render_enabled;
    v->num_levels = 0;
    v->start_level = 0;
    v->chain_seed.assign(n, 0);
    v->drawn.assign(n, 0);
    for (int i = 0; i < n; i++) {
        const uint32_t seed = seed_base + static_cast<uint32_t>(env_offset + i);
        Env* e = static_cast<Env*>(pgo_make_config(game, seed, render_enabled, 0, 0));
        if (!e) {
            for (Env* made : v->envs) delete made;
            delete v;
            return nullptr;
        }
        v->envs.push_back(e);
        v->chain_seed[i] = seed;
    }
    v->pending_reset.assign(n, 0);
    return v;
}

void pgo_vec_close(void* h);

// The same with the envs made by `threads` threads (env objects are independent; the texture bank is only read): the
// full-size parity tests make 65 536 of them.
void* pgo_vec_make_threads(const char* game, int n, uint32_t seed_base, int env_offset, int render_enabled,
                           int num_levels, int start_level, int mode, uint32_t flags, int threads) {
    auto* v = new VecState();
    v->game = game;
    v->mode = mode;
    v->flags = flags;
    v->render = render_enabled;
    v->num_levels = num_levels;
    v->start_level = start_level;
    v->chain_seed.assign(n, 0);
    v->drawn.assign(n, 0);
    v->envs.assign(n, nullptr);
    v->pending_reset.assign(n, 0);
    auto work = [&](int lo, int hi) {
        for (int i = lo; i < hi; i++) {
            const uint32_t seed = seed_base + static_cast<uint32_t>(env_offset + i);
            v->envs[i] = static_cast<Env*>(pgo_make_config(game, seed, render_enabled, mode, flags));
            if (!v->envs[i]) return;
            v->chain_seed[i] = seed;
            v->drawn[i] = num_levels > 0 ? 1 : 0;
            vec_new_level(v, i, false, 0);
        }
    };
    if (threads < 1) threads = 1;
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back(work, static_cast<int>(int64_t(n) * t / threads), static_cast<int>(int64_t(n) * (t + 1) / threads));
    for (auto& th : pool) th.join();
    for (Env* e : v->envs)
        if (!e) {
            pgo_vec_close(v);
            return nullptr;
        }
    return v;
}

void* pgo_vec_make_config(const char* game, int n, uint32_t seed_base, int env_offset, int render_enabled,
                          int num_levels, int start_level, int mode) {
    return pgo_vec_make_flags(game, n, seed_base, env_offset, render_enabled, num_levels, start_level, mode, 0);
}
void* pgo_vec_make_levels(const char* game, int n, uint32_t seed_base, int env_offset, int render_enabled,
                          int num_levels, int start_level) {
    return pgo_vec_make_config(game, n, seed_base, env_offset, render_enabled, num_levels, start_level, 0);
}
void* pgo_vec_make(const char* game, int n, uint32_t seed_base, int env_offset, int render_enabled) {
    return pgo_vec_make_config(game, n, seed_base, env_offset, render_enabled, 0, 0, 0);
}

// cenv_reset of the envs with mask[i] != 0 (nullptr = all); seeds nullptr = keep the streams.
void pgo_vec_reset(void* h, const uint8_t* mask, const int32_t* seeds) {
    auto* v = static_cast<VecState*>(h);
    for (size_t i = 0; i < v->envs.size(); i++) {
        if (mask && !mask[i]) continue;
        vec_new_level(v, static_cast<int>(i), seeds != nullptr, seeds ? static_cast<uint32_t>(seeds[i]) : 0u);
        v->envs[i]->reward = 0.0f;
        v->envs[i]->terminated = false;
        v->pending_reset[i] = 0;
    }
}

void pgo_vec_close(void* h) {
    auto* v = static_cast<VecState*>(h);
    for (Env* e : v->envs) delete e;
    delete v;
}

// The single-env cenv shim (pgo_cenv.cpp): the reference never resets on its own.
void pgo_vec_clear_pending(void* h) {
    auto* v = static_cast<VecState*>(h);
    std::fill(v->pending_reset.begin(), v->pending_reset.end(), 0);
}

// Drawing on / off for every env of the batch from now on (Painter::enabled gates the pixel work only: game state, draw
// lists and the camera evolve the same either way).  The full-size parity tests step with drawing off and switch it on
// for the steps whose observations they compare.
void pgo_vec_set_render(void* h, int on) {
    auto* v = static_cast<VecState*>(h);
    v->render = on;
    for (Env* e : v->envs) e->set_render_enabled(on != 0);
}

void pgo_vec_reset_threads(void* h, int threads) {  // pgo_vec_reset(h, nullptr, nullptr) by `threads` threads
    auto* v = static_cast<VecState*>(h);
    const int n = static_cast<int>(v->envs.size());
    auto work = [&](int lo, int hi) {
        for (int i = lo; i < hi; i++) {
            vec_new_level(v, i, false, 0u);
            v->envs[i]->reward = 0.0f;
            v->envs[i]->terminated = false;
            v->pending_reset[i] = 0;
        }
    };
    if (threads < 1) threads = 1;
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back(work, static_cast<int>(int64_t(n) * t / threads), static_cast<int>(int64_t(n) * (t + 1) / threads));
    for (auto& th : pool) th.join();
}

// One vector step over envs [lo, hi) with explicit actions (nullptr → synthetic hash with
// run_seed and the global env index env_offset + i).
static void vec_step_range(VecState* v, int lo, int hi, const int32_t* actions, uint32_t run_seed, int env_offset,
                           uint8_t* obs_out, float* rew_out, uint8_t* done_out) {
    for (int i = lo; i < hi; i++) {
        Env* e = v->envs[i];
        if (v->pending_reset[i]) {
            vec_new_level(v, i, false, 0);
            e = v->envs[i];
            e->reward = 0.0f;
            e->terminated = false;
            v->pending_reset[i] = 0;
        } else {
            int a = actions ? actions[i] : pgo_synthetic_action(run_seed, v->step_counter, env_offset + i);
            e->step(a);
            if (e->terminated) v->pending_reset[i] = 1;
        }
        if (obs_out) std::memcpy(obs_out + size_t(i) * pgo::kObsBytes, e->obs, pgo::kObsBytes);
        if (rew_out) rew_out[i] = e->reward;
        if (done_out) done_out[i] = e->terminated ? 1 : 0;
    }
}

void pgo_vec_step(void* h, const int32_t* actions, uint32_t run_seed, int env_offset, int threads, uint8_t* obs_out,
                  float* rew_out, uint8_t* done_out) {
    auto* v = static_cast<VecState*>(h);
    const int n = static_cast<int>(v->envs.size());
    if (threads <= 1) {
        vec_step_range(v, 0, n, actions, run_seed, env_offset, obs_out, rew_out, done_out);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) {
            int lo = static_cast<int>(int64_t(n) * t / threads), hi = static_cast<int>(int64_t(n) * (t + 1) / threads);
            pool.emplace_back(vec_step_range, v, lo, hi, actions, run_seed, env_offset, obs_out, rew_out, done_out);
        }
        for (auto& th : pool) th.join();
    }
    v->step_counter++;
}

void pgo_vec_obs(void* h, uint8_t* obs_out) {
    auto* v = static_cast<VecState*>(h);
    for (size_t i = 0; i < v->envs.size(); i++)
        std::memcpy(obs_out + i * pgo::kObsBytes, v->envs[i]->obs, pgo::kObsBytes);
}

int pgo_vec_dump_state(void* h, int env, float* out, int cap) {
    return static_cast<VecState*>(h)->envs[env]->dump_state(out, cap);
}
int pgo_vec_dump_tiles(void* h, int env, uint8_t* out, int cap) {
    return static_cast<VecState*>(h)->envs[env]->dump_tiles(out, cap);
}

// Times `steps` vector steps (synthetic actions) and returns env-steps per second.
double pgo_vec_bench(void* h, int steps, uint32_t run_seed, int threads) {
    auto* v = static_cast<VecState*>(h);
    const int n = static_cast<int>(v->envs.size());
    const uint32_t first = v->step_counter;
    auto t0 = std::chrono::steady_clock::now();
    // Envs never interact, so each thread runs all `steps` over its own slice: no per-step barrier,
    // which is the most favourable schedule for the CPU side.
    auto slice = [&](int lo, int hi) {
        for (int s = 0; s < steps; s++) {
            for (int i = lo; i < hi; i++) {
                Env* e = v->envs[i];
                if (v->pending_reset[i]) {
                    vec_new_level(v, i, false, 0);
                    e = v->envs[i];
                    e->reward = 0.0f;
                    e->terminated = false;
                    v->pending_reset[i] = 0;
                } else {
                    e->step(pgo_synthetic_action(run_seed, first + s, i));
                    if (e->terminated) v->pending_reset[i] = 1;
                }
            }
        }
    };
    if (threads <= 1) {
        slice(0, n);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++)
            pool.emplace_back(slice, static_cast<int>(int64_t(n) * t / threads),
                              static_cast<int>(int64_t(n) * (t + 1) / threads));
        for (auto& th : pool) th.join();
    }
    v->step_counter += steps;
    auto t1 = std::chrono::steady_clock::now();
    double sec = std::chrono::duration<double>(t1 - t0).count();
    return double(steps) * double(v->envs.size()) / sec;
}

}  // extern "C"
