// ORACLE — TEST INFRASTRUCTURE ONLY.  The reference's single-image `cenv` C ABI (cenv/cenv.h:122-133: four data symbols,
// six functions) over the CPU restatement's vector API (pgo_api.cpp pgo_vec_*), so that BASELINE.json configs[0] —
// "maze, 64 envs, CPU reference engine via cenv (plumbing, no GPU)" — can be driven through the very wrapper the HIP
// engine's libMaze.so is driven through (procgen2_amd/cenv.py, same call shapes as cenv/cenv.py:184-380), with no GPU.
// Same batch conventions as the engine's shim (include/procgen2_cenv.h): options "seed", "num_envs", "game" (default 1 =
// maze); num_envs == 1: "screen" only, no auto-reset (the reference); num_envs > 1: "screen" BYTE[N*12288], "reward"
// FLOAT[N], "terminated" BYTE[N], next-step auto-reset, step_data.reward.f = the batch mean.
// Built as oracle/libpgoracle_cenv.so beside libpgoracle.so (which it links); nothing in procgen2_amd/ loads either.
#include <cstring>
#include <string>
#include <vector>

#include "../include/procgen2_cenv.h"

extern "C" {
void* pgo_vec_make_only(const char* game, int n, uint32_t seed_base, int env_offset, int render_enabled);
void pgo_vec_reset(void* h, const uint8_t* mask, const int32_t* seeds);
void pgo_vec_step(void* h, const int32_t* actions, uint32_t run_seed, int env_offset, int threads, uint8_t* obs_out, float* rew_out,
                  uint8_t* done_out);
void pgo_vec_obs(void* h, uint8_t* obs_out);
void pgo_vec_clear_pending(void* h);
void pgo_vec_close(void* h);

cenv_make_data make_data;
cenv_reset_data reset_data;
cenv_step_data step_data;
cenv_render_data render_data;
}

namespace {
constexpr int kObsBytes = 64 * 64 * 3;
const char* const kGames[7] = {"coinrun", "maze", "bossfight", "climber", "caveflyer", "chaser", "jumper"};
struct Shim {
    void* vec = nullptr;
    int n = 0;
    cenv_key_value obs_space{}, act_space{}, observations[3]{};
    float box_bounds[2] = {0.0f, 255.0f};
    int32_t nvec[1] = {15};
    std::vector<uint8_t> obs, done, frame;
    std::vector<float> reward;
    std::vector<int32_t> actions;
} g;

int opt_int(const cenv_option& o, int* out) {
    if (o.value_type == CENV_VALUE_TYPE_INT) return *out = o.value.i, 0;
    if (o.value_type == CENV_VALUE_TYPE_DOUBLE) return *out = static_cast<int>(o.value.d), 0;  // python float (cenv.py:39-42)
    return 1;
}
}  // namespace

extern "C" {

int32_t cenv_get_env_version(void) { return 100; }  // maze.cpp:9

int32_t cenv_make(const char* render_mode, cenv_option* options, int32_t options_size) {
    (void)render_mode;
    if (g.vec) cenv_close();
    int seed = 0, num_envs = 1, game = 1;
    for (int i = 0; i < options_size; i++) {
        const std::string name(options[i].name ? options[i].name : "");
        int v = 0;
        if ((name == "seed" || name == "num_envs" || name == "game") && opt_int(options[i], &v)) return 1;
        if (name == "seed") seed = v;
        if (name == "num_envs") num_envs = v;
        if (name == "game") game = v;
    }
    if (game < 0 || game >= 7 || num_envs < 1) return 1;
    g.vec = pgo_vec_make_only(kGames[game], num_envs, static_cast<uint32_t>(seed), 0, 1);  // env i: rng.seed(seed + i), level 0 built (D1)
    if (!g.vec) return 1;
    g.n = num_envs;
    g.obs.assign(size_t(num_envs) * kObsBytes, 0);
    g.reward.assign(num_envs, 0.0f);
    g.done.assign(num_envs, 0);
    g.actions.assign(num_envs, 0);
    g.frame.assign(size_t(512) * 512 * 3, 0);
    g.obs_space = {"screen", CENV_SPACE_TYPE_BOX, 2, {}};  // maze.cpp:154-172
    g.obs_space.value_buffer.f = g.box_bounds;
    g.act_space = {"action", CENV_SPACE_TYPE_MULTI_DISCRETE, 1, {}};
    g.act_space.value_buffer.i = g.nvec;
    make_data = {1, &g.obs_space, 1, &g.act_space};
    g.observations[0] = {"screen", CENV_VALUE_TYPE_BYTE, num_envs * kObsBytes, {}};
    g.observations[0].value_buffer.b = g.obs.data();
    g.observations[1] = {"reward", CENV_VALUE_TYPE_FLOAT, num_envs, {}};
    g.observations[1].value_buffer.f = g.reward.data();
    g.observations[2] = {"terminated", CENV_VALUE_TYPE_BYTE, num_envs, {}};
    g.observations[2].value_buffer.b = g.done.data();
    const int nobs = num_envs == 1 ? 1 : 3;
    reset_data = {nobs, g.observations, 0, nullptr};
    step_data = {};
    step_data.observations_size = nobs;
    step_data.observations = g.observations;
    render_data = {CENV_VALUE_TYPE_BYTE, 512, 512, 3, {}};
    render_data.value_buffer.b = g.frame.data();
    return 0;
}

int32_t cenv_reset(cenv_option* options, int32_t options_size) {
    if (!g.vec) return 1;
    bool reseed = false;
    for (int i = 0; i < options_size; i++)
        if (options[i].name && !std::strcmp(options[i].name, "seed")) {
            int v = 0;
            if (opt_int(options[i], &v)) return 1;
            for (int k = 0; k < g.n; k++) g.actions[k] = v + k;  // env i reseeds with seed + i
            reseed = true;
        }
    pgo_vec_reset(g.vec, nullptr, reseed ? g.actions.data() : nullptr);
    pgo_vec_obs(g.vec, g.obs.data());
    std::fill(g.reward.begin(), g.reward.end(), 0.0f);
    std::fill(g.done.begin(), g.done.end(), uint8_t(0));
    return 0;
}

int32_t cenv_step(cenv_key_value* actions, int32_t actions_size) {
    if (!g.vec) return 1;
    std::fill(g.actions.begin(), g.actions.end(), 0);  // `int action = 0;` (maze.cpp:278)
    for (int i = 0; i < actions_size; i++) {
        if (!actions[i].key || std::strcmp(actions[i].key, "action")) continue;
        if (actions[i].value_type != CENV_VALUE_TYPE_INT) return 1;
        const int m = actions[i].value_buffer_size < g.n ? actions[i].value_buffer_size : g.n;
        for (int k = 0; k < m; k++) g.actions[k] = actions[i].value_buffer.i[k];
    }
    if (g.n == 1) pgo_vec_clear_pending(g.vec);  // the reference has no auto-reset (game_test.py:36-40 resets from Python)
    pgo_vec_step(g.vec, g.actions.data(), 0, 0, 1, g.obs.data(), g.reward.data(), g.done.data());
    double sum = 0.0;
    bool all = true;
    for (int i = 0; i < g.n; i++) {
        sum += g.reward[i];
        all = all && g.done[i];
    }
    step_data.reward.f = static_cast<float>(g.n == 1 ? g.reward[0] : sum / g.n);
    step_data.terminated = g.n == 1 ? g.done[0] != 0 : all;
    step_data.truncated = false;
    return 0;
}

int32_t cenv_render(void) {  // plumbing only: env 0's observation, nearest-neighbour enlarged to the 512×512 window
    if (!g.vec) return 1;
    for (int y = 0; y < 512; y++)
        for (int x = 0; x < 512; x++)
            for (int c = 0; c < 3; c++) g.frame[c + 3 * (x + 512 * y)] = g.obs[c + 3 * ((x / 8) + 64 * (y / 8))];
    return 0;
}

void cenv_close(void) {
    if (g.vec) pgo_vec_close(g.vec);
    g.vec = nullptr;
}

}  // extern "C"
