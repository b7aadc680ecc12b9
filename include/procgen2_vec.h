/* procgen2_vec.h — vector extension of the cenv ABI for the MI355X engine.
 *
 * The reference ABI (cenv/cenv.h:122-133) is single-env with process-global state and host buffers;
 * at 65 536 envs one step's observations are 805 MB, which cannot cross PCIe at the target rate
 * (SURVEY.md §7 hard part 5, §8b "Consequence for a vector engine").  This extension keeps the same
 * life cycle — make / reset / step / close, games/<g>/<g>.cpp cenv_make:127, cenv_reset:308,
 * cenv_step:341, cenv_close:413 — but for N envs at once, with explicit handles and DEVICE pointers.
 *
 * Plain C ABI: pointers and sizes only.  Pointers documented as "device" are HIP device pointers on
 * the env's GPU.  All work is enqueued on the env's stream (its own, or the one given at make);
 * pgv_sync waits for it.  Every function returns 0 on success; on failure the message is in
 * pgv_last_error().  There is no CPU fallback: without a usable GPU pgv_make fails.
 *
 * Semantics that mirror the reference:
 *   - env i is seeded with seed_base + env_offset + i and builds one level that is never observed
 *     (cenv_make calls reset(), coinrun.cpp:235,303 — SURVEY.md D1);
 *   - pgv_reset ≙ cenv_reset on the selected envs (optional reseed), renders the reset frame;
 *   - pgv_step ≙ cenv_step.  An env that reported terminated performs its reset on the NEXT
 *     pgv_step instead of stepping (action ignored, reward 0, done 0, observation = reset frame),
 *     i.e. the observation sequence equals the reference caller's `if term: env.reset()` loop
 *     (game_test.py:36-40) with the per-env mt19937 stream continuing;
 *   - truncated is always false in the reference games (coinrun.cpp:367) and is not materialised.
 */
#ifndef PROCGEN2_VEC_H
#define PROCGEN2_VEC_H

#ifdef __cplusplus
extern "C" {
#endif

#include <stdint.h>

#define PGV_API __attribute__((__visibility__("default")))

#define PGV_OBS_BYTES 12288 /* 64*64*3, coinrun.cpp:24-25,185 */
#define PGV_NUM_ACTIONS 15  /* coinrun.cpp:26 */

typedef struct pgv_env pgv_env;

/* Games available in this build: 0 "coinrun", 1 "maze", 2 "bossfight", 3 "climber", 4 "caveflyer", 5 "chaser", 6 "jumper".  Returns NULL past the end. */
PGV_API const char* pgv_game_name(int32_t game_id);
PGV_API int32_t pgv_game_id(const char* name); /* -1 if unknown */

/* Create N envs of `game` on HIP device `device`.  `stream` is a hipStream_t to enqueue on, or NULL
 * for an internally created stream.  Asset root: $PROCGEN2_ASSETS, else <dir of the .so>/../assets. */
PGV_API int32_t pgv_make(const char* game, int32_t num_envs, int32_t device, uint32_t seed_base, int32_t env_offset,
                         void* stream, pgv_env** out);
/* Same with a finite level set (the reference has no such option, the semantics follow the
 * original procgen's num_levels / start_level).  num_levels = 0 is pgv_make: every level is new.  With
 * num_levels > 0 the k-th level an env builds since it was made / reseeded is level number
 *     start_level + mix32(mix32(seed of the env) + k) % num_levels      (mix32: the hash of pgv_synthetic_action)
 * and level number L is exactly what a fresh cenv_make(seed = L) of the reference builds as its level 0 (fresh
 * containers and camera, rng.seed(L): coinrun.cpp:130-152,219-262), so equal numbers give equal levels. */
PGV_API int32_t pgv_make_levels(const char* game, int32_t num_envs, int32_t device, uint32_t seed_base,
                                int32_t env_offset, void* stream, int32_t num_levels, int32_t start_level,
                                pgv_env** out);

/* The general form.  Distribution modes are compile-time `System_Tilemap::Config` constants in the
 * reference (e.g. maze/tilemap.h:13-17,40-42, chaser/tilemap.cpp:85-99, climber/tilemap.h:33); PGV_MODE_DEFAULT is the
 * one each reference game is compiled with.  pgv_game_modes(game_id) = bit mask (1 << mode) of what a game offers;
 * asking for anything else fails.  Zero-initialise the struct and set struct_size = sizeof(pgv_config). */
#define PGV_MODE_DEFAULT 0
#define PGV_MODE_EASY 1
#define PGV_MODE_HARD 2
#define PGV_MODE_MEMORY 3
#define PGV_MODE_EXTREME 4
typedef struct pgv_config {
    uint32_t struct_size;
    int32_t num_envs;
    const char* game;
    void* stream; /* hipStream_t or NULL */
    int32_t device;
    uint32_t seed_base;
    int32_t env_offset;
    int32_t num_levels, start_level; /* pgv_make_levels */
    int32_t mode;                    /* PGV_MODE_* */
    uint32_t game_flags;             /* game-specific generator switches, 0 = the reference's defaults (below) */
} pgv_config;
/* coinrun: System_Tilemap::Config's allow_pit / allow_crate / allow_dy / allow_mobs (coinrun/tilemap.h:42-45, all true
 * in the reference; tilemap.cpp:158,174,250,258), as switches that turn a feature OFF. */
#define PGV_COINRUN_NO_PIT 1u
#define PGV_COINRUN_NO_CRATE 2u
#define PGV_COINRUN_NO_DY 4u
#define PGV_COINRUN_NO_MOBS 8u
/* chaser, jumper: how the reference's unqualified `abs(<float>)` calls resolve (games/chaser/common_systems.cpp:165-166,
 * 206,346-420; games/jumper/common_systems.cpp:198).  0 (default): glibc's `int abs(int)` — the argument is truncated
 * first — which is what g++/libstdc++ picks when no header of the translation unit includes <stdlib.h> or <math.h>
 * (SDL3's SDL_stdinc.h stopped including them).  1: `float std::abs(float)`, what the same sources mean as soon as any
 * header pulls libstdc++'s <stdlib.h> wrapper in, and on libc++ / MSVC always.  Both readings have recorded traces of
 * the unmodified reference (tests/golden/appendix_c.json). */
#define PGV_CHASER_FLOAT_ABS 1u
#define PGV_JUMPER_FLOAT_ABS 1u
PGV_API int32_t pgv_make_config(const pgv_config* config, pgv_env** out);
PGV_API uint32_t pgv_game_modes(int32_t game_id);
PGV_API int32_t pgv_mode(pgv_env* env); /* the resolved mode (never PGV_MODE_DEFAULT) */
PGV_API void pgv_close(pgv_env* env);

/* Reset the envs whose mask byte is non-zero (device u8[N]; NULL = all).  seeds: device int32[N] to
 * reseed each selected env's mt19937 (coinrun.cpp:313-317), or NULL to keep the streams running. */
PGV_API int32_t pgv_reset(pgv_env* env, const uint8_t* d_mask, const int32_t* d_seeds);

/* One step of every env.  d_actions: device int32[N], values 0..14 (others are no-ops except in maze,
 * SURVEY.md D6/D20). */
PGV_API int32_t pgv_step(pgv_env* env, const int32_t* d_actions);

/* Same, with actions generated on the device: a = (mix(run_seed, step_index, env_offset+i) * 15) >> 32
 * where step_index counts pgv_step/pgv_step_synthetic calls since make.  Used by the benchmark so
 * the CPU oracle sees identical streams without a host transfer. */
PGV_API int32_t pgv_step_synthetic(pgv_env* env, uint32_t run_seed);
PGV_API int32_t pgv_synthetic_action(uint32_t run_seed, uint32_t step_index, uint32_t global_env);

/* Host-pointer conveniences (synchronous upload, then the calls above): for callers without device
 * memory of their own, e.g. the cenv shim and the parity tests.  h_mask / h_seeds may be NULL. */
/* `steps` synthetic steps of `count` envs side by side — step s of every env is enqueued (each on its own stream) before
 * step s+1 of any: the launch loop of a mixed workload (several games on one device) without a host round trip through
 * the caller's language per env and step.  Nothing is synchronised; pgv_sync the envs afterwards. */
PGV_API int32_t pgv_step_synthetic_many(pgv_env* const* envs, int32_t count, int32_t steps, uint32_t run_seed);

PGV_API int32_t pgv_step_host(pgv_env* env, const int32_t* h_actions);
PGV_API int32_t pgv_reset_host(pgv_env* env, const uint8_t* h_mask, const int32_t* h_seeds);

/* PNG → RGBA8 through the engine's own decoder (host only, no GPU needed): lets tests compare the
 * atlas loader with an independent decoder.  Returns 0 and fills w/h; copies min(cap, w*h*4) bytes. */
PGV_API int32_t pgv_decode_png(const char* path, int32_t* w, int32_t* h, uint8_t* h_rgba, int64_t cap);

PGV_API int32_t pgv_sync(pgv_env* env);

/* How many times the level generator has been launched on the env's side stream since make (pg_prefetch.h).  The cadence
 * is a function of the step count alone — every Game::pregen_every()-th step, plus one launch per make / reset — whatever
 * the caller's synchronisation pattern: a measurement / test tap, -1 for a NULL env. */
PGV_API int64_t pgv_generator_launches(pgv_env* env);

/* Result buffers (device pointers, valid until pgv_close or the next pgv_bind_outputs):
 *   obs    u8 [N][64][64][3]   row-major HWC, one contiguous slab
 *   reward f32[N]
 *   done   u8 [N]              terminated flag of the last step */
PGV_API uint8_t* pgv_obs(pgv_env* env);
PGV_API float* pgv_reward(pgv_env* env);
PGV_API uint8_t* pgv_done(pgv_env* env);
/* Let the caller own the result buffers (e.g. torch tensors): any pointer may be NULL to keep the
 * engine's own allocation. */
PGV_API int32_t pgv_bind_outputs(pgv_env* env, uint8_t* d_obs, float* d_reward, uint8_t* d_done);

PGV_API int32_t pgv_num_envs(pgv_env* env);
PGV_API int32_t pgv_device(pgv_env* env);
PGV_API void* pgv_stream(pgv_env* env);

/* Synchronous copies to host memory (any pointer may be NULL). */
PGV_API int32_t pgv_copy_out(pgv_env* env, uint8_t* h_obs, float* h_reward, uint8_t* h_done);

/* Whole-batch snapshot / restore: game state (incl. RNG streams and prefetched levels), reward, done, the pending
 * auto-resets, the step counter and the observations, as one HOST buffer of pgv_snapshot_bytes().  A snapshot loads
 * only into an env made with the same game, num_envs and env_offset.  The reference has no counterpart (its state
 * lives in process globals, games/coinrun/coinrun.cpp:22-70); this is the vector engine's checkpoint/resume. */
PGV_API int64_t pgv_snapshot_bytes(pgv_env* env);
PGV_API int32_t pgv_save_state(pgv_env* env, void* h_buffer, int64_t capacity);
PGV_API int32_t pgv_load_state(pgv_env* env, const void* h_buffer, int64_t size);

/* cenv_render for one env of the batch (games/coinrun/coinrun.cpp:393-411, render_game(false)): the human-size frame,
 * width x height x 3 bytes row-major RGB into a HOST buffer.  Synchronises the env's stream.  Debug / viewer path. */
PGV_API int32_t pgv_render_frame(pgv_env* env, int32_t index, int32_t width, int32_t height, uint8_t* h_rgb);

/* Measurement helper for bench.py: runs `steps` synthetic steps and returns, from HIP events recorded
 * on the env's stream, the total time of the region and the summed time of the dominant (render)
 * kernel launches inside it. */
PGV_API int32_t pgv_timed_steps(pgv_env* env, int32_t steps, uint32_t run_seed, double* total_ms,
                                double* render_kernel_ms);
/* render_kernel_ms == NULL: the region holds nothing but the steps between its two events (no per-launch events): the
 * form `value` is measured with.  render_kernel_ms is the render KERNEL's launches alone: a game's render pre-pass
 * (setup_kernel, since round 4 a launch of its own in front of the render kernel) is NOT in it — pgv_step_phases
 * reports that one beside it.
 *
 * Per-step detail for the same kind of run: h_step_ms[s] = time from the start of step s to the start of step s+1 (to
 * the end of the run for the last), h_render_ms[s] = its render launch (again without the pre-pass), both from HIP events
 * on the env's stream (host arrays of `steps` floats, either may be NULL).  For latency percentiles and the roofline
 * window — never for `value`: events sit inside the region. */
PGV_API int32_t pgv_step_times(pgv_env* env, int32_t steps, uint32_t run_seed, float* h_step_ms, float* h_render_ms);
/* The same run with a step cut into its four phases by events on the env's stream (any pointer may be NULL):
 *   h_logic_ms    the logic kernels (level install / auto-reset, agent, entities, resolve — whatever the game launches
 *                 in front of its frame), including the side-stream launch of the level generator;
 *   h_prepass_ms  the render pre-pass (setup_kernel; 0 for a game or debug mode without one);
 *   h_render_ms   the render kernel (jumper: + the list kernel that walks the frames the pre-pass handed back);
 *   h_late_ms     what follows the render launch inside the step (chaser: the join with its reset stream and the late
 *                 pass over the envs that were reset).
 * logic + prepass + render + late = step.  bench.py's roofline.render_path is prepass + render + late. */
PGV_API int32_t pgv_step_phases(pgv_env* env, int32_t steps, uint32_t run_seed, float* h_step_ms, float* h_logic_ms,
                                float* h_prepass_ms, float* h_render_ms, float* h_late_ms);
/* … of `count` envs stepping side by side, each on its own stream (the mixed workload: step s of every env is enqueued
 * before step s+1 of any, as pgv_step_synthetic_many does).  h_ms: host floats [count][5][steps] — per env: step, logic,
 * prepass, render, late.  The streams overlap on the device, so an env's phases are what ITS stream saw, not a share of
 * the wall clock. */
PGV_API int32_t pgv_step_phases_many(pgv_env* const* envs, int32_t count, int32_t steps, uint32_t run_seed, float* h_ms);

/* Debug switches (tests only); none changes a result.  Bit 0: render the background and tile layer by replaying the
 * draw list one blit at a time instead of the fused row composer.  Bit 8: no level prefetch — every reset generates its
 * level inside the step.  Bit 21: no render pre-pass — every frame's workgroup does its own set-up, as the frames the
 * pre-pass hands back do anyway.  Bit 23: the pre-pass hands back every third env's frame (the way for tests to the
 * hand-back path of games that never take it in a normal run).  Bit 24: coinrun works every hazard's boxes out behind
 * the agent instead of the few its entity lanes pre-selected (the fallback a normal run never takes).  Bit 25: chaser's
 * enemies take their turns one after the other on the env's random stream itself instead of side by side on outputs
 * peeked from it (what the last few words of a 624-word block take in a normal run).  Any other bit is refused. */
PGV_API int32_t pgv_set_debug(pgv_env* env, int32_t flags);

/* Parity taps (host pointers): game-defined state vector / tile ids of one env; return the full
 * length, copy at most `cap` items. */
PGV_API int32_t pgv_dump_state(pgv_env* env, int32_t index, float* h_out, int32_t cap);
PGV_API int32_t pgv_dump_tiles(pgv_env* env, int32_t index, uint8_t* h_out, int32_t cap);

PGV_API const char* pgv_last_error(void);

#ifdef __cplusplus
}
#endif

#endif /* PROCGEN2_VEC_H */
