/* procgen2_cenv.h — the single-env "cenv" C ABI of Procgen2, re-declared for the MI355X engine.
 *
 * This header declares exactly the symbols the reference's FFI binds
 * (reference: cenv/cenv.h:26-133, bound by cenv/cenv.py:157-182 through ctypes):
 *   4 data symbols   make_data, reset_data, step_data, render_data      (cenv.h:122-125)
 *   6 functions      cenv_get_env_version, cenv_make, cenv_reset, cenv_step, cenv_render, cenv_close
 *                                                                        (cenv.h:128-133)
 * Struct layouts are the x86-64 layouts cenv.py mirrors with ctypes (cenv.py:62-111):
 *   cenv_value 8 B, cenv_value_buffer 8 B, cenv_key_value 24 B, cenv_option 24 B,
 *   cenv_step_data 40 B, cenv_render_data 24 B.
 *
 * libprocgen2_hip.so implements this ABI on top of the vector engine (procgen2_vec.h): one loaded
 * image = one vector env of `num_envs` (option, default 1) envs of game `game` (option, default
 * from the library name: libCoinRun.so / libMaze.so aliases).  With num_envs == 1 the behaviour is
 * the reference's: "screen" is BYTE[12288], reward/terminated are scalars, no auto-reset.
 * With num_envs > 1 the same call shapes carry batches the way the unmodified cenv.py allows
 * (SURVEY.md §8b): action {"action": int32[N]}, observations "screen" BYTE[N*12288] plus
 * "reward" FLOAT[N] and "terminated" BYTE[N]; step_data.reward.f is the batch mean.
 *
 * Error convention (cenv.py:208-209,285-286,340-341): 0 = OK, non-zero = failure; the message is
 * available through pgv_last_error().  A missing/failed GPU is a failure, never a CPU fallback.
 */
#ifndef PROCGEN2_CENV_H
#define PROCGEN2_CENV_H

#ifdef __cplusplus
extern "C" {
#endif

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#define CENV_API __attribute__((__visibility__("default")))
#define CENV_VERSION 1

/* cenv.h:29-39 */
typedef enum {
    CENV_VALUE_TYPE_INT = 0,
    CENV_VALUE_TYPE_FLOAT = 1,
    CENV_VALUE_TYPE_DOUBLE = 2,
    CENV_VALUE_TYPE_BYTE = 3,
    CENV_SPACE_TYPE_BOX = 4,
    CENV_SPACE_TYPE_MULTI_DISCRETE = 5
} cenv_value_type;

/* cenv.h:42-47 */
typedef union {
    int32_t i;
    float f;
    double d;
    uint8_t b;
} cenv_value;

/* cenv.h:50-55 */
typedef union {
    int32_t* i;
    float* f;
    double* d;
    uint8_t* b;
} cenv_value_buffer;

/* cenv.h:57-65 */
typedef struct {
    const char* key;
    cenv_value_type value_type;
    int32_t value_buffer_size;
    cenv_value_buffer value_buffer;
} cenv_key_value;

/* cenv.h:68-74 */
typedef struct {
    const char* name;
    cenv_value_type value_type;
    cenv_value value;
} cenv_option;

/* cenv.h:77-83 */
typedef struct {
    int32_t observation_spaces_size;
    cenv_key_value* observation_spaces;
    int32_t action_spaces_size;
    cenv_key_value* action_spaces;
} cenv_make_data;

/* cenv.h:86-92 */
typedef struct {
    int32_t observations_size;
    cenv_key_value* observations;
    int32_t infos_size;
    cenv_key_value* infos;
} cenv_reset_data;

/* cenv.h:95-105 */
typedef struct {
    int32_t observations_size;
    cenv_key_value* observations;
    cenv_value reward;
    bool terminated;
    bool truncated;
    int32_t infos_size;
    cenv_key_value* infos;
} cenv_step_data;

/* cenv.h:108-119 — image addressed channel + channels*(x + width*y) */
typedef struct {
    cenv_value_type value_type;
    int32_t value_buffer_width;
    int32_t value_buffer_height;
    int32_t value_buffer_channels;
    cenv_value_buffer value_buffer;
} cenv_render_data;

/* cenv.h:122-125 */
CENV_API extern cenv_make_data make_data;
CENV_API extern cenv_reset_data reset_data;
CENV_API extern cenv_step_data step_data;
CENV_API extern cenv_render_data render_data;

/* cenv.h:128-133.
 * cenv_make options (games/coinrun/coinrun.cpp:133-151): "seed" INT, "width"/"height" INT (human frame
 * size); engine additions, INT: "num_envs", "game" (0 coinrun, 1 maze, 2 bossfight, 3 climber, 4 caveflyer, 5 chaser, 6 jumper), "device", "env_offset".
 * cenv_reset options (coinrun.cpp:310-318): "seed" INT (env i reseeds with seed + i). */
CENV_API int32_t cenv_get_env_version(void);
CENV_API int32_t cenv_make(const char* render_mode, cenv_option* options, int32_t options_size);
CENV_API int32_t cenv_reset(cenv_option* options, int32_t options_size);
CENV_API int32_t cenv_step(cenv_key_value* actions, int32_t actions_size);
CENV_API int32_t cenv_render(void);
CENV_API void cenv_close(void);

#ifdef __cplusplus
}
#endif

#endif /* PROCGEN2_CENV_H */
