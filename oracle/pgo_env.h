// ORACLE — TEST INFRASTRUCTURE ONLY (see pgo_common.h header).
//
// One env instance = what one loaded reference .so image holds in its globals
// (SURVEY.md §0 fact 1).  Life cycle mirrors the cenv entry points:
//   make(seed)  ≙ cenv_make   (games/<g>/<g>.cpp: seeds rng, builds level 0 which is never observed — D1)
//   reset()     ≙ cenv_reset  (optional reseed, new level, render, pack)
//   step(a)     ≙ cenv_step   (sub-steps, render, pack)
#pragma once

#include "pgo_raster.h"

namespace pgo {

constexpr int kObsW = 64;
constexpr int kObsH = 64;
constexpr int kObsBytes = kObsW * kObsH * 3;

class Env {
   public:
    Env() : surface_(kObsW, kObsH) { painter_.target = &surface_; }
    virtual ~Env() = default;

    void make(uint32_t seed) {
        rng_.seed(seed);
        on_make();
        new_level();
    }
    void reset(bool reseed, int32_t seed) {
        // `rng.seed(options[i].value.i)`: int → unsigned long → mod 2^32 (coinrun.cpp:316)
        if (reseed) rng_.seed(static_cast<uint32_t>(seed));
        new_level();
        observe();
    }
    void step(int action) {
        advance(action);
        observe();
    }

    // Level-seed mode (include/procgen2_vec.h pgv_make_levels): the level make() built is the one that is played, so it
    // gets presented without the reset() that normally follows.
    void present() { observe(); }

    void set_render_enabled(bool on) { painter_.enabled = on; }

    // Distribution mode = the reference's compile-time `System_Tilemap::Config` / `Distribution_Mode` of the game
    // (include/procgen2_vec.h PGV_MODE_*; always a resolved value here, never "default").  Set before make().
    enum Mode { kEasy = 1, kHard = 2, kMemory = 3, kExtreme = 4 };
    void set_mode(int mode) { mode_ = mode; }
    // include/procgen2_vec.h pgv_config.game_flags (coinrun: PGV_COINRUN_NO_*).  Set before make().
    void set_flags(uint32_t flags) { flags_ = flags; }

    // cenv_render (coinrun.cpp:393-411): render_game(false) into a width×height target, packed RGB.  Like the
    // reference it leaves the camera scale/size at the window's values afterwards: the next observation render sets
    // them again, and before that only bossfight reads them (reset() and both update()s, D15).
    void render_frame(int width, int height, uint8_t* out_rgb) {
        Surface big(width, height);
        Surface* keep_target = painter_.target;
        painter_.target = &big;
        view_w_ = width;
        view_h_ = height;
        paint();
        pack_rgb(big, out_rgb);
        painter_.target = keep_target;
        view_w_ = kObsW;
        view_h_ = kObsH;
    }

    float reward = 0.0f;
    bool terminated = false;
    bool truncated = false;
    uint8_t obs[kObsBytes] = {0};

    // Debug/parity taps: a flat float dump of the logical game state, game-defined layout.
    virtual int dump_state(float* out, int cap) const = 0;
    // Level dump: tile ids as the game stores them (column-major y + x*H), returns count.
    virtual int dump_tiles(uint8_t* out, int cap) const = 0;
    long draw_calls() const { return painter_.draw_calls; }
    uint32_t rng_peek() {  // next raw output without consuming
        std::mt19937 copy = rng_.eng;
        return static_cast<uint32_t>(copy());
    }

   protected:
    virtual void on_make() = 0;         // load textures, fixed setup
    virtual void new_level() = 0;       // the game's reset()
    virtual void advance(int action) = 0;  // the sub-step loop of cenv_step
    virtual void paint() = 0;           // render_game(true)

    void observe() {
        if (!painter_.enabled) return;  // logic-only traces: no textures needed
        paint();
        pack_rgb(surface_, obs);
    }

    int mode_ = kHard;
    uint32_t flags_ = 0;
    int view_w_ = kObsW, view_h_ = kObsH;  // `width`, `height` of render_game(is_obs): the target being painted
    Rng rng_;
    Surface surface_;
    Painter painter_;
};

Env* new_coinrun();
Env* new_maze();
Env* new_bossfight();
Env* new_climber();
Env* new_caveflyer();
Env* new_chaser();
Env* new_jumper();

}  // namespace pgo
