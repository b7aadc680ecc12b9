// bossfight on gfx950: a shooter without a tile map — boss phase machine, 64 + 32
// bullets, 8 explosions, and the per-env mt19937 consumed INSIDE the step.
//
// Reference:
//   step   games/bossfight/bossfight.cpp:294-345, common_systems.cpp:199-390 (boss), :494-683 (agent),
//          :103-185 (fire_pattern), :187-197 (show_damage)
//   render games/bossfight/bossfight.cpp:401-424, common_systems.cpp:392-450, :685-721, :22-48
//   reset  games/bossfight/bossfight.cpp:426-504, common_systems.cpp:452-470, :724-737
// Config = the reference's compile-time default, hard_mode (common_systems.h:63-65).
//
// Machine mapping: logic = a gang of kGang adjacent lanes per env (pg_gang.h): the env's scalars uniformly in all of
// them, the three bullet rings walked kGang slots at a time; render = two wavefronts per env with the target in LDS.  std::cos/std::sin(float) are pg_sincos.h (bit-identical to the glibc the
// reference links); every `M_PI` expression keeps the reference's float/double promotion points.
//
// Entity ids are fixed by creation order (player 0, boss 1, barriers 2..): the hazard set therefore iterates
// barriers newest-first, then the boss (libstdc++ unordered_set with 13 buckets, one id per bucket, each insert
// goes to the list head — SURVEY.md T3), which is the order first-hit tests below use.
#include "pg_engine.h"
#include "pg_frame.h"
#include "pg_gang.h"
#include "pg_geom.h"
// (Measured and rejected, round 4 — commit bd6a4f0 has it, PG_REPLAY_SHARE_AREA: the boss and its shield (35×29 and 23×21
// pixels, mostly in the upper wave's rows) done by both waves with their rows dealt alternately.  Bit-exact; render
// 0.783 -> 0.806 ms: the two barriers per shared draw cost more than the idle lower wave.)
#include "pg_render.h"
#include "pg_prepass.h"
#include "pg_stamps.h"
#include "pg_rng.h"
#include "pg_sincos.h"

namespace pg {
namespace PG_VARIANT_NS {
namespace bossfight {

// common_systems.cpp:104,202 (`config.mode == hard_mode ? … : …`; common_systems.h:64: hard_mode is the default)
#if PG_VARIANT == 0
constexpr float kBossBulletSpeed = 0.1f, kShieldedSpread = 80.0f;
#elif PG_VARIANT == 1  // easy_mode
constexpr float kBossBulletSpeed = 0.05f, kShieldedSpread = 30.0f;
#else
#error "bossfight: unknown PG_VARIANT"
#endif

constexpr int kAgentShots = 32, kBossShots = 64, kBooms = 8, kRocks = 4;
// What the render pre-pass's tables hold (setup_kernel → render_kernel, no complete-path fallback in the lean kernel): the
// second list — the boss, its shield, explosions, barriers, the agent's bullets, the agent — as draws of rank < kPrepDraws
// (prep_draws_flush drops any beyond), and the boss's bullets one per lane of a wavefront with their count in a byte of
// the meta word.  True of these constants; a change to them has to give the lean kernel a fallback first.
static_assert(2 + kBooms + kRocks + kAgentShots + 1 <= 64 /* pg_prepass.h kPrepDraws */ && kBossShots <= 64,
              "bossfight's lean render kernel has no path for draws beyond its tables");
constexpr double kPi = 3.14159265358979323846;  // M_PI

enum Tex {
    kTexSpace = 0,    // 13 backgrounds
    kTexRock = 13,    // 8 barrier sprites
    kTexBoss = 21,    // 4
    kTexPlayer = 25,  // 4
    kTexLaser = 29,   // 3
    kTexBoom = 32,    // 5
    kTexShield = 37,
    kTexCount = 38
};

// per-env scalar floats
enum {
    F_AX, F_AY, F_AVX, F_AVY, F_ATIMER, F_BX, F_BY, F_BVX, F_BVY, F_PHASE_T, F_ATTACK_T, F_EXPLO_T, F_DAMAGE_T,
    F_MOVE_T,
    F_CAMW, F_CAMH,  // gr.camera_size as the last render_game left it (D15): 64×64 after an observation render
    F_COUNT
};
// per-env scalar ints
enum {
    I_FLAGS, I_A_NEXT, I_A_COUNT, I_PHASE, I_WEAPON, I_HP, I_B_NEXT, I_B_COUNT, I_X_NEXT, I_X_COUNT, I_SKINS, I_NROCKS,
    I_COUNT
};
constexpr int kFlagAlive = 1, kFlagListed = 2;
// I_SKINS: a_ship | a_laser<<4 | b_ship<<8 | b_laser<<12 | backdrop<<16
// shot fields (floats); agent shots also have a "bouncing" byte
enum { S_X, S_Y, S_VX, S_VY, S_FRAME, S_BOUNCE_T, S_ROT, S_SN, S_CS, S_COUNT };  // S_SN/S_CS (boss bullets): the drawing angle as raster spec S6 takes it, int bits, fixed when fired

// The rings are per-env contiguous ([env][field][slot]): a gang's lanes and the render wavefront's lanes both index
// them by slot, so one field of one env is one coalesced request.  Scalars are [field][env].
struct State {
    int n;
    uint32_t* mt;   // [n][625]
    float* f;       // [F_COUNT][n]
    int32_t* i;     // [I_COUNT][n]
    float* ashot;   // [n][S_COUNT][32]
    uint8_t* abnc;  // [n][32]  bouncing flag
    float* bshot;   // [n][S_COUNT][64]   (S_BOUNCE_T unused)
    float* boom;    // [n][3][8]  x, y, frame
    float* rock;    // [3][4][n]  x, y, texture index
    struct Prep {   // the render pre-pass's hand-over (setup_kernel → render_kernel; scratch memory, not state)
        uint32_t* backdrops;  // [13][128]  per backdrop: bg_offset of pixel columns 0-63, of pixel rows 0-63 (backdrop_kernel, once)
        uint32_t* backdrop_px;  // [13][64 × 64]  … and the backdrop as it lands on the frame, composed once (backdrop_kernel): the camera never moves
        uint32_t* meta;       // [n]        boss bullets | other draws << 8 | backdrop << 16
        uint32_t* bullets;    // [n][64][kBulletWords]  visible boss bullets in drawing order
        uint32_t* draws;      // [n][64][kBlitWords]    the visible draws of the second list in drawing order
    } prep;
    // A second buffer for every env's random stream and its selector (pg_gang.h GangRng): the stream's next 624 words are
    // worked out ahead of the step that needs them (a few extra workgroups of setup_kernel, pg_rng.h mt_next_block_wave)
    // into whichever buffer is not current, and the gang that runs out of numbers changes buffers instead of regenerating
    // the block (19 of the logic kernel's 126 µs were wavefronts waiting for one of their four gangs to do that).  Scratch
    // memory, not state: begin_level and prepare_save make mt[env] the current block again.
    uint32_t* mt_other;  // [n][kMtN]
    uint8_t* mt_sel;     // [n]  bit 1: the current words are in mt_other[env]; bit 0: the other buffer holds the next block
    uint32_t stamps;  // word offset of the stamp table in the atlas (pg_stamps.h; BossfightGame::extend_atlas), 0 = none
};
constexpr int kBulletWords = 10;  // pg_render.h BlitWords, sine, cosine (16.16), bounding box on the target (x, y, w, h: a byte each), one spare
constexpr int kBackdrops = 13;

PG_D float& SF(const State& s, int field, int env) { return s.f[size_t(field) * s.n + env]; }
PG_D int32_t& SI(const State& s, int field, int env) { return s.i[size_t(field) * s.n + env]; }
PG_D float& AS(const State& s, int field, int k, int env) { return s.ashot[(size_t(env) * S_COUNT + field) * kAgentShots + k]; }
PG_D uint8_t& AB(const State& s, int k, int env) { return s.abnc[size_t(env) * kAgentShots + k]; }
PG_D float& BS(const State& s, int field, int k, int env) { return s.bshot[(size_t(env) * S_COUNT + field) * kBossShots + k]; }
PG_D float& BM(const State& s, int field, int k, int env) { return s.boom[(size_t(env) * 3 + field) * kBooms + k]; }
PG_D float& RK(const State& s, int field, int k, int env) { return s.rock[(size_t(field) * kRocks + k) * s.n + env]; }

// Observation camera: size 64, scale 1 (bossfight.cpp:412-413).
constexpr float kCamSize = 64.0f, kCamScale = 1.0f;

// D15: reset() (bossfight.cpp:434,458) and both update()s (common_systems.cpp:227-228,513-514) read gr.camera_size and
// gr.camera_scale as the LAST render_game left them: the observation's 64 / 1.0 — or, right after a human-size
// cenv_render, the window's W×H and 1.0·W/64, which for a non-square window moves the spawn row, the barriers and the
// screen rectangle of the next step or reset.  F_CAMW/F_CAMH hold that size per env: the frame kernel writes the
// window's, whoever reads them (logic / level code, always followed by an observation render) puts 64 back.
struct View {  // (reading: View{w, h, 1·w/64}; whoever reads a size other than 64 puts 64 back)
    float w, h, sc;
};
PG_D Box screen_box(const View& v) {  // common_systems.cpp:224-226, 513-515
    return Box{-v.w / v.sc * kPxUnit * 0.5f, -v.h / v.sc * kPxUnit * 0.5f, v.w / v.sc * kPxUnit, v.h / v.sc * kPxUnit};
}

// ------------------------------------------------------------------------------------------------
// reset (bossfight.cpp:426-504)
// ------------------------------------------------------------------------------------------------
// R: the env's stream, by one lane (LaneRng: make / reset kernels) or by a gang whose lanes all run this and draw the
// same numbers (pg_gang.h GangRng: the auto-reset inside the logic kernel); `lead` = this lane does the writing.
struct LaneRng {
    uint32_t* mt;
    PG_D float real(float a, float b) { return rng_real(mt, a, b); }
    PG_D int integer(int lo, int hi) { return rng_int(mt, lo, hi); }
};
template <class R>
PG_D void new_level(const State& s, int env, R& rng, bool lead) {
    const float cam_w = SF(s, F_CAMW, env), cam_h = SF(s, F_CAMH, env);  // take_view: every lane reads, the lead puts 64 back
    const View view{cam_w, cam_h, 1.0f * cam_w / 64.0f};
    const float ax = rng.real(-1.0f, 1.0f) * view.w / view.sc * kPxUnit * 0.5f;
    const int want = rng.integer(1, 4);
    Box placed[kRocks];
    float rock_x[kRocks], rock_y[kRocks], rock_t[kRocks];
    int n_rocks = 0;
#pragma unroll
    for (int k = 0; k < kRocks; k++) {
        if (k < want) {
            const float px = rng.real(-1.0f, 1.0f) * view.w / view.sc * kPxUnit * 0.5f * 0.9f;
            const float py = view.h / view.sc * kPxUnit * 0.5f - rng.real(0.7f, 1.2f);
            const Box wc{px + -0.1f, py + -0.1f, 0.2f, 0.2f};
            bool clash = false;
#pragma unroll
            for (int j = 0; j < kRocks; j++)
                if (j < k && box_hit(wc, placed[j])) clash = true;
            if (!clash) {
                const float t = static_cast<float>(rng.integer(0, 7));
#pragma unroll
                for (int j = 0; j < kRocks; j++)  // (selects: the table stays in registers)
                    if (j == n_rocks) {
                        rock_x[j] = px;
                        rock_y[j] = py;
                        rock_t[j] = t;
                    }
                n_rocks++;
                placed[k] = wc;
            } else {
                placed[k] = Box{0.0f, 0.0f, 0.0f, 0.0f};
            }
        }
    }
    const int backdrop = rng.integer(0, 12);
    rng.real(0.0f, 1.0f);  // current_background_offset_x / _y: drawn, never used
    rng.real(0.0f, 1.0f);
    // System_Agent::reset, then System_Mob_AI::reset
    const int a_ship = rng.integer(0, 3);
    const int a_laser = rng.integer(0, 2);
    const int b_ship = rng.integer(0, 3);
    const int b_laser = rng.integer(0, 2);
    if (!lead) return;
    if (cam_w != kCamSize || cam_h != kCamSize) {
        SF(s, F_CAMW, env) = kCamSize;
        SF(s, F_CAMH, env) = kCamSize;
    }
    SF(s, F_AX, env) = ax;
    SF(s, F_AY, env) = view.h / view.sc * kPxUnit * 0.5f;
    SF(s, F_AVX, env) = 0.0f;
    SF(s, F_AVY, env) = 0.0f;
    SF(s, F_BX, env) = 0.0f;
    SF(s, F_BY, env) = 0.0f;
    SF(s, F_BVX, env) = 0.0f;
    SF(s, F_BVY, env) = 0.0f;
    SF(s, F_PHASE_T, env) = 0.0f;
    SF(s, F_ATTACK_T, env) = 0.0f;
    SI(s, I_PHASE, env) = 0;
    SI(s, I_WEAPON, env) = 0;
    SI(s, I_HP, env) = 0;
#pragma unroll
    for (int j = 0; j < kRocks; j++)
        if (j < n_rocks) {
            RK(s, 0, j, env) = rock_x[j];
            RK(s, 1, j, env) = rock_y[j];
            RK(s, 2, j, env) = rock_t[j];
        }
    SI(s, I_NROCKS, env) = n_rocks;
    SI(s, I_A_NEXT, env) = 0;
    SI(s, I_A_COUNT, env) = 0;
    SF(s, F_ATIMER, env) = 0.0f;
    SI(s, I_B_NEXT, env) = 0;
    SI(s, I_X_NEXT, env) = 0;
    SI(s, I_B_COUNT, env) = 0;
    SI(s, I_X_COUNT, env) = 0;
    SF(s, F_EXPLO_T, env) = 0.0f;
    SF(s, F_DAMAGE_T, env) = 0.0f;
    SF(s, F_MOVE_T, env) = 0.0f;
    SI(s, I_SKINS, env) = a_ship | (a_laser << 4) | (b_ship << 8) | (b_laser << 12) | (backdrop << 16);
    SI(s, I_FLAGS, env) = kFlagAlive;  // agent alive; sprite draw list cleared (D2)
}

// ------------------------------------------------------------------------------------------------
// step (bossfight.cpp:308-325)
// ------------------------------------------------------------------------------------------------
// One env = one gang (pg_gang.h).  Everything in Live is uniform over the gang; `q.g` only decides which ring slots a
// lane owns (slot mod width) — it writes the bullets fired into them and is the one that moves them.
#ifndef PG_BOSSFIGHT_GANG
#define PG_BOSSFIGHT_GANG 8  // (16 until the random streams changed buffers instead of regenerating, round 6: 8 then 126.5 -> 130.8 M; 4: 124.6; 32: 118.3)
#endif
#ifndef PG_BOSSFIGHT_WAVES
#define PG_BOSSFIGHT_WAVES 3  // wavefronts per SIMD the logic kernel's registers are capped for
#endif
constexpr int kGang = PG_BOSSFIGHT_GANG;
constexpr int kBoomGang = kGang < kBooms ? kGang : kBooms;  // lanes of a trip over the explosions
using Q = Gang<kGang>;
using Rng = GangRng<kGang>;

// The three rings of an env while its gang steps it: the slots that hold something are read once, before the first
// sub-step, into LDS — indexable registers: a slot is only ever touched by the lane that owns it — and the slots that
// changed go back after the last, so the eight trips of a step wait for LDS instead of for a store and a load each.
#ifndef PG_BOSSFIGHT_AGENT_RING_LDS
#define PG_BOSSFIGHT_AGENT_RING_LDS 0  // 1: the agent's ring is staged in LDS too (896 bytes a gang more)
#endif
struct RingsLds {  // one per gang
    float b[S_FRAME + 1][kBossShots];       // S_X .. S_FRAME of the boss's bullets
#if PG_BOSSFIGHT_AGENT_RING_LDS
    float a[S_BOUNCE_T + 1][kAgentShots];   // … and S_BOUNCE_T of the agent's
    uint8_t a_bouncing[kAgentShots];
#endif
    float boom[kBooms];                     // frame
};
static_assert(S_X == 0 && S_FRAME == 4 && S_BOUNCE_T == 5, "the ring fields staged in LDS come first");
#if PG_BOSSFIGHT_AGENT_RING_LDS
#define PG_AS(R, s, f, k, env) (R).lds->a[f][k]
#define PG_AB(R, s, k, env) (R).lds->a_bouncing[k]
#else  // straight from memory: few of them are in flight, and a slot is only ever touched by the lane that owns it
#define PG_AS(R, s, f, k, env) AS(s, f, k, env)
#define PG_AB(R, s, k, env) AB(s, k, env)
#endif
struct Rings {  // a lane's view: the gang's rings, and which of ITS slots it has written (bit j: slot g + width·j)
    RingsLds* lds;
    uint32_t dirty_a, dirty_b, dirty_x;
};
struct Live {  // the hot scalars of one env, kept in registers over the four sub-steps
    float ax, ay, avx, avy, atimer;
    float bx, by, bvx, bvy, phase_t, attack_t, explo_t, damage_t, move_t;
    int a_next, a_count, phase, weapon, hp, b_next, b_count, x_next, x_count, n_rocks;
    bool a_alive;
    Box scr;  // the screen rectangle both update()s clamp against (screen_box of the env's View, D15)
    float rock_x[kRocks], rock_y[kRocks];  // the barriers (fixed for the episode): read once a step, not once per bullet
};

PG_D bool hit(const Box& a, const Box& b) {  // box_hit without short circuits (no branches)
    return (a.x < b.x + b.w) & (a.x + a.w > b.x) & (a.y < b.y + b.h) & (a.y + a.h > b.y);
}
PG_D bool hits_a_rock(const Live& v, const Box& sb) {
    bool rock = false;
#pragma unroll
    for (int k = 0; k < kRocks; k++)
        rock = rock | ((k < v.n_rocks) & hit(sb, Box{v.rock_x[k] + -0.1f, v.rock_y[k] + -0.1f, 0.2f, 0.2f}));
    return rock;
}

// `shots` calls of fire_bullet one after the other (common_systems.cpp:75-88; each fires only while fewer than 64 are
// in flight): shot j goes into ring slot b_next + j, written by the lane that owns the slot.
template <class Rot>
PG_D void boss_volley(const State& s, Rings& R, int env, Live& v, Q q, int shots, float speed, Rot rotation_of_shot) {
    const int room = kBossShots - v.b_count;
    const int fired = shots < room ? shots : room;
    for (int j = (q.g - v.b_next) & (kGang - 1); j < fired; j += kGang) {
        const int k = (v.b_next + j) & (kBossShots - 1);
        const float rotation = rotation_of_shot(j);
        BS(s, S_ROT, k, env) = rotation;  // (angle, sine and cosine: for the render kernels only, straight to memory)
        {   // what System_Mob_AI::render's angle (rotation + π/2) comes to in the raster (pg_render.h rotation_of): once
            // per bullet here, instead of sinf and cosf in every lane of both render wavefronts every frame
            int sn, cs;
            rotation_of(static_cast<float>(rotation + kPi * 0.5f), sn, cs);
            BS(s, S_SN, k, env) = __int_as_float(sn);
            BS(s, S_CS, k, env) = __int_as_float(cs);
        }
        R.lds->b[S_VX][k] = sc_cosf(rotation) * speed;
        R.lds->b[S_VY][k] = -sc_sinf(rotation) * speed;
        R.lds->b[S_X][k] = v.bx;
        R.lds->b[S_Y][k] = v.by;
        R.lds->b[S_FRAME][k] = 0.0f;
        R.dirty_b |= 1u << (k / kGang);
    }
    v.b_next = (v.b_next + fired) & (kBossShots - 1);
    v.b_count += fired;
}

PG_D void fire_pattern(const State& s, Rings& R, int env, Live& v, Rng& rng, Q q, int pattern, float dt) {  // :103-185
    const float bullet_speed = kBossBulletSpeed;
    float& timer = v.attack_t;
    switch (pattern) {
        case -1:
            if (rng.real(0.0f, 1.0f) < 0.1f * dt) {
                const float rot = static_cast<float>(kPi * (1.0f + rng.real(0.0f, 1.0f)));
                boss_volley(s, R, env, v, q, 1, bullet_speed, [&](int) { return rot; });
            }
            break;
        case 0:
            if (timer >= 8.0f) {
                timer = 0.0f;
                boss_volley(s, R, env, v, q, 5, bullet_speed,
                            [](int k) { return static_cast<float>(kPi * 1.5f + (k - 2) * kPi * 0.125f); });
            } else
                timer += dt;
            break;
        case 1:
            if (timer >= 5.0f) {
                timer = 0.0f;
                int w = static_cast<int>(timer / 5.0f);
                w = abs(8 - (w % 16));
                boss_volley(s, R, env, v, q, 4, bullet_speed,
                            [w](int k) { return static_cast<float>(kPi * (1.25f + w * 0.0625f) + k * kPi * 0.5f); });
            } else
                timer += dt;
            break;
        case 2:
            if (timer >= 10.0f) {
                timer = 0.0f;
                const float offset = static_cast<float>(rng.real(0.0f, 1.0f) * 2.0f * kPi);
                boss_volley(s, R, env, v, q, 8, bullet_speed,
                            [offset](int k) { return static_cast<float>(kPi * 0.25f * k + offset); });
            } else
                timer += dt;
            break;
        case 3:
            if (timer >= 4.0f) {
                timer = 0.0f;
                const float rot = static_cast<float>(kPi * (1.0f + rng.real(0.0f, 1.0f)));
                boss_volley(s, R, env, v, q, 1, bullet_speed, [&](int) { return rot; });
            } else
                timer += dt;
            break;
    }
}

PG_D bool agent_update(const State& s, Rings& R, int env, Live& v, Rng& rng, Q q, float dt, int action) {  // :494-683
    const float mixrate = 0.5f, speed = 0.1f, bullet_time = 5.0f, bullet_speed = 0.1f;
    const float bounce_speed = 0.05f, bounce_time = 10.0f, explosion_rate = 0.3f;
    const Box scr = v.scr;
    const float mx = static_cast<float>((action == 6 || action == 7 || action == 8) -
                                        (action == 0 || action == 1 || action == 2));
    const float my = static_cast<float>((action == 2 || action == 5 || action == 8) -
                                        (action == 0 || action == 3 || action == 6));
    const bool fire = action == 9;
    v.avx += mixrate * (mx * speed - v.avx) * dt;
    v.avy += mixrate * (-my * speed - v.avy) * dt;
    v.ax += v.avx * dt;
    v.ay += v.avy * dt;
    Box wc{v.ax + -0.15f, v.ay + -0.1f, 0.3f, 0.2f};
    if (wc.x < scr.x) {
        v.ax += scr.x - wc.x;
        v.avx = 0.0f;
    } else if (wc.x + wc.w > scr.x + scr.w) {
        v.ax += scr.x + scr.w - (wc.x + wc.w);
        v.avx = 0.0f;
    }
    if (wc.y < scr.y) {
        v.ay += scr.y - wc.y;
        v.avy = 0.0f;
    } else if (wc.y + wc.h > scr.y + scr.h) {
        v.ay += scr.y + scr.h - (wc.y + wc.h);
        v.avy = 0.0f;
    }
    wc = Box{v.ax + -0.15f, v.ay + -0.1f, 0.3f, 0.2f};
    if (fire) {
        if (v.atimer == 0.0f && v.a_count < kAgentShots) {
            v.atimer = bullet_time;
            const int k = v.a_next;
            if ((k & (kGang - 1)) == q.g) {
                AS(s, S_ROT, k, env) = 0.0f;
                PG_AS(R, s, S_VX, k, env) = 0.0f;
                PG_AS(R, s, S_VY, k, env) = -bullet_speed;
                PG_AS(R, s, S_X, k, env) = v.ax;
                PG_AS(R, s, S_Y, k, env) = v.ay;
                PG_AS(R, s, S_FRAME, k, env) = 0.0f;
                PG_AS(R, s, S_BOUNCE_T, k, env) = 0.0f;
                PG_AB(R, s, k, env) = 0;
                R.dirty_a |= 1u << (k / kGang);
            }
            v.a_next = (v.a_next + 1) % kAgentShots;
            v.a_count++;
        } else {
            v.atimer = fmaxf(0.0f, v.atimer - dt);
        }
    }
    // any hazard: a boolean, order-free
    const bool crashed = hits_a_rock(v, wc) || hit(wc, Box{v.bx + -0.6f, v.by + -0.4f, 1.2f, 0.8f});
    v.a_alive = v.a_alive && !crashed;
    // The reference visits the bullets newest first, one after the other, while `count` shrinks under the loop's feet
    // (positions beyond it are never reached), and a bounce off the shield draws a random number.  A trip does kGang
    // positions at once: every lane works out what its bullet would do if it is reached; positions are reached as long
    // as i < count − (bullets destroyed before i) — a prefix of the list, because i + destroyed(i) only grows — and the
    // k-th bounce in visiting order takes the k-th next output of the stream.
    const bool shielded = v.phase % 2 == 0;
    const Box boss{v.bx + -0.6f, v.by + -0.4f, 1.2f, 0.8f};
    int count = v.a_count;
    for (int i0 = 0; i0 < count; i0 += kGang) {
        const Q::Trip t = q.trip<kAgentShots>(v.a_next, i0);
        const int k = t.slot;
        const bool mine = t.i < count;
        float frame0 = -1.0f, px = 0.0f, py = 0.0f, vx = 0.0f, vy = 0.0f, btimer = 0.0f;
        bool bouncing = false;
        if (mine) {
            frame0 = PG_AS(R, s, S_FRAME, k, env);
            px = PG_AS(R, s, S_X, k, env);
            py = PG_AS(R, s, S_Y, k, env);
            vx = PG_AS(R, s, S_VX, k, env);
            vy = PG_AS(R, s, S_VY, k, env);
            btimer = PG_AS(R, s, S_BOUNCE_T, k, env);
            bouncing = PG_AB(R, s, k, env) != 0;
        }
        const bool live = mine & (frame0 != -1.0f);
        const Box sb{px - 0.01f, py - 0.01f, 0.02f, 0.02f};
        const bool fly = live & (frame0 == 0.0f);
        const bool off = fly & !hit(sb, scr);
        // hazard set order: barriers newest-first, then the boss — the first one hit decides, and every barrier does the same
        const bool rock = hits_a_rock(v, sb);
        const bool at_boss = fly & !off & !rock & hit(sb, boss);
        const bool bounces = at_boss & shielded;
        const bool bursts = (fly & !off & rock) | (at_boss & !shielded);
        btimer = bounces ? bounce_time : btimer;
        bouncing = bouncing | bounces;
        float frame = off ? 5.0f : (bursts ? 1.0f : frame0);
        const bool ticking = btimer > 0.0f;
        const bool destroyed = live & ((frame >= 5.0f) | (bouncing & !ticking));
        const bool act = live & (t.i < count - Q::before(q.ranked(t, destroyed), t.rank));
        const uint32_t bounce_list = q.ranked(t, bounces & act);
        if (bounce_list) {  // bounce off the shield
            const uint32_t word = rng.next_in_order(bounces & act, Q::before(bounce_list, t.rank), __popc(bounce_list));
            vx = bounces ? (canonical_of(word) * (1.0f - -1.0f) + -1.0f) * bounce_speed : vx;
            vy = bounces ? bounce_speed : vy;
        }
        const int wounds = __popc(q.ballot(at_boss & !shielded & act));
        v.hp = v.hp - wounds < 0 ? 0 : v.hp - wounds;
        vx = (off | bursts) ? 0.0f : vx;
        vy = (off | bursts) ? 0.0f : vy;
        px = px + vx * dt;
        py = py + vy * dt;
        frame = (!destroyed & (frame >= 1.0f)) ? frame + explosion_rate * dt : frame;
        btimer = (bouncing & ticking) ? fmaxf(0.0f, btimer - dt) : btimer;
        frame = destroyed ? -1.0f : frame;
        if (act) {
            PG_AS(R, s, S_X, k, env) = px;
            PG_AS(R, s, S_Y, k, env) = py;
            PG_AS(R, s, S_VX, k, env) = vx;
            PG_AS(R, s, S_VY, k, env) = vy;
            PG_AS(R, s, S_FRAME, k, env) = frame;
            PG_AS(R, s, S_BOUNCE_T, k, env) = btimer;
            PG_AB(R, s, k, env) = bouncing ? 1 : 0;
            R.dirty_a |= 1u << (k / kGang);
        }
        count -= __popc(q.ballot(destroyed & act));
    }
    v.a_count = count;
    return v.a_alive;
}

PG_D bool boss_update(const State& s, Rings& R, int env, Live& v, Rng& rng, Q q, float dt) {  // :199-390
    const float shielded_time = 180.0f + rng.real(0.0f, 1.0f) * kShieldedSpread;  // drawn every sub-step (D14)
    const float unshielded_time = 300.0f, explosion_rate = 0.3f, move_time = 70.0f, damage_time = 80.0f;
    const int boss_hp = 3;
    bool alive = true;
    const Box agent_rect{v.ax + -0.15f, v.ay + -0.1f, 0.3f, 0.2f};
    const Box scr = v.scr;

    if (v.phase_t == 0.0f) {
        v.weapon = rng.integer(0, 3);
        v.attack_t = 0.0f;
        v.hp = boss_hp;
    }
    if (v.phase % 2 == 0) {
        if (v.phase_t >= shielded_time) {
            v.phase_t = 0.0f;
            v.phase++;
        } else
            v.phase_t += dt;
        fire_pattern(s, R, env, v, rng, q, v.weapon, dt);
    } else {
        if (v.phase_t >= unshielded_time) {
            v.phase_t = 0.0f;
            v.phase++;
        } else
            v.phase_t += dt;
        fire_pattern(s, R, env, v, rng, q, -1, dt);
        if (v.hp == 0) {
            if (v.explo_t >= 8.0f) {  // show_damage → explode (:187-197, :91-101)
                v.explo_t = 0.0f;
                const float ox = rng.real(-0.5f, 0.5f) + v.bx;
                const float oy = rng.real(-0.5f, 0.5f) + v.by;
                if (v.x_count < kBooms) {
                    if ((v.x_next & (kBoomGang - 1)) == q.g) {
                        BM(s, 0, v.x_next, env) = ox;
                        BM(s, 1, v.x_next, env) = oy;
                        R.lds->boom[v.x_next] = 0.0f;
                        R.dirty_x |= 1u << (v.x_next / kBoomGang);
                    }
                    v.x_next = (v.x_next + 1) % kBooms;
                    v.x_count++;
                }
            } else
                v.explo_t += dt;
            if (v.damage_t >= damage_time) {
                v.damage_t = 0.0f;
                v.phase++;
                v.hp = boss_hp;
            } else
                v.damage_t += dt;
        }
    }
    if (v.move_t >= move_time) {
        v.move_t = 0.0f;
        const float tx = (rng.real(0.0f, 1.0f) * 2.0f - 1.0f) * 0.5f * scr.w * 0.7f;
        const float ty = ((rng.real(0.0f, 1.0f) * 2.0f - 1.0f) * 0.5f - 0.3f) * scr.h * 0.5f;
        v.bvx = (tx - v.bx) / move_time;
        v.bvy = (ty - v.by) / move_time;
    } else
        v.move_t += dt;
    v.bx += v.bvx * dt;
    v.by += v.bvy * dt;

    // The boss's bullets, as the agent's above; in addition a bullet that meets the agent explodes where it is and ends
    // the loop (D14: the bullets behind it skip this sub-step): only the positions up to the first such one are reached.
    {
        int count = v.b_count;
        bool struck = false;
        for (int i0 = 0; i0 < count && !struck; i0 += kGang) {
            const Q::Trip t = q.trip<kBossShots>(v.b_next, i0);
            const int k = t.slot;
            const bool mine = t.i < count;
            float frame0 = -1.0f, px = 0.0f, py = 0.0f, vx = 0.0f, vy = 0.0f;
            if (mine) {
                frame0 = R.lds->b[S_FRAME][k];
                px = R.lds->b[S_X][k];
                py = R.lds->b[S_Y][k];
                vx = R.lds->b[S_VX][k];
                vy = R.lds->b[S_VY][k];
            }
            const bool live = mine & (frame0 != -1.0f);
            const Box sb{px - 0.01f, py - 0.01f, 0.02f, 0.02f};
            const bool fly = live & (frame0 == 0.0f);
            const bool off = fly & !hit(sb, scr);
            const bool strikes = fly & !off & hit(sb, agent_rect);
            const bool rock = hits_a_rock(v, sb);  // barriers (the boss skips itself); which one is first does not matter
            const bool bursts = strikes | (fly & !off & rock);
            vx = (off | bursts) ? 0.0f : vx;
            vy = (off | bursts) ? 0.0f : vy;
            float frame = off ? 5.0f : (bursts ? 1.0f : frame0);
            const bool moves = live & !strikes;  // (the one that met the agent stays where it is, at frame 1)
            const bool gone = moves & (frame >= 5.0f);
            const bool reached = t.i < count - Q::before(q.ranked(t, gone), t.rank);
            const uint32_t strike_list = q.ranked(t, strikes & reached);
            const int first = strike_list ? __ffs(strike_list) - 1 : kGang;
            const bool act = live & reached & (t.rank <= first);
            px = moves ? px + vx * dt : px;
            py = moves ? py + vy * dt : py;
            const bool burns = moves & !gone & (frame >= 1.0f);
            frame = gone ? -1.0f : (burns ? frame + explosion_rate * dt : frame);
            if (act) {
                R.lds->b[S_X][k] = px;
                R.lds->b[S_Y][k] = py;
                R.lds->b[S_VX][k] = vx;
                R.lds->b[S_VY][k] = vy;
                R.lds->b[S_FRAME][k] = frame;
                R.dirty_b |= 1u << (k / kGang);
            }
            count -= __popc(q.ballot(gone & act));
            struck = strike_list != 0;  // later bullets skip this sub-step (D14)
        }
        v.b_count = count;
        v.a_alive = v.a_alive & !struck;
    }
    {
        int count = v.x_count;
        for (int i0 = 0; i0 < count; i0 += kBoomGang) {
            const Q::Trip t = q.trip<kBooms, kBoomGang>(v.x_next, i0);
            const bool mine = (q.g < kBoomGang) & (t.i < count);
            float frame = mine ? R.lds->boom[t.slot] : -1.0f;
            const bool live = mine & (frame != -1.0f);
            const bool gone = live & (frame >= 4.0f);
            const bool act = live & (t.i < count - Q::before(q.ranked<kBoomGang>(t, gone), t.rank));
            frame = gone ? -1.0f : (frame >= 0.0f ? frame + explosion_rate * dt : frame);
            if (act) {
                R.lds->boom[t.slot] = frame;
                R.dirty_x |= 1u << (t.slot / kBoomGang);
            }
            count -= __popc(q.ballot(gone & act));
        }
        v.x_count = count;
    }
    if (v.phase >= 6) alive = false;
    return alive;
}

PG_D void advance(const State& s, Rings& R, Q q, int env, int action, float& reward_out, bool& terminated_out) {
    Rng rng = Rng::open(s.mt + size_t(env) * kMtWords, q, s.mt_other + size_t(env) * kMtN, s.mt_sel + env);
    const int flags = SI(s, I_FLAGS, env);
    Live v;
    v.ax = SF(s, F_AX, env);
    v.ay = SF(s, F_AY, env);
    v.avx = SF(s, F_AVX, env);
    v.avy = SF(s, F_AVY, env);
    v.atimer = SF(s, F_ATIMER, env);
    v.bx = SF(s, F_BX, env);
    v.by = SF(s, F_BY, env);
    v.bvx = SF(s, F_BVX, env);
    v.bvy = SF(s, F_BVY, env);
    v.phase_t = SF(s, F_PHASE_T, env);
    v.attack_t = SF(s, F_ATTACK_T, env);
    v.explo_t = SF(s, F_EXPLO_T, env);
    v.damage_t = SF(s, F_DAMAGE_T, env);
    v.move_t = SF(s, F_MOVE_T, env);
    v.a_next = SI(s, I_A_NEXT, env);
    v.a_count = SI(s, I_A_COUNT, env);
    v.phase = SI(s, I_PHASE, env);
    v.weapon = SI(s, I_WEAPON, env);
    v.hp = SI(s, I_HP, env);
    v.b_next = SI(s, I_B_NEXT, env);
    v.b_count = SI(s, I_B_COUNT, env);
    v.x_next = SI(s, I_X_NEXT, env);
    v.x_count = SI(s, I_X_COUNT, env);
    v.n_rocks = SI(s, I_NROCKS, env);
    v.a_alive = (flags & kFlagAlive) != 0;
    const float cam_w = SF(s, F_CAMW, env), cam_h = SF(s, F_CAMH, env);  // take_view by a gang: all read, lane 0 puts 64 back below
    v.scr = screen_box(View{cam_w, cam_h, 1.0f * cam_w / 64.0f});
#pragma unroll
    for (int k = 0; k < kRocks; k++) {
        v.rock_x[k] = RK(s, 0, k, env);
        v.rock_y[k] = RK(s, 1, k, env);
    }

    // stage the slots that hold something (list positions below the counts), each by its owner
    R.dirty_a = R.dirty_b = R.dirty_x = 0u;
#pragma unroll
    for (int j = 0; j < kBossShots / kGang; j++) {
        const int k = q.g + kGang * j;
        if (((v.b_next - 1 - k) & (kBossShots - 1)) < v.b_count) {
#pragma unroll
            for (int f = S_X; f <= S_FRAME; f++) R.lds->b[f][k] = BS(s, f, k, env);
        }
    }
#if PG_BOSSFIGHT_AGENT_RING_LDS
#pragma unroll
    for (int j = 0; j < (kAgentShots + kGang - 1) / kGang; j++) {
        const int k = q.g + kGang * j;
        if (k < kAgentShots && ((v.a_next - 1 - k) & (kAgentShots - 1)) < v.a_count) {
#pragma unroll
            for (int f = S_X; f <= S_BOUNCE_T; f++) R.lds->a[f][k] = AS(s, f, k, env);
            R.lds->a_bouncing[k] = AB(s, k, env);
        }
    }
#endif
#pragma unroll
    for (int j = 0; j < kBooms / kBoomGang; j++) {
        const int k = q.g + kBoomGang * j;
        if (q.g < kBoomGang && ((v.x_next - 1 - k) & (kBooms - 1)) < v.x_count) R.lds->boom[k] = BM(s, 2, k, env);
    }

    const float dt = 1.0f / 4;
    float reward = 0.0f;
    bool terminated = false;
    for (int ss = 0; ss < 4; ss++) {
        const bool agent_alive = agent_update(s, R, env, v, rng, q, dt, action);
        const bool boss_alive = boss_update(s, R, env, v, rng, q, dt);
        reward = (!agent_alive) * -10.0f + (!boss_alive) * 10.0f;
        terminated = !agent_alive || !boss_alive;
        if (terminated) break;
    }
    rng.close();
    // the slots this lane wrote, back to memory
#pragma unroll
    for (int j = 0; j < kBossShots / kGang; j++)
        if ((R.dirty_b >> j) & 1u) {
            const int k = q.g + kGang * j;
#pragma unroll
            for (int f = S_X; f <= S_FRAME; f++) BS(s, f, k, env) = R.lds->b[f][k];
        }
#if PG_BOSSFIGHT_AGENT_RING_LDS
#pragma unroll
    for (int j = 0; j < (kAgentShots + kGang - 1) / kGang; j++)
        if ((R.dirty_a >> j) & 1u) {
            const int k = q.g + kGang * j;
#pragma unroll
            for (int f = S_X; f <= S_BOUNCE_T; f++) AS(s, f, k, env) = R.lds->a[f][k];
            AB(s, k, env) = R.lds->a_bouncing[k];
        }
#endif
#pragma unroll
    for (int j = 0; j < kBooms / kBoomGang; j++)
        if ((R.dirty_x >> j) & 1u) BM(s, 2, q.g + kBoomGang * j, env) = R.lds->boom[q.g + kBoomGang * j];
    if (q.g == 0) {
        if (cam_w != kCamSize || cam_h != kCamSize) {
            SF(s, F_CAMW, env) = kCamSize;
            SF(s, F_CAMH, env) = kCamSize;
        }
        SF(s, F_AX, env) = v.ax;
        SF(s, F_AY, env) = v.ay;
        SF(s, F_AVX, env) = v.avx;
        SF(s, F_AVY, env) = v.avy;
        SF(s, F_ATIMER, env) = v.atimer;
        SF(s, F_BX, env) = v.bx;
        SF(s, F_BY, env) = v.by;
        SF(s, F_BVX, env) = v.bvx;
        SF(s, F_BVY, env) = v.bvy;
        SF(s, F_PHASE_T, env) = v.phase_t;
        SF(s, F_ATTACK_T, env) = v.attack_t;
        SF(s, F_EXPLO_T, env) = v.explo_t;
        SF(s, F_DAMAGE_T, env) = v.damage_t;
        SF(s, F_MOVE_T, env) = v.move_t;
        SI(s, I_A_NEXT, env) = v.a_next;
        SI(s, I_A_COUNT, env) = v.a_count;
        SI(s, I_PHASE, env) = v.phase;
        SI(s, I_WEAPON, env) = v.weapon;
        SI(s, I_HP, env) = v.hp;
        SI(s, I_B_NEXT, env) = v.b_next;
        SI(s, I_B_COUNT, env) = v.b_count;
        SI(s, I_X_NEXT, env) = v.x_next;
        SI(s, I_X_COUNT, env) = v.x_count;
        SI(s, I_FLAGS, env) = (v.a_alive ? kFlagAlive : 0) | kFlagListed;  // sprite list built by the first update
    }
    reward_out = reward;
    terminated_out = terminated;
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// What cenv_make leaves in an env besides the seeded RNG: the three bullet pools, all "dead".
PG_D void fresh_env(const State& s, int env) {
    SF(s, F_CAMW, env) = kCamSize;  // Renderer gr: camera_size{64, 64}, camera_scale = 1 (renderer.h:18-20)
    SF(s, F_CAMH, env) = kCamSize;
    for (int k = 0; k < kAgentShots; k++) {  // std::vector<Bullet>(32): frame = -1 ("dead"), rest zero
        for (int f = 0; f < S_COUNT; f++) AS(s, f, k, env) = (f == S_FRAME) ? -1.0f : 0.0f;
        AB(s, k, env) = 0;
    }
    for (int k = 0; k < kBossShots; k++)
        for (int f = 0; f < S_COUNT; f++) BS(s, f, k, env) = (f == S_FRAME) ? -1.0f : 0.0f;
    for (int k = 0; k < kBooms; k++) {
        BM(s, 0, k, env) = 0.0f;
        BM(s, 1, k, env) = 0.0f;
        BM(s, 2, k, env) = -1.0f;
    }
}

// One new level for env.  restart: the env's RNG stream starts over from chain_seed.  Level-seed mode
// (pg_engine.h LevelPlan) instead rebuilds the env as a fresh cenv_make(seed = level number) would.
PG_D void begin_level(const State& s, const LevelPlan& plan, int env, bool restart, uint32_t chain_seed) {
    uint32_t* mt = s.mt + size_t(env) * kMtWords;
    if (s.mt_sel[env] & 2) {  // the stream's current words are where its gang left them (State::mt_sel): home first
        const uint32_t* other = s.mt_other + size_t(env) * kMtN;
        for (int k = 0; k < kMtN; k++) mt[k] = other[k];
    }
    s.mt_sel[env] = 0;  // (… and whatever was made ahead is dropped: this lane draws, or reseeds)
    if (restart) {
        plan.chain_seed[env] = chain_seed;
        plan.drawn[env] = 0;
    }
    if (plan.num_levels > 0) {
        const uint32_t k = plan.drawn[env];
        plan.drawn[env] = k + 1;
        mt_seed(mt, level_number(plan.num_levels, plan.start_level, plan.chain_seed[env], k));
        fresh_env(s, env);
    } else if (restart) {
        mt_seed(mt, chain_seed);
    }
    LaneRng rng{mt};
    new_level(s, env, rng, true);
}

__global__ void __launch_bounds__(64) make_kernel(State s, uint32_t seed_base, int env_offset, LevelPlan plan) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= s.n) return;
    fresh_env(s, env);
    begin_level(s, plan, env, true, seed_base + static_cast<uint32_t>(env_offset + env));  // level 0, never observed (D1)
}

__global__ void __launch_bounds__(64) reset_kernel(State s, const uint8_t* mask, const int32_t* seeds, StepIO io,
                                                   LevelPlan plan) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= s.n) return;
    if (mask && !mask[env]) return;
    begin_level(s, plan, env, seeds != nullptr, seeds ? static_cast<uint32_t>(seeds[env]) : 0u);
    io.reward[env] = 0.0f;
    io.done[env] = 0;
    io.pending[env] = 0;
}

__global__ void __launch_bounds__(64, PG_BOSSFIGHT_WAVES) logic_kernel(State s, const int32_t* actions, uint32_t run_seed,
                                                      uint32_t step_index, int env_offset, StepIO io, LevelPlan plan) {
    const int env = (blockIdx.x * 64 + threadIdx.x) / kGang;
    if (env >= s.n) return;
    const Q q = Q::at(threadIdx.x);
    if (io.pending[env]) {
        if (plan.num_levels > 0) {  // level-seed mode rebuilds the env: one lane (begin_level)
            if (q.g == 0) begin_level(s, plan, env, false, 0u);
        } else {
            Rng rng = Rng::open(s.mt + size_t(env) * kMtWords, q, s.mt_other + size_t(env) * kMtN, s.mt_sel + env);
            new_level(s, env, rng, q.g == 0);
            rng.close();
        }
        if (q.g == 0) {
            io.reward[env] = 0.0f;
            io.done[env] = 0;
            io.pending[env] = 0;
        }
        return;
    }
    const int action =
        actions ? actions[env] : synthetic_action(run_seed, step_index, static_cast<uint32_t>(env_offset + env));
    __shared__ RingsLds rings[64 / kGang];
    Rings R{&rings[(threadIdx.x & 63) / kGang], 0u, 0u, 0u};
    float reward;
    bool terminated;
    advance(s, R, q, env, action, reward, terminated);
    if (q.g == 0) {
        io.reward[env] = reward;
        io.done[env] = terminated ? 1 : 0;
        io.pending[env] = terminated ? 1 : 0;
    }
}

// render_game(true) (bossfight.cpp:401-424): one workgroup of two wavefronts per env (pg_render.h).  The complete frame,
// set-up included: every frame before the pre-pass existed; still the draw-list replay (flags bit 0), kDebugNoPrepass and
// the timing experiments.
__global__ void __launch_bounds__(128, 4) render_full_kernel(State s, AtlasView atlas, const uint8_t* mask, StepIO io,
                                                         int flags) {
    const int env = blockIdx.x;
    if (mask && !mask[env]) return;
    const int lane = threadIdx.x & 63, half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // two wavefronts per env (pg_render.h)
    constexpr int halves = 2;
    __shared__ alignas(16) uint32_t fb[kFbWords];

    const Camera cam{0.0f, 0.0f, kCamSize, kCamSize, kCamScale};  // camera_position stays {0,0} (renderer.h:18)
    const DescRegs descs = DescRegs::load(atlas, lane);
    const int skins = SI(s, I_SKINS, env), sflags = SI(s, I_FLAGS, env);
    const int a_ship = skins & 15, a_laser = (skins >> 4) & 15, b_ship = (skins >> 8) & 15, b_laser = (skins >> 12) & 15;
    const int backdrop = (skins >> 16) & 255;
    const int a_next = SI(s, I_A_NEXT, env), a_count = SI(s, I_A_COUNT, env);
    const int b_next = SI(s, I_B_NEXT, env), b_count = SI(s, I_B_COUNT, env);
    const int x_next = SI(s, I_X_NEXT, env), x_count = SI(s, I_X_COUNT, env);
    const int n_rocks = SI(s, I_NROCKS, env), phase = SI(s, I_PHASE, env);
    const float bx = SF(s, F_BX, env), by = SF(s, F_BY, env), ax = SF(s, F_AX, env), ay = SF(s, F_AY, env);
    Blit mine;

    int4 bg_d;  // the background draw, bossfight.cpp:416-419: texture, world position, scale — each wave resolves the axis it needs (pg_render.h BgAxis)
    float bg_px, bg_py, bg_sc;
    {
        const int4 d = descs.uniform(kTexSpace + backdrop);
        bg_d = d;
        bg_px = -kCamSize / kCamScale * 0.5f;
        bg_py = -kCamSize / kCamScale * 0.5f;
        bg_sc = 1.0f / d.z * kCamSize / kCamScale;
    }
    bool composed = false;
    if (PG_ABL(flags, 0x10000)) {  // (timing experiment, -DPG_ABLATE builds only: no background)
        wave_clear(fb, lane, half, halves);
        composed = true;
    } else if (!(flags & 1)) {  // no tile layer in this game: the background over black (pg_render.h)
        compose_background(fb, atlas, bg_axis(cam, bg_d, bg_px, bg_py, bg_sc, half), lane, half, halves);
        composed = true;
    }
    if (!composed) {
        wave_clear(fb, lane, half, halves);
        const bool has_bg = resolve_draw(cam, bg_d.y, bg_d.z, bg_d.x, bg_px, bg_py, bg_sc, 1.0f, false, false, mine);
        wave_replay(fb, atlas, mine, has_bg ? 1ull : 0ull, lane, half, halves);
    }

    // negative-z sprites: none.  System_Mob_AI::render (common_systems.cpp:392-450): boss bullets, rotated
    {
        bool has = false;
        int want_tex = kTexLaser + b_laser;
        float px = 0.0f, py = 0.0f;
        int rot_sn = 0, rot_cs = 0;  // (boss_fire worked them out)
        if (lane < b_count) {
            const int k = (kBossShots + b_next - 1 - lane) % kBossShots;
            const float frame = BS(s, S_FRAME, k, env);
            if (frame != -1.0f) {
                has = true;
                if (frame != 0.0f) want_tex = kTexBoom + static_cast<int>(frame - 1.0f);
                px = BS(s, S_X, k, env);
                py = BS(s, S_Y, k, env);
                rot_sn = __float_as_int(BS(s, S_SN, k, env));
                rot_cs = __float_as_int(BS(s, S_CS, k, env));
            }
        }
        const int4 d = descs.at(want_tex);
        if (has) {
            const float size = 0.1f;
            has = resolve_rotated_at(cam, d.y, d.z, d.x, px * kUnitPx - size * d.y * 0.5f, py * kUnitPx - size * d.z * 0.5f,
                                     rot_sn, rot_cs, size, 1.0f, mine);
        }
        if (PG_ABL(flags, 0x20000)) has = false;  // (timing experiment: no boss bullets)
        wave_replay_rows<4, true>(fb, atlas, mine, __ballot(has), lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
    }
    // second list, one draw per lane: boss ship, shield, explosions, barriers (positive-z sprites), agent bullets, agent
    {
        const int first_boom = 2, first_rock = first_boom + x_count;
        const int n_listed = (sflags & kFlagListed) ? n_rocks : 0;  // empty draw list right after a reset (D2)
        const int first_shot = first_rock + n_listed, agent_lane = first_shot + a_count;
        bool has = false;
        int want_tex = 0;
        float px = 0.0f, py = 0.0f, size = 0.0f, alpha = 1.0f;
        bool sprite = false;
        if (lane == 0) {
            has = true;
            want_tex = kTexBoss + b_ship;
            px = bx;
            py = by;
            size = 0.25f;
        } else if (lane == 1) {
            has = (phase % 2 == 0);
            want_tex = kTexShield;
            px = bx;
            py = by;
            size = 0.25f;
            alpha = 0.7f;
        } else if (lane < first_rock) {
            const int k = (kBooms + x_next - 1 - (lane - first_boom)) % kBooms;
            const float frame = BM(s, 2, k, env);
            if (frame != -1.0f) {
                has = true;
                want_tex = kTexBoom + static_cast<int>(frame);
                px = BM(s, 0, k, env);
                py = BM(s, 1, k, env);
                size = 0.3f;
            }
        } else if (lane < first_shot) {
            const int r = n_rocks - 1 - (lane - first_rock);  // sprite set order: newest barrier first
            has = true;
            sprite = true;
            want_tex = kTexRock + static_cast<int>(RK(s, 2, r, env));
            px = RK(s, 0, r, env);
            py = RK(s, 1, r, env);
        } else if (lane < agent_lane) {
            const int k = (kAgentShots + a_next - 1 - (lane - first_shot)) % kAgentShots;
            const float frame = AS(s, S_FRAME, k, env);
            if (frame != -1.0f) {
                has = true;
                want_tex = (frame == 0.0f) ? kTexLaser + a_laser : kTexBoom + static_cast<int>(frame - 1.0f);
                px = AS(s, S_X, k, env);
                py = AS(s, S_Y, k, env);
                size = 0.05f;
            }
        } else if (lane == agent_lane) {
            has = true;
            want_tex = kTexPlayer + a_ship;
            px = ax;
            py = ay;
            size = 0.05f;
        }
        const int4 d = descs.at(want_tex);
        if (has) {  // the two kinds of draw differ in their parameters only: pick per lane, resolve once
            float wx, wy, sc, al;
            if (sprite) {  // common_systems.cpp:22-48: offset (-0.15,-0.15), scale 0.3 (bossfight.cpp:479)
                const float scale = 1.0f * 0.3f;
                wx = (px + -0.15f) * kUnitPx;
                wy = (py + -0.15f) * kUnitPx;
                sc = scale * kUnitPx / d.y;
                al = 1.0f;
            } else {
                wx = px * kUnitPx - size * d.y * 0.5f;
                wy = py * kUnitPx - size * d.z * 0.5f;
                sc = size;
                al = alpha;
            }
            has = resolve_draw(cam, d.y, d.z, d.x, wx, wy, sc, al, false, false, mine);
        }
        if (PG_ABL(flags, 0x40000)) has = has && lane >= 2;    // (timing experiments: no boss ship and shield …
        if (PG_ABL(flags, 0x80000)) has = has && lane < 2;     //  … nothing but them)
        wave_replay_rows<4, true>(fb, atlas, mine, __ballot(has), lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
    }
    // each wave stores the rows it owns (pg_render.h wave_replay_rows): no barrier
    wave_store_rows(fb, io.obs + size_t(env) * kObsBytes, lane, half * (kObsH / halves), (half + 1) * (kObsH / halves));
}

// ---- the render pre-pass ---------------------------------------------------------------------------------------------
// Everything the frame needs that does not depend on the pixel: which draws there are, each one's rectangle on the target
// and in its texture, a rotated bullet's bounding box.  Both wavefronts of an env's render workgroup used to work all of
// it out, one draw per lane, most lanes idle (a frame has some dozens of draws for 128 lanes); here a wavefront takes the
// lists of two envs, dealt densely over its lanes, once.  The background's 64 column and 64 row offsets depend on the
// backdrop only (the camera never moves): a table per backdrop, made once.

// Every stream whose current words are in the second buffer back into mt[env] (BossfightGame::prepare_save); a wavefront
// per 64 envs.  The block made ahead, if any, is dropped: setup_kernel makes it again.
__global__ void __launch_bounds__(64) streams_home_kernel(State s) {
    const int lane = threadIdx.x, env0 = blockIdx.x * 64, e = env0 + lane;
    unsigned long long todo = __ballot(e < s.n && (s.mt_sel[e] & 2));
    while (todo) {  // (wave-uniform)
        const int env = env0 + __builtin_ctzll(todo);
        todo &= todo - 1;
        uint32_t* home = s.mt + size_t(env) * kMtWords;
        const uint32_t* other = s.mt_other + size_t(env) * kMtN;
        for (int k = lane; k < kMtN; k += 64) home[k] = other[k];
        if (lane == 0) s.mt_sel[env] = 0;
    }
}

// bg_axis + bg_offset of compose_background for each of the 13 backdrops (bossfight.cpp:416-419).
// … and, from those, the backdrop's 64 × 64 pixels as the frame target holds them (pg_render.h compose_background_from, the
// words it leaves in LDS): thirteen pictures of 16 KB for every env of the batch, which a frame then starts from as a
// straight copy (render_kernel) instead of 32 gathers and their offsets — the camera of this game never moves.
__global__ void __launch_bounds__(128) backdrop_kernel(State s, AtlasView atlas) {
    const int backdrop = blockIdx.x, lane = threadIdx.x & 63, axis = threadIdx.x >> 6;
    const Camera cam{0.0f, 0.0f, kCamSize, kCamSize, kCamScale};
    const int4 d = atlas.desc[kTexSpace + backdrop];
    const float pos = -kCamSize / kCamScale * 0.5f, scale = 1.0f / d.z * kCamSize / kCamScale;
    const uint32_t mine = bg_offset(bg_axis(cam, d, pos, pos, scale, axis), lane, axis);
    s.prep.backdrops[backdrop * 128 + axis * 64 + lane] = mine;
    __shared__ uint32_t offsets[128];
    __shared__ alignas(16) uint32_t fb[kFbWords];
    offsets[axis * 64 + lane] = mine;
    __syncthreads();
    const int half = axis;  // (wave h composes rows [32h, 32h + 32) of the picture)
    compose_background_from<kObsH / 2, false>(fb, atlas, offsets[lane], offsets[64 + lane], lane, half);
    __syncthreads();
    for (int k = threadIdx.x; k < kFbWords; k += 128) s.prep.backdrop_px[size_t(backdrop) * kFbWords + k] = fb[k];
}

constexpr int kPrepEnvs = 8, kPrepThreads = 256;
enum { PE_SKINS, PE_FLAGS, PE_A_NEXT, PE_A_COUNT, PE_B_NEXT, PE_B_COUNT, PE_X_NEXT, PE_X_COUNT, PE_NROCKS, PE_PHASE,
       PE_BX, PE_BY, PE_AX, PE_AY, PE_COUNT };
struct SetupLds {
    int4 desc[kTexCount];
    uint32_t env[kPrepEnvs][PE_COUNT];       // the scalars of the envs of this workgroup (float bits where floats)
    PrepDrawQueue queue[kPrepThreads / 64];  // one worklist per wavefront (pg_prepass.h)
    uint4 stamp[kTexCount * kStampsPerTex];  // pg_stamps.h: the sizes each texture is drawn at, pre-scaled
};
// Workgroups from n_groups on (a step's launch only): the random streams' next blocks for the envs that have none
// (State::mt_other, mt_sel), a wavefront per 64 envs.  The logic kernel is done and nothing else draws: the streams stand still.
__global__ void __launch_bounds__(kPrepThreads) setup_kernel(State s, AtlasView atlas, const uint8_t* mask, int n_groups) {
    __shared__ SetupLds S;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (static_cast<int>(blockIdx.x) >= n_groups) {  // (workgroup-uniform)
        const int first = ((static_cast<int>(blockIdx.x) - n_groups) * (kPrepThreads / 64) + wave) * 64, e = first + lane;
        const int sel = e < s.n ? s.mt_sel[e] : 1;
        unsigned long long todo = __ballot(e < s.n && !(sel & 1));
        while (todo) {  // (wave-uniform)
            const int k = __builtin_ctzll(todo), env = first + k;
            todo &= todo - 1;
            const int sel_env = __shfl(sel, k);
            uint32_t* home = s.mt + size_t(env) * kMtWords;
            uint32_t* other = s.mt_other + size_t(env) * kMtN;
            const bool in_other = (sel_env & 2) != 0;
            mt_next_block_wave(in_other ? other : home, in_other ? home : other, lane);
            if (lane == 0) s.mt_sel[env] = static_cast<uint8_t>(sel_env | 1);
        }
        return;
    }
    const int env0 = prep_block(blockIdx.x, n_groups) * kPrepEnvs;  // (pg_prepass.h: the groups of one XCD are consecutive)
    if (tid < kTexCount) S.desc[tid] = atlas.desc[tid];
    if (tid < kTexCount * kStampsPerTex)
        S.stamp[tid] = s.stamps ? reinterpret_cast<const uint4*>(atlas.texels + s.stamps)[tid] : make_uint4(0u, 0u, 0u, 0u);
    static_assert(kTexCount * kStampsPerTex <= kPrepThreads, "one entry per thread");
    if (tid >= 64 && tid < 64 + kPrepEnvs * PE_COUNT) {
        const int q = tid - 64, f = q / kPrepEnvs, e = q - f * kPrepEnvs, env = env0 + e;  // (the envs of a field side by side)
        constexpr int kInts[10] = {I_SKINS, I_FLAGS, I_A_NEXT, I_A_COUNT, I_B_NEXT, I_B_COUNT, I_X_NEXT, I_X_COUNT, I_NROCKS, I_PHASE};
        constexpr int kFloats[4] = {F_BX, F_BY, F_AX, F_AY};
        uint32_t v = 0;
        if (env < s.n) v = f < 10 ? static_cast<uint32_t>(SI(s, kInts[f], env)) : __float_as_uint(SF(s, kFloats[f - 10], env));
        S.env[e][f] = v;
    }
    __syncthreads();
    static_assert(kPrepEnvs == 2 * (kPrepThreads / 64), "two envs per wavefront");
    const Camera cam{0.0f, 0.0f, kCamSize, kCamSize, kCamScale};  // camera_position stays {0,0} (renderer.h:18)
    const int ea = 2 * wave, eb = 2 * wave + 1;
    const bool on_a = env0 + ea < s.n && !(mask && !mask[env0 + ea]), on_b = env0 + eb < s.n && !(mask && !mask[env0 + eb]);
    int done_bullets[2];
    // System_Mob_AI::render (common_systems.cpp:392-450): the boss's bullets, newest first, rotated
    {
        const int cnt_a = on_a ? static_cast<int>(S.env[ea][PE_B_COUNT]) : 0, cnt_b = on_b ? static_cast<int>(S.env[eb][PE_B_COUNT]) : 0;
        int done[2] = {0, 0};
        for (int base = 0; base < cnt_a + cnt_b; base += 64) {  // wave-uniform
            const int q = base + lane;
            const bool is_b = q >= cnt_a, valid = q < cnt_a + cnt_b;
            const int e = is_b ? eb : ea, env = env0 + e, slot = is_b ? q - cnt_a : q;
            const uint32_t* pe = S.env[e];
            bool has = false;
            int want_tex = kTexLaser + static_cast<int>((pe[PE_SKINS] >> 12) & 15u);
            float px = 0.0f, py = 0.0f;
            int rot_sn = 0, rot_cs = 0;  // (boss_fire worked them out)
            if (valid) {
                const int k = (kBossShots + static_cast<int>(pe[PE_B_NEXT]) - 1 - slot) % kBossShots;
                const float frame = BS(s, S_FRAME, k, env);
                if (frame != -1.0f) {
                    has = true;
                    if (frame != 0.0f) want_tex = kTexBoom + static_cast<int>(frame - 1.0f);
                    px = BS(s, S_X, k, env);
                    py = BS(s, S_Y, k, env);
                    rot_sn = __float_as_int(BS(s, S_SN, k, env));
                    rot_cs = __float_as_int(BS(s, S_CS, k, env));
                }
            }
            Blit b;
            RotBox box{0, 0, 0, 0};
            if (has) {
                const int4 d = S.desc[want_tex];
                const float size = 0.1f;
                has = resolve_rotated_at(cam, d.y, d.z, d.x, px * kUnitPx - size * d.y * 0.5f, py * kUnitPx - size * d.z * 0.5f,
                                         rot_sn, rot_cs, size, 1.0f, b);
                if (has) {
                    // the bullet's stamp, and with it the part of the bullet that can show at all: a laser is a few opaque
                    // texels in a transparent rim, and the box is that of the core (pg_stamps.h rot_box_core) — 3 × 3 or
                    // 4 × 4 pixels instead of 6 × 6, four bullets to a render wavefront's slot (pg_render.h, tiny draws)
                    int core_w, core_h;
                    has = stamp_substitute(&S.stamp[want_tex * kStampsPerTex], d.y, d.z, b, core_w, core_h);
                    if (has && (b.flip_mod & kRotated)) {  // no pixel of it on the target: not listed
                        box = rot_box_core(b, core_w, core_h);
                        has = box.bw > 0 && box.bh > 0;
                    }
                }
            }
            const unsigned long long m_a = __ballot(has && !is_b), m_b = __ballot(has && is_b);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (has) {
                const int rank = is_b ? done[1] + __popcll(m_b & below) : done[0] + __popcll(m_a & below);
                const BlitWords w = blit_pack(b);
                uint2* at = reinterpret_cast<uint2*>(s.prep.bullets + (size_t(env) * kBossShots + rank) * kBulletWords);
                at[0] = make_uint2(w.w[0], w.w[1]);
                at[1] = make_uint2(w.w[2], w.w[3]);
                at[2] = make_uint2(w.w[4], w.w[5]);
                at[3] = make_uint2(static_cast<uint32_t>(b.rot_sn), static_cast<uint32_t>(b.rot_cs));
                at[4] = make_uint2(static_cast<uint32_t>(box.x_lo) | static_cast<uint32_t>(box.y_lo) << 8 |
                                       static_cast<uint32_t>(box.bw) << 16 | static_cast<uint32_t>(box.bh) << 24, 0u);
            }
            done[0] += __popcll(m_a);
            done[1] += __popcll(m_b);
        }
        done_bullets[0] = done[0];
        done_bullets[1] = done[1];
    }
    // second list: boss ship, shield, explosions, barriers (positive-z sprites), agent bullets, agent — culled first,
    // the survivors finished densely (pg_prepass.h prep_draws_pass)
    {
        auto slots_of = [&](int e) {
            const uint32_t* pe = S.env[e];
            const int n_listed = (pe[PE_FLAGS] & kFlagListed) ? static_cast<int>(pe[PE_NROCKS]) : 0;  // empty draw list right after a reset (D2)
            return 2 + static_cast<int>(pe[PE_X_COUNT]) + n_listed + static_cast<int>(pe[PE_A_COUNT]) + 1;
        };
        const int cnt_a = on_a ? slots_of(ea) : 0, cnt_b = on_b ? slots_of(eb) : 0;
        uint32_t* const draws_a = s.prep.draws + size_t(env0 + ea) * kPrepDraws * kBlitWords;
        uint32_t* const draws_b = s.prep.draws + size_t(env0 + eb) * kPrepDraws * kBlitWords;
        PrepDrawPass st{0, {0, 0}, {0, 0}};
        PrepDrawQueue& Q = S.queue[wave];
        for (int base = 0; base < cnt_a + cnt_b; base += 64) {  // wave-uniform
            const int q = base + lane;
            const bool is_b = q >= cnt_a, valid = q < cnt_a + cnt_b;
            const int e = is_b ? eb : ea, env = env0 + e, slot = is_b ? q - cnt_a : q;
            const uint32_t* pe = S.env[e];
            const int skins = static_cast<int>(pe[PE_SKINS]);
            const int a_ship = skins & 15, a_laser = (skins >> 4) & 15, b_ship = (skins >> 8) & 15;
            const int x_count = static_cast<int>(pe[PE_X_COUNT]), n_rocks = static_cast<int>(pe[PE_NROCKS]);
            const int first_boom = 2, first_rock = first_boom + x_count;
            const int n_listed = (pe[PE_FLAGS] & kFlagListed) ? n_rocks : 0;
            const int first_shot = first_rock + n_listed, agent_slot = first_shot + static_cast<int>(pe[PE_A_COUNT]);
            const float bx = __uint_as_float(pe[PE_BX]), by = __uint_as_float(pe[PE_BY]);
            bool has = false, sprite = false;
            int want_tex = 0;
            float px = 0.0f, py = 0.0f, size = 0.0f, alpha = 1.0f;
            if (!valid) {
            } else if (slot == 0) {
                has = true;
                want_tex = kTexBoss + b_ship;
                px = bx;
                py = by;
                size = 0.25f;
            } else if (slot == 1) {
                has = (static_cast<int>(pe[PE_PHASE]) % 2 == 0);
                want_tex = kTexShield;
                px = bx;
                py = by;
                size = 0.25f;
                alpha = 0.7f;
            } else if (slot < first_rock) {
                const int k = (kBooms + static_cast<int>(pe[PE_X_NEXT]) - 1 - (slot - first_boom)) % kBooms;
                const float frame = BM(s, 2, k, env);
                if (frame != -1.0f) {
                    has = true;
                    want_tex = kTexBoom + static_cast<int>(frame);
                    px = BM(s, 0, k, env);
                    py = BM(s, 1, k, env);
                    size = 0.3f;
                }
            } else if (slot < first_shot) {
                const int r = n_rocks - 1 - (slot - first_rock);  // sprite set order: newest barrier first
                has = true;
                sprite = true;
                want_tex = kTexRock + static_cast<int>(RK(s, 2, r, env));
                px = RK(s, 0, r, env);
                py = RK(s, 1, r, env);
            } else if (slot < agent_slot) {
                const int k = (kAgentShots + static_cast<int>(pe[PE_A_NEXT]) - 1 - (slot - first_shot)) % kAgentShots;
                const float frame = AS(s, S_FRAME, k, env);
                if (frame != -1.0f) {
                    has = true;
                    want_tex = (frame == 0.0f) ? kTexLaser + a_laser : kTexBoom + static_cast<int>(frame - 1.0f);
                    px = AS(s, S_X, k, env);
                    py = AS(s, S_Y, k, env);
                    size = 0.05f;
                }
            } else {
                has = true;
                want_tex = kTexPlayer + a_ship;
                px = __uint_as_float(pe[PE_AX]);
                py = __uint_as_float(pe[PE_AY]);
                size = 0.05f;
            }
            PrepDraw p{has, false, false, want_tex, 0.0f, 0.0f, 1.0f, 1.0f};
            if (has) {  // the two kinds of draw differ in their parameters only
                const int4 d = S.desc[want_tex];
                if (sprite) {  // common_systems.cpp:22-48: offset (-0.15,-0.15), scale 0.3 (bossfight.cpp:479)
                    const float scale = 1.0f * 0.3f;
                    p.wx = (px + -0.15f) * kUnitPx;
                    p.wy = (py + -0.15f) * kUnitPx;
                    p.scale = scale * kUnitPx / d.y;
                } else {
                    p.wx = px * kUnitPx - size * d.y * 0.5f;
                    p.wy = py * kUnitPx - size * d.z * 0.5f;
                    p.scale = size;
                    p.alpha = alpha;
                }
            }
            prep_draws_pass(Q, st, S.desc, cam, cam, draws_a, draws_b, valid, is_b, p, lane, nullptr, S.stamp);
        }
        prep_draws_flush(Q, st, S.desc, cam, cam, draws_a, draws_b, lane, nullptr, S.stamp);
        if (lane < 2) {
            const int e = lane ? eb : ea;
            const bool on = lane ? on_b : on_a;
            const uint32_t word = static_cast<uint32_t>(lane ? done_bullets[1] : done_bullets[0]) |
                                  static_cast<uint32_t>(lane ? st.done[1] : st.done[0]) << 8 | ((S.env[e][PE_SKINS] >> 16) & 255u) << 16;
            if (on) s.prep.meta[env0 + e] = word;
        }
    }
}

// The frame from what setup_kernel left.  Nothing here is shared between the two wavefronts of an env: no barrier.
// A wavefront per workgroup: the env's upper and lower 32 rows are two workgroups with 8 KB of frame target each (render
// 0.660 -> 0.640 ms against one workgroup of two waves: a wave that is done — the lower rows rarely hold the boss — gives its
// slot and memory back at once.  Four parts of 16 rows: 0.84 ms, what a wave does before its first pixel is a third of its
// work; one wave for all 64 rows: 0.86 ms, 16 KB a wave leaves 2.5 waves per SIMD).
constexpr int kRenderParts = 2;
// Draws per memory round trip of the lean kernel's sprite passes (pg_render.h kGroup: small draws in flight together;
// kLone: trips of a big draw in flight together).  A render wavefront's life is a chain of such round trips.
#ifndef PG_BOSSFIGHT_GROUP
#define PG_BOSSFIGHT_GROUP 4
#endif
#ifndef PG_BOSSFIGHT_LONE
#define PG_BOSSFIGHT_LONE 4
#endif
#ifndef PG_BOSSFIGHT_QUARTERS
#define PG_BOSSFIGHT_QUARTERS false  // (tiny draws four to a slot, pg_render.h: measured, 2 % slower here — a half frame holds five or six bullets)
#endif
__global__ void __launch_bounds__(64, 5) render_kernel(State s, AtlasView atlas, const uint8_t* mask, StepIO io) {
    constexpr int halves = kRenderParts;
    const int env = blockIdx.x / halves;
    if (mask && !mask[env]) return;
    const int lane = threadIdx.x & 63, half = blockIdx.x % halves;
    __shared__ alignas(16) uint32_t fb[kFbWords / halves];  // this wave's rows only: frame row row_lo is its row 0
    PG_TL_BEGIN(6);
    PG_TL(0);
    const uint32_t meta = __builtin_amdgcn_readfirstlane(s.prep.meta[env]);
    const int n_bullets = meta & 0xffu, n_draws = (meta >> 8) & 0xffu, backdrop = meta >> 16;
    const int row_lo = half * (kObsH / halves), row_hi = (half + 1) * (kObsH / halves);
    const bool has_bullet = lane < n_bullets, has_draw = lane < n_draws;
    uint2 b3 = make_uint2(0u, 65536u), b4 = make_uint2(0u, 0u);
    const uint32_t* mine_at = s.prep.bullets + (size_t(env) * kBossShots + lane) * kBulletWords;
    Blit bullet = prep_draw_load(mine_at, has_bullet);
    if (has_bullet) {
        b3 = reinterpret_cast<const uint2*>(mine_at)[3];
        b4 = reinterpret_cast<const uint2*>(mine_at)[4];
    }
    bullet.rot_sn = static_cast<int32_t>(b3.x);
    bullet.rot_cs = static_cast<int32_t>(b3.y);
    // The draws in the coordinates of the wave's own target: everything moved up by row_lo.  A draw's pixels are found from
    // their offsets to its corner (pg_render.h wave_blit, rotated_pixel), so moving the corner moves them and nothing else.
    bullet.dy -= row_lo;
    const RotBox box{static_cast<int>(b4.x & 0xffu), static_cast<int>((b4.x >> 8) & 0xffu) - row_lo, static_cast<int>((b4.x >> 16) & 0xffu),
                     static_cast<int>(b4.x >> 24)};
    Blit draw = prep_draw_load(s.prep.draws + (size_t(env) * kPrepDraws + lane) * kBlitWords, has_draw);
    draw.dy -= row_lo;
    constexpr int kOwnRows = kObsH / halves;
    PG_TL(1);
    {   // the backdrop: a straight copy of this wave's 32 rows of its picture (8 KB, backdrop_kernel), memory to LDS without
        // passing through registers — `buffer_load_dwordx4 … lds`: lane l's 16 bytes land at M0 + 16·l, 1 KB an instruction
        using lds_ptr = __attribute__((address_space(3))) void*;
        const __amdgpu_buffer_rsrc_t px_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t*>(s.prep.backdrop_px + size_t(backdrop) * kFbWords + half * (kFbWords / halves)), 0,
            kFbWords / halves * 4, 0x00020000);
#pragma unroll
        for (int k = 0; k < kFbWords / halves / 4 / 64; k++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(px_rsrc, (lds_ptr)(fb + k * 256), 16, lane * 16, k * 1024, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler does not track LDS-direct loads: the copy has landed)
        wave_order();  // the copy, lane by lane, before the draws of other lanes on the same words
    }
    PG_TL(2);
    const unsigned long long mb = __ballot(has_bullet), md = __ballot(has_draw);
    wave_replay_rows<PG_BOSSFIGHT_GROUP, true, true, PG_BOSSFIGHT_LONE, true, PG_BOSSFIGHT_QUARTERS>(fb, atlas, bullet, mb, lane, 0, kOwnRows, &box);  // (…, stamps: setup_kernel substitutes them; tiny draws four to a slot)
    PG_TL(3);
    wave_replay_rows<PG_BOSSFIGHT_GROUP, true, true, PG_BOSSFIGHT_LONE, true, PG_BOSSFIGHT_QUARTERS>(fb, atlas, draw, md, lane, 0, kOwnRows);
    PG_TL(4);
    wave_store_rows(fb, io.obs + size_t(env) * kObsBytes, lane, row_lo, row_hi, 0);
    PG_TL_END(6, true, io.obs + size_t(env) * kObsBytes + half * (kObsBytes / 2));
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// cenv_render's frame (render_game(false)) for one env: pg_frame.h; the draw list of render_kernel, one draw at a time.
__global__ void __launch_bounds__(kFrameThreads) frame_kernel(State s, AtlasView atlas, int env, FrameTarget t) {
    const float fw = static_cast<float>(t.w), fh = static_cast<float>(t.h);
    const float sc = 1.0f * fw / 64.0f;
    FramePainter P{t, atlas, Camera{0.0f, 0.0f, fw, fh, sc}, static_cast<int>(threadIdx.x), kFrameThreads};
    if (threadIdx.x == 0) {  // render_game(false) leaves the window's size in gr (D15, see take_view)
        SF(s, F_CAMW, env) = fw;
        SF(s, F_CAMH, env) = fh;
    }
    const int skins = SI(s, I_SKINS, env), sflags = SI(s, I_FLAGS, env);
    const int a_ship = skins & 15, a_laser = (skins >> 4) & 15, b_ship = (skins >> 8) & 15, b_laser = (skins >> 12) & 15;
    const int backdrop = (skins >> 16) & 255;
    const int a_next = SI(s, I_A_NEXT, env), a_count = SI(s, I_A_COUNT, env);
    const int b_next = SI(s, I_B_NEXT, env), b_count = SI(s, I_B_COUNT, env);
    const int x_next = SI(s, I_X_NEXT, env), x_count = SI(s, I_X_COUNT, env);
    const int n_rocks = SI(s, I_NROCKS, env), phase = SI(s, I_PHASE, env);
    const float bx = SF(s, F_BX, env), by = SF(s, F_BY, env), ax = SF(s, F_AX, env), ay = SF(s, F_AY, env);
    P.clear();
    {
        const int tex = kTexSpace + backdrop;
        P.draw(tex, -fw / sc * 0.5f, -fh / sc * 0.5f, 1.0f / P.desc(tex).z * fh / sc);
    }
    for (int i = 0; i < b_count; i++) {
        const int k = (kBossShots + b_next - 1 - i) % kBossShots;
        const float frame = BS(s, S_FRAME, k, env);
        if (frame == -1.0f) continue;
        const int tex = (frame == 0.0f) ? kTexLaser + b_laser : kTexBoom + static_cast<int>(frame - 1.0f);
        const int4 d = P.desc(tex);
        const float size = 0.1f;
        P.draw_rotated(tex, BS(s, S_X, k, env) * kUnitPx - size * d.y * 0.5f, BS(s, S_Y, k, env) * kUnitPx - size * d.z * 0.5f,
                       static_cast<float>(BS(s, S_ROT, k, env) + kPi * 0.5f), size);
    }
    {
        const int tex = kTexBoss + b_ship;
        const int4 d = P.desc(tex);
        const float size = 0.25f;
        P.draw(tex, bx * kUnitPx - size * d.y * 0.5f, by * kUnitPx - size * d.z * 0.5f, size);
    }
    if (phase % 2 == 0) {
        const int4 d = P.desc(kTexShield);
        const float size = 0.25f;
        P.draw(kTexShield, bx * kUnitPx - size * d.y * 0.5f, by * kUnitPx - size * d.z * 0.5f, size, 0.7f);
    }
    for (int i = 0; i < x_count; i++) {
        const int k = (kBooms + x_next - 1 - i) % kBooms;
        const float frame = BM(s, 2, k, env);
        if (frame == -1.0f) continue;
        const int tex = kTexBoom + static_cast<int>(frame);
        const int4 d = P.desc(tex);
        const float size = 0.3f;
        P.draw(tex, BM(s, 0, k, env) * kUnitPx - size * d.y * 0.5f, BM(s, 1, k, env) * kUnitPx - size * d.z * 0.5f, size);
    }
    if (sflags & kFlagListed)
        for (int k = 0; k < n_rocks; k++) {
            const int r = n_rocks - 1 - k;  // sprite set order: newest barrier first
            const int tex = kTexRock + static_cast<int>(RK(s, 2, r, env));
            const float scale = 1.0f * 0.3f;
            P.draw(tex, (RK(s, 0, r, env) + -0.15f) * kUnitPx, (RK(s, 1, r, env) + -0.15f) * kUnitPx,
                   scale * kUnitPx / P.desc(tex).y);
        }
    for (int i = 0; i < a_count; i++) {
        const int k = (kAgentShots + a_next - 1 - i) % kAgentShots;
        const float frame = AS(s, S_FRAME, k, env);
        if (frame == -1.0f) continue;
        const int tex = (frame == 0.0f) ? kTexLaser + a_laser : kTexBoom + static_cast<int>(frame - 1.0f);
        const int4 d = P.desc(tex);
        const float size = 0.05f;
        P.draw(tex, AS(s, S_X, k, env) * kUnitPx - size * d.y * 0.5f, AS(s, S_Y, k, env) * kUnitPx - size * d.z * 0.5f, size);
    }
    {
        const int tex = kTexPlayer + a_ship;
        const int4 d = P.desc(tex);
        const float size = 0.05f;
        P.draw(tex, ax * kUnitPx - size * d.y * 0.5f, ay * kUnitPx - size * d.z * 0.5f, size);
    }
}

class BossfightGame final : public Game {
   public:
    const char* name() const override { return "bossfight"; }
    std::vector<std::string> texture_names() const override {
        std::vector<std::string> v;
        for (const char* n : {"deep_space_01", "spacegen_01", "milky_way_01", "ez_space_lite_01", "meyespace_v1_01",
                              "eye_nebula_01", "deep_sky_01", "space_nebula_01", "Background-1", "Background-2",
                              "Background-3", "Background-4", "parallax-space-backgound"})
            v.push_back(std::string("space_backgrounds/") + n + ".png");
        for (const char* n : {"spaceMeteors_001", "spaceMeteors_002", "spaceMeteors_003", "spaceMeteors_004",
                              "meteorGrey_big1", "meteorGrey_big2", "meteorGrey_big3", "meteorGrey_big4",
                              "enemyShipBlack1", "enemyShipBlue2", "enemyShipGreen3", "enemyShipRed4",
                              "playerShip1_blue", "playerShip1_green", "playerShip2_orange", "playerShip3_red",
                              "laserGreen14", "laserRed11", "laserBlue09", "explosion1", "explosion2", "explosion3",
                              "explosion4", "explosion5", "shield2"})
            v.push_back(std::string("misc_assets/") + n + ".png");
        return v;
    }
    std::string check_atlas(const std::vector<std::pair<int, int>>& sizes) const override {
        return static_cast<int>(sizes.size()) == kTexCount ? "" : "bossfight: unexpected texture count";
    }
    // The observation camera never moves or zooms (kCamSize, kCamScale), so every sprite lands at one size per texture
    // and scale: pre-scaled stamps (pg_stamps.h) for every draw of setup_kernel's two lists — the sizes are what the
    // kernels' own arithmetic gives (a size predicted wrongly would only never match).
    void extend_atlas(Atlas& atlas) override {
        std::vector<StampSpec> specs;
        auto plain = [&](int tex, float scale, float alpha) {
            const int4 d = atlas.desc_host(tex);
            specs.push_back({tex, stamp_len_plain(kCamSize, kCamScale, d.y, scale), stamp_len_plain(kCamSize, kCamScale, d.z, scale), stamp_mod(alpha)});
        };
        auto rotated = [&](int tex, float scale) {
            const int4 d = atlas.desc_host(tex);
            specs.push_back({tex, stamp_len_rotated(kCamScale, d.y, scale), stamp_len_rotated(kCamScale, d.z, scale), 255});
        };
        for (int k = 0; k < 3; k++) rotated(kTexLaser + k, 0.1f);   // the boss's bullets …
        for (int k = 0; k < 5; k++) rotated(kTexBoom + k, 0.1f);    // … and their explosion frames
        for (int k = 0; k < 4; k++) plain(kTexBoss + k, 0.25f, 1.0f);
        plain(kTexShield, 0.25f, 0.7f);
        for (int k = 0; k < 5; k++) plain(kTexBoom + k, 0.3f, 1.0f);
        for (int k = 0; k < 8; k++) plain(kTexRock + k, 1.0f * 0.3f * kUnitPx / atlas.desc_host(kTexRock + k).y, 1.0f);
        for (int k = 0; k < 3; k++) plain(kTexLaser + k, 0.05f, 1.0f);  // the agent's bullets, their explosion frames, the agent
        for (int k = 0; k < 5; k++) plain(kTexBoom + k, 0.05f, 1.0f);
        for (int k = 0; k < 4; k++) plain(kTexPlayer + k, 0.05f, 1.0f);
        stamps_at_ = append_stamps(atlas, kTexCount, specs);
    }
    static size_t align256(size_t x) { return (x + 255) & ~size_t(255); }
    struct Layout {
        size_t mt, f, i, ashot, abnc, bshot, boom, rock, total;
    };
    static Layout layout(int n) {
        Layout l{};
        size_t off = 0;
        auto take = [&](size_t bytes) {
            size_t at = off;
            off += align256(bytes);
            return at;
        };
        l.mt = take(size_t(n) * kMtWords * 4);
        l.f = take(size_t(F_COUNT) * n * 4);
        l.i = take(size_t(I_COUNT) * n * 4);
        l.ashot = take(size_t(S_COUNT) * kAgentShots * n * 4);
        l.abnc = take(size_t(kAgentShots) * n);
        l.bshot = take(size_t(S_COUNT) * kBossShots * n * 4);
        l.boom = take(size_t(3) * kBooms * n * 4);
        l.rock = take(size_t(3) * kRocks * n * 4);
        l.total = off;
        return l;
    }
    size_t state_bytes(int n) const override { return layout(n).total; }
    void bind(void* d_state, int n, AtlasView atlas) override {
        uint8_t* p = static_cast<uint8_t*>(d_state);
        const Layout l = layout(n);
        s_.n = n;
        s_.mt = reinterpret_cast<uint32_t*>(p + l.mt);
        s_.f = reinterpret_cast<float*>(p + l.f);
        s_.i = reinterpret_cast<int32_t*>(p + l.i);
        s_.ashot = reinterpret_cast<float*>(p + l.ashot);
        s_.abnc = p + l.abnc;
        s_.bshot = reinterpret_cast<float*>(p + l.bshot);
        s_.boom = reinterpret_cast<float*>(p + l.boom);
        s_.rock = reinterpret_cast<float*>(p + l.rock);
        s_.stamps = stamps_at_;
        atlas_ = atlas;
    }
    int blocks() const { return (s_.n + 63) / 64; }
    void launch_make(hipStream_t st, uint32_t seed_base, int env_offset) override {
        hipLaunchKernelGGL(make_kernel, dim3(blocks()), dim3(64), 0, st, s_, seed_base, env_offset, plan);
        hipLaunchKernelGGL(backdrop_kernel, dim3(kBackdrops), dim3(128), 0, st, s_, atlas_);  // (the engine binds the scratch memory first)
    }
    void launch_reset(hipStream_t st, const uint8_t* mask, const int32_t* seeds, StepIO io) override {
        hipLaunchKernelGGL(reset_kernel, dim3(blocks()), dim3(64), 0, st, s_, mask, seeds, io, plan);
    }
    void launch_logic(hipStream_t st, const int32_t* actions, uint32_t run_seed, uint32_t step_index, int env_offset,
                      StepIO io) override {
        hipLaunchKernelGGL(logic_kernel, dim3((s_.n * kGang + 63) / 64), dim3(64), 0, st, s_, actions, run_seed, step_index,
                           env_offset, io, plan);
    }
    bool launch_frame(hipStream_t st, int env, uint32_t* d_px, int w, int h) override {
        hipLaunchKernelGGL(frame_kernel, dim3(1), dim3(kFrameThreads), 0, st, s_, atlas_, env, FrameTarget{d_px, w, h});
        return true;
    }
    // (every flag that changes the frame's path — the draw-list replay, kDebugNoPrepass, the timing experiments — takes the complete kernel)
    bool lean() const { return (debug_flags & ~kDebugNoPrefetch) == 0; }
    void launch_prepass(hipStream_t st, const uint8_t* mask) override {
        if (lean()) {
            const int groups = (s_.n + kPrepEnvs - 1) / kPrepEnvs;  // … and a wavefront per 64 envs for the random streams' next blocks
            hipLaunchKernelGGL(setup_kernel, dim3(groups + (s_.n + kPrepThreads - 1) / kPrepThreads), dim3(kPrepThreads), 0, st, s_, atlas_, mask, groups);
        }
    }
    void launch_render(hipStream_t st, const uint8_t* mask, StepIO io) override {
        if (lean())
            hipLaunchKernelGGL(render_kernel, dim3(kRenderParts * s_.n), dim3(64), 0, st, s_, atlas_, mask, io);
        else
            hipLaunchKernelGGL(render_full_kernel, dim3(s_.n), dim3(128), 0, st, s_, atlas_, mask, io, debug_flags);
    }
    static size_t up256(size_t b) { return (b + 255) & ~size_t(255); }
    size_t scratch_bytes(int n) const override {
        return up256(size_t(kBackdrops) * 128 * 4) + up256(size_t(n) * 4) + up256(size_t(n) * kBossShots * kBulletWords * 4) +
               up256(size_t(n) * kPrepDraws * kBlitWords * 4) + up256(size_t(n) * kMtN * 4) + up256(size_t(n)) +
               up256(size_t(kBackdrops) * kFbWords * 4);
    }
    void state_loaded(hipStream_t st) override {
        hipMemsetAsync(s_.mt_sel, 0, size_t(s_.n), st);  // the streams that were just loaded are in mt[env]; what was made ahead is not theirs
    }
    // A snapshot takes the streams from mt[env]: the ones whose gang has moved on to the second buffer come home first.
    void prepare_save(hipStream_t st) override {
        hipLaunchKernelGGL(streams_home_kernel, dim3((s_.n + 63) / 64), dim3(64), 0, st, s_);
    }
    void bind_scratch(void* d_scratch, int n) override {
        uint8_t* p = static_cast<uint8_t*>(d_scratch);
        s_.prep.backdrops = reinterpret_cast<uint32_t*>(p);
        p += up256(size_t(kBackdrops) * 128 * 4);
        s_.prep.meta = reinterpret_cast<uint32_t*>(p);
        p += up256(size_t(n) * 4);
        s_.prep.bullets = reinterpret_cast<uint32_t*>(p);
        p += up256(size_t(n) * kBossShots * kBulletWords * 4);
        s_.prep.draws = reinterpret_cast<uint32_t*>(p);
        p += up256(size_t(n) * kPrepDraws * kBlitWords * 4);
        s_.mt_other = reinterpret_cast<uint32_t*>(p);
        p += up256(size_t(n) * kMtN * 4);
        s_.mt_sel = p;  // (the engine zeroes the scratch block: every stream is at home, nothing is made ahead)
        p += up256(size_t(n));
        s_.prep.backdrop_px = reinterpret_cast<uint32_t*>(p);
    }
    // Same layout as oracle/pgo_bossfight.cpp Bossfight::dump_state.
    int dump_state(hipStream_t st, int env, float* out, int cap) override {
        hipStreamSynchronize(st);
        const size_t n = s_.n;
        auto rf = [&](const float* base, size_t idx) {
            float v;
            hipMemcpy(&v, base + idx, 4, hipMemcpyDeviceToHost);
            return v;
        };
        auto f = [&](int field) { return rf(s_.f, size_t(field) * n + env); };
        auto iv = [&](int field) {
            int32_t v;
            hipMemcpy(&v, s_.i + size_t(field) * n + env, 4, hipMemcpyDeviceToHost);
            return static_cast<float>(v);
        };
        int32_t flags;
        hipMemcpy(&flags, s_.i + size_t(I_FLAGS) * n + env, 4, hipMemcpyDeviceToHost);
        std::vector<float> v = {f(F_AX), f(F_AY), f(F_AVX), f(F_AVY), (flags & kFlagAlive) ? 1.0f : 0.0f, f(F_ATIMER),
                                iv(I_A_NEXT), iv(I_A_COUNT), f(F_BX), f(F_BY), f(F_BVX), f(F_BVY), f(F_PHASE_T),
                                iv(I_PHASE), iv(I_WEAPON), f(F_ATTACK_T), iv(I_HP), iv(I_B_NEXT), iv(I_B_COUNT),
                                iv(I_X_NEXT), iv(I_X_COUNT), f(F_EXPLO_T), f(F_DAMAGE_T), f(F_MOVE_T), iv(I_NROCKS)};
        for (int k = 0; k < kAgentShots; k++)
            for (int fld : {S_X, S_Y, S_FRAME}) v.push_back(rf(s_.ashot, (size_t(env) * S_COUNT + fld) * kAgentShots + k));
        for (int k = 0; k < kBossShots; k++)
            for (int fld : {S_X, S_Y, S_FRAME}) v.push_back(rf(s_.bshot, (size_t(env) * S_COUNT + fld) * kBossShots + k));
        const int m = cap < static_cast<int>(v.size()) ? cap : static_cast<int>(v.size());
        for (int k = 0; k < m; k++) out[k] = v[k];
        return static_cast<int>(v.size());
    }
    int dump_tiles(hipStream_t, int, uint8_t*, int) override { return 0; }

   private:
    State s_{};
    AtlasView atlas_{};
    uint32_t stamps_at_ = 0;  // extend_atlas → bind
};

}  // namespace bossfight

}  // namespace PG_VARIANT_NS

std::unique_ptr<Game> PG_FACTORY(make_bossfight)() { return std::make_unique<PG_VARIANT_NS::bossfight::BossfightGame>(); }

}  // namespace pg
