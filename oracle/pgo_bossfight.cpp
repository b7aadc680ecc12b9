// ORACLE — TEST INFRASTRUCTURE ONLY (see pgo_common.h header).
//
// bossfight: CPU restatement of SURVEY.md rows G4s / G4r/g.
//   step   games/bossfight/bossfight.cpp:294-345, common_systems.cpp:199-390 (boss), :494-683 (agent),
//          :103-185 (fire_pattern), :187-197 (show_damage)
//   render games/bossfight/bossfight.cpp:401-424, common_systems.cpp:392-450, :685-721, :22-48
//   reset  games/bossfight/bossfight.cpp:426-504, common_systems.cpp:452-470, :724-737
// Config = the reference's compile-time default, hard_mode (common_systems.h:63-65).
// sinf/cosf are the process's libm (glibc), exactly what the reference's std::cos/std::sin(float) resolve to.
#include <algorithm>
#include <cmath>

#include "pgo_env.h"

namespace pgo {
namespace {

const char* const kSpace[13] = {"deep_space_01", "spacegen_01", "milky_way_01", "ez_space_lite_01", "meyespace_v1_01",
                                "eye_nebula_01", "deep_sky_01", "space_nebula_01", "Background-1", "Background-2",
                                "Background-3", "Background-4", "parallax-space-backgound"};  // bossfight.cpp:54-68
const char* const kRocks[8] = {"spaceMeteors_001", "spaceMeteors_002", "spaceMeteors_003", "spaceMeteors_004",
                               "meteorGrey_big1",  "meteorGrey_big2",  "meteorGrey_big3",  "meteorGrey_big4"};
const char* const kBossShips[4] = {"enemyShipBlack1", "enemyShipBlue2", "enemyShipGreen3", "enemyShipRed4"};
const char* const kPlayerShips[4] = {"playerShip1_blue", "playerShip1_green", "playerShip2_orange", "playerShip3_red"};
const char* const kLasers[3] = {"laserGreen14", "laserRed11", "laserBlue09"};

struct Shot {  // common_systems.h:46-51 / 108-116
    V2 pos, vel;
    float rotation = 0.0f;
    float frame = -1.0f;  // -1 = dead
    bool bouncing = false;
    float bounce_timer = 0.0f;
};
struct Boom {  // common_systems.h:53-56
    V2 pos;
    float frame = -1.0f;
};

class Bossfight final : public Env {
   public:
    int dump_state(float* out, int cap) const override {
        std::vector<float> v = {a_pos.x, a_pos.y, a_vel.x, a_vel.y, static_cast<float>(a_alive), a_timer,
                                static_cast<float>(a_next), static_cast<float>(a_count), b_pos.x, b_pos.y, b_vel.x,
                                b_vel.y, phase_timer, static_cast<float>(phase_index),
                                static_cast<float>(weapon_index), attack_timer, static_cast<float>(hp),
                                static_cast<float>(b_next), static_cast<float>(b_count), static_cast<float>(x_next),
                                static_cast<float>(x_count), explosion_timer, damage_timer, move_timer,
                                static_cast<float>(n_rocks)};
        for (int i = 0; i < 32; i++) {
            v.push_back(a_shots[i].pos.x);
            v.push_back(a_shots[i].pos.y);
            v.push_back(a_shots[i].frame);
        }
        for (int i = 0; i < 64; i++) {
            v.push_back(b_shots[i].pos.x);
            v.push_back(b_shots[i].pos.y);
            v.push_back(b_shots[i].frame);
        }
        int n = std::min<int>(cap, static_cast<int>(v.size()));
        std::memcpy(out, v.data(), n * sizeof(float));
        return static_cast<int>(v.size());
    }
    int dump_tiles(uint8_t*, int) const override { return 0; }  // no tile map in this game

   protected:
    void on_make() override {
        auto& bank = TextureBank::global();
        auto T = [&](const std::string& n) { return bank.find("assets/" + n + ".png"); };
        for (int i = 0; i < 13; i++) tex_space_[i] = T(std::string("space_backgrounds/") + kSpace[i]);
        for (int i = 0; i < 8; i++) tex_rock_[i] = T(std::string("misc_assets/") + kRocks[i]);
        for (int i = 0; i < 4; i++) tex_boss_[i] = T(std::string("misc_assets/") + kBossShips[i]);
        for (int i = 0; i < 4; i++) tex_player_[i] = T(std::string("misc_assets/") + kPlayerShips[i]);
        for (int i = 0; i < 3; i++) tex_laser_[i] = T(std::string("misc_assets/") + kLasers[i]);
        for (int i = 0; i < 5; i++) tex_boom_[i] = T("misc_assets/explosion" + std::to_string(i + 1));
        tex_shield_ = T("misc_assets/shield2");
    }

    void new_level() override {  // bossfight.cpp:426-504
        ids_.refill();
        in_hazard_.clear();
        in_sprite_.clear();
        const V2 cs = painter_.cam_size;  // D15: whatever the last render left (64 / 1.0 for observation-only use)
        const float sc = painter_.cam_scale;

        ids_.take();  // player
        a_pos = {rng_.frange(-1.0f, 1.0f) * cs.x / sc * kPxUnit * 0.5f, cs.y / sc * kPxUnit * 0.5f};
        a_vel = {0.0f, 0.0f};
        boss_id_ = ids_.take();
        in_hazard_.insert(boss_id_);
        b_pos = {0.0f, 0.0f};
        b_vel = {0.0f, 0.0f};
        phase_timer = 0.0f;
        phase_index = 0;
        weapon_index = 0;
        attack_timer = 0.0f;
        hp = 0;

        const int want = rng_.irange(1, 4);
        Box placed[4];
        n_rocks = 0;
        for (int i = 0; i < want; i++) {
            V2 p;
            p.x = rng_.frange(-1.0f, 1.0f) * cs.x / sc * kPxUnit * 0.5f * 0.9f;
            p.y = cs.y / sc * kPxUnit * 0.5f - rng_.frange(0.7f, 1.2f);
            Box wc{p.x + -0.1f, p.y + -0.1f, 0.2f, 0.2f};
            bool clash = false;
            for (int j = 0; j < i; j++)
                if (boxes_touch(wc, placed[j])) {
                    clash = true;
                    break;
                }
            if (!clash) {
                int id = ids_.take();
                in_hazard_.insert(id);
                in_sprite_.insert(id);
                rock_id_[n_rocks] = id;
                rock_pos_[n_rocks] = p;
                rock_tex_[n_rocks] = rng_.irange(0, 7);
                n_rocks++;
                placed[i] = wc;
            } else {
                placed[i] = Box{0.0f, 0.0f, 0.0f, 0.0f};
            }
        }
        backdrop_ = rng_.irange(0, 12);
        rng_.unit();  // current_background_offset_x / _y are drawn but never used
        rng_.unit();
        draw_list_.clear();

        // System_Agent::reset (common_systems.cpp:724-737)
        a_next = a_count = 0;
        a_timer = 0.0f;
        a_ship = rng_.irange(0, 3);
        a_laser = rng_.irange(0, 2);
        a_alive = true;
        // System_Mob_AI::reset (common_systems.cpp:452-470)
        b_next = x_next = b_count = x_count = 0;
        explosion_timer = damage_timer = move_timer = 0.0f;
        b_ship = rng_.irange(0, 3);
        b_laser = rng_.irange(0, 2);
    }

    Box screen() const {  // common_systems.cpp:224-226, 513-515
        const V2 cs = painter_.cam_size;
        const float sc = painter_.cam_scale;
        return Box{-cs.x / sc * kPxUnit * 0.5f, -cs.y / sc * kPxUnit * 0.5f, cs.x / sc * kPxUnit, cs.y / sc * kPxUnit};
    }
    Box hazard_box(int id) const {
        if (id == boss_id_) return Box{b_pos.x + -0.6f, b_pos.y + -0.4f, 1.2f, 0.8f};
        for (int k = 0; k < n_rocks; k++)
            if (rock_id_[k] == id) return Box{rock_pos_[k].x + -0.1f, rock_pos_[k].y + -0.1f, 0.2f, 0.2f};
        return Box{};
    }

    bool agent_update(float dt, int action) {  // common_systems.cpp:494-683
        const float mixrate = 0.5f, speed = 0.1f, bullet_time = 5.0f, bullet_speed = 0.1f;
        const float bounce_speed = 0.05f, bounce_time = 10.0f, explosion_rate = 0.3f;
        const Box scr = screen();
        float mx = (action == 6 || action == 7 || action == 8) - (action == 0 || action == 1 || action == 2);
        float my = (action == 2 || action == 5 || action == 8) - (action == 0 || action == 3 || action == 6);
        const bool fire = action == 9;
        a_vel.x += mixrate * (mx * speed - a_vel.x) * dt;
        a_vel.y += mixrate * (-my * speed - a_vel.y) * dt;
        a_pos.x += a_vel.x * dt;
        a_pos.y += a_vel.y * dt;
        Box wc{a_pos.x + -0.15f, a_pos.y + -0.1f, 0.3f, 0.2f};
        if (wc.x < scr.x) {
            a_pos.x += scr.x - wc.x;
            a_vel.x = 0.0f;
        } else if (wc.x + wc.w > scr.x + scr.w) {
            a_pos.x += scr.x + scr.w - (wc.x + wc.w);
            a_vel.x = 0.0f;
        }
        if (wc.y < scr.y) {
            a_pos.y += scr.y - wc.y;
            a_vel.y = 0.0f;
        } else if (wc.y + wc.h > scr.y + scr.h) {
            a_pos.y += scr.y + scr.h - (wc.y + wc.h);
            a_vel.y = 0.0f;
        }
        wc = Box{a_pos.x + -0.15f, a_pos.y + -0.1f, 0.3f, 0.2f};
        if (fire) {
            if (a_timer == 0.0f && a_count < 32) {
                a_timer = bullet_time;
                Shot& s = a_shots[a_next];
                s.rotation = 0.0f;
                s.vel = {0.0f, -bullet_speed};
                s.pos = a_pos;
                s.frame = 0.0f;
                s.bouncing = false;
                s.bounce_timer = 0.0f;
                a_next = (a_next + 1) % 32;
                a_count++;
            } else {
                a_timer = std::max(0.0f, a_timer - dt);
            }
        }
        for (int h : in_hazard_)
            if (boxes_touch(wc, hazard_box(h))) {
                a_alive = false;
                break;
            }
        for (int i = 0; i < a_count; i++) {  // NB: a_count shrinks inside the loop, as in the reference
            const int k = (32 + a_next - 1 - i) % 32;
            Shot& s = a_shots[k];
            if (s.frame == -1.0f) continue;
            if (s.frame == 0.0f) {
                Box sb{s.pos.x - 0.01f, s.pos.y - 0.01f, 0.02f, 0.02f};
                if (!boxes_touch(sb, scr)) {
                    s.vel = {0.0f, 0.0f};
                    s.frame = 5.0f;
                } else {
                    for (int h : in_hazard_) {
                        if (!boxes_touch(sb, hazard_box(h))) continue;
                        if (h == boss_id_) {
                            if (phase_index % 2 == 0) {
                                s.vel = {rng_.frange(-1.0f, 1.0f) * bounce_speed, bounce_speed};
                                s.bounce_timer = bounce_time;
                                s.bouncing = true;
                            } else {
                                s.vel = {0.0f, 0.0f};
                                s.frame = 1.0f;
                                if (hp > 0) hp--;
                            }
                        } else {
                            s.vel = {0.0f, 0.0f};
                            s.frame = 1.0f;
                        }
                        break;
                    }
                }
            }
            s.pos.x += s.vel.x * dt;
            s.pos.y += s.vel.y * dt;
            bool destroy = false;
            if (s.frame >= 5.0f)
                destroy = true;
            else if (s.frame >= 1.0f)
                s.frame += explosion_rate * dt;
            if (s.bouncing) {
                if (s.bounce_timer > 0.0f)
                    s.bounce_timer = std::max(0.0f, s.bounce_timer - dt);
                else
                    destroy = true;
            }
            if (destroy) {
                a_count--;
                s.frame = -1.0f;
            }
        }
        return a_alive;
    }

    void boss_fire(V2 pos, float rotation, float speed) {  // common_systems.cpp:75-88
        if (b_count < 64) {
            Shot& s = b_shots[b_next];
            s.rotation = rotation;
            s.vel = {std::cos(rotation) * speed, -std::sin(rotation) * speed};
            s.pos = pos;
            s.frame = 0.0f;
            b_next = (b_next + 1) % 64;
            b_count++;
        }
    }
    void fire_pattern(V2 pos, int pattern, float& timer, float dt) {  // common_systems.cpp:103-185
        const float bullet_speed = mode_ == kHard ? 0.1f : 0.05f;  // common_systems.cpp:104
        switch (pattern) {
            case -1:
                if (rng_.unit() < 0.1f * dt) boss_fire(pos, M_PI * (1.0f + rng_.unit()), bullet_speed);
                break;
            case 0:
                if (timer >= 8.0f) {
                    timer = 0.0f;
                    for (int i = 0; i < 5; i++) {
                        float rotation = M_PI * 1.5f + (i - 2) * M_PI * 0.125f;
                        boss_fire(pos, rotation, bullet_speed);
                    }
                } else
                    timer += dt;
                break;
            case 1:
                if (timer >= 5.0f) {
                    timer = 0.0f;
                    int k = timer / 5.0f;
                    k = std::abs(8 - (k % 16));
                    for (int i = 0; i < 4; i++) {
                        float rotation = M_PI * (1.25f + k * 0.0625f) + i * M_PI * 0.5f;
                        boss_fire(pos, rotation, bullet_speed);
                    }
                } else
                    timer += dt;
                break;
            case 2:
                if (timer >= 10.0f) {
                    timer = 0.0f;
                    float offset = rng_.unit() * 2.0f * M_PI;
                    for (int i = 0; i < 8; i++) {
                        float rotation = M_PI * 0.25f * i + offset;
                        boss_fire(pos, rotation, bullet_speed);
                    }
                } else
                    timer += dt;
                break;
            case 3:
                if (timer >= 4.0f) {
                    timer = 0.0f;
                    boss_fire(pos, M_PI * (1.0f + rng_.unit()), bullet_speed);
                } else
                    timer += dt;
                break;
        }
    }
    void explode(V2 pos) {  // common_systems.cpp:91-101
        if (x_count < 8) {
            booms[x_next].pos = pos;
            booms[x_next].frame = 0.0f;
            x_next = (x_next + 1) % 8;
            x_count++;
        }
    }

    bool boss_update(float dt) {  // common_systems.cpp:199-390
        const float shielded_time =
            180.0f + rng_.unit() * (mode_ == kHard ? 80.0f : 30.0f);  // drawn every sub-step (D14); common_systems.cpp:202
        const float unshielded_time = 300.0f, explosion_rate = 0.3f, move_time = 70.0f, damage_time = 80.0f;
        const int boss_hp = 3;
        bool alive = true;
        const Box agent_rect{a_pos.x + -0.15f, a_pos.y + -0.1f, 0.3f, 0.2f};
        const Box scr = screen();

        if (phase_timer == 0.0f) {
            weapon_index = rng_.irange(0, 3);
            attack_timer = 0.0f;
            hp = boss_hp;
        }
        if (phase_index % 2 == 0) {
            if (phase_timer >= shielded_time) {
                phase_timer = 0.0f;
                phase_index++;
            } else
                phase_timer += dt;
            fire_pattern(b_pos, weapon_index, attack_timer, dt);
        } else {
            if (phase_timer >= unshielded_time) {
                phase_timer = 0.0f;
                phase_index++;
            } else
                phase_timer += dt;
            fire_pattern(b_pos, -1, attack_timer, dt);
            if (hp == 0) {
                if (explosion_timer >= 8.0f) {  // show_damage, common_systems.cpp:187-197
                    explosion_timer = 0.0f;
                    float ox = rng_.frange(-0.5f, 0.5f) + b_pos.x;
                    float oy = rng_.frange(-0.5f, 0.5f) + b_pos.y;
                    explode(V2{ox, oy});
                } else
                    explosion_timer += dt;
                if (damage_timer >= damage_time) {
                    damage_timer = 0.0f;
                    phase_index++;
                    hp = boss_hp;
                } else
                    damage_timer += dt;
            }
        }
        if (move_timer >= move_time) {
            move_timer = 0.0f;
            float tx = (rng_.unit() * 2.0f - 1.0f) * 0.5f * scr.w * 0.7f;
            float ty = ((rng_.unit() * 2.0f - 1.0f) * 0.5f - 0.3f) * scr.h * 0.5f;
            b_vel.x = (tx - b_pos.x) / move_time;
            b_vel.y = (ty - b_pos.y) / move_time;
        } else
            move_timer += dt;
        b_pos.x += b_vel.x * dt;
        b_pos.y += b_vel.y * dt;

        for (int i = 0; i < b_count; i++) {
            const int k = (64 + b_next - 1 - i) % 64;
            Shot& s = b_shots[k];
            if (s.frame == -1.0f) continue;
            if (s.frame == 0.0f) {
                Box sb{s.pos.x - 0.01f, s.pos.y - 0.01f, 0.02f, 0.02f};
                if (!boxes_touch(sb, scr)) {
                    s.vel = {0.0f, 0.0f};
                    s.frame = 5.0f;
                } else {
                    if (boxes_touch(sb, agent_rect)) {
                        s.vel = {0.0f, 0.0f};
                        s.frame = 1.0f;
                        a_alive = false;
                        break;  // later bullets skip this sub-step (D14)
                    }
                    for (int h : in_hazard_) {
                        if (h == boss_id_) continue;
                        if (boxes_touch(sb, hazard_box(h))) {
                            s.vel = {0.0f, 0.0f};
                            s.frame = 1.0f;
                            break;
                        }
                    }
                }
            }
            s.pos.x += s.vel.x * dt;
            s.pos.y += s.vel.y * dt;
            if (s.frame >= 5.0f) {
                b_count--;
                s.frame = -1.0f;
            } else if (s.frame >= 1.0f)
                s.frame += explosion_rate * dt;
        }
        for (int i = 0; i < x_count; i++) {
            const int k = (8 + x_next - 1 - i) % 8;
            Boom& b = booms[k];
            if (b.frame == -1.0f) continue;
            if (b.frame >= 4.0f) {
                x_count--;
                b.frame = -1.0f;
            } else if (b.frame >= 0.0f)
                b.frame += explosion_rate * dt;
        }
        if (phase_index >= 6) alive = false;
        return alive;
    }

    void advance(int action) override {  // bossfight.cpp:308-325
        const float dt = 1.0f / 4;
        for (int ss = 0; ss < 4; ss++) {
            bool agent_alive = agent_update(dt, action);
            bool boss_alive = boss_update(dt);
            // System_Sprite_Render::update: list the barrier sprites (set order, then std::sort on z)
            draw_list_.resize(in_sprite_.size());
            int k = 0;
            for (int id : in_sprite_) draw_list_[k++] = {0.0f, id};
            std::sort(draw_list_.begin(), draw_list_.end(),
                      [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first < b.first; });
            reward = (!agent_alive) * -10.0f + (!boss_alive) * 10.0f;
            terminated = !agent_alive || !boss_alive;
            truncated = false;
            if (terminated) break;
        }
    }

    void paint() override {  // bossfight.cpp:401-424
        painter_.target->clear_black();
        painter_.cam_scale = 1.0f * static_cast<float>(view_w_) / static_cast<float>(kObsW);
        painter_.cam_size = {static_cast<float>(view_w_), static_cast<float>(view_h_)};
        const V2 cs = painter_.cam_size;
        const float sc = painter_.cam_scale;
        const Texture* bg = tex_space_[backdrop_];
        painter_.draw(bg, V2{-cs.x / sc * 0.5f, -cs.y / sc * 0.5f}, 1.0f / bg->h * cs.y / sc);

        // negative-z sprites: none (barriers have z = 0)
        // System_Mob_AI::render (common_systems.cpp:392-450)
        for (int i = 0; i < b_count; i++) {
            const int k = (64 + b_next - 1 - i) % 64;
            const Shot& s = b_shots[k];
            if (s.frame == -1.0f) continue;
            const Texture* t = (s.frame == 0.0f) ? tex_laser_[b_laser] : tex_boom_[static_cast<int>(s.frame - 1.0f)];
            const float size = 0.1f;
            painter_.draw_rotated(t, V2{s.pos.x * kUnitPx - size * t->w * 0.5f, s.pos.y * kUnitPx - size * t->h * 0.5f},
                                  s.rotation + M_PI * 0.5f, size);
        }
        {
            const float size = 0.25f;
            const Texture* t = tex_boss_[b_ship];
            painter_.draw(t, V2{b_pos.x * kUnitPx - size * t->w * 0.5f, b_pos.y * kUnitPx - size * t->h * 0.5f}, size);
        }
        if (phase_index % 2 == 0) {
            const float size = 0.25f;
            painter_.draw(tex_shield_,
                          V2{b_pos.x * kUnitPx - size * tex_shield_->w * 0.5f, b_pos.y * kUnitPx - size * tex_shield_->h * 0.5f},
                          size, 0.7f);
        }
        for (int i = 0; i < x_count; i++) {
            const int k = (8 + x_next - 1 - i) % 8;
            const Boom& b = booms[k];
            if (b.frame == -1.0f) continue;
            const Texture* t = tex_boom_[static_cast<int>(b.frame)];
            const float size = 0.3f;
            painter_.draw(t, V2{b.pos.x * kUnitPx - size * t->w * 0.5f, b.pos.y * kUnitPx - size * t->h * 0.5f}, size);
        }
        // positive-z sprites: the barriers (bossfight.cpp:479), common_systems.cpp:22-48
        for (auto& zi : draw_list_) {
            for (int r = 0; r < n_rocks; r++)
                if (rock_id_[r] == zi.second) {
                    const Texture* t = tex_rock_[rock_tex_[r]];
                    float scale = 1.0f * 0.3f;
                    painter_.draw(t, V2{(rock_pos_[r].x + -0.15f) * kUnitPx, (rock_pos_[r].y + -0.15f) * kUnitPx},
                                  scale * kUnitPx / t->w, 1.0f, false, false);
                }
        }
        // System_Agent::render (common_systems.cpp:685-721)
        for (int i = 0; i < a_count; i++) {
            const int k = (32 + a_next - 1 - i) % 32;
            const Shot& s = a_shots[k];
            if (s.frame == -1.0f) continue;
            const Texture* t = (s.frame == 0.0f) ? tex_laser_[a_laser] : tex_boom_[static_cast<int>(s.frame - 1.0f)];
            const float size = 0.05f;
            painter_.draw(t, V2{s.pos.x * kUnitPx - size * t->w * 0.5f, s.pos.y * kUnitPx - size * t->h * 0.5f}, size);
        }
        {
            const float size = 0.05f;
            const Texture* t = tex_player_[a_ship];
            painter_.draw(t, V2{a_pos.x * kUnitPx - size * t->w * 0.5f, a_pos.y * kUnitPx - size * t->h * 0.5f}, size);
        }
    }

   private:
    IdPool ids_;
    IdSet in_hazard_, in_sprite_;
    std::vector<std::pair<float, int>> draw_list_;
    int boss_id_ = 1;
    // agent
    V2 a_pos, a_vel;
    Shot a_shots[32];
    int a_next = 0, a_count = 0, a_ship = 0, a_laser = 0;
    float a_timer = 0.0f;
    bool a_alive = true;
    // boss
    V2 b_pos, b_vel;
    float phase_timer = 0.0f, attack_timer = 0.0f;
    int phase_index = 0, weapon_index = 0, hp = 0;
    Shot b_shots[64];
    Boom booms[8];
    int b_next = 0, x_next = 0, b_count = 0, x_count = 0, b_ship = 0, b_laser = 0;
    float explosion_timer = 0.0f, damage_timer = 0.0f, move_timer = 0.0f;
    // barriers
    int n_rocks = 0, rock_id_[4] = {}, rock_tex_[4] = {};
    V2 rock_pos_[4];
    int backdrop_ = 0;

    const Texture* tex_space_[13] = {};
    const Texture* tex_rock_[8] = {};
    const Texture* tex_boss_[4] = {};
    const Texture* tex_player_[4] = {};
    const Texture* tex_laser_[3] = {};
    const Texture* tex_boom_[5] = {};
    const Texture* tex_shield_ = nullptr;
};

}  // namespace

Env* new_bossfight() { return new Bossfight(); }

}  // namespace pgo
