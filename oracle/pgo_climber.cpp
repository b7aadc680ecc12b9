// ORACLE — TEST INFRASTRUCTURE ONLY (see pgo_common.h header).
//
// climber: CPU restatement of SURVEY.md row G7.
//   step   games/climber/climber.cpp:323-376, common_systems.cpp:184-270 (agent), :109-168 (mobs), :66-107 (points),
//          :8-39 (sprite list)
//   render games/climber/climber.cpp:431-459, tilemap.cpp:172-198, common_systems.cpp:41-63, :272-298
//   reset  games/climber/climber.cpp:461-497, tilemap.cpp:40-70, :75-170
// Config = the reference's compile-time default (easy_mode = false, climber/tilemap.h:32-34).
// The per-step `std::cout << "REWARD…"` of the reference (D18) is not reproduced.
#include <algorithm>

#include "pgo_env.h"

namespace pgo {
namespace {

enum Tile : uint8_t { kEmpty = 0, kWallTop, kWallMid };  // tilemap.h:12-17

const char* const kBackdrops[10] = {  // climber.cpp:60-71
    "platform_backgrounds/alien_bg.png",    "platform_backgrounds/another_world_bg.png",
    "platform_backgrounds_2/fantasy1.png",  "platform_backgrounds_2/fantasy2.png",
    "platform_backgrounds_2/fantasy3.png",  "platform_backgrounds_2/fantasy4.png",
    "platform_backgrounds_2/candy1.png",    "platform_backgrounds_2/candy2.png",
    "platform_backgrounds_2/candy3.png",    "platform_backgrounds_2/candy4.png"};
const char* const kTops[4] = {"tileBlue_05", "tileGreen_05", "tileYellow_06", "tileBrown_06"};  // tilemap.cpp:10-13
const char* const kMids[4] = {"tileBlue_08", "tileGreen_08", "tileYellow_09", "tileBrown_09"};  // tilemap.cpp:15-18
const char* const kSuits[4] = {"Blue", "Green", "Grey", "Red"};                                 // common_systems.h

struct Thing {
    bool is_mob = false, alive = false;
    V2 pos;
    // mob
    float vel_x = 0.15f;
    int spawn_x = 0;
    bool flip_x = false;
    int frame = 0;
    float anim_t = 0.0f;
    bool tex_set = false;
};

struct Hit {
    V2 at;
    bool any;
};

class Climber final : public Env {
   public:
    static constexpr int W = 20, H = 64;

    int dump_state(float* out, int cap) const override {
        std::vector<float> v = {a_pos.x, a_pos.y, a_vel.x, a_vel.y, static_cast<float>(a_ground),
                                static_cast<float>(a_forward), a_phase, painter_.cam_pos.x, painter_.cam_pos.y,
                                static_cast<float>(backdrop_), backdrop_shift_, static_cast<float>(suit_),
                                static_cast<float>(theme_), static_cast<float>(n_things_)};
        for (int id = 0; id < n_things_; id++) {
            const Thing& t = things_[id];
            v.push_back(t.alive ? 1.0f : 0.0f);
            v.push_back(t.pos.x);
            v.push_back(t.pos.y);
            v.push_back(t.is_mob ? t.vel_x : 0.0f);
            v.push_back(static_cast<float>(t.frame));
            v.push_back(t.anim_t);
        }
        int n = std::min<int>(cap, static_cast<int>(v.size()));
        std::memcpy(out, v.data(), n * sizeof(float));
        return static_cast<int>(v.size());
    }
    int dump_tiles(uint8_t* out, int cap) const override {
        int n = std::min<int>(cap, W * H);
        std::memcpy(out, tiles_.data(), n);
        return n;
    }

   protected:
    void on_make() override {
        auto& bank = TextureBank::global();
        auto T = [&](const std::string& n) { return bank.find("assets/" + n); };
        for (int i = 0; i < 4; i++) {
            tex_top_[i] = T(std::string("platformer/") + kTops[i] + ".png");
            tex_mid_[i] = T(std::string("platformer/") + kMids[i] + ".png");
            std::string p = std::string("platformer/player") + kSuits[i];
            tex_stand_[i] = T(p + "_stand.png");
            tex_jump_[i] = T(p + "_walk4.png");  // common_systems.cpp:178
            tex_walk1_[i] = T(p + "_walk1.png");
            tex_walk2_[i] = T(p + "_walk2.png");
        }
        tex_fish_[0] = T("platformer/enemySwimming_1.png");
        tex_fish_[1] = T("platformer/enemySwimming_2.png");
        tex_gem_ = T("misc_assets/yellowCrystal.png");
        for (int i = 0; i < 10; i++) tex_backdrop_[i] = T(kBackdrops[i]);
    }

    void put(int x, int y, Tile t) {
        if (x < 0 || y < 0 || x >= W || y >= H) return;
        tiles_[y + x * H] = t;
    }
    Tile at(int x, int y) const {
        if (x < 0 || y < 0 || x >= W || y >= H) return kWallMid;
        return static_cast<Tile>(tiles_[y + x * H]);
    }
    void fill(int x, int y, int w, int h, Tile t) {
        for (int i = 0; i < w; i++)
            for (int j = 0; j < h; j++) put(x + i, y + j, t);
    }
    void fill_capped(int x, int y, int w, int h, Tile body, Tile cap) {
        fill(x, y, w, h - 1, body);
        fill(x, y + h - 1, w, 1, cap);
    }

    int spawn(const Thing& t) {
        int id = ids_.take();
        things_[id] = t;
        things_[id].alive = true;
        n_things_ = std::max(n_things_, id + 1);
        in_tilemap_.insert(id);
        return id;
    }
    void add_mob(int x, int y) {  // tilemap.cpp:40-58
        Thing t;
        t.is_mob = true;
        t.pos = {static_cast<float>(x) + 0.5f, static_cast<float>(H - 1 - y) + 0.5f};
        t.vel_x = 0.15f * (rng_.irange(0, 1) * 2.0f - 1.0f);
        t.spawn_x = x;
        int id = spawn(t);
        in_sprite_.insert(id);
        in_mob_.insert(id);
    }
    void add_gem(int x, int y) {  // tilemap.cpp:60-70
        Thing t;
        t.pos = {static_cast<float>(x) + 0.5f, static_cast<float>(H - 1 - y) + 0.5f};
        t.tex_set = true;
        int id = spawn(t);
        in_sprite_.insert(id);
        in_point_.insert(id);
    }

    void new_level() override {  // climber.cpp:461-497
        ids_.refill();
        in_sprite_.clear();
        in_tilemap_.clear();
        in_mob_.clear();
        in_point_.clear();
        in_agent_.clear();
        n_things_ = 0;

        // tilemap.cpp:75-170
        const float max_jump = 1.5f, gravity = 0.2f;
        std::fill(tiles_.begin(), tiles_.end(), kEmpty);
        fill_capped(0, 0, W, 1, kWallMid, kWallTop);
        fill(0, 0, 1, H, kWallMid);
        fill(W - 1, 0, 1, H, kWallMid);
        fill(0, H - 1, W, 1, kWallMid);
        const int difficulty = rng_.irange(1, 3);
        const int lo = difficulty * difficulty + 1, hi = (difficulty + 1) * (difficulty + 1) + 1;
        const int platforms = rng_.irange(lo, hi);
        int cx = rng_.irange(2, W - 3), cy = 1;
        const int margin = 3;
        const float enemy_prob = mode_ == kEasy ? .2 : .5;  // tilemap.cpp:118
        float reach_y = max_jump * max_jump / (2.0f * gravity);
        const int max_dy = reach_y - 0.5f;
        for (int p = 0; p < platforms; p++) {
            const int dy = rng_.irange(3, max_dy - 1);
            const bool roomy = (cx >= margin) && (cx <= W - 1 - margin);
            if (roomy && (rng_.unit() < enemy_prob)) {
                const int my = cy + rng_.irange(0, 1) + 2;
                add_mob(cx, my);
            }
            cy += dy;
            const int len = 2 + rng_.irange(0, 9);
            int vx = rng_.irange(0, 1) * 2 - 1;
            if (cx < margin) vx = 1;
            if (cx > W - margin) vx = -1;
            std::vector<int> spots;
            for (int j = 0; j < len; j++) {
                int nx = cx + (j + 1) * vx;
                if (nx <= 0 || nx >= W - 1) break;
                spots.push_back(nx);
                fill_capped(nx, cy, 1, 1, kWallMid, kWallTop);
            }
            const int last = static_cast<int>(spots.size()) - 1;
            if (rng_.unit() < .5 || p == platforms - 1) add_gem(spots[rng_.irange(0, last)], cy + 1);
            cx = spots[rng_.irange(0, last)];
        }

        painter_.cam_pos.x = W / 2.0f * kUnitPx;
        backdrop_ = rng_.irange(0, 9);
        backdrop_shift_ = rng_.unit();
        int agent_id = ids_.take();
        in_tilemap_.insert(agent_id);
        in_agent_.insert(agent_id);
        a_pos = {1.5f, H - 2 + 1.0f};
        a_vel = {0.0f, 0.0f};
        a_ground = false;
        a_forward = true;
        a_phase = 0.0f;
        suit_ = rng_.irange(0, 3);
        theme_ = rng_.irange(0, 3);
        draw_list_.clear();
    }

    template <class Pred>
    Hit collide(Box r, Pred solid) const {  // tilemap.cpp:200-258 (variant B)
        bool any = false;
        const int x0 = std::floor(r.x), y0 = std::floor(r.y);
        const int x1 = std::ceil(r.x + r.w), y1 = std::ceil(r.y + r.h);
        const V2 mid{r.x + r.w * 0.5f, r.y + r.h * 0.5f};
        Box cell{0.0f, 0.0f, 1.0f, 1.0f};
        for (int y = y0; y <= y1; y++)
            for (int x = x0; x <= x1; x++) {
                if (!solid(at(x, H - 1 - y))) continue;
                cell.x = x;
                cell.y = y;
                const Box o = overlap_box(r, cell);
                if (o.w == 0.0f && o.h == 0.0f) continue;
                if (o.w > o.h) {
                    r.y = (o.y + o.h * 0.5f > mid.y ? cell.y - r.h : cell.y + cell.h);
                    any = true;
                }
            }
        for (int y = y0; y <= y1; y++)
            for (int x = x0; x <= x1; x++) {
                if (!solid(at(x, H - 1 - y))) continue;
                cell.x = x;
                cell.y = y;
                const Box o = overlap_box(r, cell);
                if (o.w == 0.0f && o.h == 0.0f) continue;
                if (o.w <= o.h) {
                    r.x = (o.x + o.w * 0.5f > mid.x ? cell.x - r.w : cell.x + cell.w);
                    any = true;
                }
            }
        return {{r.x, r.y}, any};
    }
    static bool is_wall(Tile t) { return t == kWallMid || t == kWallTop; }

    void agent_update(float dt, int action) {  // common_systems.cpp:184-270
        const float max_jump = 1.55f, gravity = 0.2f, max_speed = 0.5f, mix = 0.2f, air_control = 0.15f;
        float move_x = (action == 6 || action == 7 || action == 8) - (action == 0 || action == 1 || action == 2);
        const bool jump = (action == 2 || action == 5 || action == 8);
        float mix_x = a_ground ? mix : (mix * air_control);
        a_vel.x += mix_x * (max_speed * move_x - a_vel.x) * dt;
        if (std::abs(a_vel.x) < mix_x * max_speed * dt) a_vel.x = 0.0f;
        if (jump && a_ground) a_vel.y = -max_jump;
        a_vel.y += gravity * dt;
        if (std::abs(a_vel.y) > max_jump) a_vel.y = (a_vel.y > 0.0f ? 1.0f : -1.0f) * max_jump;
        a_pos.x += a_vel.x * dt;
        a_pos.y += a_vel.y * dt;
        Box body{a_pos.x + -0.5f, a_pos.y + -1.0f, 1.0f, 1.0f};
        Hit h = collide(body, is_wall);
        V2 moved{h.at.x - body.x, h.at.y - body.y};
        a_ground = moved.y < 0.0f && h.any;
        a_pos.x = h.at.x - -0.5f;
        a_pos.y = h.at.y - -1.0f;
        if (moved.x != 0.0f) a_vel.x = 0.0f;
        if (a_ground) a_vel.y = 0.0f;
        painter_.cam_pos.y = (a_pos.y - 8 - 0.5f) * kUnitPx;
        a_phase += 0.1f * dt;
        a_phase = std::fmod(a_phase, 1.0f);
        if (move_x > 0.0f)
            a_forward = true;
        else if (move_x < 0.0f)
            a_forward = false;
    }

    bool mobs_update(float dt) {  // common_systems.cpp:109-168
        bool hit = false;
        const int patrol = 4;
        const Box agent{a_pos.x + -0.5f, a_pos.y + -1.0f, 1.0f, 1.0f};
        for (int id : in_mob_) {
            Thing& m = things_[id];
            m.pos.x += m.vel_x * dt;
            Box probe{m.pos.x - 0.5f, m.pos.y - 0.6f, 1.0f, 0.5f};
            Hit w = collide(probe, is_wall);
            m.pos.x = w.at.x + 0.5f;
            Box mb{m.pos.x + -0.4f, m.pos.y + -0.4f, 0.8f, 0.8f};
            if (boxes_touch(agent, mb)) hit = true;
            bool end_patrol = m.pos.x > m.spawn_x + patrol || m.pos.x < m.spawn_x - patrol;
            if (w.any || end_patrol) m.vel_x *= -1.0f;
            m.flip_x = m.vel_x < 0.0f;
        }
        return hit;
    }

    void points_update(int& delta, int& available) {  // common_systems.cpp:66-107
        const Box agent{a_pos.x + -0.5f, a_pos.y + -1.0f, 1.0f, 1.0f};
        delta = 0;
        available = 0;
        std::vector<int> gone;
        for (int id : in_point_) {
            const Thing& t = things_[id];
            Box b{t.pos.x + -0.5f, t.pos.y + -0.5f, 1.0f, 1.0f};
            if (boxes_touch(agent, b)) {
                delta++;
                gone.push_back(id);
            } else
                available++;
        }
        for (int id : gone) {  // Coordinator::destroy_entity (ecs.cpp:85-90)
            ids_.give_back(id);
            things_[id].alive = false;
            in_sprite_.erase(id);
            in_point_.erase(id);
            in_tilemap_.erase(id);
        }
    }

    void sprites_update(float dt) {  // common_systems.cpp:8-39
        draw_list_.resize(in_sprite_.size());
        int k = 0;
        for (int id : in_sprite_) {
            Thing& t = things_[id];
            if (t.is_mob) {
                t.anim_t += dt;
                int adv = t.anim_t * 0.2f;
                t.anim_t -= adv / 0.2f;
                t.frame = (t.frame + adv) % 2;
                t.tex_set = true;
            }
            draw_list_[k++] = {1.0f, id};
        }
        std::sort(draw_list_.begin(), draw_list_.end(),
                  [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first < b.first; });
    }

    void advance(int action) override {  // climber.cpp:338-355
        const float dt = 1.0f / 4;
        for (int ss = 0; ss < 4; ss++) {
            agent_update(dt, action);
            bool dead = mobs_update(dt);
            int delta, available;
            points_update(delta, available);
            sprites_update(dt);
            reward = delta + (available == 0) * 10.0f;
            terminated = dead || (available == 0);
            truncated = false;
            if (terminated) break;
        }
    }

    void paint() override {  // climber.cpp:431-459
        painter_.target->clear_black();
        painter_.cam_scale = 0.2f * static_cast<float>(view_w_) / static_cast<float>(kObsW);
        painter_.cam_size = {static_cast<float>(view_w_), static_cast<float>(view_h_)};
        const Texture* bg = tex_backdrop_[backdrop_];
        float aspect = static_cast<float>(bg->w) / static_cast<float>(bg->h);
        float extra = aspect - 1.0f;
        painter_.draw(bg, V2{-backdrop_shift_ * extra, 0.0f}, 64.0f * kUnitPx / bg->h);
        {  // tilemap.cpp:172-198
            const V2& cp = painter_.cam_pos;
            const V2& cs = painter_.cam_size;
            const float sc = painter_.cam_scale;
            Box view{(cp.x - cs.x * 0.5f / sc) * kPxUnit, (cp.y - cs.y * 0.5f / sc) * kPxUnit, cs.x * kPxUnit / sc,
                     cs.y * kPxUnit / sc};
            int x0 = std::floor(view.x), y0 = std::floor(view.y);
            int x1 = std::ceil(view.x + view.w), y1 = std::ceil(view.y + view.h);
            for (int y = y0; y <= y1; y++)
                for (int x = x0; x <= x1; x++) {
                    Tile t = at(x, H - 1 - y);
                    if (t == kEmpty) continue;
                    const Texture* tex = (t == kWallTop) ? tex_top_[theme_] : tex_mid_[theme_];
                    painter_.draw(tex, V2{x * kUnitPx, y * kUnitPx}, kUnitPx / tex->w);
                }
        }
        for (auto& zi : draw_list_) {  // positive z: mobs (offset -0.4) and gems (offset -0.5)
            const Thing& t = things_[zi.second];
            if (!t.tex_set) continue;
            const Texture* tex = t.is_mob ? tex_fish_[t.frame] : tex_gem_;
            const float off = t.is_mob ? -0.4f : -0.5f;
            float scale = 1.0f * 1.0f;
            painter_.draw(tex, V2{(t.pos.x + off) * kUnitPx, (t.pos.y + off) * kUnitPx}, scale * kUnitPx / tex->w, 1.0f,
                          t.flip_x);
        }
        {  // common_systems.cpp:272-298
            const Texture* tex;
            if (std::abs(a_vel.x) < 0.01f && a_ground)
                tex = tex_stand_[suit_];
            else if (!a_ground)
                tex = tex_jump_[suit_];
            else if (a_phase > 0.5f)
                tex = tex_walk2_[suit_];
            else
                tex = tex_walk1_[suit_];
            V2 p{a_pos.x - 0.5f, a_pos.y - 1.0f};
            painter_.draw(tex, V2{p.x * kUnitPx, p.y * kUnitPx}, 0.8f * kUnitPx / tex->w, 1.0f, !a_forward);
        }
    }

   private:
    std::vector<uint8_t> tiles_ = std::vector<uint8_t>(W * H, 0);
    std::vector<Thing> things_ = std::vector<Thing>(IdPool::kMax);
    int n_things_ = 0;
    IdPool ids_;
    IdSet in_sprite_, in_tilemap_, in_mob_, in_point_, in_agent_;
    std::vector<std::pair<float, int>> draw_list_;
    V2 a_pos, a_vel;
    bool a_ground = false, a_forward = true;
    float a_phase = 0.0f;
    int backdrop_ = 0, suit_ = 0, theme_ = 0;
    float backdrop_shift_ = 0.0f;
    const Texture* tex_top_[4] = {};
    const Texture* tex_mid_[4] = {};
    const Texture* tex_stand_[4] = {};
    const Texture* tex_jump_[4] = {};
    const Texture* tex_walk1_[4] = {};
    const Texture* tex_walk2_[4] = {};
    const Texture* tex_fish_[2] = {};
    const Texture* tex_gem_ = nullptr;
    const Texture* tex_backdrop_[10] = {};
};

}  // namespace

Env* new_climber() { return new Climber(); }

}  // namespace pgo
