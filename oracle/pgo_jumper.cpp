// ORACLE — TEST INFRASTRUCTURE ONLY (see pgo_common.h header).
//
// jumper: CPU restatement of SURVEY.md row G6.
//   step   games/jumper/jumper.cpp:340-389, common_systems.cpp:57-202 (agent), :255-283 (particles), :7-24 (sprites)
//   render games/jumper/jumper.cpp:445-509 (incl. the compass HUD), tilemap.cpp:255-281, common_systems.cpp:26-48,
//          :204-247 (agent), :285-308 (particles)
//   reset  games/jumper/jumper.cpp:511-533, tilemap.cpp:79-253, maze_generator.cpp:47-173, room_generator.cpp:4-202
// Config = the reference's compile-time default, hard_mode (40×40, pruned; jumper/tilemap.h:44-46).
// atan2f/sqrtf/fmodf are the process's libm (glibc), what std::atan2/std::sqrt/std::fmod(float) resolve to.
// D21 applies to one line here: `abs(dynamics.velocity.x) > 0.01f` (common_systems.cpp:198) truncates to int first.
#include <algorithm>
#include <cmath>

#include "pgo_env.h"
#include "pgo_kruskal.h"
#include "pgo_rooms.h"

namespace pgo {
namespace {

const char* const kBackdrops[49] = {  // jumper.cpp:59-109 (the list coinrun uses)
    "platform_backgrounds/alien_bg",          "platform_backgrounds/another_world_bg",
    "platform_backgrounds/back_cave",         "platform_backgrounds/caverns",
    "platform_backgrounds/cyberpunk_bg",      "platform_backgrounds/parallax_forest",
    "platform_backgrounds/scifi_bg",          "platform_backgrounds/scifi2_bg",
    "platform_backgrounds/living_tissue_bg",  "platform_backgrounds/airadventurelevel1",
    "platform_backgrounds/airadventurelevel2", "platform_backgrounds/airadventurelevel3",
    "platform_backgrounds/airadventurelevel4", "platform_backgrounds/cave_background",
    "platform_backgrounds/blue_desert",       "platform_backgrounds/blue_grass",
    "platform_backgrounds/blue_land",         "platform_backgrounds/blue_shroom",
    "platform_backgrounds/colored_desert",    "platform_backgrounds/colored_grass",
    "platform_backgrounds/colored_land",      "platform_backgrounds/colored_shroom",
    "platform_backgrounds/landscape1",        "platform_backgrounds/landscape2",
    "platform_backgrounds/landscape3",        "platform_backgrounds/landscape4",
    "platform_backgrounds/battleback1",       "platform_backgrounds/battleback2",
    "platform_backgrounds/battleback3",       "platform_backgrounds/battleback4",
    "platform_backgrounds/battleback5",       "platform_backgrounds/battleback6",
    "platform_backgrounds/battleback7",       "platform_backgrounds/battleback8",
    "platform_backgrounds/battleback9",       "platform_backgrounds/battleback10",
    "platform_backgrounds/sunrise",           "platform_backgrounds_2/beach1",
    "platform_backgrounds_2/beach2",          "platform_backgrounds_2/beach3",
    "platform_backgrounds_2/beach4",          "platform_backgrounds_2/fantasy1",
    "platform_backgrounds_2/fantasy2",        "platform_backgrounds_2/fantasy3",
    "platform_backgrounds_2/fantasy4",        "platform_backgrounds_2/candy1",
    "platform_backgrounds_2/candy2",          "platform_backgrounds_2/candy3",
    "platform_backgrounds_2/candy4"};
const char* const kTops[4] = {"tileBlue_05", "tileGreen_05", "tileYellow_06", "tileBrown_06"};  // tilemap.cpp:13-16
const char* const kMids[4] = {"tileBlue_08", "tileGreen_08", "tileYellow_09", "tileBrown_09"};  // tilemap.cpp:18-21

struct Hit {
    V2 at;
    bool any;
};
struct Puff {  // common_components.h:54-57
    V2 pos;
    float life = 0.0f;
};

class Jumper final : public Env {
   public:
    int W = 40, H = 40;  // tilemap.cpp world_dim: hard 40 (the default), easy 20
    enum Tile : uint8_t { kEmpty = 0, kWallTop = 1, kWallMid = 2, kSpike = 3 };  // tilemap.h:19-26

    int dump_state(float* out, int cap) const override {
        std::vector<float> v = {a_pos.x, a_pos.y, a_vel.x, a_vel.y, static_cast<float>(a_ground),
                                static_cast<float>(a_forward), a_phase, a_jump_timer, static_cast<float>(a_jumps),
                                painter_.cam_pos.x, painter_.cam_pos.y, to_goal.x, to_goal.y,
                                static_cast<float>(backdrop_), backdrop_shift_, static_cast<float>(theme_),
                                puff_timer, static_cast<float>(puff_on), goal_pos.x, goal_pos.y,
                                static_cast<float>(spikes_.size())};
        for (int i = 0; i < 10; i++) {
            v.push_back(puffs[i].pos.x);
            v.push_back(puffs[i].pos.y);
            v.push_back(puffs[i].life);
        }
        for (const V2& s : spikes_) {
            v.push_back(s.x);
            v.push_back(s.y);
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
        if (mode_ == kEasy) W = H = 20;  // tilemap.cpp:82-87: world_dim by Distribution_Mode
        if (mode_ == kMemory) W = H = 45;
        tiles_.assign(W * H, 0);
        auto& bank = TextureBank::global();
        auto T = [&](const std::string& n) { return bank.find("assets/" + n + ".png"); };
        for (int i = 0; i < 49; i++) tex_backdrop_[i] = T(kBackdrops[i]);
        for (int i = 0; i < 4; i++) {
            tex_top_[i] = T(std::string("platformer/") + kTops[i]);
            tex_mid_[i] = T(std::string("platformer/") + kMids[i]);
        }
        tex_spike_ = T("misc_assets/spikeMan_stand");
        tex_carrot_ = T("misc_assets/carrot");
        tex_stand_ = T("misc_assets/bunny2_ready");
        tex_jump_ = T("misc_assets/bunny2_jump");
        tex_walk1_ = T("misc_assets/bunny2_walk1");
        tex_walk2_ = T("misc_assets/bunny2_walk2");
        tex_puff_ = T("misc_assets/iconCircle_white");
        tex_circle_ = T("custom/jumper_compass_circle");
        tex_needle_ = T("custom/jumper_compass_needle");
        tex_bar_ = T("custom/jumper_compass_bar");
    }

    int at(int x, int y) const {  // tilemap.h:84-89: out of bounds is a wall
        if (x < 0 || y < 0 || x >= W || y >= H) return kWallMid;
        return tiles_[y + x * H];
    }
    void put(int x, int y, int id) {
        if (x < 0 || y < 0 || x >= W || y >= H) return;
        tiles_[y + x * H] = static_cast<uint8_t>(id);
    }
    bool space_on_ground(int x, int y) const {  // tilemap.cpp:52-62
        if (at(x, y) != kEmpty) return false;
        if (at(x, y + 1) != kEmpty) return false;
        const int below = at(x, y - 1);
        return below == kWallMid || below == kWallTop;
    }
    bool left_wall(int x, int y) const { return at(x, y) == kWallMid && at(x + 1, y) == kEmpty; }
    bool right_wall(int x, int y) const { return at(x, y) == kWallMid && at(x - 1, y) == kEmpty; }

    void new_level() override {  // jumper.cpp:511-533
        spikes_.clear();
        in_sprite_.clear();
        ids_.refill();

        // tilemap.cpp:79-253
        std::uniform_real_distribution<float> dist01(0.0f, 1.0f);
        std::mt19937& rng = rng_.eng;
        std::fill(tiles_.begin(), tiles_.end(), static_cast<uint8_t>(kEmpty));
        const int maze_scale = 3, maze_dim = W / maze_scale;
        std::vector<int> maze;
        carve_merged(maze_dim, maze, rng_);  // generate_maze_no_dead_ends = generate_maze + the dead-end pass
        open_dead_ends(maze_dim, maze, rng_);  // (maze_generator.cpp:47-173, pgo_kruskal.h)
        Rooms rooms;
        rooms.gw = W;
        rooms.gh = H;
        rooms.grid.assign(W * H, 0);
        for (int i = 0; i < W * H; i++) {
            const int obj = maze[((i % H) / maze_scale + 1) + (maze_dim + 2) * ((i / H) / maze_scale + 1)];
            const float prob = obj == 1 ? 0.8f : 0.2f;
            tiles_[i] = dist01(rng) < prob ? kWallMid : kEmpty;
            rooms.grid[i] = tiles_[i] == kWallMid ? 1 : 0;
        }
        for (int it = 0; it < 2; it++) rooms.update();
        for (int i = 0; i < W; i++) {
            rooms.grid[0 + H * i] = 1;
            rooms.grid[(H - 1) + H * i] = 1;
        }
        for (int i = 0; i < H; i++) {
            rooms.grid[i + H * 0] = 1;
            rooms.grid[i + H * (W - 1)] = 1;
        }
        std::unordered_set<int> best_room;
        rooms.find_best_room(best_room);
        for (int i = 0; i < W * H; i++) tiles_[i] = kWallMid;
        std::vector<int> free_cells;
        for (int i : best_room) {
            tiles_[i] = kEmpty;
            free_cells.push_back(i);
        }
        const int goal_cell = free_cells[rng_.irange(0, static_cast<int>(free_cells.size()) - 1)];
        std::vector<int> candidates;
        for (int x = 0; x < W; x++)
            for (int y = 0; y < H; y++) {
                const int i = y + H * x;
                if (space_on_ground(x, y) && i != goal_cell) candidates.push_back(i);
            }
        const int agent_cell = candidates[rng_.irange(0, static_cast<int>(candidates.size()) - 1)];
        std::vector<int> goal_path;
        rooms.find_path(agent_cell, goal_cell, goal_path);
        if (mode_ != kMemory) {  // should_prune (tilemap.cpp:176-187)
            std::unordered_set<int> wide;
            wide.insert(goal_path.begin(), goal_path.end());
            rooms.expand_room(wide, 4);
            for (int i = 0; i < W * H; i++) tiles_[i] = kWallMid;
            for (int i : wide) tiles_[i] = kEmpty;
        }
        const int goal_id = ids_.take();
        in_sprite_.insert(goal_id);
        goal_pos = {static_cast<float>(goal_cell / H) + 0.5f, static_cast<float>(H - 1 - goal_cell % H) + 0.5f};

        const float spike_prob = mode_ == kMemory ? 0.0f : 0.2f;  // tilemap.cpp:205
        for (int x = 0; x < W; x++)
            for (int y = 0; y < H; y++)
                if (space_on_ground(x, y) && space_on_ground(x - 1, y) && space_on_ground(x + 1, y))
                    if (dist01(rng) < spike_prob) put(x, y, kSpike);
        for (int x = 0; x < W; x++)  // no long vertical walls (:215-224)
            for (int y = 0; y < H; y++) {
                if (left_wall(x, y) && left_wall(x, y + 1) && left_wall(x, y + 2)) put(x, y + rng_.irange(0, 2), kEmpty);
                if (right_wall(x, y) && right_wall(x, y + 1) && right_wall(x, y + 2)) put(x, y + rng_.irange(0, 2), kEmpty);
            }
        ids_.take();  // the agent entity
        a_pos = {static_cast<float>(static_cast<int>(agent_cell / H)) + 0.5f, static_cast<float>(H - 1 - (agent_cell % H))};
        a_vel = {0.0f, 0.0f};
        a_ground = false;
        a_forward = true;
        a_phase = 0.0f;
        a_jump_timer = 0.0f;
        a_jumps = 2;
        for (auto& p : puffs) p = Puff{};
        puff_timer = 0.0f;
        puff_on = true;
        for (int i = 0; i < W * H; i++)
            if (tiles_[i] == kSpike) {
                tiles_[i] = kEmpty;
                if (i != agent_cell && i != goal_cell) {
                    const int id = ids_.take();
                    in_sprite_.insert(id);
                    if (static_cast<int>(spikes_.size()) <= id - 2) spikes_.resize(id - 1);
                    spikes_[id - 2] = {static_cast<float>(i / H) + 0.5f, static_cast<float>(H - 1 - i % H) + 0.5f};
                }
            }
        for (int x = 0; x < W; x++)
            for (int y = 0; y < H; y++)
                if (at(x, y) == kWallMid && at(x, y + 1) == kEmpty) put(x, y, kWallTop);

        backdrop_ = rng_.irange(0, 48);
        backdrop_shift_ = rng_.unit();
        theme_ = rng_.irange(0, 3);
        draw_list_.clear();
        // camera and System_Agent::info.to_goal keep their previous values until the first update (D3)
    }

    template <class Pred>
    Hit collide(Box r, Pred solid) const {  // tilemap.cpp:283-345 (variant B)
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

    void agent_update(float dt, int action, bool& alive, bool& achieved_goal) {  // common_systems.cpp:57-202
        alive = true;
        achieved_goal = false;
        const float max_jump = 0.92f, gravity = 0.1f, max_speed = 0.5f, mix = 0.2f, air_control = 1.0f,
                    jump_cooldown = 3.0f;
        float movement_x = (action == 6 || action == 7 || action == 8) - (action == 0 || action == 1 || action == 2);
        const bool jump = (action == 2 || action == 5 || action == 8);
        const float mix_x = a_ground ? mix : (mix * air_control);
        a_vel.x += mix_x * (max_speed * movement_x - a_vel.x) * dt;
        if (std::abs(a_vel.x) < mix_x * max_speed * dt) a_vel.x = 0.0f;
        if (a_ground) a_jumps = 2;
        if (jump && a_jumps > 0 && a_jump_timer == 0.0f) {
            a_vel.y = -max_jump;
            a_jumps--;
            a_jump_timer = jump_cooldown;
        }
        if (a_jump_timer > 0.0f) a_jump_timer = std::max(0.0f, a_jump_timer - dt);
        a_vel.y += gravity * dt;
        if (std::abs(a_vel.y) > max_jump) a_vel.y = (a_vel.y > 0.0f ? 1.0f : -1.0f) * max_jump;
        a_pos.x += a_vel.x * dt;
        a_pos.y += a_vel.y * dt;
        const Box bounds{-0.25f, -0.8f, 0.5f, 0.8f};
        Box wc{a_pos.x + bounds.x, a_pos.y + bounds.y, bounds.w, bounds.h};
        const Hit hit = collide(wc, [](int t) { return t == kWallMid || t == kWallTop; });
        const V2 delta{hit.at.x - wc.x, hit.at.y - wc.y};
        a_ground = delta.y < 0.0f && hit.any;
        a_pos.x = hit.at.x - bounds.x;
        a_pos.y = hit.at.y - bounds.y;
        wc.x = a_pos.x + bounds.x;
        wc.y = a_pos.y + bounds.y;
        if (delta.x != 0.0f) a_vel.x = 0.0f;
        if (delta.y > 0.0f && hit.any) a_vel.y = 0.0f;
        if (a_ground) a_vel.y = 0.0f;
        for (const V2& s : spikes_)
            if (boxes_touch(wc, Box{s.x + -0.25f, s.y + -0.25f, 0.5f, 0.5f})) {
                alive = false;
                break;
            }
        if (boxes_touch(wc, Box{goal_pos.x + -0.5f, goal_pos.y + -0.5f, 1.0f, 1.0f})) achieved_goal = true;
        painter_.cam_pos.x = a_pos.x * kUnitPx;
        painter_.cam_pos.y = (a_pos.y - 0.5f) * kUnitPx;
        a_phase += 0.1f * dt;
        a_phase = std::fmod(a_phase, 1.0f);
        if (movement_x > 0.0f)
            a_forward = true;
        else if (movement_x < 0.0f)
            a_forward = false;
        to_goal = {goal_pos.x - a_pos.x, goal_pos.y - a_pos.y};
        // `abs` = int abs(int) here (D21); game_flags bit 0 (PGV_JUMPER_FLOAT_ABS): the float overload (see pgo_chaser.cpp)
        puff_on = !a_ground || ((flags_ & 1u) ? std::fabs(a_vel.x) : static_cast<float>(std::abs(static_cast<int>(a_vel.x)))) > 0.01f;
    }

    void puffs_update(float dt) {  // common_systems.cpp:255-283
        const float lifespan = 5.0f, spawn_time = 0.5f;
        int dead_index = -1;
        for (int i = 0; i < 10; i++) {
            puffs[i].life -= dt;
            if (puffs[i].life <= 0.0f) dead_index = i;
        }
        puff_timer += dt;
        if (dead_index != -1 && puff_timer >= spawn_time && puff_on) {
            puff_timer = std::fmod(puff_timer, spawn_time);
            Puff& p = puffs[dead_index];
            p.life = lifespan;
            p.pos.x = a_pos.x + 0.0f;
            p.pos.y = a_pos.y + -0.2f;
        }
    }

    void sprites_update() {  // common_systems.cpp:7-24
        draw_list_.resize(in_sprite_.size());
        int k = 0;
        for (int id : in_sprite_) draw_list_[k++] = {1.0f, id};
        std::sort(draw_list_.begin(), draw_list_.end(),
                  [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first < b.first; });
    }

    void advance(int action) override {  // jumper.cpp:355-371
        const float dt = 1.0f / 4;
        for (int ss = 0; ss < 4; ss++) {
            bool alive, achieved_goal;
            agent_update(dt, action, alive, achieved_goal);
            puffs_update(dt);
            sprites_update();
            reward = achieved_goal * 10.0f;
            terminated = !alive || achieved_goal;
            truncated = false;
            if (terminated) break;
        }
    }

    void paint() override {  // jumper.cpp:445-509
        painter_.target->clear_black();
        const float game_zoom = 0.3f;
        painter_.cam_scale = game_zoom * static_cast<float>(view_w_) / static_cast<float>(kObsW);
        painter_.cam_size = {static_cast<float>(view_w_), static_cast<float>(view_h_)};
        const Texture* bg = tex_backdrop_[backdrop_];
        const float aspect = static_cast<float>(bg->w) / static_cast<float>(bg->h);
        const float extra = aspect - 1.0f;
        painter_.draw(bg, V2{-backdrop_shift_ * extra, 0.0f}, 64.0f * kUnitPx / bg->h);
        {  // tilemap.cpp:255-281
            const V2& cp = painter_.cam_pos;
            const V2& cs = painter_.cam_size;
            const float sc = painter_.cam_scale;
            Box view{(cp.x - cs.x * 0.5f / sc) * kPxUnit, (cp.y - cs.y * 0.5f / sc) * kPxUnit, cs.x * kPxUnit / sc,
                     cs.y * kPxUnit / sc};
            int x0 = std::floor(view.x), y0 = std::floor(view.y);
            int x1 = std::ceil(view.x + view.w), y1 = std::ceil(view.y + view.h);
            for (int y = y0; y <= y1; y++)
                for (int x = x0; x <= x1; x++) {
                    const int t = at(x, H - 1 - y);
                    if (t != kWallTop && t != kWallMid) continue;
                    const Texture* tex = (t == kWallTop) ? tex_top_[theme_] : tex_mid_[theme_];
                    painter_.draw(tex, V2{x * kUnitPx, y * kUnitPx}, kUnitPx / tex->w);
                }
        }
        {  // System_Particles::render (common_systems.cpp:285-308)
            const float base_alpha = 0.5f, base_scale = 0.45f, lifespan = 5.0f;
            for (int i = 0; i < 10; i++) {
                const Puff& p = puffs[i];
                if (p.life <= 0.0f) continue;
                float life_ratio = (lifespan - p.life) / lifespan;
                float alpha = base_alpha * (1.0f - life_ratio);
                float scale = base_scale * (0.4f * life_ratio + 0.6f);
                float offset_y = -life_ratio * 0.17f;
                painter_.draw(tex_puff_,
                              V2{p.pos.x * kUnitPx - 0.5f * tex_puff_->w * scale,
                                 (p.pos.y + offset_y) * kUnitPx - 0.5f * tex_puff_->h * scale},
                              scale * kUnitPx / tex_puff_->w, alpha);
            }
        }
        for (auto& zi : draw_list_) {  // positive-z sprites (common_systems.cpp:26-48): carrot and spikes
            const int id = zi.second;
            if (id == 0) {
                float scale = 1.0f * 1.0f;
                painter_.draw(tex_carrot_, V2{(goal_pos.x + -0.5f) * kUnitPx, (goal_pos.y + -0.5f) * kUnitPx},
                              scale * kUnitPx / tex_carrot_->w, 1.0f, false);
            } else {
                const V2& s = spikes_[id - 2];
                float scale = 1.0f * 0.4f;
                painter_.draw(tex_spike_, V2{(s.x + -0.25f) * kUnitPx, (s.y + -0.25f) * kUnitPx},
                              scale * kUnitPx / tex_spike_->w, 1.0f, false);
            }
        }
        {  // System_Agent::render (common_systems.cpp:204-247)
            const Texture* tex;
            float agent_scale = 1.0f;
            V2 off{0.0f, 0.0f};
            if (std::abs(a_vel.x) < 0.01f && a_ground) {
                tex = tex_stand_;
                agent_scale = 0.5f;
                off = {0.0f, 0.2f};
            } else if (!a_ground) {
                tex = tex_jump_;
                agent_scale = 0.6f;
                off = {-0.05f, 0.25f};
            } else if (a_phase > 0.5f) {
                tex = tex_walk2_;
                agent_scale = 0.5f;
                off = {0.0f, 0.2f};
            } else {
                tex = tex_walk1_;
                agent_scale = 0.5f;
                off = {0.0f, 0.2f};
            }
            const V2 position{a_pos.x - 0.25f, a_pos.y - 1.0f};
            painter_.draw(tex, V2{(position.x + off.x) * kUnitPx, (position.y + off.y) * kUnitPx},
                          kUnitPx / tex->w * agent_scale, 1.0f, !a_forward);
        }
        if (!painter_.enabled) return;
        // compass HUD: raw SDL_RenderTextureRotated calls in screen space (jumper.cpp:473-509)
        const float width = static_cast<float>(view_w_);
        const float compass_size = 200.0f;
        const V2 compass_offset{-32.0f, 32.0f};
        float angle = std::atan2(to_goal.y, to_goal.x) * 180.0f / M_PI;
        float dist = std::sqrt(to_goal.x * to_goal.x + to_goal.y * to_goal.y);
        float dist_inv = 1.0f / std::max(0.0001f, dist);
        V2 dir{to_goal.x * dist_inv, to_goal.y * dist_inv};
        float ratio = std::min(1.0f, dist / (W * 1.414f));
        auto whole = [&](const Texture* t, float dx, float dy, float dw, float dh, double deg) {
            painter_.draw_calls++;
            spec_blit(*painter_.target, *t, 0.0f, 0.0f, static_cast<float>(t->w), static_cast<float>(t->h), dx, dy, dw, dh, deg,
                      kFlipNone, 255);
        };
        whole(tex_circle_, width - compass_size * game_zoom + compass_offset.x * game_zoom, compass_offset.y * game_zoom,
              compass_size * game_zoom, compass_size * game_zoom, 0.0f);
        {
            float dx = width - compass_size * 0.75f * game_zoom + compass_offset.x * game_zoom;
            float dy = compass_size * 0.5f * game_zoom + compass_offset.y * game_zoom;
            dx += compass_size * 0.25f * dir.x * game_zoom;
            dy += compass_size * 0.25f * dir.y * game_zoom;
            whole(tex_needle_, dx, dy, compass_size * 0.5f * game_zoom, compass_size * 0.1f * game_zoom, angle);
        }
        whole(tex_bar_, width - compass_size * game_zoom + compass_offset.x * game_zoom,
              compass_size * game_zoom + compass_offset.y * game_zoom, compass_size * game_zoom * ratio,
              compass_size * 0.15f * game_zoom, 0.0f);
    }

   private:
    std::vector<uint8_t> tiles_;
    std::vector<V2> spikes_;  // entity id − 2
    IdPool ids_;
    IdSet in_sprite_;
    std::vector<std::pair<float, int>> draw_list_;
    V2 a_pos, a_vel, goal_pos, to_goal;
    bool a_ground = false, a_forward = true;
    float a_phase = 0.0f, a_jump_timer = 0.0f;
    int a_jumps = 2;
    Puff puffs[10];
    float puff_timer = 0.0f;
    bool puff_on = true;
    int backdrop_ = 0, theme_ = 0;
    float backdrop_shift_ = 0.0f;
    const Texture* tex_backdrop_[49] = {};
    const Texture* tex_top_[4] = {};
    const Texture* tex_mid_[4] = {};
    const Texture* tex_spike_ = nullptr;
    const Texture* tex_carrot_ = nullptr;
    const Texture* tex_stand_ = nullptr;
    const Texture* tex_jump_ = nullptr;
    const Texture* tex_walk1_ = nullptr;
    const Texture* tex_walk2_ = nullptr;
    const Texture* tex_puff_ = nullptr;
    const Texture* tex_circle_ = nullptr;
    const Texture* tex_needle_ = nullptr;
    const Texture* tex_bar_ = nullptr;
};

}  // namespace

Env* new_jumper() { return new Jumper(); }

}  // namespace pgo
