// ORACLE — TEST INFRASTRUCTURE ONLY (see pgo_common.h header).
//
// caveflyer: CPU restatement of SURVEY.md rows G3s / G3r / G3g.
//   step   games/caveflyer/caveflyer.cpp:301-357, common_systems.cpp:90-289 (agent + bullets), :50-75 (enemies),
//          :333-372 (exhaust particles), :7-24 (sprite list)
//   render games/caveflyer/caveflyer.cpp:413-440, tilemap.cpp:280-303, common_systems.cpp:26-48, :291-327, :374-397
//   reset  games/caveflyer/caveflyer.cpp:442-460, tilemap.cpp:118-278, room_generator.cpp:4-202
// Config = the reference's compile-time default, hard_mode (40×40, pruned; caveflyer/tilemap.h:43-45).
// cosf/sinf/fmodf are the process's libm (glibc), what the reference's std::cos/std::sin/std::fmod(float) resolve to.
// The level generator's std::unordered_set<int> is the real container: the iteration order of the largest room decides
// where the goal and the ship spawn (tilemap.cpp:158-169).
#include <algorithm>
#include <cmath>
#include <queue>

#include "pgo_env.h"
#include "pgo_rooms.h"

namespace pgo {
namespace {

const char* const kSpace[13] = {"deep_space_01", "spacegen_01", "milky_way_01", "ez_space_lite_01", "meyespace_v1_01",
                                "eye_nebula_01", "deep_sky_01", "space_nebula_01", "Background-1", "Background-2",
                                "Background-3", "Background-4", "parallax-space-backgound"};  // caveflyer.cpp:58-72

enum Kind { kMeteor = 0, kTarget = 1, kEnemy = 2, kGoal = 3 };

struct Thing {
    int kind = kMeteor;
    bool alive = false;
    V2 pos, vel;
    Box bounds;
};
struct Shot {  // common_systems.h:55-60
    V2 pos, vel;
    float rotation = 0.0f;
    float frame = -1.0f;
};
struct Puff {  // common_components.h:47-52
    V2 pos, dir;
    float rotation = 0.0f;
    float life = 0.0f;
};
struct Hit {
    V2 at;
    bool any;
};

class Caveflyer final : public Env {
   public:
    int W = 40, H = 40;  // tilemap.cpp world_dim: hard 40 (the default), easy 20
    enum Tile : uint8_t { kEmpty = 0, kWall = 1, kMarker = 2 };

    int dump_state(float* out, int cap) const override {
        std::vector<float> v = {a_pos.x, a_pos.y, a_vel.x, a_vel.y, a_rot, painter_.cam_pos.x, painter_.cam_pos.y,
                                static_cast<float>(backdrop_), backdrop_shift_, static_cast<float>(s_next),
                                static_cast<float>(s_count), s_timer, puff_timer, static_cast<float>(puff_on),
                                static_cast<float>(n_things_)};
        for (int i = 0; i < 32; i++) {
            v.push_back(shots[i].pos.x);
            v.push_back(shots[i].pos.y);
            v.push_back(shots[i].frame);
        }
        for (int i = 0; i < 10; i++) {
            v.push_back(puffs[i].pos.x);
            v.push_back(puffs[i].pos.y);
            v.push_back(puffs[i].life);
        }
        for (int id = 0; id < n_things_; id++) {
            if (id == 1) continue;  // the ship
            const Thing& t = things_[id];
            v.push_back(t.alive ? 1.0f : 0.0f);
            v.push_back(static_cast<float>(t.kind));
            v.push_back(t.pos.x);
            v.push_back(t.pos.y);
            v.push_back(t.vel.x);
            v.push_back(t.vel.y);
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
        if (mode_ == kEasy) W = H = 20;  // tilemap.cpp:121-126: world_dim by Distribution_Mode
        if (mode_ == kMemory) W = H = 45;
        tiles_.assign(W * H, 0);
        auto& bank = TextureBank::global();
        auto T = [&](const std::string& n) { return bank.find("assets/" + n + ".png"); };
        for (int i = 0; i < 13; i++) tex_space_[i] = T(std::string("space_backgrounds/") + kSpace[i]);
        tex_wall_ = T("misc_assets/groundA");
        tex_kind_[kMeteor] = T("misc_assets/meteorBrown_big1");
        tex_kind_[kTarget] = T("misc_assets/ufoRed2");
        tex_kind_[kEnemy] = T("misc_assets/enemyShipBlue4");
        tex_kind_[kGoal] = T("misc_assets/ufoGreen2");
        tex_ship_ = T("misc_assets/playerShip1_red");
        tex_laser_ = T("misc_assets/laserBlue02");
        for (int i = 0; i < 5; i++) tex_boom_[i] = T("misc_assets/explosion" + std::to_string(i + 1));
        tex_puff_ = T("misc_assets/towerDefense_tile295");
    }

    uint8_t at(int x, int y) const {  // tilemap.h:78-83
        if (x < 0 || y < 0 || x >= W || y >= H) return kWall;
        return tiles_[y + x * H];
    }

    int spawn(int kind, int cell, Box bounds) {  // tilemap.cpp:34-66
        const int x = cell / H, y = cell % H;
        const int id = ids_.take();
        Thing& t = things_[id];
        t = Thing{};
        t.kind = kind;
        t.alive = true;
        t.pos = {static_cast<float>(x) + 0.5f, static_cast<float>(H - 1 - y) + 0.5f};
        t.bounds = bounds;
        n_things_ = std::max(n_things_, id + 1);
        in_tilemap_.insert(id);
        in_sprite_.insert(id);
        if (kind != kGoal) in_hazard_.insert(id);
        return id;
    }

    static int check_neighbors(const V2& p0, const V2& p1) {  // tilemap.cpp:103-115
        const float neighborhood = 2.0f, epsilon = 0.001f;
        if (std::abs(p0.x - p1.x) <= epsilon && std::abs(p0.y - p1.y) <= neighborhood) return 1;
        if (std::abs(p0.x - p1.x) <= neighborhood && std::abs(p0.y - p1.y) <= epsilon) return 2;
        return 0;
    }

    void new_level() override {  // caveflyer.cpp:442-460
        ids_.refill();
        in_sprite_.clear();
        in_tilemap_.clear();
        in_hazard_.clear();
        in_mob_.clear();
        in_goal_.clear();
        n_things_ = 0;

        // tilemap.cpp:118-278
        std::uniform_real_distribution<float> dist01(0.0f, 1.0f);
        std::mt19937& rng = rng_.eng;
        Rooms rooms;
        rooms.gw = W;
        rooms.gh = H;
        rooms.grid.assign(W * H, 0);
        for (int i = 0; i < W * H; i++) rooms.grid[i] = dist01(rng) < 0.5f ? 1 : 0;
        for (int it = 0; it < 2; it++) rooms.update();
        std::unordered_set<int> best_room;
        rooms.find_best_room(best_room);
        for (int i = 0; i < W * H; i++) tiles_[i] = rooms.grid[i] == 1 ? kWall : kEmpty;
        std::vector<int> free_cells;
        for (int i : best_room) {
            tiles_[i] = kEmpty;
            free_cells.push_back(i);
        }
        int goal_index, agent_index;
        {
            std::uniform_int_distribution<int> d(0, static_cast<int>(free_cells.size()) - 1);
            goal_index = d(rng);
            agent_index = d(rng);
        }
        if (agent_index == goal_index) agent_index = (agent_index + 1) % static_cast<int>(free_cells.size());
        const int goal_cell = free_cells[goal_index], agent_cell = free_cells[agent_index];

        goal_id_ = spawn(kGoal, goal_cell, Box{-0.4f, -0.4f, 0.8f, 0.8f});
        in_goal_.insert(goal_id_);
        const V2 agent_pos{static_cast<float>(static_cast<int>(agent_cell / H)) + 0.5f,
                           static_cast<float>(H - 1 - (agent_cell % H))};  // no +0.5 on y (D13)
        const int agent_id = ids_.take();
        n_things_ = std::max(n_things_, agent_id + 1);
        in_tilemap_.insert(agent_id);
        a_pos = agent_pos;
        a_vel = {0.0f, 0.0f};
        a_rot = 0.0f;
        for (auto& p : puffs) p = Puff{};
        puff_timer = 0.0f;
        puff_on = true;

        std::vector<int> goal_path;
        rooms.find_path(agent_cell, goal_cell, goal_path);
        if (mode_ != kMemory) {  // should_prune (tilemap.cpp:203-215)
            std::unordered_set<int> wide;
            wide.insert(goal_path.begin(), goal_path.end());
            rooms.expand_room(wide, 4);
            for (int i = 0; i < W * H; i++) tiles_[i] = kWall;
            for (int i : wide) tiles_[i] = kEmpty;
        }
        // four more automaton iterations on the generator's grid: never copied back (D13)
        for (int i : goal_path) tiles_[i] = kMarker;
        free_cells.clear();
        for (int i = 0; i < W * H; i++)
            if (tiles_[i] == kEmpty) free_cells.push_back(i);
        const int chunk = static_cast<int>(free_cells.size()) / 80;
        const int num_objects = 3 * chunk;
        std::vector<int> picked(num_objects);
        for (int i = 0; i < num_objects; i++) {
            std::uniform_int_distribution<int> d(0, static_cast<int>(free_cells.size()) - 1);
            int index = d(rng);
            bool repeat;
            do {
                repeat = false;
                for (int j = 0; j < i; j++)
                    if (picked[j] == index) {
                        index = (index + 1) % static_cast<int>(free_cells.size());
                        repeat = true;
                        break;
                    }
            } while (repeat);
            picked[i] = index;
            const int cell = free_cells[index];
            if (i < chunk)
                spawn(kMeteor, cell, Box{-0.25f, -0.25f, 0.5f, 0.5f});
            else if (i < 2 * chunk)
                spawn(kTarget, cell, Box{-0.25f, -0.25f, 0.5f, 0.5f});
            else {  // tilemap.cpp:68-101
                const int id = spawn(kEnemy, cell, Box{-0.4f, -0.4f, 0.8f, 0.8f});
                Thing& t = things_[id];
                float vel_component = (0.1f * dist01(rng) + 0.1f) * (dist01(rng) < 0.5f ? 1.0f : -1.0f);
                const int clash = check_neighbors(t.pos, agent_pos);
                if (clash == 0) {
                    if (dist01(rng) < 0.5f)
                        t.vel.x = vel_component;
                    else
                        t.vel.y = vel_component;
                } else if (clash == 1)
                    t.vel.x = vel_component;
                else
                    t.vel.y = vel_component;
                in_mob_.insert(id);
            }
        }
        for (int i = 0; i < W * H; i++)
            if (tiles_[i] == kMarker) tiles_[i] = kEmpty;

        backdrop_ = rng_.irange(0, 12);
        backdrop_shift_ = rng_.unit();
        draw_list_.clear();
        s_next = 0;  // System_Agent::reset (common_systems.h:83-87); the 32 slots keep their contents
        s_count = 0;
        s_timer = 0.0f;
    }

    template <class Pred>
    Hit collide(Box r, Pred solid) const {  // tilemap.cpp:305-366 (variant B)
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
    static bool is_wall(uint8_t t) { return t == kWall; }
    Box world_box(const Thing& t) const { return Box{t.pos.x + t.bounds.x, t.pos.y + t.bounds.y, t.bounds.w, t.bounds.h}; }

    void destroy(int id) {  // Coordinator::destroy_entity (ecs.cpp:85-90)
        ids_.give_back(id);
        things_[id].alive = false;
        in_sprite_.erase(id);
        in_hazard_.erase(id);
        in_mob_.erase(id);
        in_goal_.erase(id);
        in_tilemap_.erase(id);
    }

    // common_systems.cpp:90-289
    void agent_update(float dt, int action, bool& alive, bool& achieved_goal, int& targets_destroyed) {
        alive = true;
        achieved_goal = false;
        targets_destroyed = 0;
        const float accel = 0.05f, spin_rate = 0.05f, vel_decay = 0.1f, reverse_mul = 0.5f, bullet_time = 0.5f,
                    bullet_speed = 1.0f, explosion_rate = 0.5f;
        float movement_x = (action == 6 || action == 7 || action == 8) - (action == 0 || action == 1 || action == 2);
        float movement_y = (action == 2 || action == 5 || action == 8) - (action == 0 || action == 3 || action == 6);
        const bool fire = action == 9;
        if (movement_y < 0.0f) movement_y *= reverse_mul;
        a_rot += movement_x * spin_rate * dt;
        const V2 dir{std::cos(a_rot), std::sin(a_rot)};
        if (fire) {
            if (s_timer == 0.0f && s_count < 32) {
                s_timer = bullet_time;
                Shot& b = shots[s_next];
                b.rotation = a_rot;
                b.vel = {dir.x * bullet_speed, dir.y * bullet_speed};
                b.pos = a_pos;
                b.frame = 0.0f;
                s_next = (s_next + 1) % 32;
                s_count++;
            } else
                s_timer = std::max(0.0f, s_timer - dt);
        }
        const V2 acc{dir.x * movement_y * accel, dir.y * movement_y * accel};
        a_vel.x += (acc.x - a_vel.x * vel_decay) * dt;
        a_vel.y += (acc.y - a_vel.y * vel_decay) * dt;
        a_pos.x += a_vel.x * dt;
        a_pos.y += a_vel.y * dt;
        const Box bounds{-0.4f, -0.4f, 0.8f, 0.8f};
        Box wc{a_pos.x + bounds.x, a_pos.y + bounds.y, bounds.w, bounds.h};
        const Hit hit = collide(wc, is_wall);
        const V2 delta{hit.at.x - wc.x, hit.at.y - wc.y};
        a_pos.x = hit.at.x - bounds.x;
        a_pos.y = hit.at.y - bounds.y;
        wc.x = a_pos.x + bounds.x;
        wc.y = a_pos.y + bounds.y;
        if (delta.x != 0.0f) a_vel.x = 0.0f;
        if (delta.y != 0.0f) a_vel.y = 0.0f;
        for (int h : in_hazard_)
            if (boxes_touch(wc, world_box(things_[h]))) {
                alive = false;
                break;
            }
        for (int g : in_goal_)
            if (boxes_touch(wc, world_box(things_[g]))) {
                achieved_goal = true;
                break;
            }
        painter_.cam_pos.x = a_pos.x * kUnitPx;
        painter_.cam_pos.y = a_pos.y * kUnitPx;

        for (int i = 0; i < s_count; i++) {
            const int k = (32 + s_next - 1 - i) % 32;
            Shot& b = shots[k];
            if (b.frame == -1.0f) continue;
            if (b.frame == 0.0f) {
                const Box sb{b.pos.x - 0.01f, b.pos.y - 0.01f, 0.02f, 0.02f};
                if (collide(sb, is_wall).any) {
                    b.vel = {0.0f, 0.0f};
                    b.frame = 1.0f;
                }
                std::vector<int> to_destroy;
                for (int h : in_hazard_) {
                    const Thing& t = things_[h];
                    if (boxes_touch(sb, world_box(t))) {
                        b.vel = {0.0f, 0.0f};
                        b.frame = 1.0f;
                        if (t.kind == kTarget) {
                            to_destroy.push_back(h);
                            targets_destroyed++;
                        }
                        break;
                    }
                }
                for (int h : to_destroy) destroy(h);
            }
            b.pos.x += b.vel.x * dt;
            b.pos.y += b.vel.y * dt;
            if (b.frame >= 5.0f) {
                s_count--;
                b.frame = -1.0f;
            } else if (b.frame >= 1.0f)
                b.frame += explosion_rate * dt;
        }
        puff_on = movement_y > 0.0f;
    }

    void mobs_update(float dt) {  // common_systems.cpp:50-75
        for (int id : in_mob_) {
            Thing& t = things_[id];
            t.pos.x += t.vel.x * dt;
            t.pos.y += t.vel.y * dt;
            if (collide(world_box(t), is_wall).any) t.vel = {-t.vel.x, -t.vel.y};
        }
    }

    void puffs_update(float dt) {  // common_systems.cpp:333-372
        const float lifespan = 3.0f, spawn_time = 0.3f;
        const V2 offset{0.0f, 0.3f};
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
            p.rotation = a_rot + M_PI * 0.5f;
            float c = std::cos(p.rotation);
            float s = std::sin(p.rotation);
            p.dir = {-std::cos(a_rot), -std::sin(a_rot)};
            V2 rotated{c * offset.x - s * offset.y, s * offset.x + c * offset.y};
            p.pos.x = a_pos.x + rotated.x;
            p.pos.y = a_pos.y + rotated.y;
        }
    }

    void sprites_update() {  // common_systems.cpp:7-24
        draw_list_.resize(in_sprite_.size());
        int k = 0;
        for (int id : in_sprite_) draw_list_[k++] = {1.0f, id};
        std::sort(draw_list_.begin(), draw_list_.end(),
                  [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first < b.first; });
    }

    void advance(int action) override {  // caveflyer.cpp:316-337
        const float dt = 1.0f / 4;
        for (int ss = 0; ss < 4; ss++) {
            bool alive, achieved_goal;
            int targets_destroyed;
            agent_update(dt, action, alive, achieved_goal, targets_destroyed);
            mobs_update(dt);
            puffs_update(dt);
            sprites_update();
            reward = achieved_goal * 10.0f + targets_destroyed * 3.0f;
            terminated = !alive || achieved_goal;
            truncated = false;
            if (terminated) break;
        }
    }

    void paint() override {  // caveflyer.cpp:413-440
        painter_.target->clear_black();
        painter_.cam_scale = 0.5f * static_cast<float>(view_w_) / static_cast<float>(kObsW);
        painter_.cam_size = {static_cast<float>(view_w_), static_cast<float>(view_h_)};
        const Texture* bg = tex_space_[backdrop_];
        const float aspect = static_cast<float>(bg->w) / static_cast<float>(bg->h);
        const float extra = aspect - 1.0f;
        painter_.draw(bg, V2{-backdrop_shift_ * extra, 0.0f}, 64.0f * kUnitPx / bg->h);
        {  // tilemap.cpp:280-303
            const V2& cp = painter_.cam_pos;
            const V2& cs = painter_.cam_size;
            const float sc = painter_.cam_scale;
            Box view{(cp.x - cs.x * 0.5f / sc) * kPxUnit, (cp.y - cs.y * 0.5f / sc) * kPxUnit, cs.x * kPxUnit / sc,
                     cs.y * kPxUnit / sc};
            int x0 = std::floor(view.x), y0 = std::floor(view.y);
            int x1 = std::ceil(view.x + view.w), y1 = std::ceil(view.y + view.h);
            for (int y = y0; y <= y1; y++)
                for (int x = x0; x <= x1; x++) {
                    if (at(x, H - 1 - y) == kEmpty) continue;
                    painter_.draw(tex_wall_, V2{x * kUnitPx, y * kUnitPx}, kUnitPx / tex_wall_->w);
                }
        }
        {  // System_Particles::render (common_systems.cpp:374-397)
            const float base_alpha = 0.5f, base_scale = 1.0f, lifespan = 3.0f;
            for (int i = 0; i < 10; i++) {
                const Puff& p = puffs[i];
                if (p.life <= 0.0f) continue;
                float life_ratio = (lifespan - p.life) / lifespan;
                float alpha = base_alpha * (1.0f - life_ratio);
                float scale = base_scale * (0.4f * life_ratio + 0.6f);
                float shift = life_ratio * 2.0f;
                float size = scale * kUnitPx / tex_puff_->w;
                painter_.draw_rotated(tex_puff_,
                                      V2{(p.pos.x + p.dir.x * shift) * kUnitPx - size * tex_puff_->w * 0.5f,
                                         (p.pos.y + p.dir.y * shift) * kUnitPx - size * tex_puff_->h * 0.5f},
                                      p.rotation, size, alpha);
            }
        }
        for (auto& zi : draw_list_) {  // positive-z sprites (common_systems.cpp:26-48)
            const Thing& t = things_[zi.second];
            const Texture* tex = tex_kind_[t.kind];
            float scale = 1.0f * 0.8f;
            painter_.draw(tex, V2{(t.pos.x + -0.4f) * kUnitPx, (t.pos.y + -0.4f) * kUnitPx}, scale * kUnitPx / tex->w,
                          1.0f, false);
        }
        // System_Agent::render (common_systems.cpp:291-327)
        for (int i = 0; i < s_count; i++) {
            const int k = (32 + s_next - 1 - i) % 32;
            const Shot& b = shots[k];
            if (b.frame == -1.0f) continue;
            const Texture* t = (b.frame == 0.0f) ? tex_laser_ : tex_boom_[static_cast<int>(b.frame - 1.0f)];
            const float size = 0.1f;
            painter_.draw_rotated(t, V2{b.pos.x * kUnitPx - size * t->w * 0.5f, b.pos.y * kUnitPx - size * t->h * 0.5f},
                                  b.rotation + M_PI * 0.5f, size);
        }
        {
            const float size = 0.15f;
            painter_.draw_rotated(tex_ship_,
                                  V2{a_pos.x * kUnitPx - size * tex_ship_->w * 0.5f,
                                     a_pos.y * kUnitPx - size * tex_ship_->h * 0.5f},
                                  a_rot + M_PI * 0.5f, size);
        }
    }

   private:
    std::vector<uint8_t> tiles_;
    std::vector<Thing> things_ = std::vector<Thing>(IdPool::kMax);
    int n_things_ = 0;
    int goal_id_ = 0;
    IdPool ids_;
    IdSet in_sprite_, in_tilemap_, in_hazard_, in_mob_, in_goal_;
    std::vector<std::pair<float, int>> draw_list_;
    V2 a_pos, a_vel;
    float a_rot = 0.0f;
    Shot shots[32];
    int s_next = 0, s_count = 0;
    float s_timer = 0.0f;
    Puff puffs[10];
    float puff_timer = 0.0f;
    bool puff_on = true;
    int backdrop_ = 0;
    float backdrop_shift_ = 0.0f;
    const Texture* tex_space_[13] = {};
    const Texture* tex_wall_ = nullptr;
    const Texture* tex_kind_[4] = {};
    const Texture* tex_ship_ = nullptr;
    const Texture* tex_laser_ = nullptr;
    const Texture* tex_boom_[5] = {};
    const Texture* tex_puff_ = nullptr;
};

}  // namespace

Env* new_caveflyer() { return new Caveflyer(); }

}  // namespace pgo
