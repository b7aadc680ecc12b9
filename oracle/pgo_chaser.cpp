// ORACLE — TEST INFRASTRUCTURE ONLY (see pgo_common.h header).
//
// chaser: CPU restatement of SURVEY.md row G5.
//   step   games/chaser/chaser.cpp:282-334, common_systems.cpp:305-444 (agent), :117-295 (enemies), :66-106 (points),
//          :8-39 (sprite list)
//   render games/chaser/chaser.cpp:390-416, tilemap.cpp:245-267, common_systems.cpp:41-63, :446-460
//   reset  games/chaser/chaser.cpp:418-443, tilemap.cpp:80-243, maze_generator.cpp:47-130
// Config = the reference's compile-time default, easy_mode (11×11, 3 enemies; chaser/tilemap.h:39-41).
// Quirk kept on purpose (D21): common_systems.cpp calls the unqualified `abs` on floats, and in that translation unit
// only glibc's `int abs(int)` is visible at global scope (<cmath> reaches <stdlib.h> through #include_next, which
// skips the libstdc++ wrapper that would export std::abs's float overloads; cenv.h, which includes <stdlib.h>
// properly, is not included there).  Every such argument is therefore truncated to int first: the "close to the cell
// centre" tests are always true and the Manhattan distances are sums of truncated integers.  The Appendix C traces
// (built from the unmodified sources with g++ 11) pin this: with float abs they do not reproduce.
#include <algorithm>
#include <cmath>

#include "pgo_env.h"
#include "pgo_kruskal.h"

namespace pgo {
namespace {

const char* const kFloors[9] = {"floortiles",          "backgrounddetailed1", "backgrounddetailed2",
                                "backgrounddetailed3", "backgrounddetailed4", "backgrounddetailed5",
                                "backgrounddetailed6", "backgrounddetailed7", "backgrounddetailed8"};  // chaser.cpp:57-67

enum Kind { kOrb = 0, kPoint = 1, kEgg = 2 };

struct Thing {
    int kind = kPoint;
    bool alive = false;
    V2 pos, vel;
    float hatch_timer = 0.0f;
    int tex = 0;  // eggs/enemies: 0 egg, 1..3 flying frames, 4 walking
};


int sign(float x) {  // helpers.h:31-36
    if (x == 0.0f) return 0;
    return (x > 0.0f) * 2 - 1;
}

class Chaser final : public Env {
   public:
    // What the unqualified `abs(<float>)` of common_systems.cpp:165-166,206,346-420 computes (D21).  Default: glibc's
    // `int abs(int)` — the argument is truncated first — as g++/libstdc++ resolves it when nothing in the translation
    // unit includes <stdlib.h> / <math.h> (the result is a small integer, exact as a float).  With game_flags bit 0
    // (PGV_CHASER_FLOAT_ABS): `float std::abs(float)`, what the same sources give once any header brings libstdc++'s
    // <stdlib.h> wrapper in (`using std::abs;`), and what libc++ / MSVC give always.  Both are "the reference";
    // tests/golden/appendix_c.json holds traces of the unmodified sources for each.
    float qabs(float v) const {
        return (flags_ & 1u) ? std::fabs(v) : static_cast<float>(std::abs(static_cast<int>(v)));
    }

    int W = 11, H = 11, total_enemies_ = 3, extra_orb_sign_ = 0;  // tilemap.cpp:85-99, easy_mode is the default
    enum Tile : uint8_t { kEmpty = 0, kWall = 1, kMarker = 2 };

    int dump_state(float* out, int cap) const override {
        std::vector<float> v = {a_pos.x, a_pos.y, a_vel.x, a_vel.y, a_next.x, a_next.y, input_timer, anim_timer,
                                static_cast<float>(anim_index), eat_timer, static_cast<float>(backdrop_), backdrop_shift_,
                                static_cast<float>(n_things_)};
        for (int id = 0; id < n_things_; id++) {
            const Thing& t = things_[id];
            v.push_back(t.alive ? 1.0f : 0.0f);
            v.push_back(static_cast<float>(t.kind));
            v.push_back(t.pos.x);
            v.push_back(t.pos.y);
            v.push_back(t.vel.x);
            v.push_back(t.vel.y);
            v.push_back(t.hatch_timer);
            v.push_back(static_cast<float>(t.tex));
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
        if (mode_ == kHard) {  // tilemap.cpp:85-94
            W = H = 13;
            extra_orb_sign_ = -1;
        } else if (mode_ == kExtreme) {
            W = H = 19;
            total_enemies_ = 5;
            extra_orb_sign_ = 1;
        }
        tiles_.assign(W * H, 0);
        auto& bank = TextureBank::global();
        auto T = [&](const std::string& n) { return bank.find("assets/" + n + ".png"); };
        for (int i = 0; i < 9; i++) tex_floor_[i] = T(std::string("topdown_backgrounds/") + kFloors[i]);
        tex_wall_ = T("misc_assets/tileStone_slope");
        tex_orb_ = T("misc_assets/yellowCrystal");
        tex_point_ = T("custom/chaser_point");
        tex_enemy_[0] = T("misc_assets/enemySpikey_1b");
        tex_enemy_[1] = T("misc_assets/enemyFlying_1");
        tex_enemy_[2] = T("misc_assets/enemyFlying_2");
        tex_enemy_[3] = T("misc_assets/enemyFlying_3");
        tex_enemy_[4] = T("misc_assets/enemyWalking_1b");
        tex_agent_ = T("misc_assets/enemyFloating_1b");
    }

    int at(int x, int y) const {  // tilemap.h:79-84: out of bounds is its own id (-1), neither empty nor wall
        if (x < 0 || y < 0 || x >= W || y >= H) return -1;
        return tiles_[y + x * H];
    }

    int spawn(int kind, int cell) {  // tilemap.cpp:30-78
        const int x = cell / H, y = cell % H;
        const int id = ids_.take();
        Thing& t = things_[id];
        t = Thing{};
        t.kind = kind;
        t.alive = true;
        t.pos = {static_cast<float>(x) + 0.5f, static_cast<float>(H - 1 - y) + 0.5f};
        n_things_ = std::max(n_things_, id + 1);
        in_sprite_.insert(id);
        if (kind == kEgg)
            in_mob_.insert(id);
        else
            in_point_.insert(id);
        return id;
    }

    void new_level() override {  // chaser.cpp:418-443
        ids_.refill();
        in_sprite_.clear();
        in_point_.clear();
        in_mob_.clear();
        n_things_ = 0;

        // tilemap.cpp:80-243
        const int total_enemies = total_enemies_, extra_orb_sign = extra_orb_sign_;
        std::fill(tiles_.begin(), tiles_.end(), static_cast<uint8_t>(kEmpty));
        std::vector<int> grid;
        carve_merged(W, grid, rng_);  // maze_generator.cpp:47-130 (pgo_kruskal.h)
        const int extra_quad = rng_.irange(0, 3);
        std::vector<std::vector<int>> quadrants(4);
        int orbs_for[4];
        for (int i = 0; i < 4; i++) orbs_for[i] = 1 + (i == extra_quad ? extra_orb_sign : 0);
        for (int x = 0; x < W; x++)
            for (int y = 0; y < H; y++) {
                const int obj = grid[(y + 1) + (H + 2) * (x + 1)];
                tiles_[y + x * H] = obj == 1 ? kWall : kEmpty;
                if (obj == 0) quadrants[(x >= W / 2) * 2 + (y >= H / 2)].push_back(y + x * H);
            }
        for (int i = 0; i < 4; i++) {
            const std::vector<int>& quadrant = quadrants[i];
            std::uniform_int_distribution<int> pos_dist(0, static_cast<int>(quadrant.size()) - 1);
            std::unordered_set<int> selected;
            for (int j = 0; j < orbs_for[i]; j++) {
                int pos = pos_dist(rng_.eng);
                while (std::find(selected.begin(), selected.end(), pos) != selected.end())
                    pos = (pos + 1) % static_cast<int>(quadrant.size());
                selected.insert(pos);
            }
            for (int j : selected) {
                spawn(kOrb, quadrant[j]);
                tiles_[quadrant[j]] = kMarker;
            }
        }
        free_cells_.clear();
        for (int i = 0; i < W * H; i++)
            if (tiles_[i] == kEmpty) free_cells_.push_back(i);
        int agent_x, agent_y;
        {
            std::uniform_int_distribution<int> pos_dist(0, static_cast<int>(free_cells_.size()) - 1);
            std::unordered_set<int> selected;
            for (int j = 0; j < total_enemies + 1; j++) {
                int pos = pos_dist(rng_.eng);
                while (std::find(selected.begin(), selected.end(), pos) != selected.end())
                    pos = (pos + 1) % static_cast<int>(free_cells_.size());
                selected.insert(pos);
            }
            auto it = selected.begin();
            const int start = free_cells_[*it];
            agent_x = start / H;
            agent_y = start % H;
            tiles_[start] = kMarker;
            for (int i = 0; i < total_enemies; i++) {
                ++it;
                const int cell = free_cells_[*it];
                spawn(kEgg, cell);
                tiles_[cell] = kMarker;
            }
        }
        free_cells_.clear();
        for (int i = 0; i < W * H; i++)
            if (tiles_[i] == kEmpty) free_cells_.push_back(i);
        for (int cell : free_cells_) spawn(kPoint, cell);
        for (int i = 0; i < W * H; i++)
            if (tiles_[i] == kMarker) tiles_[i] = kEmpty;
        ids_.take();  // the agent entity
        a_pos = {static_cast<float>(agent_x) + 0.5f, static_cast<float>(H - 1 - agent_y) + 0.5f};
        a_vel = {0.0f, 0.0f};
        a_next = {0.0f, 0.0f};

        backdrop_ = rng_.irange(0, 8);
        backdrop_shift_ = rng_.unit();
        input_timer = 0.0f;  // System_Agent::reset
        anim_timer = 0.0f;   // System_Mob_AI::reset
        anim_index = 0;
        eat_timer = 0.0f;
        draw_list_.clear();
        painter_.cam_pos.x = W * 0.5f * kUnitPx;
        painter_.cam_pos.y = H * 0.5f * kUnitPx;
    }

    void agent_update(float dt, int action) {  // common_systems.cpp:305-444
        const float speed = 0.2f;
        const float input_reset_time = 1.0f / speed * 0.5f;
        float movement_x = (action == 7) - (action == 1);
        float movement_y = (action == 3) - (action == 5);
        if (movement_x != 0.0f && movement_y != 0.0f) movement_y = 0.0f;
        if (movement_x != 0.0f || movement_y != 0.0f) {
            a_next = {movement_x, movement_y};
            input_timer = 0.0f;
        }
        V2& p = a_pos;
        auto frac_x = [&] { return qabs(p.x - (static_cast<int>(p.x) + 0.5f)); };
        auto frac_y = [&] { return qabs(p.y - (static_cast<int>(p.y) + 0.5f)); };
        if (a_next.x > 0.0f) {
            if (frac_y() <= speed * dt && at(static_cast<int>(p.x) + 1, H - 1 - static_cast<int>(p.y)) == kEmpty) {
                p.y = static_cast<int>(p.y) + 0.5f;
                a_vel = a_next;
            }
        } else if (a_next.x < 0.0f) {
            if (frac_y() <= speed * dt && at(static_cast<int>(p.x) - 1, H - 1 - static_cast<int>(p.y)) == kEmpty) {
                p.y = static_cast<int>(p.y) + 0.5f;
                a_vel = a_next;
            }
        }
        if (a_next.y > 0.0f) {
            if (frac_x() <= speed * dt && at(static_cast<int>(p.x), H - 1 - (static_cast<int>(p.y) + 1)) == kEmpty) {
                p.x = static_cast<int>(p.x) + 0.5f;
                a_vel = a_next;
            }
        } else if (a_next.y < 0.0f) {
            if (frac_x() <= speed * dt && at(static_cast<int>(p.x), H - 1 - (static_cast<int>(p.y) - 1)) == kEmpty) {
                p.x = static_cast<int>(p.x) + 0.5f;
                a_vel = a_next;
            }
        }
        if (a_vel.x < 0.0f) {
            if (frac_x() <= speed * dt && at(static_cast<int>(p.x) - 1, H - 1 - static_cast<int>(p.y)) != kEmpty) {
                p.x = static_cast<int>(p.x) + 0.5f;
                a_vel.x = 0.0f;
            }
        } else if (a_vel.x > 0.0f) {
            if (frac_x() <= speed * dt && at(static_cast<int>(p.x) + 1, H - 1 - static_cast<int>(p.y)) != kEmpty) {
                p.x = static_cast<int>(p.x) + 0.5f;
                a_vel.x = 0.0f;
            }
        }
        if (a_vel.y < 0.0f) {
            if (frac_y() <= speed * dt && at(static_cast<int>(p.x), H - 1 - (static_cast<int>(p.y) - 1)) != kEmpty) {
                p.y = static_cast<int>(p.y) + 0.5f;
                a_vel.y = 0.0f;
            }
        } else if (a_vel.y > 0.0f) {
            if (frac_y() <= speed * dt && at(static_cast<int>(p.x), H - 1 - (static_cast<int>(p.y) + 1)) != kEmpty) {
                p.y = static_cast<int>(p.y) + 0.5f;
                a_vel.y = 0.0f;
            }
        }
        p.x += a_vel.x * speed * dt;
        p.y += a_vel.y * speed * dt;
        if (input_timer >= input_reset_time)
            a_next = {0.0f, 0.0f};
        else
            input_timer += dt;
    }

    bool mobs_update(float dt) {  // common_systems.cpp:117-295
        const float hatch_time = 50.0f, anim_time = 1.0f, speed_low = 0.125f, speed_high = 0.25f;
        const V2 directions[4] = {{-1.0f, 0.0f}, {1.0f, 0.0f}, {0.0f, -1.0f}, {0.0f, 1.0f}};
        bool player_hit = false;
        const Box agent_rect{-0.5f + a_pos.x, -0.5f + a_pos.y, 1.0f, 1.0f};
        std::uniform_real_distribution<float> dist01(0.0f, 1.0f);
        for (int e : in_mob_) {
            Thing& t = things_[e];
            if (t.hatch_timer >= hatch_time) {
                float speed;
                if (eat_timer == 0.0f) {
                    t.tex = anim_index < 3 ? 1 + anim_index : 1 + (5 - anim_index);
                    speed = speed_high;
                } else {
                    t.tex = 4;
                    speed = speed_low;
                }
                const bool at_junction = std::max(qabs(t.pos.x - (static_cast<int>(t.pos.x) + 0.5f)),
                                                  qabs(t.pos.y - (static_cast<int>(t.pos.y) + 0.5f))) < speed * dt;
                if ((t.vel.x == 0.0f && t.vel.y == 0.0f) || at_junction) {
                    bool possible[4];
                    int k = 0, n_possible = 0;
                    for (int dx = -1; dx <= 1; dx += 2) {
                        const int id = at(static_cast<int>(t.pos.x) + dx, H - 1 - static_cast<int>(t.pos.y));
                        possible[k] = (id == kEmpty && dx != -sign(t.vel.x));
                        n_possible += possible[k] ? 1 : 0;
                        k++;
                    }
                    for (int dy = -1; dy <= 1; dy += 2) {
                        const int id = at(static_cast<int>(t.pos.x), H - 1 - (static_cast<int>(t.pos.y) + dy));
                        possible[k] = (id == kEmpty && dy != -sign(t.vel.y));
                        n_possible += possible[k] ? 1 : 0;
                        k++;
                    }
                    const bool be_aggressive = dist01(rng_.eng) < 0.5f;
                    int select = 0;
                    if (be_aggressive) {
                        float min_dist = 999999.0f;
                        for (int i = 0; i < 4; i++)
                            if (possible[i]) {
                                float d = qabs(t.pos.x + directions[i].x - a_pos.x) +
                                          qabs(t.pos.y + directions[i].y - a_pos.y);
                                if (eat_timer > 0.0f) d = -d;
                                if (d < min_dist) {
                                    min_dist = d;
                                    select = i;
                                }
                            }
                    } else if (n_possible > 0) {
                        const int cusp = rng_.irange(0, n_possible - 1);
                        int sum = 0;
                        for (int i = 0; i < 4; i++) {
                            sum += possible[i];
                            if (sum > cusp) {
                                select = i;
                                break;
                            }
                        }
                    }
                    t.vel.x = directions[select].x * speed;
                    t.vel.y = directions[select].y * speed;
                    if (directions[select].x == 0.0f) t.pos.x = static_cast<int>(t.pos.x) + 0.5f;
                    if (directions[select].y == 0.0f) t.pos.y = static_cast<int>(t.pos.y) + 0.5f;
                }
                t.pos.x += t.vel.x * dt;
                t.pos.y += t.vel.y * dt;
                const Box rect{-0.5f + t.pos.x, -0.5f + t.pos.y, 1.0f, 1.0f};
                if (boxes_touch(agent_rect, rect)) {
                    if (eat_timer == 0.0f)
                        player_hit = true;
                    else {  // back to an egg somewhere (no world-y flip here, D16)
                        t.hatch_timer = 0.0f;
                        const int cell = free_cells_[rng_.irange(0, static_cast<int>(free_cells_.size()) - 1)];
                        t.pos.x = cell / H + 0.5f;
                        t.pos.y = cell % H + 0.5f;
                        t.tex = 0;
                    }
                }
            } else
                t.hatch_timer += dt;
        }
        if (anim_timer < anim_time)
            anim_timer += dt;
        else {
            anim_timer -= anim_time;
            anim_index = (anim_index + 1) % 6;
        }
        if (eat_timer > 0.0f) eat_timer = std::max(0.0f, eat_timer - dt);
        return player_hit;
    }

    void points_update(int& delta, int& available) {  // common_systems.cpp:66-106
        const Box agent_rect{-0.5f + a_pos.x, -0.5f + a_pos.y, 1.0f, 1.0f};
        available = 0;
        delta = 0;
        std::vector<int> gone;
        for (int e : in_point_) {
            const Thing& t = things_[e];
            const Box rect = t.kind == kOrb ? Box{-0.5f + t.pos.x, -0.5f + t.pos.y, 1.0f, 1.0f}
                                            : Box{-0.3f + t.pos.x, -0.3f + t.pos.y, 0.6f, 0.6f};
            if (boxes_touch(agent_rect, rect)) {
                if (t.kind == kOrb) eat_timer = 75.0f;
                delta++;
                gone.push_back(e);
            } else
                available++;
        }
        for (int e : gone) {  // Coordinator::destroy_entity
            ids_.give_back(e);
            things_[e].alive = false;
            in_sprite_.erase(e);
            in_point_.erase(e);
        }
    }

    void sprites_update() {  // common_systems.cpp:8-39 (no entity carries an animation component)
        draw_list_.resize(in_sprite_.size());
        int k = 0;
        for (int id : in_sprite_) draw_list_[k++] = {0.0f, id};
        std::sort(draw_list_.begin(), draw_list_.end(),
                  [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first < b.first; });
    }

    void advance(int action) override {  // chaser.cpp:297-314
        const float dt = 1.0f / 4;
        for (int ss = 0; ss < 4; ss++) {
            agent_update(dt, action);
            const bool dead = mobs_update(dt);
            int delta, available;
            points_update(delta, available);
            sprites_update();
            reward = delta * 0.04f + (available == 0) * 10.0f;
            terminated = dead || (available == 0);
            truncated = false;
            if (terminated) break;
        }
    }

    void paint() override {  // chaser.cpp:390-416
        painter_.target->clear_black();
        painter_.cam_scale = static_cast<float>(view_w_) * kPxUnit / static_cast<float>(W);
        painter_.cam_size = {static_cast<float>(view_w_), static_cast<float>(view_h_)};
        const Texture* bg = tex_floor_[backdrop_];
        const float aspect = static_cast<float>(bg->w) / static_cast<float>(bg->h);
        const float extra = aspect - 1.0f;
        painter_.draw(bg, V2{-backdrop_shift_ * extra, 0.0f}, 64.0f * kUnitPx / bg->h);
        {  // tilemap.cpp:245-267
            const V2& cp = painter_.cam_pos;
            const V2& cs = painter_.cam_size;
            const float sc = painter_.cam_scale;
            Box view{(cp.x - cs.x * 0.5f / sc) * kPxUnit, (cp.y - cs.y * 0.5f / sc) * kPxUnit, cs.x * kPxUnit / sc,
                     cs.y * kPxUnit / sc};
            int x0 = std::floor(view.x), y0 = std::floor(view.y);
            int x1 = std::ceil(view.x + view.w), y1 = std::ceil(view.y + view.h);
            for (int y = y0; y <= y1; y++)
                for (int x = x0; x <= x1; x++) {
                    if (at(x, H - 1 - y) != kWall) continue;  // empty and out-of-bounds are skipped
                    painter_.draw(tex_wall_, V2{x * kUnitPx, y * kUnitPx}, kUnitPx / tex_wall_->w);
                }
        }
        for (auto& zi : draw_list_) {  // every sprite has z = 0: the positive pass (common_systems.cpp:41-63)
            const Thing& t = things_[zi.second];
            const Texture* tex = t.kind == kOrb ? tex_orb_ : t.kind == kPoint ? tex_point_ : tex_enemy_[t.tex];
            float scale = 1.0f * 1.0f;
            painter_.draw(tex, V2{(t.pos.x + -0.5f) * kUnitPx, (t.pos.y + -0.5f) * kUnitPx}, scale * kUnitPx / tex->w, 1.0f,
                          false);
        }
        painter_.draw(tex_agent_, V2{(a_pos.x + -0.5f) * kUnitPx, (a_pos.y + -0.5f) * kUnitPx},
                      kUnitPx / tex_agent_->w * 1.0f, 1.0f, false);
    }

   private:
    std::vector<uint8_t> tiles_;
    std::vector<Thing> things_ = std::vector<Thing>(IdPool::kMax);
    std::vector<int> free_cells_;
    int n_things_ = 0;
    IdPool ids_;
    IdSet in_sprite_, in_point_, in_mob_;
    std::vector<std::pair<float, int>> draw_list_;
    V2 a_pos, a_vel, a_next;
    float input_timer = 0.0f, anim_timer = 0.0f, eat_timer = 0.0f;
    int anim_index = 0;
    int backdrop_ = 0;
    float backdrop_shift_ = 0.0f;
    const Texture* tex_floor_[9] = {};
    const Texture* tex_wall_ = nullptr;
    const Texture* tex_orb_ = nullptr;
    const Texture* tex_point_ = nullptr;
    const Texture* tex_enemy_[5] = {};
    const Texture* tex_agent_ = nullptr;
};

}  // namespace

Env* new_chaser() { return new Chaser(); }

}  // namespace pgo
