// ORACLE — TEST INFRASTRUCTURE ONLY (see pgo_common.h header).
//
// maze: CPU restatement of SURVEY.md row G2.
//   step   games/maze/maze.cpp:279-330, common_systems.cpp:69-136
//   render games/maze/maze.cpp:386-414, tilemap.cpp:111-133, common_systems.cpp:41-63,138-150
//   reset  games/maze/maze.cpp:416-438, tilemap.cpp:31-109, maze_generator.cpp:55-139,183-195
// Config: mode_ picks the reference's `Distribution_Mode` (tilemap.cpp:31-48); hard_mode (25×25 world, all visible,
// fixed camera) is its compile-time default.
#include <algorithm>

#include "pgo_env.h"
#include "pgo_kruskal.h"

namespace pgo {
namespace {

enum Cell : uint8_t { kOpen = 0, kWall = 1 };

const char* const kFloors[9] = {  // maze.cpp:62-72
    "topdown_backgrounds/floortiles.png",          "topdown_backgrounds/backgrounddetailed1.png",
    "topdown_backgrounds/backgrounddetailed2.png", "topdown_backgrounds/backgrounddetailed3.png",
    "topdown_backgrounds/backgrounddetailed4.png", "topdown_backgrounds/backgrounddetailed5.png",
    "topdown_backgrounds/backgrounddetailed6.png", "topdown_backgrounds/backgrounddetailed7.png",
    "topdown_backgrounds/backgrounddetailed8.png"};

class Maze final : public Env {
   public:
    static constexpr int kTimeout = 500, kGoalMark = 2;
    int W = 25, H = 25, visible_ = 25;  // tilemap.cpp:31-48
    bool centred_ = false;

    int dump_state(float* out, int cap) const override {
        float v[10] = {a_pos.x, a_pos.y, static_cast<float>(a_forward), goal_pos.x, goal_pos.y,
                       static_cast<float>(steps_), static_cast<float>(floor_), floor_shift_, painter_.cam_pos.x,
                       painter_.cam_pos.y};
        int n = std::min(cap, 10);
        std::memcpy(out, v, n * sizeof(float));
        return 10;
    }
    int dump_tiles(uint8_t* out, int cap) const override {
        int n = std::min<int>(cap, W * H);
        std::memcpy(out, tiles_.data(), n);
        return n;
    }

   protected:
    void on_make() override {
        if (mode_ == kMemory) {
            W = H = 31;
            visible_ = 8;
            centred_ = true;
        } else if (mode_ == kEasy) {
            W = H = visible_ = 15;
        }
        tiles_.assign(W * H, kWall);
        auto& bank = TextureBank::global();
        tex_wall_ = bank.find("assets/kenney/Ground/Sand/sandCenter.png");
        tex_cheese_ = bank.find("assets/misc_assets/cheese.png");
        tex_mouse_ = bank.find("assets/kenney/Enemies/mouse_move.png");
        for (int i = 0; i < 9; i++) tex_floor_[i] = bank.find(std::string("assets/") + kFloors[i]);
    }

    uint8_t at(int x, int y) const {
        if (x < 0 || y < 0 || x >= W || y >= H) return kWall;
        return tiles_[y + x * H];
    }

    void new_level() override {  // maze.cpp:416-438
        ids_.refill();
        in_sprite_.clear();
        in_goal_.clear();
        in_agent_.clear();
        in_tilemap_.clear();

        // tilemap.cpp:31-109
        std::fill(tiles_.begin(), tiles_.end(), kWall);
        const int dim = rng_.irange(0, (W - 1) / 2 - 1) * 2 + 3;
        const int margin = (W - dim) / 2;
        Carver carver;
        carver.carve(dim, dim, rng_);
        carver.drop(kGoalMark, rng_);
        int gx = 0, gy = 0;
        for (int i = 0; i < dim; i++)
            for (int j = 0; j < dim; j++) {
                int t = carver.get(i + Carver::kPad, j + Carver::kPad);
                tiles_[(j + margin) + (i + margin) * H] = (t == Carver::kBrick) ? kWall : kOpen;
                if (t == kGoalMark) {
                    gx = i + margin;
                    gy = j + margin;
                }
            }
        int goal_id = ids_.take();
        in_tilemap_.insert(goal_id);
        in_sprite_.insert(goal_id);
        in_goal_.insert(goal_id);
        goal_pos = {static_cast<float>(gx) + 0.5f, static_cast<float>(H - 1 - gy) + 0.5f};

        int agent_id = ids_.take();
        in_tilemap_.insert(agent_id);
        in_agent_.insert(agent_id);
        a_pos = {static_cast<float>(margin) + 0.5f, static_cast<float>(H - 1 - margin) + 0.5f};
        a_forward = true;

        steps_ = 0;
        floor_ = rng_.irange(0, 8);
        floor_shift_ = rng_.unit();
        goal_listed_ = false;  // sprite_render->clear_render()  (D2)
        painter_.cam_pos = {W * 0.5f * kUnitPx, H * 0.5f * kUnitPx};
    }

    void advance(int action) override {  // maze.cpp:293-310, common_systems.cpp:69-136
        int mx = action / 3 - 1;
        int my = mx ? 0 : -(action % 3 - 1);
        if (mx) {
            if (at(static_cast<int>(a_pos.x + mx), H - 1 - static_cast<int>(a_pos.y)) == kOpen)
                a_pos.x = static_cast<int>(a_pos.x + mx) + 0.5f;
        } else if (my) {
            if (at(static_cast<int>(a_pos.x), H - 1 - static_cast<int>(a_pos.y + my)) == kOpen)
                a_pos.y = static_cast<int>(a_pos.y + my) + 0.5f;
        }
        Box body{a_pos.x - 0.5f, a_pos.y - 0.5f, 1.0f, 1.0f};
        Box goal{goal_pos.x - 0.5f, goal_pos.y - 0.5f, 1.0f, 1.0f};
        bool reached = boxes_touch(body, goal);
        if (centred_) painter_.cam_pos = {a_pos.x * kUnitPx, a_pos.y * kUnitPx};  // common_systems.cpp:119-123
        if (mx > 0.0f)
            a_forward = true;
        else if (mx < 0.0f)
            a_forward = false;
        goal_listed_ = true;  // sprite_render->update(dt): the draw list now holds the cheese

        reward = reached * 10.0f;
        terminated = reached;
        truncated = false;
        if (++steps_ >= kTimeout) terminated = true;
    }

    void paint() override {  // maze.cpp:386-414
        painter_.target->clear_black();
        float zoom = static_cast<float>(view_w_) / (kUnitPx * static_cast<float>(visible_));
        painter_.cam_scale = zoom;
        painter_.cam_size = {static_cast<float>(view_w_), static_cast<float>(view_h_)};

        const Texture* bg = tex_floor_[floor_];
        float aspect = static_cast<float>(bg->w) / static_cast<float>(bg->h);
        float extra = aspect - 1.0f;
        painter_.draw(bg, V2{-floor_shift_ * extra, 0.0f}, 64.0f * kUnitPx / bg->h);

        {  // tilemap.cpp:111-133
            const V2& cp = painter_.cam_pos;
            const V2& cs = painter_.cam_size;
            const float sc = painter_.cam_scale;
            Box view{(cp.x - cs.x * 0.5f / sc) * kPxUnit, (cp.y - cs.y * 0.5f / sc) * kPxUnit, cs.x * kPxUnit / sc,
                     cs.y * kPxUnit / sc};
            int x0 = std::floor(view.x), y0 = std::floor(view.y);
            int x1 = std::ceil(view.x + view.w), y1 = std::ceil(view.y + view.h);
            for (int y = y0; y <= y1; y++)
                for (int x = x0; x <= x1; x++) {
                    if (at(x, H - 1 - y) == kOpen) continue;
                    painter_.draw(tex_wall_, V2{x * kUnitPx, y * kUnitPx}, kUnitPx / tex_wall_->w);
                }
        }
        if (goal_listed_) {  // cheese sprite: offset (-0.48,-0.5), scale 0.95, z=1 (tilemap.cpp:88)
            float scale = 1.0f * 0.95f;
            painter_.draw(tex_cheese_, V2{(goal_pos.x + -0.48f) * kUnitPx, (goal_pos.y + -0.5f) * kUnitPx},
                          scale * kUnitPx / tex_cheese_->w, 1.0f, false);
        }
        {  // common_systems.cpp:138-150
            float agent_scale = 1.0f;
            V2 off{-0.5f, -0.5f};
            painter_.draw(tex_mouse_, V2{(a_pos.x + off.x) * kUnitPx, (a_pos.y + off.y) * kUnitPx},
                          kUnitPx / tex_mouse_->w * agent_scale, 1.0f, a_forward);
        }
    }

   private:
    std::vector<uint8_t> tiles_;
    IdPool ids_;
    IdSet in_sprite_, in_goal_, in_agent_, in_tilemap_;
    V2 a_pos, goal_pos;
    bool a_forward = true, goal_listed_ = false;
    int steps_ = 0, floor_ = 0;
    float floor_shift_ = 0.0f;
    const Texture* tex_wall_ = nullptr;
    const Texture* tex_cheese_ = nullptr;
    const Texture* tex_mouse_ = nullptr;
    const Texture* tex_floor_[9] = {};
};

}  // namespace

Env* new_maze() { return new Maze(); }

}  // namespace pgo
