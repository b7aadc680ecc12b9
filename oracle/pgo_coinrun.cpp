// ORACLE — TEST INFRASTRUCTURE ONLY (see pgo_common.h header).
//
// coinrun: CPU restatement of SURVEY.md rows G1s / G1r / G1g.
//   step   games/coinrun/coinrun.cpp:341-391, common_systems.cpp:7-39,65-105,121-252,284-313
//   render games/coinrun/coinrun.cpp:443-470, tilemap.cpp:294-321, common_systems.cpp:41-63,254-278,315-337
//   reset  games/coinrun/coinrun.cpp:472-507, tilemap.cpp:52-94,97-292
// The reference keeps entities in an ECS; here each entity is one record in a flat table indexed by
// the id the reference's allocator would have handed out, and each System's std::unordered_set<Entity>
// is kept as a real std::unordered_set<int> so its iteration order (T3) is libstdc++'s own.
#include <algorithm>
#include <array>

#include "pgo_env.h"

namespace pgo {
namespace {

enum Tile : uint8_t { kEmpty = 0, kWallTop, kWallMid, kLavaTop, kLavaMid, kCrate };  // tilemap.h:13-21
enum Solid { kPass = 0, kFull, kOneWay };                                            // tilemap.h:23-27

const char* const kGroundThemes[6] = {"Dirt", "Grass", "Planet", "Sand", "Snow", "Stone"};  // tilemap.h:29
const char* const kWalkers[9] = {"slimeBlock", "slimePurple", "slimeBlue", "slimeGreen", "mouse",
                                 "snail",      "ladybug",     "wormGreen", "wormPink"};  // tilemap.h:30
const char* const kCrates[4] = {"boxCrate", "boxCrate_double", "boxCrate_single", "boxCrate_warning"};
const char* const kAliens[5] = {"Beige", "Blue", "Green", "Pink", "Yellow"};  // common_systems.h:62

const char* const kBackdrops[49] = {  // coinrun.cpp:60-110
    "platform_backgrounds/alien_bg.png", "platform_backgrounds/another_world_bg.png",
    "platform_backgrounds/back_cave.png", "platform_backgrounds/caverns.png",
    "platform_backgrounds/cyberpunk_bg.png", "platform_backgrounds/parallax_forest.png",
    "platform_backgrounds/scifi_bg.png", "platform_backgrounds/scifi2_bg.png",
    "platform_backgrounds/living_tissue_bg.png", "platform_backgrounds/airadventurelevel1.png",
    "platform_backgrounds/airadventurelevel2.png", "platform_backgrounds/airadventurelevel3.png",
    "platform_backgrounds/airadventurelevel4.png", "platform_backgrounds/cave_background.png",
    "platform_backgrounds/blue_desert.png", "platform_backgrounds/blue_grass.png",
    "platform_backgrounds/blue_land.png", "platform_backgrounds/blue_shroom.png",
    "platform_backgrounds/colored_desert.png", "platform_backgrounds/colored_grass.png",
    "platform_backgrounds/colored_land.png", "platform_backgrounds/colored_shroom.png",
    "platform_backgrounds/landscape1.png", "platform_backgrounds/landscape2.png",
    "platform_backgrounds/landscape3.png", "platform_backgrounds/landscape4.png",
    "platform_backgrounds/battleback1.png", "platform_backgrounds/battleback2.png",
    "platform_backgrounds/battleback3.png", "platform_backgrounds/battleback4.png",
    "platform_backgrounds/battleback5.png", "platform_backgrounds/battleback6.png",
    "platform_backgrounds/battleback7.png", "platform_backgrounds/battleback8.png",
    "platform_backgrounds/battleback9.png", "platform_backgrounds/battleback10.png",
    "platform_backgrounds/sunrise.png", "platform_backgrounds_2/beach1.png",
    "platform_backgrounds_2/beach2.png", "platform_backgrounds_2/beach3.png",
    "platform_backgrounds_2/beach4.png", "platform_backgrounds_2/fantasy1.png",
    "platform_backgrounds_2/fantasy2.png", "platform_backgrounds_2/fantasy3.png",
    "platform_backgrounds_2/fantasy4.png", "platform_backgrounds_2/candy1.png",
    "platform_backgrounds_2/candy2.png", "platform_backgrounds_2/candy3.png",
    "platform_backgrounds_2/candy4.png"};

std::string lower(std::string s) {
    for (auto& ch : s) ch = static_cast<char>(std::tolower(static_cast<unsigned char>(ch)));
    return s;
}

struct Spark {  // common_components.h:60-63
    V2 pos;
    float life = 0.0f;
};

struct Thing {
    bool has_sprite = false, has_anim = false, is_mob = false;
    V2 pos;
    Box bounds{-0.5f, -0.5f, 1.0f, 1.0f};
    // sprite
    V2 sprite_off{-0.5f, -0.5f};
    float z = 1.0f;
    bool flip_x = false;
    const Texture* tex = nullptr;
    bool tex_set = false;  // "texture != nullptr" even in logic-only runs
    // animation (common_components.h:37-43)
    const Texture* frames[2] = {nullptr, nullptr};
    int frame = 0;
    float rate = 0.1f, anim_t = 0.0f;
    // mob
    float vel_x = 0.15f;
    // particles (common_components.h:65-72)
    std::array<Spark, 10> sparks;
    float lifespan = 5.0f, spawn_timer = 0.0f, spawn_time = 0.5f;
    V2 spark_off{0.0f, 0.34f};
};

struct Hit {
    V2 at;
    bool any;
};

class Coinrun final : public Env {
   public:
    static constexpr int W = 64, H = 64;

    int dump_state(float* out, int cap) const override;
    int dump_tiles(uint8_t* out, int cap) const override {
        int n = std::min<int>(cap, W * H);
        std::memcpy(out, tiles_.data(), n);
        return n;
    }

   protected:
    void on_make() override;
    void new_level() override;
    void advance(int action) override;
    void paint() override;

   private:
    // ---- tile map (tilemap.h:62-85) ----
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
    template <class Pred>
    Hit collide(Box r, Pred solid, bool fallthrough = false, float step_y = 0.0f) const;

    int spawn(Thing t);
    void add_saw(int x, int y);
    void add_mob(int x, int y);
    void build_level();

    void tick_mobs(float dt);
    void tick_agent(float dt, int action, bool& alive, bool& got_coin);
    void tick_sparks(float dt);
    void tick_sprites(float dt);

    std::vector<uint8_t> tiles_ = std::vector<uint8_t>(W * H, 0);
    std::vector<int> crate_kind_ = std::vector<int>(W * H, 0);

    std::vector<Thing> things_ = std::vector<Thing>(IdPool::kMax);
    IdPool ids_;
    // One set per reference System (coinrun.cpp:249-292); tilemap's signature {0} matches every id (D19).
    IdSet in_sprite_, in_tilemap_, in_mob_, in_hazard_, in_goal_, in_agent_, in_sparks_;
    std::vector<std::pair<float, int>> draw_list_;  // common_systems.h:22

    // agent (Component_Transform/Dynamics/Agent of the player entity)
    int agent_id_ = -1;
    V2 a_pos, a_vel;
    bool a_ground = false, a_forward = true;
    float a_phase = 0.0f;
    const Box a_bounds{-0.5f, -1.0f, 1.0f, 1.0f};

    int backdrop_ = 0, alien_ = 0, ground_ = 0;
    float backdrop_shift_ = 0.0f;

    // textures
    const Texture* tex_top_[6] = {};
    const Texture* tex_mid_[6] = {};
    const Texture* tex_lava_top_ = nullptr;
    const Texture* tex_lava_ = nullptr;
    const Texture* tex_crate_[4] = {};
    const Texture* tex_walk_[9][2] = {};
    const Texture* tex_saw_[2] = {};
    const Texture* tex_coin_ = nullptr;
    const Texture* tex_stand_[5] = {};
    const Texture* tex_jump_[5] = {};
    const Texture* tex_walk1_[5] = {};
    const Texture* tex_walk2_[5] = {};
    const Texture* tex_spark_ = nullptr;
    const Texture* tex_backdrop_[49] = {};
};

void Coinrun::on_make() {
    auto& bank = TextureBank::global();
    auto T = [&](const std::string& n) { return bank.find("assets/" + n); };
    for (int i = 0; i < 6; i++) {  // tilemap.cpp:10-13
        std::string th = kGroundThemes[i];
        tex_top_[i] = T("kenney/Ground/" + th + "/" + lower(th) + "Mid.png");
        tex_mid_[i] = T("kenney/Ground/" + th + "/" + lower(th) + "Center.png");
    }
    tex_lava_top_ = T("kenney/Tiles/lavaTop_low.png");
    tex_lava_ = T("kenney/Tiles/lava.png");
    for (int i = 0; i < 4; i++) tex_crate_[i] = T(std::string("kenney/Tiles/") + kCrates[i] + ".png");
    for (int i = 0; i < 9; i++) {
        tex_walk_[i][0] = T(std::string("kenney/Enemies/") + kWalkers[i] + ".png");
        tex_walk_[i][1] = T(std::string("kenney/Enemies/") + kWalkers[i] + "_move.png");
    }
    tex_saw_[0] = T("kenney/Enemies/sawHalf.png");
    tex_saw_[1] = T("kenney/Enemies/sawHalf_move.png");
    tex_coin_ = T("kenney/Items/coinGold.png");
    for (int i = 0; i < 5; i++) {  // common_systems.cpp:113-118
        std::string a = kAliens[i];
        std::string base = "kenney/Players/128x256/" + a + "/alien" + a;
        tex_stand_[i] = T(base + "_stand.png");
        tex_jump_[i] = T(base + "_jump.png");
        tex_walk1_[i] = T(base + "_walk1.png");
        tex_walk2_[i] = T(base + "_walk2.png");
    }
    tex_spark_ = T("misc_assets/iconCircle_white.png");
    for (int i = 0; i < 49; i++) tex_backdrop_[i] = T(kBackdrops[i]);
}

int Coinrun::spawn(Thing t) {
    int id = ids_.take();
    things_[id] = t;
    in_tilemap_.insert(id);
    return id;
}

void Coinrun::add_saw(int x, int y) {  // tilemap.cpp:52-68
    Thing t;
    t.pos = {static_cast<float>(x) + 0.5f, static_cast<float>(H - 1 - y) + 0.5f};
    t.has_sprite = t.has_anim = true;
    t.frames[0] = tex_saw_[0];
    t.frames[1] = tex_saw_[1];
    t.rate = 1.0f;
    int id = spawn(t);
    in_sprite_.insert(id);
    in_hazard_.insert(id);
}

void Coinrun::add_mob(int x, int y) {  // tilemap.cpp:70-94
    Thing t;
    t.pos = {static_cast<float>(x) + 0.5f, static_cast<float>(H - 1 - y) + 0.5f};
    int which = rng_.irange(0, 8);
    t.has_sprite = t.has_anim = t.is_mob = true;
    t.frames[0] = tex_walk_[which][0];
    t.frames[1] = tex_walk_[which][1];
    t.rate = 0.2f;
    t.bounds = {-0.5f, -0.48f, 1.0f, 0.98f};
    t.vel_x = 0.15f * ((rng_.unit() < 0.5f) * 2.0f - 1.0f);
    int id = spawn(t);
    in_sprite_.insert(id);
    in_hazard_.insert(id);
    in_mob_.insert(id);
    in_sparks_.insert(id);
}

void Coinrun::build_level() {  // tilemap.cpp:97-292
    const float max_jump = 1.5f, gravity = 0.2f, max_speed = 0.5f;
    std::fill(tiles_.begin(), tiles_.end(), kEmpty);
    fill(0, 0, W, 1, kWallTop);
    fill(0, 0, 1, H, kWallMid);
    fill(W - 1, 0, 1, H, kWallMid);
    fill(0, H - 1, W, 1, kWallMid);

    const int difficulty = rng_.irange(1, 3);
    const int sections = rng_.irange(difficulty, 2 * difficulty - 1);
    int cx = 5, cy = 1;
    const int pit_thresh = difficulty;
    const int danger = rng_.irange(0, 2);

    float reach_x = max_speed * 2.0f * max_jump / gravity;
    float reach_y = max_jump * max_jump / (2.0f * gravity);
    const int max_dx = reach_x - 0.5f;
    const int max_dy = reach_y - 0.5f;

    for (int s = 0; s < sections; s++) {
        if (cx + 15 >= W) break;
        const int bump = difficulty / 3;
        int dy = (flags_ & 4u) ? 0 : rng_.irange(1 + bump, 4 + bump);  // cfg.allow_dy ? dy_dist(rng) : 0 (tilemap.cpp:158)
        dy = std::min(dy, max_dy);
        if (cy >= 20 || (cy >= 5 && rng_.unit() < 0.5f)) dy = -dy;
        const int dx = rng_.irange(3 + bump, 2 * difficulty + 2 + bump);
        cy = std::max(1, cy + dy);

        const bool pit = !(flags_ & 1u) && (dx > 7) && (cy > 3) && (rng_.irange(0, 19) >= pit_thresh);  // allow_pit (:174)
        if (pit) {
            int x1 = rng_.irange(1, 3);
            int x2 = rng_.irange(1, 3);
            int gap = dx - x1 - x2;
            if (gap > max_dx) {
                gap = max_dx;
                x2 = dx - x1 - gap;
            }
            fill_capped(cx, 0, x1, cy, kWallMid, kWallTop);
            fill_capped(cx + dx - x2, 0, x2, cy, kWallMid, kWallTop);
            int lava_h = rng_.irange(1, cy - 3);
            if (danger == 0) {
                fill_capped(cx + x1, 1, gap, lava_h, kLavaMid, kLavaTop);
            } else if (danger == 1) {
                for (int i = 0; i < gap; i++) add_saw(cx + x1 + i, 1);
            } else {
                for (int i = 0; i < gap; i++) add_mob(cx + x1 + i, 1);
            }
            if (gap > 4) {  // stepping stone
                int x3, w1;
                if (gap == 5) {
                    x3 = rng_.irange(1, 2);
                    w1 = rng_.irange(1, 2);
                } else if (gap == 6) {
                    x3 = rng_.irange(1, 2) + 1;
                    w1 = rng_.irange(1, 2);
                } else {
                    x3 = rng_.irange(1, 2) + 1;
                    int x4 = rng_.irange(1, 2) + 1;
                    w1 = gap - x3 - x4;
                }
                fill_capped(cx + x1 + x3, cy - 1, w1, 1, kWallMid, kWallTop);
            }
        } else {
            fill_capped(cx, 0, dx, cy, kWallMid, kWallTop);
            int ob1 = -1, ob2 = -1;
            if (rng_.irange(0, 9) < 2 * difficulty && dx > 3) {
                ob1 = cx + rng_.irange(1, dx - 2);
                add_saw(ob1, cy);
            }
            if (!(flags_ & 8u) && rng_.irange(0, 9) < difficulty && dx > 3 && max_dx >= 4) {  // allow_mobs (:250)
                ob1 = cx + rng_.irange(1, dx - 2);
                add_mob(ob1, cy);
            }
            for (int i = 0; i < ((flags_ & 2u) ? 0 : 2); i++) {  // allow_crate (:258)
                int crate_x = cx + rng_.irange(1, dx - 2);
                if (rng_.unit() < 0.5f && ob1 != crate_x && ob2 != crate_x) {
                    int pile = rng_.irange(1, 3);
                    for (int j = 0; j < pile; j++) {
                        put(crate_x, cy + j, kCrate);
                        crate_kind_[cy + j + crate_x * H] = rng_.irange(0, 3);
                    }
                }
            }
        }
        cx += dx;
    }

    Thing coin;
    coin.pos = {static_cast<float>(cx) + 0.5f, static_cast<float>(H - 1 - cy) + 0.5f};
    coin.has_sprite = true;
    coin.tex = tex_coin_;
    coin.tex_set = true;
    int id = spawn(coin);
    in_sprite_.insert(id);
    in_goal_.insert(id);

    fill_capped(cx, 0, 1, cy, kWallMid, kWallTop);
    fill(cx + 1, 0, W - cx, H, kWallMid);
}

void Coinrun::new_level() {  // coinrun.cpp:472-507
    ids_.refill();  // ecs.cpp:52-97 clear_entities: sets keep their bucket arrays
    in_sprite_.clear();
    in_tilemap_.clear();
    in_mob_.clear();
    in_hazard_.clear();
    in_goal_.clear();
    in_agent_.clear();
    in_sparks_.clear();

    build_level();

    backdrop_ = rng_.irange(0, 48);
    backdrop_shift_ = rng_.unit();

    Thing player;
    agent_id_ = spawn(player);
    in_agent_.insert(agent_id_);
    a_pos = {1.5f, H - 1 - 1.0f};
    a_vel = {0.0f, 0.0f};
    a_ground = false;
    a_forward = true;
    a_phase = 0.0f;

    alien_ = rng_.irange(0, 4);
    ground_ = rng_.irange(0, 5);
    draw_list_.clear();
}

template <class Pred>
Hit Coinrun::collide(Box r, Pred solid, bool fallthrough, float step_y) const {  // tilemap.cpp:323-396
    bool any = false;
    const int x0 = std::floor(r.x), y0 = std::floor(r.y);
    const int x1 = std::ceil(r.x + r.w), y1 = std::ceil(r.y + r.h);
    const V2 mid{r.x + r.w * 0.5f, r.y + r.h * 0.5f};
    Box cell{0.0f, 0.0f, 1.0f, 1.0f};

    for (int y = y0; y <= y1; y++)
        for (int x = x0; x <= x1; x++) {
            const int kind = solid(at(x, H - 1 - y));
            if (kind == kPass) continue;
            cell.x = x;
            cell.y = y;
            const Box o = overlap_box(r, cell);
            if (o.w == 0.0f && o.h == 0.0f) continue;
            const float oy = o.y + o.h * 0.5f;
            if (o.w > o.h) {
                if (kind == kOneWay) {
                    const bool inside = (r.y + r.h - step_y > cell.y);
                    if (step_y > 0.01f && !fallthrough && !inside) {
                        r.y = (oy > mid.y ? cell.y - r.h : cell.y + cell.h);
                        any = true;
                    }
                } else {
                    r.y = (oy > mid.y ? cell.y - r.h : cell.y + cell.h);
                    any = true;
                }
            }
        }
    for (int y = y0; y <= y1; y++)
        for (int x = x0; x <= x1; x++) {
            const int kind = solid(at(x, H - 1 - y));
            if (kind == kPass) continue;
            cell.x = x;
            cell.y = y;
            const Box o = overlap_box(r, cell);
            if (o.w == 0.0f && o.h == 0.0f) continue;
            const float ox = o.x + o.w * 0.5f;
            if (o.w <= o.h && kind != kOneWay) {
                r.x = (ox > mid.x ? cell.x - r.w : cell.x + cell.w);
                any = true;
            }
        }
    return {{r.x, r.y}, any};
}

void Coinrun::tick_mobs(float dt) {  // common_systems.cpp:65-105
    for (int id : in_mob_) {
        Thing& m = things_[id];
        m.pos.x += m.vel_x * dt;
        Box wall_probe{m.pos.x - 0.5f, m.pos.y - 0.6f, 1.0f, 0.5f};
        Box floor_probe{m.pos.x - 0.5f, m.pos.y + 0.6f, 1.0f, 0.5f};
        Hit wall = collide(wall_probe, [](Tile t) { return (t == kWallMid || t == kWallTop) ? kFull : kPass; });
        Hit gap = collide(floor_probe, [](Tile t) { return t == kEmpty ? kFull : kPass; });
        float nx = wall.at.x + 0.5f;
        if (gap.any) nx = gap.at.x + 0.5f;
        m.pos.x = nx;
        if (wall.any || gap.any) m.vel_x *= -1.0f;
        m.flip_x = m.vel_x > 0.0f;
    }
}

void Coinrun::tick_agent(float dt, int action, bool& alive, bool& got_coin) {  // common_systems.cpp:121-252
    alive = true;
    got_coin = false;
    const float max_jump = 1.55f, gravity = 0.2f, max_speed = 0.5f, mix = 0.2f, air_control = 0.15f;

    float move_x = (action == 6 || action == 7 || action == 8) - (action == 0 || action == 1 || action == 2);
    const bool jump = (action == 2 || action == 5 || action == 8);
    const bool drop = (action == 0 || action == 3 || action == 6);

    float mix_x = a_ground ? mix : (mix * air_control);
    a_vel.x += mix_x * (max_speed * move_x - a_vel.x) * dt;
    if (std::abs(a_vel.x) < mix_x * max_speed * dt) a_vel.x = 0.0f;
    if (jump && a_ground) a_vel.y = -max_jump;
    a_vel.y += gravity * dt;
    if (std::abs(a_vel.y) > max_jump) a_vel.y = (a_vel.y > 0.0f ? 1.0f : -1.0f) * max_jump;

    a_pos.x += a_vel.x * dt;
    a_pos.y += a_vel.y * dt;

    Box body{a_pos.x + a_bounds.x, a_pos.y + a_bounds.y, a_bounds.w, a_bounds.h};
    Hit h = collide(
        body,
        [](Tile t) { return (t == kWallMid || t == kWallTop) ? kFull : (t == kCrate ? kOneWay : kPass); }, drop,
        a_vel.y * dt);
    V2 moved{h.at.x - body.x, h.at.y - body.y};
    a_ground = moved.y < 0.0f && h.any;
    a_pos.x = h.at.x - a_bounds.x;
    a_pos.y = h.at.y - a_bounds.y;
    body.x = a_pos.x + a_bounds.x;
    body.y = a_pos.y + a_bounds.y;
    if (moved.x != 0.0f) a_vel.x = 0.0f;
    if (a_ground) a_vel.y = 0.0f;

    for (int id : in_hazard_) {
        const Thing& t = things_[id];
        Box hb{t.pos.x + t.bounds.x, t.pos.y + t.bounds.y, t.bounds.w, t.bounds.h};
        if (boxes_touch(body, hb)) {
            alive = false;
            break;
        }
    }
    Hit lava = collide(body, [](Tile t) { return (t == kLavaMid || t == kLavaTop) ? kFull : kPass; });
    if (lava.any) alive = false;
    for (int id : in_goal_) {
        const Thing& t = things_[id];
        Box gb{t.pos.x + t.bounds.x, t.pos.y + t.bounds.y, t.bounds.w, t.bounds.h};
        if (boxes_touch(body, gb)) {
            got_coin = true;
            break;
        }
    }

    painter_.cam_pos.x = a_pos.x * kUnitPx;  // common_systems.cpp:238-239
    painter_.cam_pos.y = (a_pos.y - 0.5f) * kUnitPx;

    a_phase += 0.1f * dt;  // Component_Agent::rate = 0.1f
    a_phase = std::fmod(a_phase, 1.0f);
    if (move_x > 0.0f)
        a_forward = true;
    else if (move_x < 0.0f)
        a_forward = false;
}

void Coinrun::tick_sparks(float dt) {  // common_systems.cpp:284-313
    for (int id : in_sparks_) {
        Thing& m = things_[id];
        int dead = -1;
        for (int i = 0; i < 10; i++) {
            m.sparks[i].life -= dt;
            if (m.sparks[i].life <= 0.0f) dead = i;
        }
        m.spawn_timer += dt;
        if (dead != -1 && m.spawn_timer >= m.spawn_time) {
            m.spawn_timer = std::fmod(m.spawn_timer, m.spawn_time);
            m.sparks[dead].life = m.lifespan;
            m.sparks[dead].pos.x = m.pos.x + m.spark_off.x;
            m.sparks[dead].pos.y = m.pos.y + m.spark_off.y;
        }
    }
}

void Coinrun::tick_sprites(float dt) {  // common_systems.cpp:7-39
    if (draw_list_.size() != in_sprite_.size()) draw_list_.resize(in_sprite_.size());
    int k = 0;
    for (int id : in_sprite_) {
        Thing& t = things_[id];
        if (t.has_anim) {
            t.anim_t += dt;
            int adv = t.anim_t * t.rate;
            t.anim_t -= adv / t.rate;
            t.frame = (t.frame + adv) % 2;
            t.tex = t.frames[t.frame];
            t.tex_set = true;
        }
        draw_list_[k++] = {t.z, id};
    }
    std::sort(draw_list_.begin(), draw_list_.end(),
              [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first < b.first; });
}

void Coinrun::advance(int action) {  // coinrun.cpp:356-371
    const float dt = 1.0f / 4;
    for (int ss = 0; ss < 4; ss++) {
        bool alive, coin;
        tick_mobs(dt);
        tick_agent(dt, action, alive, coin);
        tick_sparks(dt);
        tick_sprites(dt);
        reward = coin * 10.0f;
        terminated = !alive || coin;
        truncated = false;
        if (terminated) break;
    }
}

void Coinrun::paint() {  // coinrun.cpp:443-470
    painter_.target->clear_black();
    painter_.cam_scale = 0.3f * static_cast<float>(view_w_) / static_cast<float>(kObsW);
    painter_.cam_size = {static_cast<float>(view_w_), static_cast<float>(view_h_)};

    const Texture* bg = tex_backdrop_[backdrop_];
    float aspect = static_cast<float>(bg->w) / static_cast<float>(bg->h);
    float extra = aspect - 1.0f;
    painter_.draw(bg, V2{-backdrop_shift_ * extra, 0.0f}, 64.0f * kUnitPx / bg->h);

    // negative-z sprites: none in coinrun (every sprite has z = 1), loop kept for the contract
    for (auto& zi : draw_list_) {
        const Thing& t = things_[zi.second];
        if (!t.tex_set) continue;
        if (t.z >= 0.0f) break;
        painter_.draw(t.tex, V2{(t.pos.x + t.sprite_off.x) * kUnitPx, (t.pos.y + t.sprite_off.y) * kUnitPx},
                      1.0f * kUnitPx / t.tex->w, 1.0f, t.flip_x);
    }

    {  // tilemap.cpp:294-321
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
                const Texture* tex = nullptr;
                if (t == kWallMid)
                    tex = tex_mid_[ground_];
                else if (t == kWallTop)
                    tex = tex_top_[ground_];
                else if (t == kLavaMid)
                    tex = tex_lava_;
                else if (t == kLavaTop)
                    tex = tex_lava_top_;
                else
                    tex = tex_crate_[crate_kind_[H - 1 - y + x * H]];
                painter_.draw(tex, V2{x * kUnitPx, y * kUnitPx}, kUnitPx / tex->w);
            }
    }

    {  // common_systems.cpp:315-337
        const float base_alpha = 0.5f, base_scale = 0.45f;
        for (int id : in_sparks_) {
            const Thing& m = things_[id];
            for (int i = 0; i < 10; i++) {
                const Spark& p = m.sparks[i];
                if (p.life <= 0.0f) continue;
                float lr = (m.lifespan - p.life) / m.lifespan;
                float alpha = base_alpha * (1.0f - lr);
                float scale = base_scale * (0.4f * lr + 0.6f);
                float oy = -lr * 0.17f;
                painter_.draw(tex_spark_,
                              V2{p.pos.x * kUnitPx - 0.5f * tex_spark_->w * scale,
                                 (p.pos.y + oy) * kUnitPx - 0.5f * tex_spark_->h * scale},
                              scale * kUnitPx / tex_spark_->w, alpha);
            }
        }
    }

    for (auto& zi : draw_list_) {  // common_systems.cpp:41-63, positive_z
        const Thing& t = things_[zi.second];
        if (!t.tex_set) continue;
        if (t.z < 0.0f) continue;
        float scale = 1.0f * 1.0f;
        painter_.draw(t.tex, V2{(t.pos.x + t.sprite_off.x) * kUnitPx, (t.pos.y + t.sprite_off.y) * kUnitPx},
                      scale * kUnitPx / t.tex->w, 1.0f, t.flip_x);
    }

    {  // common_systems.cpp:254-278
        const Texture* tex;
        if (std::abs(a_vel.x) < 0.01f && a_ground)
            tex = tex_stand_[alien_];
        else if (!a_ground)
            tex = tex_jump_[alien_];
        else if (a_phase > 0.5f)
            tex = tex_walk2_[alien_];
        else
            tex = tex_walk1_[alien_];
        V2 p{a_pos.x - 0.5f, a_pos.y - 2.0f};
        painter_.draw(tex, V2{p.x * kUnitPx, p.y * kUnitPx}, kUnitPx / tex->w, 1.0f, !a_forward);
    }
}

int Coinrun::dump_state(float* out, int cap) const {
    std::vector<float> v;
    v.push_back(a_pos.x);
    v.push_back(a_pos.y);
    v.push_back(a_vel.x);
    v.push_back(a_vel.y);
    v.push_back(a_ground);
    v.push_back(a_forward);
    v.push_back(a_phase);
    v.push_back(painter_.cam_pos.x);
    v.push_back(painter_.cam_pos.y);
    v.push_back(static_cast<float>(backdrop_));
    v.push_back(backdrop_shift_);
    v.push_back(static_cast<float>(alien_));
    v.push_back(static_cast<float>(ground_));
    v.push_back(static_cast<float>(agent_id_));  // = number of non-agent entities
    for (int id = 0; id < agent_id_; id++) {
        const Thing& t = things_[id];
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

}  // namespace

Env* new_coinrun() { return new Coinrun(); }

}  // namespace pgo
