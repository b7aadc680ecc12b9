// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Thin C-ABI drivers around the parts of the REFERENCE that compile from their own sources with plain g++ and no
// stand-in of any kind (they include no SDL header): games/maze/maze_generator.cpp, games/chaser/maze_generator.cpp
// (jumper's copy is byte-identical), games/caveflyer/room_generator.cpp (jumper's copy is byte-identical),
// games/coinrun/helpers.cpp and games/coinrun/ecs.cpp (all seven games hold byte-identical copies).
//
// This file holds no reference code: it includes the reference's headers with -I$(REF)/games/<game> and is linked
// with the reference's .cpp files compiled WHERE THEY LIE (oracle/Makefile: ref); outputs go to oracle/_ref/ only.
// One shared object per family (-DREF_FAMILY_…), because the families define classes of the same name.
// tests/test_reference_pin.py runs every function below next to the oracle's restatement (oracle/pgo_hooks.cpp,
// same signatures with a pgo_hook_ prefix) on the same inputs, and tests/golden/make_ref_fixtures.py records a few
// of their outputs as fixtures for the GPU box, where /root/reference does not exist.
#include <cstdint>
#include <cstring>
#include <random>
#include <unordered_set>
#include <vector>

#define REF_API extern "C" __attribute__((visibility("default")))

static uint32_t next_draw(std::mt19937& rng) { return static_cast<uint32_t>(rng()); }

#if defined(REF_FAMILY_MAZE)
#include "maze_generator.h"  // games/maze

// generate_maze(w, h) on a freshly seeded engine, then n_objects × place_object(2 + k).
// grid: (w+2)·(h+2) ints; free_cells: the first num_free_cells entries (cap w·h); *next = the engine's next output.
REF_API int ref_maze_generate(uint32_t seed, int w, int h, int n_objects, int* grid, int* free_cells, int* n_free,
                              uint32_t* next) {
    std::mt19937 rng;
    rng.seed(seed);
    Maze_Generator gen;
    gen.generate_maze(w, h, rng);
    for (int k = 0; k < n_objects; k++) gen.place_object(2 + k, rng);
    std::memcpy(grid, gen.grid.data(), gen.grid.size() * sizeof(int));
    for (int k = 0; k < gen.num_free_cells; k++) free_cells[k] = gen.free_cells[k];
    *n_free = gen.num_free_cells;
    *next = next_draw(rng);
    return static_cast<int>(gen.grid.size());
}

// What games/maze/tilemap.cpp:65-70 does with a fresh engine (rng.seed(seed) is cenv_make's, maze.cpp): draw the maze
// side, generate, place the goal (GOAL = 2, tilemap.cpp:6).  Returns the side; grid as above.
REF_API int ref_maze_level(uint32_t seed, int world_dim, int* grid, uint32_t* next) {
    std::mt19937 rng;
    rng.seed(seed);
    std::uniform_int_distribution<int> n_dist(0, (world_dim - 1) / 2 - 1);
    const int maze_dim = n_dist(rng) * 2 + 3;
    Maze_Generator gen;
    gen.generate_maze(maze_dim, maze_dim, rng);
    gen.place_object(2, rng);
    std::memcpy(grid, gen.grid.data(), gen.grid.size() * sizeof(int));
    *next = next_draw(rng);
    return maze_dim;
}
#endif

#if defined(REF_FAMILY_SETMAZE)
#include "maze_generator.h"  // games/chaser (== games/jumper)

// generate_maze / generate_maze_no_dead_ends(dim, dim) on a freshly seeded engine.  grid: (dim+2)² ints.
REF_API int ref_setmaze_generate(uint32_t seed, int dim, int no_dead_ends, int* grid, uint32_t* next) {
    std::mt19937 rng;
    rng.seed(seed);
    Maze_Generator gen;
    if (no_dead_ends)
        gen.generate_maze_no_dead_ends(dim, dim, rng);
    else
        gen.generate_maze(dim, dim, rng);
    std::memcpy(grid, gen.grid.data(), gen.grid.size() * sizeof(int));
    *next = next_draw(rng);
    return static_cast<int>(gen.grid.size());
}
#endif

#if defined(REF_FAMILY_ROOMS)
#include "room_generator.h"  // games/caveflyer (== games/jumper)

// `iters` cellular-automaton updates of a gw × gh grid (index y + gh·x), in place.
REF_API void ref_rooms_update(int gw, int gh, int* grid, int iters) {
    Room_Generator r;
    r.init(gw, gh);
    r.grid.assign(grid, grid + gw * gh);
    for (int k = 0; k < iters; k++) r.update();
    std::memcpy(grid, r.grid.data(), sizeof(int) * gw * gh);
}

// find_best_room → its ITERATION ORDER (what `for (int i : best_room)` of caveflyer/tilemap.cpp:158 sees); then
// src = order[src_sel % n], dst = order[dst_sel % n]; find_path(src, dst); a set filled by range-insert of the path
// and widened with expand_room(set, expand_n) → its iteration order.  Returns n (0: no room at all).
REF_API int ref_rooms_analyse(int gw, int gh, const int* grid, int* best_order, uint32_t src_sel, uint32_t dst_sel,
                              int* path, int* n_path, int expand_n, int* wide_order, int* n_wide) {
    Room_Generator r;
    r.init(gw, gh);
    r.grid.assign(grid, grid + gw * gh);
    std::unordered_set<int> best;
    r.find_best_room(best);
    int n = 0;
    for (int i : best) best_order[n++] = i;
    *n_path = 0;
    *n_wide = 0;
    if (n == 0) return 0;
    std::vector<int> p;
    r.find_path(best_order[src_sel % n], best_order[dst_sel % n], p);
    for (int i : p) path[(*n_path)++] = i;
    std::unordered_set<int> wide;
    wide.insert(p.begin(), p.end());
    r.expand_room(wide, expand_n);
    for (int i : wide) wide_order[(*n_wide)++] = i;
    return n;
}
#endif

#if defined(REF_FAMILY_ECS)
#include "ecs.h"      // games/coinrun
#include "helpers.h"  // games/coinrun

// helpers.cpp:40-108 over n rectangle pairs (x, y, w, h each).
REF_API void ref_collisions(int n, const float* a, const float* b, uint8_t* hit, float* overlap) {
    for (int i = 0; i < n; i++) {
        const Rectangle r1{a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]};
        const Rectangle r2{b[4 * i], b[4 * i + 1], b[4 * i + 2], b[4 * i + 3]};
        hit[i] = check_collision(r1, r2) ? 1 : 0;
        const Rectangle o = get_collision_overlap(r1, r2);
        overlap[4 * i] = o.x;
        overlap[4 * i + 1] = o.y;
        overlap[4 * i + 2] = o.width;
        overlap[4 * i + 3] = o.height;
    }
}

// A script of entity operations against the reference's Coordinator (ecs.h / ecs.cpp) with two component types and
// three systems: SysA wants {A}, SysAB wants {A, B}, SysAny has the empty signature and so mirrors every entity that
// ever had a component added (System_Tilemap's `{0}` signature).
//   op 0, arg m (1..3): create an entity, add A if m & 1, then B if m & 2          → out_id = the new id
//   op 1, arg k:        destroy the (k mod live)-th live entity in creation order  → out_id = its id (-1: none alive)
//   op 2:               clear_entities                                             → out_id = -1
//   op 3, arg k:        remove component B from the (k mod nB)-th live B-holder    → out_id = its id (-1: none)
// After every op the iteration order of the three systems' `entities` sets is appended to `orders`, each as
// count, ids…  Returns the number of ints written (or -1 if cap is too small).
struct CompA {
    int v;
};
struct CompB {
    int v;
};
class SysA : public System {};
class SysAB : public System {};
class SysAny : public System {};

REF_API int ref_ecs_script(int n_ops, const int* ops, const int* args, int* out_ids, int* orders, int cap) {
    static bool registered = false;
    static std::shared_ptr<SysA> sa;
    static std::shared_ptr<SysAB> sab;
    static std::shared_ptr<SysAny> sany;
    if (!registered) {
        c.register_component<CompA>();
        c.register_component<CompB>();
        sa = c.register_system<SysA>();
        sab = c.register_system<SysAB>();
        sany = c.register_system<SysAny>();
        Signature s;
        s.set(c.get_component_type<CompA>());
        c.set_system_signature<SysA>(s);
        s.set(c.get_component_type<CompB>());
        c.set_system_signature<SysAB>(s);
        c.set_system_signature<SysAny>(Signature());
        registered = true;
    }
    c.clear_entities();
    // NOTE: clear() keeps each set's bucket count, so a script's orders depend on the scripts run before it in this
    // process — exactly the reference's behaviour across episodes.  ref_ecs_fresh() below tells.
    struct Live {
        int id;
        bool b;
    };
    std::vector<Live> live;
    int w = 0;
    for (int k = 0; k < n_ops; k++) {
        int id = -1;
        if (ops[k] == 0) {
            id = c.create_entity();
            if (args[k] & 1) c.add_component(id, CompA{k});
            if (args[k] & 2) c.add_component(id, CompB{k});
            live.push_back({id, (args[k] & 2) != 0});
        } else if (ops[k] == 1 && !live.empty()) {
            const int at = static_cast<int>(static_cast<unsigned>(args[k]) % live.size());
            id = live[at].id;
            c.destroy_entity(id);
            live.erase(live.begin() + at);
        } else if (ops[k] == 2) {
            c.clear_entities();
            live.clear();
        } else if (ops[k] == 3) {
            std::vector<int> holders;
            for (size_t i = 0; i < live.size(); i++)
                if (live[i].b) holders.push_back(static_cast<int>(i));
            if (!holders.empty()) {
                const int at = holders[static_cast<unsigned>(args[k]) % holders.size()];
                id = live[at].id;
                c.remove_component<CompB>(id);
                live[at].b = false;
            }
        }
        out_ids[k] = id;
        for (System* sys : {static_cast<System*>(sa.get()), static_cast<System*>(sab.get()), static_cast<System*>(sany.get())}) {
            if (w + 1 + static_cast<int>(sys->entities.size()) > cap) return -1;
            orders[w++] = static_cast<int>(sys->entities.size());
            for (int e : sys->entities) orders[w++] = e;
        }
    }
    return w;
}
#endif
