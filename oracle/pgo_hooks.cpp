// ORACLE — TEST INFRASTRUCTURE ONLY (see pgo_common.h header).
//
// Test hooks: the oracle's level-generator pieces, AABB helpers and entity bookkeeping behind the same C signatures
// as oracle/ref_driver.cpp gives the REFERENCE's own compiled sources (oracle/_ref), so tests/test_reference_pin.py can
// run both on the same inputs.  What each one restates:
//   pgo_hook_maze_generate / _maze_level   games/maze/maze_generator.cpp:55-139,183-195; maze/tilemap.cpp:65-70
//   pgo_hook_setmaze_generate              games/{chaser,jumper}/maze_generator.cpp:47-173
//   pgo_hook_rooms_update / _rooms_analyse games/{caveflyer,jumper}/room_generator.cpp:4-202
//   pgo_hook_collisions                    games/*/helpers.cpp:40-108
//   pgo_hook_ecs_script                    games/*/ecs.h:24-61,212-215, ecs.cpp:3-83 (id queue, per-system entity sets)
#include <cstring>

#include "pgo_common.h"
#include "pgo_kruskal.h"
#include "pgo_rooms.h"

#define HOOK extern "C" __attribute__((visibility("default")))

using namespace pgo;

HOOK int pgo_hook_maze_generate(uint32_t seed, int w, int h, int n_objects, int* grid, int* free_cells, int* n_free,
                                uint32_t* next) {
    Rng rng;
    rng.seed(seed);
    Carver cv;
    cv.carve(w, h, rng);
    for (int k = 0; k < n_objects; k++) cv.drop(2 + k, rng);
    std::memcpy(grid, cv.grid.data(), cv.grid.size() * sizeof(int));
    for (int k = 0; k < cv.n_open; k++) free_cells[k] = cv.open_cells[k];
    *n_free = cv.n_open;
    *next = static_cast<uint32_t>(rng.eng());
    return static_cast<int>(cv.grid.size());
}

HOOK int pgo_hook_maze_level(uint32_t seed, int world_dim, int* grid, uint32_t* next) {
    Rng rng;
    rng.seed(seed);
    const int dim = rng.irange(0, (world_dim - 1) / 2 - 1) * 2 + 3;  // as pgo_maze.cpp new_level
    Carver cv;
    cv.carve(dim, dim, rng);
    cv.drop(2, rng);
    std::memcpy(grid, cv.grid.data(), cv.grid.size() * sizeof(int));
    *next = static_cast<uint32_t>(rng.eng());
    return dim;
}

HOOK int pgo_hook_setmaze_generate(uint32_t seed, int dim, int no_dead_ends, int* grid, uint32_t* next) {
    Rng rng;
    rng.seed(seed);
    std::vector<int> g;
    carve_merged(dim, g, rng);
    if (no_dead_ends) open_dead_ends(dim, g, rng);
    std::memcpy(grid, g.data(), g.size() * sizeof(int));
    *next = static_cast<uint32_t>(rng.eng());
    return static_cast<int>(g.size());
}

HOOK void pgo_hook_rooms_update(int gw, int gh, int* grid, int iters) {
    Rooms r;
    r.gw = gw;
    r.gh = gh;
    r.grid.assign(grid, grid + gw * gh);
    for (int k = 0; k < iters; k++) r.update();
    std::memcpy(grid, r.grid.data(), sizeof(int) * gw * gh);
}

HOOK int pgo_hook_rooms_analyse(int gw, int gh, const int* grid, int* best_order, uint32_t src_sel, uint32_t dst_sel,
                                int* path, int* n_path, int expand_n, int* wide_order, int* n_wide) {
    Rooms r;
    r.gw = gw;
    r.gh = gh;
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

HOOK void pgo_hook_collisions(int n, const float* a, const float* b, uint8_t* hit, float* overlap) {
    for (int i = 0; i < n; i++) {
        const Box r1{a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]};
        const Box r2{b[4 * i], b[4 * i + 1], b[4 * i + 2], b[4 * i + 3]};
        hit[i] = boxes_touch(r1, r2) ? 1 : 0;
        const Box o = overlap_box(r1, r2);
        overlap[4 * i] = o.x;
        overlap[4 * i + 1] = o.y;
        overlap[4 * i + 2] = o.w;
        overlap[4 * i + 3] = o.h;
    }
}

// The same script language as ref_ecs_script, on the oracle's IdPool + IdSet (what every pgo_<game>.cpp uses for its
// systems).  The three sets persist across calls like the reference's process-global Coordinator does (clear() keeps
// the bucket count).
HOOK int pgo_hook_ecs_script(int n_ops, const int* ops, const int* args, int* out_ids, int* orders, int cap) {
    static IdPool pool;
    static IdSet in_a, in_ab, in_any;
    pool.refill();
    in_a.clear();
    in_ab.clear();
    in_any.clear();
    struct Live {
        int id;
        bool a, b;
    };
    std::vector<Live> live;
    auto changed = [&](int id, bool a, bool b) {  // System_Manager::entity_signature_changed
        if (a) in_a.insert(id); else in_a.erase(id);
        if (a && b) in_ab.insert(id); else in_ab.erase(id);
        in_any.insert(id);
    };
    int w = 0;
    for (int k = 0; k < n_ops; k++) {
        int id = -1;
        if (ops[k] == 0) {
            id = pool.take();
            bool a = false, b = false;
            if (args[k] & 1) {
                a = true;
                changed(id, a, b);
            }
            if (args[k] & 2) {
                b = true;
                changed(id, a, b);
            }
            live.push_back({id, a, b});
        } else if (ops[k] == 1 && !live.empty()) {
            const int at = static_cast<int>(static_cast<unsigned>(args[k]) % live.size());
            id = live[at].id;
            pool.give_back(id);
            in_a.erase(id);
            in_ab.erase(id);
            in_any.erase(id);
            live.erase(live.begin() + at);
        } else if (ops[k] == 2) {
            pool.refill();
            in_a.clear();
            in_ab.clear();
            in_any.clear();
            live.clear();
        } else if (ops[k] == 3) {
            std::vector<int> holders;
            for (size_t i = 0; i < live.size(); i++)
                if (live[i].b) holders.push_back(static_cast<int>(i));
            if (!holders.empty()) {
                const int at = holders[static_cast<unsigned>(args[k]) % holders.size()];
                id = live[at].id;
                live[at].b = false;
                changed(id, live[at].a, false);
            }
        }
        out_ids[k] = id;
        for (const IdSet* s : {&in_a, &in_ab, &in_any}) {
            if (w + 1 + static_cast<int>(s->size()) > cap) return -1;
            orders[w++] = static_cast<int>(s->size());
            for (int e : *s) orders[w++] = e;
        }
    }
    return w;
}

// ---------------------------------------------------------------------------------------------------------------
// The genuine libstdc++ articles for the device-side primitive sweep (tests/hip/selftest.hip,
// tests/test_primitives_gpu.py): same shapes as the pgst_* entry points.
// ---------------------------------------------------------------------------------------------------------------
#include <algorithm>

HOOK void pgo_hook_mt(int n_seeds, const uint32_t* seeds, int n, uint32_t* out) {
    for (int s = 0; s < n_seeds; s++) {
        std::mt19937 e;
        e.seed(seeds[s]);
        for (int i = 0; i < n; i++) out[static_cast<size_t>(s) * n + i] = static_cast<uint32_t>(e());
    }
}

HOOK void pgo_hook_draws(uint32_t seed, int n, const uint8_t* kind, const int* lo, const int* hi, const float* fa,
                         const float* fb, int* iout, float* fout) {
    std::mt19937 e;
    e.seed(seed);
    for (int i = 0; i < n; i++) {
        iout[i] = 0;
        fout[i] = 0.0f;
        if (kind[i] == 0) {
            std::uniform_int_distribution<int> d(lo[i], hi[i]);
            iout[i] = d(e);
        } else {
            std::uniform_real_distribution<float> d(fa[i], fb[i]);
            fout[i] = d(e);
        }
    }
}

HOOK void pgo_hook_bulk(uint32_t seed, int skip, int count, float* out, uint32_t* next) {
    std::mt19937 e;
    e.seed(seed);
    for (int i = 0; i < skip; i++) e();
    std::uniform_real_distribution<float> d(0.0f, 1.0f);
    for (int i = 0; i < count; i++) out[i] = d(e);
    *next = static_cast<uint32_t>(e());
}

static uint32_t fnv(uint32_t h, uint32_t v) { return (h ^ v) * 16777619u; }

HOOK void pgo_hook_hash_script(int n_ops, const int* ops, const int* keys, uint32_t* hashes) {
    std::unordered_set<int> set;
    for (int k = 0; k < n_ops; k++) {
        if (ops[k] == 0)
            set.insert(keys[k]);
        else if (ops[k] == 1)
            set.erase(keys[k]);
        else
            set.clear();
        uint32_t f = fnv(2166136261u, static_cast<uint32_t>(set.size()));
        for (int v : set) f = fnv(f, static_cast<uint32_t>(v));
        hashes[k] = f;
    }
}

HOOK void pgo_hook_set_rounds(int n_rounds, const int* counts, const int16_t* keys_in, int16_t* keys_out) {
    std::unordered_set<int> set;
    size_t at = 0;
    for (int r = 0; r < n_rounds; r++) {
        set.clear();  // keeps the bucket array
        for (int i = 0; i < counts[r]; i++) set.insert(keys_in[at + i]);
        size_t w = at;
        for (int v : set) keys_out[w++] = static_cast<int16_t>(v);
        at += counts[r];
    }
}

HOOK void pgo_hook_sort_equal(int n, int* out) {
    std::vector<std::pair<float, int>> v(n);
    for (int i = 0; i < n; i++) v[i] = {1.0f, i};
    std::sort(v.begin(), v.end(), [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first < b.first; });
    for (int i = 0; i < n; i++) out[i] = v[i].second;
}

// A list of integer draw calls through the raster spec (pgo_raster.cpp spec_blit), for the device-side test of the
// engine's sprite replay (tests/hip/selftest.hip pgst_replay): textures are RGBA8 blobs one after the other, `bg` the
// 64×64 target to start from (0x00BBGGRR words), a draw is 12 ints {texture, dx, dy, dw, dh, sx, sy, sw, sh,
// flip (1 = horizontal, 2 = vertical), alpha modulation, rotated} plus its angle in degrees; out = 64×64 RGB.
#include "pgo_raster.h"
HOOK void pgo_hook_raster(int n_tex, const int* tex_w, const int* tex_h, const uint8_t* rgba, const uint32_t* bg,
                          int n_draws, const int32_t* draws, const double* deg, uint8_t* out_rgb) {
    std::vector<pgo::Texture> tex(n_tex);
    size_t at = 0;
    for (int t = 0; t < n_tex; t++) {
        tex[t].w = tex_w[t];
        tex[t].h = tex_h[t];
        tex[t].rgba.assign(rgba + at, rgba + at + size_t(tex_w[t]) * tex_h[t] * 4);
        at += size_t(tex_w[t]) * tex_h[t] * 4;
    }
    pgo::Surface target(64, 64);
    for (int k = 0; k < 64 * 64; k++) {
        target.px[4 * k + 0] = static_cast<uint8_t>(bg[k]);
        target.px[4 * k + 1] = static_cast<uint8_t>(bg[k] >> 8);
        target.px[4 * k + 2] = static_cast<uint8_t>(bg[k] >> 16);
        target.px[4 * k + 3] = 255;
    }
    for (int k = 0; k < n_draws; k++) {
        const int32_t* d = draws + 12 * k;
        pgo::spec_blit(target, tex[d[0]], static_cast<float>(d[5]), static_cast<float>(d[6]), static_cast<float>(d[7]),
                       static_cast<float>(d[8]), static_cast<float>(d[1]), static_cast<float>(d[2]),
                       static_cast<float>(d[3]), static_cast<float>(d[4]), d[11] ? deg[k] : 0.0, d[9], d[10]);
    }
    pgo::pack_rgb(target, out_rgb);
}
