// TEST INFRASTRUCTURE — device-side sweep of the engine's bit-exact primitives.
//
// procgen2_amd/csrc/pg_*.h are `PG_HD` headers: tests/cpp/test_primitives.cpp compiles them for the HOST and checks
// them against libstdc++ / glibc, but several have a separate __HIP_DEVICE_COMPILE__ branch (rcp-based division,
// __umul24, bit casts) and wave-cooperative forms that only exist on the device.  This library runs each primitive
// in a small gfx950 kernel over caller-provided arguments and hands the raw results back; tests/test_primitives_gpu.py
// compares them on the host with glibc (sinf/cosf/atan2f/atanf through ctypes) and the real libstdc++
// (std::mt19937, uniform_*_distribution, unordered_set<int>, std::sort — oracle/pgo_hooks.cpp).
// Built by procgen2_amd/build.py into procgen2_amd/lib/libpg_selftest.so; never loaded by the product path.
#include <hip/hip_runtime.h>

#include <vector>

#include "pg_atan2.h"
#include "pg_geom.h"
#include "pg_order.h"
#include "pg_rng.h"
#include "pg_setorder.h"
#include "pg_sincos.h"
#include "pg_engine.h"
#include "pg_render.h"
#include "pg_stamps.h"

#define ST_API extern "C" __attribute__((visibility("default")))

namespace {

template <class T>
struct Dev {
    T* p = nullptr;
    size_t n = 0;
    explicit Dev(size_t count) : n(count) {
        if (hipMalloc(reinterpret_cast<void**>(&p), (count ? count : 1) * sizeof(T)) != hipSuccess) p = nullptr;
    }
    Dev(const T* host, size_t count) : Dev(count) {
        if (p && count) hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice);
    }
    ~Dev() {
        if (p) hipFree(p);
    }
    bool down(T* host) const { return hipMemcpy(host, p, n * sizeof(T), hipMemcpyDeviceToHost) == hipSuccess; }
};

int finish() {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return e == hipSuccess ? 0 : 1;
}

__global__ void k_sincos(int n, const float* x, float* s, float* c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    s[i] = pg::sc_sinf(x[i]);
    c[i] = pg::sc_cosf(x[i]);
}
__global__ void k_atan2(int n, const float* y, const float* x, float* out, float* out1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = pg::at_atan2f(y[i], x[i]);
    out1[i] = pg::at_atanf(y[i]);
}
__global__ void k_div(int n, const int* a, const int* b, int* q, int* hm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    q[i] = pg::udiv_small(a[i], b[i]);
    hm[i] = pg::hash_mod(a[i] & 0x7fff, 1 + (b[i] & 0xfff) % 4095);
}
__global__ void k_blend(int n, const uint32_t* dst, const uint32_t* src, const int* a, uint32_t* out, uint32_t* d255) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = pg::blend_px(dst[i], src[i], a[i]);
    d255[i] = pg::div255(static_cast<uint32_t>(i) & 0xffffu);
}
__global__ void k_div255_pair(int n, const uint32_t* x, uint32_t* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = pg::div255_pair(x[i]);
}
__global__ void k_box(int n, const float* a, const float* b, uint8_t* hit, float* ov) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const pg::Box r1{a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]};
    const pg::Box r2{b[4 * i], b[4 * i + 1], b[4 * i + 2], b[4 * i + 3]};
    hit[i] = pg::box_hit(r1, r2) ? 1 : 0;
    const pg::Box o = pg::box_overlap(r1, r2);
    ov[4 * i] = o.x;
    ov[4 * i + 1] = o.y;
    ov[4 * i + 2] = o.w;
    ov[4 * i + 3] = o.h;
}

// T1: one lane, state in global scratch (what the lane-per-env logic kernels do).
__global__ void k_mt_lane(const uint32_t* seeds, int n, uint32_t* scratch, uint32_t* out) {
    uint32_t* x = scratch + size_t(blockIdx.x) * pg::kMtWords;
    pg::mt_seed(x, seeds[blockIdx.x]);
    for (int i = 0; i < n; i++) out[size_t(blockIdx.x) * n + i] = pg::mt_next(x);
}
// T1: one wavefront, state in LDS, wave-cooperative regeneration (what the level kernels do).
__global__ void __launch_bounds__(64) k_mt_wave(const uint32_t* seeds, int n, uint32_t* out) {
    __shared__ uint32_t x[pg::kMtWords];
    const int lane = threadIdx.x;
    if (lane == 0) pg::mt_seed(x, seeds[blockIdx.x]);
    __syncthreads();
    for (int i = 0; i < n; i++) {
        const uint32_t v = pg::wave_mt_next(x, lane);
        if (lane == (i & 63)) out[size_t(blockIdx.x) * n + i] = v;
    }
}
// T2: a script of draws from one stream.  kind 0: uniform_int(lo, hi); 1: uniform_real(fa, fb).
__global__ void __launch_bounds__(64) k_draws(uint32_t seed, int n, const uint8_t* kind, const int* lo, const int* hi,
                                              const float* fa, const float* fb, int* iout, float* fout, int wave) {
    __shared__ uint32_t x[pg::kMtWords];
    const int lane = threadIdx.x;
    if (lane == 0) pg::mt_seed(x, seed);
    __syncthreads();
    if (wave) {
        for (int i = 0; i < n; i++) {
            if (kind[i] == 0) {
                const int v = pg::wave_rng_int(x, lo[i], hi[i], lane);
                if (lane == 0) iout[i] = v;
            } else {
                const float v = pg::wave_rng_real(x, fa[i], fb[i], lane);
                if (lane == 0) fout[i] = v;
            }
        }
    } else if (lane == 0) {
        for (int i = 0; i < n; i++) {
            if (kind[i] == 0)
                iout[i] = pg::rng_int(x, lo[i], hi[i]);
            else
                fout[i] = pg::rng_real(x, fa[i], fb[i]);
        }
    }
}
// wave_draws / wave_coin_flips (bulk canonical draws, caveflyer's 1 600 cells)
__global__ void __launch_bounds__(64) k_bulk(uint32_t seed, int skip, int count, float* out, uint32_t* next) {
    __shared__ uint32_t x[pg::kMtWords];
    const int lane = threadIdx.x;
    if (lane == 0) pg::mt_seed(x, seed);
    __syncthreads();
    for (int i = 0; i < skip; i++) pg::wave_mt_next(x, lane);
    pg::wave_draws(x, count, lane, [&](int k, float v) { out[k] = v; });
    __syncthreads();
    const uint32_t v = pg::wave_mt_next(x, lane);
    if (lane == 0) *next = v;
}

__device__ uint32_t fnv(uint32_t h, uint32_t v) { return (h ^ v) * 16777619u; }

// T3 serial twin on the device (its hash_mod is the rcp branch): op 0 insert, 1 erase, 2 clear; keys < 2048.
// After every op: FNV-1a over (count, keys in iteration order).
__global__ void k_hash_script(int n_ops, const int* ops, const int* keys, int16_t* next, int16_t* before,
                              uint32_t* hashes) {
    pg::HashOrder h;
    pg::hash_init(h, next, before);
    for (int k = 0; k < n_ops; k++) {
        if (ops[k] == 0) {
            if (!pg::hash_contains(h, keys[k])) pg::hash_insert(h, keys[k], true);
        } else if (ops[k] == 1) {
            if (pg::hash_contains(h, keys[k])) pg::hash_erase(h, keys[k]);
        } else {
            pg::hash_clear(h);
        }
        uint32_t f = fnv(2166136261u, static_cast<uint32_t>(h.count));
        for (int16_t node = static_cast<int16_t>(h.head); node != pg::kNil; node = h.next[node])
            f = fnv(f, static_cast<uint32_t>(node));
        hashes[k] = f;
    }
}

// T3 closed form (pg_setorder.h): rounds of "clear(), insert these distinct keys"; bucket state carried across.
constexpr int kMaxKeys = 1664, kMaxBuckets = 2368;
__global__ void __launch_bounds__(64) k_set_rounds(int n_rounds, const int* counts, const int16_t* keys_in,
                                                   int16_t* keys_out) {
    __shared__ int16_t keys[kMaxKeys];
    __shared__ int32_t touch[kMaxBuckets], chain[kMaxBuckets], tail_sum[kMaxKeys + 1];
    __shared__ int16_t link[kMaxKeys], tmp[kMaxKeys];
    const int lane = threadIdx.x;
    int32_t buckets = 1, next_resize = 0;
    int at = 0;
    for (int r = 0; r < n_rounds; r++) {
        const int n = counts[r];
        for (int i = lane; i < n; i += 64) keys[i] = keys_in[at + i];
        __syncthreads();
        pg::wave_set_order(keys, n, buckets, next_resize, pg::SetOrderScratch{touch, chain, link, tail_sum, tmp}, lane);
        __syncthreads();
        for (int i = lane; i < n; i += 64) keys_out[at + i] = keys[i];
        __syncthreads();
        at += n;
    }
}

// T4: the introsort twin on n equal keys, run on the device.
__global__ void k_sort_equal(int n, int* out) {
    pg::ZItem items[pg::kRankMax];
    for (int k = 0; k < n; k++) items[k] = {1.0f, k};
    pg::sort_by_key(items, n);
    for (int k = 0; k < n; k++) out[k] = items[k].id;
}

// The sprite replay of the two-wavefront render kernels (pg_render.h wave_replay_rows) on a synthetic draw list: one
// draw per lane, in lane order, over a given 64×64 target; each wave blends and stores the 32 rows it owns.
template <bool kRotInGroups>
__global__ void __launch_bounds__(128) k_replay(pg::AtlasView atlas, const uint32_t* bg, int n_draws, const int32_t* draws,
                                                const double* deg, uint8_t* out_rgb, uint32_t stamps) {
    __shared__ alignas(16) uint32_t fb[pg::kFbWords];
    const int lane = threadIdx.x & 63, half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int k = threadIdx.x; k < pg::kFbWords; k += 128) fb[k] = bg[k] | 0x5a000000u;  // (the top byte is nobody's)
    __syncthreads();
    pg::Blit mine;
    mine.dx = mine.dy = mine.sx = mine.sy = mine.tex_off = 0;
    mine.dw = mine.dh = mine.sw = mine.sh = mine.tex_w = 1;
    mine.flip_mod = 255;
    mine.rot_sn = 0;
    mine.rot_cs = 65536;
    bool has = lane < n_draws;
    pg::RotBox whole{0, 0, 0, 0};
    if (has) {
        const int32_t* d = draws + 12 * lane;
        const int4 t = atlas.desc[d[0]];
        mine.dx = d[1];
        mine.dy = d[2];
        mine.dw = d[3];
        mine.dh = d[4];
        mine.sx = d[5];
        mine.sy = d[6];
        mine.sw = d[7];
        mine.sh = d[8];
        mine.tex_off = t.x;
        mine.tex_w = t.y;
        mine.flip_mod = d[10] | ((d[9] & 1) ? pg::kFlipH : 0) | ((d[9] & 2) ? pg::kFlipV : 0);
        if (d[11] && deg[lane] != 0.0) {
            pg::rotation_16_16(deg[lane], mine.rot_sn, mine.rot_cs);
            mine.flip_mod = d[10] | pg::kRotated;
        }
        // stamps != 0: the pre-pass's substitution (pg_stamps.h) where a stamp of this draw's size and modulation exists —
        // the draw cut down to the stamp's core, or (rotated) its box to the core's
        if (stamps) {
            int core_w, core_h;
            has = pg::stamp_substitute(reinterpret_cast<const uint4*>(atlas.texels + stamps) + d[0] * pg::kStampsPerTex, t.y, t.z, mine,
                                       core_w, core_h);
            if (mine.flip_mod & pg::kRotated) whole = pg::rot_box_core(mine, core_w, core_h);
        } else if (mine.flip_mod & pg::kRotated) {
            whole = pg::rot_box(mine);
        }
    }
    // (stamps allowed; tiny draws four to a slot)
    pg::wave_replay_rows<4, kRotInGroups, true, 4, true, true>(fb, atlas, mine, __ballot(has), lane, 32 * half, 32 * half + 32, &whole);
    pg::wave_store_rows(fb, out_rgb, lane, 32 * half, 32 * half + 32);
}

}  // namespace

// atlas: `words` RGBA texels of all textures one after the other, desc[t] = {first texel, w, h, 0}.
namespace {
struct MiniAtlas {  // what pg::append_stamps asks of an atlas (pg_engine.h Atlas has the same four members)
    std::vector<uint32_t> words;
    std::vector<int4> desc;
    const uint32_t* texels_host(int tex) const { return words.data() + desc[tex].x; }
    int4 desc_host(int tex) const { return desc[tex]; }
    uint32_t append_words(const std::vector<uint32_t>& more) {
        const uint32_t at = static_cast<uint32_t>(words.size());
        words.insert(words.end(), more.begin(), more.end());
        return at;
    }
    size_t texel_bytes() const { return words.size() * 4; }
};
}  // namespace

// `stamped` (rot_in_groups bit 1): every draw that takes its whole texture gets a stamp of its size and modulation
// (pg_stamps.h append_stamps, the first kStampsPerTex per texture) and the kernel substitutes it, as a pre-pass would.
ST_API int pgst_replay(int rot_in_groups, int n_tex, const int32_t* desc, int n_words_in, const uint32_t* words_in,
                       const uint32_t* bg, int n_draws, const int32_t* draws, const double* deg, uint8_t* out_rgb) {
    if (n_draws > 64) return 1;
    const bool stamped = (rot_in_groups & 2) != 0;
    rot_in_groups &= 1;
    MiniAtlas mini;
    mini.words.assign(words_in, words_in + n_words_in);
    mini.desc.assign(reinterpret_cast<const int4*>(desc), reinterpret_cast<const int4*>(desc) + n_tex);
    uint32_t stamps = 0;
    if (stamped) {
        std::vector<pg::StampSpec> specs;
        for (int k = 0; k < n_draws; k++) {
            const int32_t* d = draws + 12 * k;
            const int4 t = mini.desc[d[0]];
            if (d[5] == 0 && d[6] == 0 && d[7] == t.y && d[8] == t.z) specs.push_back({d[0], d[3], d[4], d[10]});
        }
        stamps = pg::append_stamps(mini, n_tex, specs);
    }
    const int n_words = static_cast<int>(mini.words.size());
    const uint32_t* words = mini.words.data();
    Dev<int4> d_desc(reinterpret_cast<const int4*>(desc), n_tex);
    Dev<uint32_t> d_words(words, n_words), d_bg(bg, 64 * 64);
    Dev<int32_t> d_draws(draws, size_t(12) * (n_draws ? n_draws : 1));
    Dev<double> d_deg(deg, n_draws ? n_draws : 1);
    Dev<uint8_t> d_out(64 * 64 * 3);
    const pg::AtlasView atlas{d_words.p, d_desc.p, n_tex, static_cast<uint32_t>(n_words) * 4u, nullptr};
    if (rot_in_groups)
        hipLaunchKernelGGL(k_replay<true>, dim3(1), dim3(128), 0, 0, atlas, d_bg.p, n_draws, d_draws.p, d_deg.p, d_out.p, stamps);
    else
        hipLaunchKernelGGL(k_replay<false>, dim3(1), dim3(128), 0, 0, atlas, d_bg.p, n_draws, d_draws.p, d_deg.p, d_out.p, stamps);
    if (finish()) return 1;
    return d_out.down(out_rgb) ? 0 : 1;
}

ST_API int pgst_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

ST_API int pgst_sincos(int n, const float* x, float* s, float* c) {
    Dev<float> dx(x, n), ds(n), dc(n);
    hipLaunchKernelGGL(k_sincos, dim3((n + 255) / 256), dim3(256), 0, 0, n, dx.p, ds.p, dc.p);
    return finish() || !ds.down(s) || !dc.down(c);
}
ST_API int pgst_atan2(int n, const float* y, const float* x, float* out_atan2, float* out_atan_of_y) {
    Dev<float> dy(y, n), dx(x, n), d2(n), d1(n);
    hipLaunchKernelGGL(k_atan2, dim3((n + 255) / 256), dim3(256), 0, 0, n, dy.p, dx.p, d2.p, d1.p);
    return finish() || !d2.down(out_atan2) || !d1.down(out_atan_of_y);
}
ST_API int pgst_div(int n, const int* a, const int* b, int* quotient, int* hash_mod) {
    Dev<int> da(a, n), db(b, n), dq(n), dm(n);
    hipLaunchKernelGGL(k_div, dim3((n + 255) / 256), dim3(256), 0, 0, n, da.p, db.p, dq.p, dm.p);
    return finish() || !dq.down(quotient) || !dm.down(hash_mod);
}
ST_API int pgst_blend(int n, const uint32_t* dst, const uint32_t* src, const int* a, uint32_t* out, uint32_t* div255) {
    Dev<uint32_t> dd(dst, n), dsrc(src, n), dout(n), d255(n);
    Dev<int> da(a, n);
    hipLaunchKernelGGL(k_blend, dim3((n + 255) / 256), dim3(256), 0, 0, n, dd.p, dsrc.p, da.p, dout.p, d255.p);
    return finish() || !dout.down(out) || !d255.down(div255);
}
ST_API int pgst_div255_pair(int n, const uint32_t* x, uint32_t* out) {
    Dev<uint32_t> dx(x, n), dout(n);
    hipLaunchKernelGGL(k_div255_pair, dim3((n + 255) / 256), dim3(256), 0, 0, n, dx.p, dout.p);
    return finish() || !dout.down(out);
}
ST_API int pgst_box(int n, const float* a, const float* b, uint8_t* hit, float* overlap) {
    Dev<float> da(a, size_t(n) * 4), db(b, size_t(n) * 4), dov(size_t(n) * 4);
    Dev<uint8_t> dh(n);
    hipLaunchKernelGGL(k_box, dim3((n + 255) / 256), dim3(256), 0, 0, n, da.p, db.p, dh.p, dov.p);
    return finish() || !dh.down(hit) || !dov.down(overlap);
}
ST_API int pgst_mt(int n_seeds, const uint32_t* seeds, int n, int wave, uint32_t* out) {
    Dev<uint32_t> ds(seeds, n_seeds), dout(size_t(n_seeds) * n), scratch(size_t(n_seeds) * pg::kMtWords);
    if (wave)
        hipLaunchKernelGGL(k_mt_wave, dim3(n_seeds), dim3(64), 0, 0, ds.p, n, dout.p);
    else
        hipLaunchKernelGGL(k_mt_lane, dim3(n_seeds), dim3(1), 0, 0, ds.p, n, scratch.p, dout.p);
    return finish() || !dout.down(out);
}
ST_API int pgst_draws(uint32_t seed, int n, const uint8_t* kind, const int* lo, const int* hi, const float* fa,
                      const float* fb, int wave, int* iout, float* fout) {
    Dev<uint8_t> dk(kind, n);
    Dev<int> dlo(lo, n), dhi(hi, n), di(n);
    Dev<float> dfa(fa, n), dfb(fb, n), df(n);
    hipMemset(di.p, 0, size_t(n) * 4);
    hipMemset(df.p, 0, size_t(n) * 4);
    hipLaunchKernelGGL(k_draws, dim3(1), dim3(64), 0, 0, seed, n, dk.p, dlo.p, dhi.p, dfa.p, dfb.p, di.p, df.p, wave);
    return finish() || !di.down(iout) || !df.down(fout);
}
ST_API int pgst_bulk(uint32_t seed, int skip, int count, float* out, uint32_t* next) {
    Dev<float> dout(count);
    Dev<uint32_t> dn(1);
    hipLaunchKernelGGL(k_bulk, dim3(1), dim3(64), 0, 0, seed, skip, count, dout.p, dn.p);
    return finish() || !dout.down(out) || !dn.down(next);
}
ST_API int pgst_hash_script(int n_ops, const int* ops, const int* keys, uint32_t* hashes) {
    Dev<int> dops(ops, n_ops), dkeys(keys, n_ops);
    Dev<int16_t> next(2048), before(2400);
    Dev<uint32_t> dh(n_ops);
    hipLaunchKernelGGL(k_hash_script, dim3(1), dim3(1), 0, 0, n_ops, dops.p, dkeys.p, next.p, before.p, dh.p);
    return finish() || !dh.down(hashes);
}
ST_API int pgst_set_rounds(int n_rounds, const int* counts, const int16_t* keys_in, int16_t* keys_out) {
    size_t total = 0;
    for (int r = 0; r < n_rounds; r++) {
        if (counts[r] < 0 || counts[r] > kMaxKeys) return 2;
        total += counts[r];
    }
    Dev<int> dc(counts, n_rounds);
    Dev<int16_t> din(keys_in, total), dout(total);
    hipLaunchKernelGGL(k_set_rounds, dim3(1), dim3(64), 0, 0, n_rounds, dc.p, din.p, dout.p);
    return finish() || !dout.down(keys_out);
}
ST_API int pgst_sort_equal(int n, int* out) {
    if (n < 1 || n > pg::kRankMax) return 2;
    Dev<int> dout(n);
    hipLaunchKernelGGL(k_sort_equal, dim3(1), dim3(1), 0, 0, n, dout.p);
    return finish() || !dout.down(out);
}
