// Micro-benchmark: what a wavefront's texel gather costs the CU's vector-memory pipe, by access
// shape — the number the render kernels' bound table needs instead of an assumed "four lanes a clock".
//
//   hipcc --offload-arch=gfx950 -O3 -o procgen2_amd/build/gather_rate tools/probe/gather_rate.hip && procgen2_amd/build/gather_rate
//
// Every mode issues the same number of wave-level load instructions (kIters × kUnroll per wave, 20 waves per CU resident,
// 5 per SIMD as coinrun's render kernel); only the addresses / the active lanes differ.  Printed: nanoseconds per
// wave-instruction per CU, and the same as clocks at the clock the run measured with s_memtime / s_memrealtime.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kUnroll = 16, kIters = 64, kThreads = 256;

enum Mode {
    kScattered64,   // 64 lanes, 64 different 128-B lines (a tile row's worst case)
    kCoalesced64,   // 64 consecutive dwords: two lines
    kStride14,      // the backdrop's shape: one texel in 3.5, 64 lanes over ≈ 900 B = 8 lines
    kStride20,      // a tile row: 13 tiles, five lanes each, their texels 25 apart in one 512-B texel row: ≈ 50 lines
    kScattered8,    // 8 lanes active (exec mask), 8 lines
    kScattered16,   // 16 lanes active
    kOob56,         // 64 lanes active, 56 of them beyond the buffer's range (return 0), 8 lines
    kLdsDirect14,   // kStride14 as buffer_load … lds
    kLdsRead,       // ds_read_b32, random words of a 16-KB LDS table (no bank-conflict control: as a mini-atlas would be read)
    kLdsReadRow,    // ds_read_b32, 64 consecutive words
    kModes
};
static const char* const kNames[kModes] = {"64 lanes, 64 lines", "64 lanes, coalesced", "64 lanes, stride 14 B (backdrop row)",
                                           "64 lanes, 13 runs of 5 (tile row)", "8 lanes active, 8 lines", "16 lanes active, 16 lines",
                                           "64 lanes, 56 out of range", "stride 14 B, buffer_load … lds", "ds_read_b32 random (LDS)",
                                           "ds_read_b32 consecutive (LDS)"};

template <int MODE>
__global__ void __launch_bounds__(kThreads) probe(const uint32_t* table, uint32_t bytes, uint32_t* out, uint64_t* clocks) {
    __shared__ uint32_t lds[4096 + 64 * kUnroll];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < 4096; k += kThreads) lds[k] = table[k];
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(table), 0, static_cast<int>(bytes), 0x00020000);
    uint32_t acc = 0;
    uint32_t seed = blockIdx.x * 2654435761u + wave * 40503u;
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < kIters; it++) {
        uint32_t v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
            seed = seed * 1664525u + 1013904223u;
            const uint32_t base = (seed >> 8) % (bytes / 2) & ~127u;  // wave-uniform, line-aligned, lower half of the table
            uint32_t off = 0;
            bool on = true;
            if (MODE == kScattered64) off = (base + lane * 128u * 5u + (lane & 31) * 4u) % bytes;
            if (MODE == kCoalesced64) off = base + lane * 4u;
            if (MODE == kStride14 || MODE == kLdsDirect14) off = base + (lane * 14u & ~3u);
            if (MODE == kStride20) off = (base + (lane / 5u) * 8192u + (lane % 5u) * 100u) % bytes;  // five texels 25 apart in one texel row, per tile
            if (MODE == kScattered8) {
                on = lane < 8;
                off = (base + lane * 128u * 5u) % bytes;
            }
            if (MODE == kScattered16) {
                on = lane < 16;
                off = (base + lane * 128u * 5u) % bytes;
            }
            if (MODE == kOob56) off = lane < 8 ? (base + lane * 128u * 5u) % bytes : 0x40000000u + lane * 4u;
            v[u] = 0;
            if (MODE == kLdsRead) {
                v[u] = lds[((seed >> 4) + lane * 37u) & 4095u];
            } else if (MODE == kLdsReadRow) {
                v[u] = lds[((seed >> 4) & 4031u) + lane];
            } else if (MODE == kLdsDirect14) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)&lds[4096 + u * 64], 4, off, 0, 0, 0);
            } else if (on) {
                v[u] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0);
            }
        }
        if (MODE == kLdsDirect14) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc ^= lds[4096 + lane];
        }
#pragma unroll
        for (int u = 0; u < kUnroll; u++) acc ^= v[u];
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * kThreads + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clocks[0] = t1 - t0;
        clocks[1] = r1 - r0;
    }
}

template <int MODE>
static void run(const uint32_t* table, uint32_t bytes, uint32_t* out, uint64_t* clocks, const char* where) {
    const int blocks = 256 * 5 * 8;  // 5 workgroups of 4 waves per CU resident, eight rounds
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(kThreads), 0, 0, table, bytes, out, clocks);
    hipEventRecord(a, 0);
    const int reps = 5;
    for (int k = 0; k < reps; k++) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(kThreads), 0, 0, table, bytes, out, clocks);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    uint64_t c[2];
    hipMemcpy(c, clocks, 16, hipMemcpyDeviceToHost);
    const double ghz = c[1] ? double(c[0]) / double(c[1]) * 0.1 : 0.0;  // s_memrealtime ticks at 100 MHz
    const double per_cu = double(blocks) / 256.0 * (kThreads / 64) * kIters * kUnroll * reps;  // wave-instructions per CU
    const double ns = ms * 1e6 / per_cu;
    printf("%-38s %-9s %7.2f ns per wave-instruction per CU = %6.1f clocks at %.2f GHz\n", kNames[MODE], where, ns, ns * ghz, ghz);
}

int main() {
    // two table sizes: 16 KB (stays in a CU's L1) and 64 MB (L2 / Infinity Cache, as the atlas)
    for (uint32_t bytes : {16u << 10, 64u << 20}) {
        uint32_t *table, *out;
        uint64_t* clocks;
        hipMalloc(&table, bytes);
        hipMalloc(&out, 256 * 5 * 8 * kThreads * 4);
        hipMalloc(&clocks, 16);
        std::vector<uint32_t> h(bytes / 4);
        for (size_t k = 0; k < h.size(); k++) h[k] = static_cast<uint32_t>(k * 2654435761u);
        hipMemcpy(table, h.data(), bytes, hipMemcpyHostToDevice);
        const char* where = bytes <= (16u << 10) ? "16 KB" : "64 MB";
        run<kScattered64>(table, bytes, out, clocks, where);
        run<kCoalesced64>(table, bytes, out, clocks, where);
        run<kStride14>(table, bytes, out, clocks, where);
        run<kStride20>(table, bytes, out, clocks, where);
        run<kScattered8>(table, bytes, out, clocks, where);
        run<kScattered16>(table, bytes, out, clocks, where);
        run<kOob56>(table, bytes, out, clocks, where);
        run<kLdsDirect14>(table, bytes, out, clocks, where);
        if (bytes <= (16u << 10)) {
            run<kLdsRead>(table, bytes, out, clocks, where);
            run<kLdsReadRow>(table, bytes, out, clocks, where);
        }
        hipFree(table);
        hipFree(out);
        hipFree(clocks);
    }
    return 0;
}
