// TEST INFRASTRUCTURE — the room pipeline of caveflyer / jumper (procgen2_amd/csrc/pg_rooms.h) on the device, on its own:
// two automaton updates, find_best_room's iteration order, find_path, expand_room's membership, for one cave per call.
// pg_rooms.h is a per-world-size header (PG_ROOMS_DIM, namespace per PG_VARIANT), so this file is compiled once per
// size (procgen2_amd/build.py: variants 0 / 1 / 2 = 40 / 20 / 45, the sizes caveflyer's and jumper's modes use) into
// lib/libpg_selftest.so as pgst_rooms_<side>.  tests/test_primitives_gpu.py holds the results to outputs of the
// REFERENCE's own room_generator.cpp recorded in tests/golden/ref_fixtures.json — no oracle in between.
#include <hip/hip_runtime.h>

#if PG_VARIANT == 0
#define PG_ROOMS_DIM 40
#define PGST_NAME pgst_rooms_40
#elif PG_VARIANT == 1
#define PG_ROOMS_DIM 20
#define PGST_NAME pgst_rooms_20
#else
#define PG_ROOMS_DIM 45
#define PGST_NAME pgst_rooms_45
#endif
#include "pg_rooms.h"

namespace R = pg::PG_VARIANT_NS::rooms;

namespace {

// in: raw[kCells] (0/1); out: cave[kCells] after two updates, best order + count, path + count, wide membership
__global__ void __launch_bounds__(64) k_rooms(const uint8_t* raw, uint32_t src_sel, uint32_t dst_sel, uint8_t* cave,
                                              int32_t* best, int32_t* path, uint8_t* wide, int32_t* counts) {
    __shared__ R::RoomsLds L;
    const int lane = threadIdx.x;
    for (int c = lane; c < R::kCells; c += 64) L.aux[c] = raw[c];
    __syncthreads();
    R::automaton(L.aux, L.grid, lane);  // caveflyer/tilemap.cpp:141-146: two iterations
    __syncthreads();
    R::automaton(L.grid, L.aux, lane);
    __syncthreads();
    for (int c = lane; c < R::kCells; c += 64) {
        L.grid[c] = L.aux[c];
        cave[c] = L.aux[c];
    }
    __syncthreads();
    const int n = R::best_room(L, lane);
    __syncthreads();
    for (int k = lane; k < n; k += 64) best[k] = L.cells[k];
    int n_path = 0;
    if (n > 0) {
        const int src = L.cells[src_sel % static_cast<uint32_t>(n)], dst = L.cells[dst_sel % static_cast<uint32_t>(n)];
        __syncthreads();
        if (src != dst) {
            R::goal_path(L, src, dst, lane);
            n_path = L.path_len;
            for (int k = lane; k < n_path; k += 64) path[k] = L.cells[k];
            __syncthreads();
            R::widen(L, lane);
            for (int c = lane; c < R::kCells; c += 64) wide[c] = L.aux[c] != 0;
        }
    }
    if (lane == 0) {
        counts[0] = n;
        counts[1] = n_path;
    }
}

}  // namespace

extern "C" __attribute__((visibility("default"))) int PGST_NAME(const uint8_t* raw, uint32_t src_sel, uint32_t dst_sel,
                                                                  uint8_t* cave, int32_t* best, int32_t* path,
                                                                  uint8_t* wide, int32_t* counts) {
    constexpr int n = R::kCells;
    uint8_t *d_raw = nullptr, *d_cave = nullptr, *d_wide = nullptr;
    int32_t *d_best = nullptr, *d_path = nullptr, *d_counts = nullptr;
    bool ok = hipMalloc(reinterpret_cast<void**>(&d_raw), n) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&d_cave), n) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&d_wide), n) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&d_best), n * 4) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&d_path), n * 4) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&d_counts), 8) == hipSuccess;
    if (ok) {
        hipMemcpy(d_raw, raw, n, hipMemcpyHostToDevice);
        hipMemset(d_wide, 0, n);
        hipLaunchKernelGGL(k_rooms, dim3(1), dim3(64), 0, 0, d_raw, src_sel, dst_sel, d_cave, d_best, d_path, d_wide, d_counts);
        ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    }
    if (ok) {
        hipMemcpy(cave, d_cave, n, hipMemcpyDeviceToHost);
        hipMemcpy(best, d_best, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(path, d_path, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(wide, d_wide, n, hipMemcpyDeviceToHost);
        hipMemcpy(counts, d_counts, 8, hipMemcpyDeviceToHost);
    }
    for (void* p : {static_cast<void*>(d_raw), static_cast<void*>(d_cave), static_cast<void*>(d_wide), static_cast<void*>(d_best),
                    static_cast<void*>(d_path), static_cast<void*>(d_counts)})
        if (p) hipFree(p);
    return ok ? 0 : 1;
}
