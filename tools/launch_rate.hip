// Micro-benchmark: what does it cost to dispatch the render kernel's grid, independent of the work?
// Empty kernels that declare the render kernel's LDS footprint, in the shapes that could carry 65 536 envs:
//   65536 x 128 (today: one workgroup of two wavefronts per env), 32768 x 256 (two envs per workgroup),
//   16384 x 512, 131072 x 64.  Prints the average launch duration of each (HIP events, 200 launches).
#include <hip/hip_runtime.h>

#include <cstdio>

template <int THREADS, int LDS_WORDS>
__global__ void __launch_bounds__(THREADS) shell(uint32_t* sink) {
    __shared__ uint32_t lds[LDS_WORDS];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (lds[(threadIdx.x + 1) % THREADS] == 0xdeadbeef) sink[blockIdx.x] = 1;  // never true; keeps the LDS alive
}

template <int THREADS, int LDS_WORDS>
static void run(const char* tag, int blocks, uint32_t* sink) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int k = 0; k < 20; k++) hipLaunchKernelGGL((shell<THREADS, LDS_WORDS>), dim3(blocks), dim3(THREADS), 0, 0, sink);
    hipEventRecord(a, 0);
    for (int k = 0; k < 200; k++) hipLaunchKernelGGL((shell<THREADS, LDS_WORDS>), dim3(blocks), dim3(THREADS), 0, 0, sink);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    printf("%-28s %7d x %3d threads, %5d B LDS: %.4f ms per launch\n", tag, blocks, THREADS, LDS_WORDS * 4, ms / 200);
}

int main() {
    uint32_t* sink;
    hipMalloc(&sink, 1 << 20);
    constexpr int kEnvLds = 4096 + 1200;  // 16 KiB frame + composer tables, in words
    run<128, kEnvLds>("one env per workgroup", 65536, sink);
    run<256, 2 * kEnvLds>("two envs per workgroup", 32768, sink);
    run<512, 4 * kEnvLds>("four envs per workgroup", 16384, sink);
    run<64, kEnvLds>("one wave per env", 65536, sink);
    run<128, 64>("128 threads, no LDS", 65536, sink);
    run<256, 64>("256 threads, no LDS", 32768, sink);
    return 0;
}
