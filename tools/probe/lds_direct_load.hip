#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint32_t* tex, uint32_t bytes, uint32_t* out) {
    __shared__ uint32_t fb[64 * 8];
    const int lane = threadIdx.x;
    for (int r = 0; r < 8; r++) fb[r * 64 + lane] = 0xAAAA0000u + r * 64 + lane;  // stale pattern
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(tex), 0, (int)bytes, 0x00020000);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        uint32_t off = (uint32_t)((lane * 13 + r * 7) % 1000) * 4u;
        if ((lane + r) % 5 == 0) off = 0x40000000u + lane * 4;  // out of range
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)&fb[r * 64], 4, off, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int r = 0; r < 8; r++) out[r * 64 + lane] = fb[r * 64 + lane];
}
int main() {
    std::vector<uint32_t> h(1000);
    for (int i = 0; i < 1000; i++) h[i] = 0xFF000000u | i;
    uint32_t *d, *o;
    hipMalloc(&d, 4000); hipMalloc(&o, 512 * 4);
    hipMemcpy(d, h.data(), 4000, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 4000u, o);
    std::vector<uint32_t> r(512);
    hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
    int bad = 0, oob_zero = 0, oob_stale = 0, oob_other = 0;
    for (int row = 0; row < 8; row++) for (int lane = 0; lane < 64; lane++) {
        uint32_t v = r[row * 64 + lane];
        if ((lane + row) % 5 == 0) { if (v == 0) oob_zero++; else if (v == 0xAAAA0000u + row * 64 + lane) oob_stale++; else oob_other++; }
        else if (v != (0xFF000000u | ((lane * 13 + row * 7) % 1000))) bad++;
    }
    printf("in-range mismatches %d; out-of-range lanes: zero %d, stale %d, other %d\n", bad, oob_zero, oob_stale, oob_other);
    return 0;
}
