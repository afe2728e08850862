// Probe: does the LDS-DMA form of the 16-byte load (global_load_lds_dwordx4: global memory -> LDS without passing the registers, the path the GEMM kernels
// bring their operands in by) take global addresses that are only element-aligned? One wave per workgroup moves 1 KiB pieces that start `off` bytes past a
// 16-byte boundary into the LDS, reads them back with ds_read_b128 and sums; the sum is checked against the host's and the rate printed.
// build: hipcc --offload-arch=gfx950 -O2 tools/cpp/unaligned_dma_probe.hip -o tools/cpp/_bin/unaligned_dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void dma_sum(const char *p, uint64_t n_kib, unsigned long long *out) {
    __shared__ u4 buf[4][4][64]; // per wave: 4 pieces of 1 KiB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long s = 0;
    const uint32_t lds0 = (uint32_t)(uintptr_t)&buf[wave][0][0];
    for (uint64_t k = (blockIdx.x * 4ull + wave) * 4ull; k + 3 < n_kib; k += (uint64_t)gridDim.x * 16ull) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const char *base = p + (k + q) * 1024ull;
            const uint32_t voff = 16u * lane;
            const uint32_t ldsq = __builtin_amdgcn_readfirstlane(lds0 + 1024u * q);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(ldsq) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u4 v = buf[wave][q][lane];
            s += (unsigned long long)v.x + v.y + v.z + v.w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) atomicAdd(out, s);
}

int main() {
    const uint64_t bytes = 1ull << 30, n_kib = bytes / 1024 - 16;
    std::vector<uint16_t> h(bytes / 2);
    uint32_t x = 12345;
    for (auto &e : h) { x = x * 1664525u + 1013904223u; e = (uint16_t)(x >> 16); }
    char *d = nullptr;
    unsigned long long *out = nullptr;
    CK(hipMalloc((void **)&d, bytes)); CK(hipMalloc((void **)&out, 8));
    CK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint64_t used_kib = (n_kib / (2048 * 16)) * (2048 * 16); // whole grid strides only
    for (int off : { 0, 2, 4, 6, 8, 12, 14 }) {
        unsigned long long want = 0, got = 0;
        const uint16_t *hp = h.data() + off / 2;
        for (uint64_t i = 0; i < used_kib * 512; i += 2) want += (unsigned long long)hp[i] + ((unsigned long long)hp[i + 1] << 16);
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemset(out, 0, 8));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(dma_sum, dim3(2048), dim3(256), 0, 0, d + off, used_kib, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        CK(hipMemcpy(&got, out, 8, hipMemcpyDeviceToHost));
        printf("lds-dma off %2d: %s  %.1f us  %.0f GB/s\n", off, got == want ? "sum OK " : "SUM WRONG", best * 1e3, used_kib * 1024.0 / (best * 1e-3) / 1e9);
    }
    return 0;
}
