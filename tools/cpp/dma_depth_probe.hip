// Probe: how much LDS-DMA (global_load_lds_dwordx4) does ONE workgroup of W waves keep in flight, and what rate does the CU get out of an L2 / Infinity-Cache
// resident working set as a function of the pieces (1 KiB each) a wave issues before it waits? One workgroup per CU (256 x W waves), working set 32 MiB re-read.
// build: hipcc --offload-arch=gfx950 -O2 tools/cpp/dma_depth_probe.hip -o tools/cpp/_bin/dma_depth_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int Q>
__global__ void probe(const char *p, uint64_t ws_kib, int iters, unsigned *sink) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds + (uint32_t)wave * Q * 1024u;
    uint64_t k = ((uint64_t)blockIdx.x * nw + wave) * Q;
    const uint64_t stride = (uint64_t)gridDim.x * nw * Q;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const char *base = p + ((k + q) % ws_kib) * 1024ull;
            const uint32_t ldsq = __builtin_amdgcn_readfirstlane(lds0 + 1024u * q);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(16u * lane), "s"(base), "s"(ldsq) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        k += stride;
    }
    if (threadIdx.x == 0 && iters < 0) sink[0] = lds[0];
}

template <int Q>
int run(const char *d, uint64_t ws_kib, int waves, unsigned *sink) {
    const int iters = 4096 / Q;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t shm = (size_t)waves * Q * 1024;
    CK(hipFuncSetAttribute((const void *)probe<Q>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<Q>, dim3(256), dim3(64 * waves), shm, 0, d, ws_kib, iters, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    const double bytes = 256.0 * waves * Q * 1024.0 * iters;
    printf("waves %d  pieces in flight per wave %2d (%3d KiB per CU): %.1f us  %.0f GB/s  (%.1f GB/s per CU)\n", waves, Q, waves * Q, best * 1e3, bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9 / 256);
    return 0;
}

int main() {
    const uint64_t bytes = 32ull << 20;
    char *d = nullptr; unsigned *sink = nullptr;
    CK(hipMalloc((void **)&d, bytes)); CK(hipMemset(d, 1, bytes)); CK(hipMalloc((void **)&sink, 4));
    for (int waves : { 4, 8, 16 }) {
        if (run<1>(d, bytes / 1024, waves, sink)) return 1;
        if (run<2>(d, bytes / 1024, waves, sink)) return 1;
        if (run<4>(d, bytes / 1024, waves, sink)) return 1;
        if (run<8>(d, bytes / 1024, waves, sink)) return 1;
        if (waves * 16 <= 160 && run<16>(d, bytes / 1024, waves, sink)) return 1;
        if (waves * 32 <= 160 && run<32>(d, bytes / 1024, waves, sink)) return 1;
    }
    return 0;
}
