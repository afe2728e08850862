// Probe: how fast can the chip READ a buffer that does not fit the 256 MiB Infinity Cache, as a function of HOW the loads are issued?
//   mode 0: global_load_dwordx4 (plain)            U loads in flight per lane, result xor-ed into a register (kept alive)
//   mode 1: global_load_dwordx4 nt                 (what gemv_n / reduce_rows4 / op_assign use)
//   mode 2: global_load_lds_dwordx4                LDS-DMA into a ring of U slots of 1 KiB per wave, never read back (counted vmcnt)
//   mode 3: global_load_lds_dwordx4 nt             (what the few-column f32 kernel's A pieces use since round 5)
//   mode 4: global_load_dwordx4 sc1 nt / mode 5: global_load_lds_dwordx4 sc0 sc1 nt
// Layout: workgroup b of G streams the contiguous slab [b, b + 1) * bytes / G (like a row block of a matrix), wave w its quarter, 1 KiB per instruction.
// build: hipcc --offload-arch=gfx950 -O2 tools/cpp/hbm_read_probe.hip -o tools/cpp/_bin/hbm_read_probe ; run: hbm_read_probe [MiB] [waves per workgroup]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

template <int MODE>
__device__ __forceinline__ uintx4 ld(const char *p) {
    uintx4 v;
    if constexpr (MODE == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if constexpr (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if constexpr (MODE == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int MODE>
__device__ __forceinline__ void dma(uint32_t voff, const char *sbase) {
    if constexpr (MODE == 2) asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase) : "memory");
    if constexpr (MODE == 3) asm volatile("global_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(sbase) : "memory");
    if constexpr (MODE == 5) asm volatile("global_load_lds_dwordx4 %0, %1 sc0 sc1 nt" ::"v"(voff), "s"(sbase) : "memory");
}

// LAYOUT (ordinary loads only): 0 = a contiguous slab per WAVE (1 KiB per instruction); 1 = the workgroup's waves interleave 1 KiB pieces of the workgroup's slab;
// 2 = the whole chip interleaves 1 KiB pieces (piece index = step * waves_total + wave_global); 3 = a contiguous slab per HALF-wave (512 B per half per instruction:
// two streams per wave, as reduce_rows4 reads two vectors); 4 .. 7 = strided: the wave reads 1 / 2 / 4 / 8 KiB of every 16 KiB (a row group of a 4096-row
// column-major matrix: gemv_n with 4 / 8 / 16 / 32 rows per lane)  [1, 2: interleaved layouts, fault at grid 512 for a reason not found: not run]
template <int MODE, int U, int LAYOUT = 0>
__global__ __launch_bounds__(512) void stream(const char *buf, uint64_t slab, uint32_t *sink) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const uint64_t mine = slab / nw;                       // bytes of this wave (a multiple of U KiB: host)
    const uint64_t pp = (uint64_t)(uintptr_t)buf + (uint64_t)blockIdx.x * slab + (uint64_t)wave * mine; // (a scalar: the DMA takes its base from SGPRs)
    const char *p = (const char *)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(pp >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pp));
    const uint64_t iters = mine / (1024u * U);
    if constexpr (MODE == 0 || MODE == 1 || MODE == 4) {
        uintx4 acc = { 0u, 0u, 0u, 0u };
        const char *q = p + lane * 16;
        uint64_t piece = 1024, step = 1024 * U; // distance between the U loads of a step / between steps
        if constexpr (LAYOUT == 1) { q = buf + (uint64_t)blockIdx.x * slab + (uint64_t)wave * 1024u + lane * 16; piece = 1024u * nw; step = piece * U; }
        if constexpr (LAYOUT == 2) { q = buf + ((uint64_t)blockIdx.x * nw + wave) * 1024u + lane * 16; piece = 1024ull * nw * gridDim.x; step = piece * U; }
        if constexpr (LAYOUT == 3) { q = p + (uint64_t)(lane >> 5) * (mine / 2) + (lane & 31) * 16; piece = 512; step = 512 * U; }
        constexpr int R = LAYOUT >= 4 ? (1 << (LAYOUT - 4)) : 1; // layouts 4 .. 7: the wave reads R KiB (R consecutive 1 KiB pieces) of every 16 KiB column, U / R columns per step
        if constexpr (LAYOUT >= 4) {
            const uint64_t groups = 16u / R;                            // row groups of a column
            const uint64_t g = (uint64_t)blockIdx.x * nw + wave;       // global wave: row group g % groups, column range g / groups
            const uint64_t ranges = (uint64_t)gridDim.x * nw / groups;
            const uint64_t cols_per = (slab * gridDim.x / 16384u) / ranges;
            q = buf + (g / groups) * cols_per * 16384u + (g % groups) * 1024u * R + lane * 16;
            (void)cols_per;
        }
        const uint64_t n_it = LAYOUT == 3 ? mine / 2 / (512u * U) : iters;
        for (uint64_t i = 0; i < n_it; ++i) {
            uintx4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if constexpr (LAYOUT >= 4) v[u] = ld<MODE>(q + (uint64_t)(u / R) * 16384u + (uint64_t)(u % R) * 1024u);
                else v[u] = ld<MODE>(q + u * piece);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u];
            if constexpr (LAYOUT >= 4) q += 16384ull * (U / R); else q += step;
        }
        if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = 1;
    } else {
        const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem + wave * (1024u * 2u * U));
        const uint32_t voff = lane * 16;
        // two half-rings of U pieces: issue a half, wait for the older half
        for (uint64_t i = 0; i < iters; ++i) {
            const uint32_t base = lds0 + (uint32_t)(i & 1) * 1024u * U;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(base + 1024u * u));
                dma<MODE>(voff, p + u * 1024);
            }
            asm volatile("s_waitcnt vmcnt(%c0)" ::"i"(U) : "memory");
            p += 1024 * U;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (smem[threadIdx.x] == 77 && smem[threadIdx.x + 512] == 78) sink[1] = 1;
    }
}

template <int MODE, int U, int LAYOUT = 0>
static int run(const char *name, const char *buf, uint64_t bytes, int grid, int waves, uint32_t *sink) {
    const uint64_t unit = (uint64_t)grid * waves * 1024u * U;
    const uint64_t use = bytes / unit * unit, slab = use / grid;
    const size_t lds = (MODE == 2 || MODE == 3 || MODE == 5) ? (size_t)waves * 2048u * U : 1024;
    CK(hipFuncSetAttribute((const void *)stream<MODE, U, LAYOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int it = 0; it < 7; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((stream<MODE, U, LAYOUT>), dim3(grid), dim3(64 * waves), lds, 0, buf, slab, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (it >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("%-34s U=%2d grid=%5d waves=%d: %8.1f GB/s (median of 5, %.3f ms, %.0f MiB)\n", name, U, grid, waves, use / (ms[2] * 1e-3) / 1e9, ms[2], use / 1048576.0);
    fflush(stdout);
    return 0;
}

int main(int argc, char **argv) {
    const uint64_t bytes = (uint64_t)(argc > 1 ? atoi(argv[1]) : 2048) << 20;
    const int waves = argc > 2 ? atoi(argv[2]) : 4;
    char *buf; uint32_t *sink;
    CK(hipMalloc((void **)&buf, bytes)); CK(hipMalloc((void **)&sink, 64));
    CK(hipMemset(buf, 1, bytes)); CK(hipMemset(sink, 0, 64));
    if (argc > 3) { // layouts of the ordinary nt load
        for (int grid : { 256, 512, 1024, 2048 }) {
            if (run<1, 8, 0>("nt, slab per wave", buf, bytes, grid, waves, sink)) return 1;
            if (run<1, 16, 3>("nt, slab per half-wave (512 B)", buf, bytes, grid, waves, sink)) return 1;
            if (run<1, 8, 4>("nt, 1 KiB of every 16 KiB column", buf, bytes, grid, waves, sink)) return 1;
            if (run<1, 8, 5>("nt, 2 KiB of every 16 KiB column", buf, bytes, grid, waves, sink)) return 1;
            if (run<1, 8, 6>("nt, 4 KiB of every 16 KiB column", buf, bytes, grid, waves, sink)) return 1;
            if (run<1, 16, 6>("nt, 4 KiB of every 16 KiB column", buf, bytes, grid, waves, sink)) return 1;
            if (run<1, 8, 7>("nt, 8 KiB of every 16 KiB column", buf, bytes, grid, waves, sink)) return 1;
            if (run<1, 16, 7>("nt, 8 KiB of every 16 KiB column", buf, bytes, grid, waves, sink)) return 1;
        }
        return 0;
    }
    for (int grid : { 256, 512, 1024, 4096 }) {
        if (run<0, 8>("global_load_dwordx4", buf, bytes, grid, waves, sink)) return 1;
        if (run<1, 8>("global_load_dwordx4 nt", buf, bytes, grid, waves, sink)) return 1;
        if (run<1, 16>("global_load_dwordx4 nt", buf, bytes, grid, waves, sink)) return 1;
        if (run<4, 8>("global_load_dwordx4 sc1 nt", buf, bytes, grid, waves, sink)) return 1;
        if (run<2, 8>("global_load_lds_dwordx4", buf, bytes, grid, waves, sink)) return 1;
        if (run<3, 8>("global_load_lds_dwordx4 nt", buf, bytes, grid, waves, sink)) return 1;
        if (run<3, 16>("global_load_lds_dwordx4 nt", buf, bytes, grid, waves, sink)) return 1;
        if (run<5, 8>("global_load_lds_dwordx4 sc0 sc1 nt", buf, bytes, grid, waves, sink)) return 1;
    }
    return 0;
}
