// Probe: how fast does ONE workgroup (4 waves, one per SIMD, alone on its CU) get a 256 x 256 f16 tile (128 KiB) out of its registers into a
// column-major matrix, as a function of which bytes one global_store_dwordx4 instruction of a wave covers?
//   pattern 0: 16 columns x 64 bytes  (the f16 Gemm epilogue: lane (i16, kg) -> column i16, rows 8 kg .. 8 kg + 7 of a 32-row group)
//   pattern 1:  8 columns x 128 bytes (whole cache lines)
//   pattern 2:  4 columns x 256 bytes
//   pattern 3:  2 columns x 512 bytes (a whole tile column per 32 lanes)
//   pattern 4:  8 columns x 128 bytes like 1, but each line's pieces on lanes 16 apart (see below)
//   pattern 5: 16 columns x 64 bytes like 0, but a column's four pieces on four consecutive lanes
// flavour 0: plain, 1: nt, 2: sc1 nt, 3: sc0 sc1 (write-through)
// build: hipcc --offload-arch=gfx950 -O2 tools/cpp/store_probe.hip -o tools/cpp/_bin/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

template <int FL>
__device__ __forceinline__ void st(char *p, uintx4 v) {
    if constexpr (FL == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if constexpr (FL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if constexpr (FL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    if constexpr (FL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

template <int PAT, int FL>
__global__ __launch_bounds__(256) void probe(char *C, uint32_t ldc_bytes, uint32_t tiles_m, uint64_t *out) {
    extern __shared__ char smem[]; // 160 KiB requested: one workgroup per CU
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    char *tile = C + (uint64_t)tn * 256u * ldc_bytes + (uint64_t)tm * 512u;
    uintx4 v = { (uint32_t)threadIdx.x, blockIdx.x, 0x3c003c00u, 0x3c003c00u };
    if (threadIdx.x == 0) smem[0] = 1;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    // every wave stores 32 KiB = 32 instructions; the wave's quadrant as in the Gemm: rows 128 wm.., columns 128 wn..
    const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        uint32_t col, rowb; // column within the tile, byte offset of the 16-byte piece within the column
        if constexpr (PAT == 0) { const int u = i >> 2, p = i & 3; col = 128 * wn + 16 * u + (lane & 15); rowb = (128 * wm + 32 * p + 8 * (lane >> 4)) * 2; }
        if constexpr (PAT == 1) { const int u = i >> 1, p = i & 1; col = 128 * wn + 8 * u + (lane >> 3); rowb = (128 * wm + 64 * p + 8 * (lane & 7)) * 2; }
        if constexpr (PAT == 2) { col = 128 * wn + 4 * i + (lane >> 4); rowb = (128 * wm + 8 * (lane & 15)) * 2; }
        if constexpr (PAT == 3) { col = 64 * wave + 2 * i + (lane >> 5); rowb = (8 * (lane & 31)) * 2; }
        // pattern 4: the same bytes per instruction as pattern 1, but a line's 8 pieces on lanes kg * 16 + half * 8 + j (what a DPP row_ror:8 exchange of the MFMA layout gives)
        if constexpr (PAT == 4) { const int u = i >> 1, pp = i & 1; col = 128 * wn + 16 * (u >> 1) + 8 * (u & 1) + (lane & 7); rowb = (128 * wm + 64 * pp + 32 * ((lane >> 3) & 1) + 8 * (lane >> 4)) * 2; }
        // pattern 5 (round 6): the SAME bytes per instruction as pattern 0 (16 columns x 64 bytes), but a column's four pieces on four CONSECUTIVE lanes (lane = 4 c + piece):
        // what one ds_bpermute_b32 per dword of the MFMA layout would give
        if constexpr (PAT == 5) { const int u = i >> 2, p = i & 3; col = 128 * wn + 16 * u + (lane >> 2); rowb = (128 * wm + 32 * p + 8 * (lane & 3)) * 2; }
        v.x += i;
        st<FL>(tile + (uint64_t)col * ldc_bytes + rowb, v);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint64_t t2 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) { out[(blockIdx.x * 4 + wave) * 2] = t1 - t0; out[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
}

template <int PAT, int FL>
int run(char *C, uint32_t n, int wgs, uint64_t *out_d, std::vector<uint64_t> &out_h) {
    CK(hipFuncSetAttribute((const void *)probe<PAT, FL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    double best_issue = 1e9, best_done = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        probe<PAT, FL><<<wgs, 256, 160 * 1024>>>(C, n * 2, n / 256, out_d);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out_h.data(), out_d, (size_t)wgs * 4 * 2 * 8, hipMemcpyDeviceToHost));
        double si = 0, sd = 0;
        for (int w = 0; w < wgs; ++w) {
            uint64_t mi = 0, md = 0;
            for (int k = 0; k < 4; ++k) { mi = std::max(mi, out_h[(w * 4 + k) * 2]); md = std::max(md, out_h[(w * 4 + k) * 2 + 1]); }
            si += mi; sd += md;
        }
        best_issue = std::min(best_issue, si / wgs / 100.0); best_done = std::min(best_done, sd / wgs / 100.0);
    }
    printf("pattern %d flavour %d, %4d workgroups: issue %.2f us, acknowledged %.2f us per tile (mean over workgroups of the slowest wave; best of 5)\n", PAT, FL, wgs, best_issue, best_done);
    return 0;
}

int main() {
    const uint32_t n = 8192;
    char *C; CK(hipMalloc((void **)&C, (size_t)n * n * 2));
    uint64_t *out_d; CK(hipMalloc((void **)&out_d, 1024 * 4 * 2 * 8));
    std::vector<uint64_t> out_h(1024 * 4 * 2);
    for (int wgs : { 32, 256, 1024 }) {
#define R(P, F) if (run<P, F>(C, n, wgs, out_d, out_h)) return 1;
        R(0, 0) R(5, 0) R(1, 0) R(4, 0) R(2, 0) R(3, 0)
        R(0, 2) R(1, 2) R(2, 2) R(3, 2)
        R(0, 1) R(1, 1) R(0, 3) R(1, 3)
    }
    return 0;
}
