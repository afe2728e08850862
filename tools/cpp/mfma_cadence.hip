// Premise check for an 8-wave (two waves per SIMD) f16 Gemm loop (experiment (c)): does a second wave on the SIMD bring the
// v_mfma_f32_16x16x32_f16 cadence from ~17 cycles (one wave alone, in-order) down to the pipe's 16, and does it hide the issue cost of
// the LDS-DMA / ds_read instructions that the one-wave loop pays for (profiles/r02_evidence.md section 3b)?
// One workgroup per CU (160 KiB of LDS claimed), 256 or 512 threads, every wave: REPS x [ NM independent MFMAs, with one `extra` instruction
// after every EVERY-th MFMA ]. extra: 0 none, 1 ds_read_b128, 2 global_load_lds_dwordx4 (1 KiB from an L2-resident buffer), 3 both alternating.
// Reports shader cycles (s_memtime) per MFMA per SIMD.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/cpp/mfma_cadence.hip -o /tmp/mfma_cadence
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NACC, int EXTRA, int EVERY, int THREADS>
__global__ __launch_bounds__(THREADS) void cadence(const char *src, unsigned long long *out, int reps) {
    __shared__ __attribute__((aligned(16))) char smem[150 * 1024];
    floatx4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = floatx4{ 0.f, 0.f, 0.f, 0.f };
    half8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(float)(threadIdx.x & 3); b[i] = (_Float16)1.f; }
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned lds_rd = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem + wave * 4096u + lane * 16u;
    const unsigned lds_wr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem + 64u * 1024u + wave * 8192u);
    const unsigned voff = lane * 16u + wave * 1024u;
    half8 sink = b;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
            if (EXTRA && (i % EVERY) == EVERY - 1) {
                const int k = i / EVERY;
                if (EXTRA == 1 || (EXTRA == 3 && (k & 1))) {
                    half8 v;
                    asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(v) : "v"(lds_rd), "i"((k % 4) * 1024));
                    sink = v; // (never waited for inside the loop: lgkmcnt(0) after it)
                } else {
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%c3" ::"v"(voff), "s"(src), "s"(lds_wr), "i"((k % 4) * 1024));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = (float)sink[0];
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0];
    if (s == 12345.678f) out[1] = 1; // keep everything alive
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { out[2 + 2 * wave] = t0; out[3 + 2 * wave] = t1; } // span over ALL waves: the SIMD favours its older wave
}

template <int NACC, int EXTRA, int EVERY, int THREADS>
static void run(const char *name, const char *src, unsigned long long *out) {
    const int reps = 2000, threads = THREADS;
    hipLaunchKernelGGL((cadence<NACC, EXTRA, EVERY, THREADS>), dim3(256), dim3(threads), 0, 0, src, out, 10);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((cadence<NACC, EXTRA, EVERY, THREADS>), dim3(256), dim3(threads), 0, 0, src, out, reps);
    hipDeviceSynchronize();
    unsigned long long h[32] = {};
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < threads / 64; ++w) { if (h[2 + 2 * w] < lo) lo = h[2 + 2 * w]; if (h[3 + 2 * w] > hi) hi = h[3 + 2 * w]; }
    const unsigned long long cyc = hi - lo;
    const int waves_per_simd = threads / 256;
    const double per_mfma_per_simd = (double)cyc / ((double)reps * NACC * waves_per_simd);
    printf("%-46s %d wave(s)/SIMD, %3d acc tiles/wave: %6.2f cycles per MFMA per SIMD (%.0f %% of the pipe's 16)\n", name, waves_per_simd, NACC, per_mfma_per_simd,
           1600.0 / per_mfma_per_simd);
}

int main() {
    char *src;
    unsigned long long *out;
    hipMalloc(&src, 1 << 20);
    hipMemset(src, 0, 1 << 20);
    hipMalloc(&out, 256);
    hipMemset(out, 0, 256);
    // one wave per SIMD, 64 accumulator tiles (the shipped kernel's shape); two waves per SIMD, 32 tiles each (an 8-wave kernel's shape)
    run<64, 0, 1, 256>("bare MFMA stream", src, out);
    run<32, 0, 1, 512>("bare MFMA stream", src, out);
    run<64, 1, 4, 256>("ds_read_b128 every 4 MFMAs", src, out);
    run<32, 1, 4, 512>("ds_read_b128 every 4 MFMAs", src, out);
    run<64, 1, 2, 256>("ds_read_b128 every 2 MFMAs", src, out);
    run<32, 1, 2, 512>("ds_read_b128 every 2 MFMAs", src, out);
    run<64, 2, 8, 256>("LDS-DMA piece every 8 MFMAs", src, out);
    run<32, 2, 8, 512>("LDS-DMA piece every 8 MFMAs", src, out);
    run<64, 2, 4, 256>("LDS-DMA piece every 4 MFMAs", src, out);
    run<32, 2, 4, 512>("LDS-DMA piece every 4 MFMAs", src, out);
    run<64, 3, 2, 256>("ds_read / LDS-DMA alternating every 2 MFMAs", src, out);
    run<32, 3, 2, 512>("ds_read / LDS-DMA alternating every 2 MFMAs", src, out);
    return 0;
}
