// Probe: where do the watts of an f16 Gemm go? The chip is pinned at its package power cap on this kernel, so throughput is set by the
// energy per tile, not by issue slots. This probe runs the Gemm's MFMA stream (v_mfma_f32_16x16x32_f16, 64 per "half-step" per wave, 4 waves
// per CU, random operands) and adds the Gemm's other activities one at a time, at the Gemm's rates, and reports the sustained TFLOP/s:
//   bit 0: fragment reads -- 16 ds_read_b128 per half-step per wave (8 A + 8 B), the values feed the next half-step's MFMAs
//   bit 1: LDS-DMA        -- 8 global_load_lds_dwordx4 (1 KiB each) per half-step per wave = 32 KiB per CU per half-step, from a window of
//                            `span` bytes per XCD-sized group of workgroups (small: L2 hits; large: Infinity Cache / HBM)
//   bit 2: a barrier per half-step
//   bit 3: (with bit 0) the NN kernel's form of the fragment reads
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/cpp/gemm_energy_probe.hip -o tools/cpp/_bin/gemm_energy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
#define AS3 __attribute__((address_space(3)))
#include <utility>
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void fill_random(_Float16 *p, size_t n, int zero) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = zero ? (_Float16)0.f : (_Float16)((float)(mix((uint32_t)i * 2654435761u + (uint32_t)(i >> 32)) >> 8) * (2.f / 16777216.f) - 1.f);
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void burn(const char *src, uint64_t span_mask, float *out, int reps) {
    extern __shared__ __attribute__((aligned(16))) char smem[]; // 160 KiB: [0, 64 Ki) read area, [64 Ki, 160 Ki) DMA landing area
    const uint32_t lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(AS3 char *)smem;
    // fill the read area with random halves (from the global buffer)
    for (uint32_t o = threadIdx.x * 16u; o < 64u * 1024u; o += 256u * 16u)
        *(AS3 uintx4 *)(uintptr_t)(lds0 + o) = *reinterpret_cast<const uintx4 *>(src + (blockIdx.x * 65536u + o));
    __syncthreads();
    half8 a[2][8], b[2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[0][i] = *(AS3 half8 *)(uintptr_t)(lds0 + wave * 8192u + i * 1024u + lane * 16u);
        b[0][i] = *(AS3 half8 *)(uintptr_t)(lds0 + 32768u + wave * 8192u + i * 1024u + lane * 16u);
        a[1][i] = a[0][i]; b[1][i] = b[0][i];
    }
    floatx4 acc[8][8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[t][u] = floatx4{ 0.f, 0.f, 0.f, 0.f };
    // DMA streams: the 32 workgroups of an XCD (ids equal mod 8) are a 4 x 8 patch of tiles -- workgroup j reads A-panel stream j & 3 (shared by 8)
    // and B-panel stream 4 + (j >> 2) (shared by 4), 16 KiB of each per half-step: 12 streams per XCD for 32 tiles, L2 hit rate ~0.8 like the Gemm.
    // A stream walks `span` bytes and wraps: 96 streams x span small -> everything stays in L2; <= 2 MiB -> Infinity Cache; 64 MiB -> HBM.
    const uint32_t xcd = blockIdx.x & 7u, local = (blockIdx.x >> 3) & 31u;
    const char *sa = src + ((uint64_t)(xcd * 12u + (local & 3u)) << 26), *sb = src + ((uint64_t)(xcd * 12u + 4u + (local >> 2)) << 26);
    uint64_t pos = 0;
    const uint32_t voff = lane * 16u + wave * 4096u;
    uint32_t rd = 0;
    uint32_t va = lds0 + wave * 8192u + lane * 16u, vb = va + 32768u;
    const uint32_t lds_dma = __builtin_amdgcn_readfirstlane(lds0 + 65536u + wave * 8192u);
    // slot-pinned like the Gemm's loop: one MFMA per slot, at most one other instruction (pair) behind it, the order fixed by sched_barrier
    auto half_step = [&](auto hs_c, int r) {
        constexpr int hs = decltype(hs_c)::value;
        const char *ba = sa + (pos & span_mask), *bb = sb + (pos & span_mask);
        static_for<64>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            constexpr int t = j >> 3, u = j & 7;
            acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[hs][t], b[hs][u], acc[t][u], 0, 0, 0);
            if constexpr ((MODE & 9) == 1 && (j % 3) == 0 && j / 3 < 16) { // 16 fragment reads, one every third slot (the TN schedule)
                constexpr int i = j / 3;
                if constexpr (i < 8) a[hs ^ 1][i] = *(AS3 half8 *)(uintptr_t)(va + i * 1024u);
                else b[hs ^ 1][i - 8] = *(AS3 half8 *)(uintptr_t)(vb + (i - 8) * 1024u);
            }
            if constexpr ((MODE & 9) == 9 && (j & 3) != 3 && 3 * (j >> 2) + (j & 3) < 40) { // the NN schedule: 16 transposing 8-byte reads of A, 8 reads of B, 16 lane swaps
                constexpr int op = 3 * (j >> 2) + (j & 3);
                if constexpr (op < 16) {
                    uintx2 v;
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%c2" : "=v"(v) : "v"(va), "i"(op * 512));
                    uintx4 &d = reinterpret_cast<uintx4 &>(a[hs ^ 1][op >> 1]);
                    d[2 * (op & 1)] = v[0]; d[2 * (op & 1) + 1] = v[1];
                } else if constexpr (op < 24) b[hs ^ 1][op - 16] = *(AS3 half8 *)(uintptr_t)(vb + (op - 16) * 1024u);
                else {
                    constexpr int w = op - 24, p2 = w >> 2, i2 = w & 3;
                    uintx4 &x = reinterpret_cast<uintx4 &>(a[hs ^ 1][2 * p2]), &y = reinterpret_cast<uintx4 &>(a[hs ^ 1][2 * p2 + 1]);
                    const uintx2 rr = __builtin_amdgcn_permlane16_swap(x[i2], y[i2], false, false);
                    x[i2] = rr[0]; y[i2] = rr[1];
                }
            }
            if constexpr ((MODE & 2) && (j & 7) == 2) { // 8 DMA pieces, M0 one slot ahead
                constexpr int q = j >> 3;
                const uint32_t m0 = lds_dma + (uint32_t)q * 1024u + (uint32_t)(r & 1) * 32768u;
                asm volatile("s_mov_b32 m0, %0" ::"s"(m0));
            }
            if constexpr ((MODE & 2) && (j & 7) == 3) {
                constexpr int q = j >> 3;
                asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(voff + (uint32_t)(q & 3) * 1024u), "s"(q < 4 ? ba : bb) : "memory");
            }
            if constexpr (j == 50) { rd = (rd + 16384u) & 0xffffu; asm volatile("" : "+s"(rd)); }
            if constexpr (j == 51) { va = lds0 + ((rd + wave * 8192u) & 0xffffu) + lane * 16u; asm volatile("" : "+v"(va)); }
            if constexpr (j == 52) { vb = lds0 + ((rd + 32768u + wave * 8192u) & 0xffffu) + lane * 16u; asm volatile("" : "+v"(vb)); }
            if constexpr (j == 53) { pos += 16384u; asm volatile("" : "+s"(pos)); }
            if constexpr (j == 58) {
                __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
                if constexpr (MODE & 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                if constexpr (MODE & 4) __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    for (int r = 0; r < reps; ++r) {
        half_step(std::integral_constant<int, 0>{}, r);
        half_step(std::integral_constant<int, 1>{}, r);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int u = 0; u < 8; ++u) s += acc[t][u][0] + acc[t][u][1] + acc[t][u][2] + acc[t][u][3];
    if (s == 123.456f) out[blockIdx.x * 256u + threadIdx.x] = s;
}

template <int MODE>
int run(const char *src, uint64_t span, float *out, int reps, const char *label) {
    CK(hipFuncSetAttribute((const void *)burn<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    burn<MODE><<<256, 256, 160 * 1024>>>(src, span - 1, out, reps / 8);
    CK(hipDeviceSynchronize());
    double sum = 0; const int N = 4; float ms = 0;
    for (int i = 0; i < N; ++i) {
        CK(hipEventRecord(e0));
        burn<MODE><<<256, 256, 160 * 1024>>>(src, span - 1, out, reps);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        sum += 256.0 * 4.0 * reps * 2.0 * 64.0 * 2.0 * 16 * 16 * 32 / (ms * 1e-3) / 1e12;
    }
    printf("%-64s %6.1f ms: %7.1f TFLOP/s\n", label, ms, sum / N);
    return 0;
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 30000;
    const size_t bytes = 96ull << 26; // 96 streams x 64 MiB
    char *src; CK(hipMalloc((void **)&src, bytes));
    const int zero = argc > 2 && atoi(argv[2]) != 0; // zero operands: the clock stays at its ceiling, the numbers show each mode's ISSUE efficiency
    printf("operands: %s\n", zero ? "zeros" : "uniform random in [-1, 1)");
    fill_random<<<4096, 256>>>((_Float16 *)src, bytes / 2, zero);
    float *out; CK(hipMalloc((void **)&out, 256 * 256 * 4));
    CK(hipDeviceSynchronize());
    if (run<0>(src, 1ull << 20, out, reps, "MFMA only")) return 1;
    if (run<1>(src, 1ull << 20, out, reps, "+ fragment reads")) return 1;
    if (run<9>(src, 1ull << 20, out, reps, "+ fragment reads, NN form (16 tr reads, 8 reads, 16 lane swaps)")) return 1;
    if (run<4>(src, 1ull << 20, out, reps, "+ barrier")) return 1;
    if (run<2>(src, 1ull << 15, out, reps, "+ LDS-DMA, streams wrap at 32 KiB (all L2 hits)")) return 1;
    if (run<2>(src, 1ull << 21, out, reps, "+ LDS-DMA, streams wrap at 2 MiB (L2 0.8, rest Infinity Cache)")) return 1;
    if (run<2>(src, 1ull << 26, out, reps, "+ LDS-DMA, streams wrap at 64 MiB (L2 0.8, rest HBM)")) return 1;
    if (run<3>(src, 1ull << 21, out, reps, "+ fragment reads + LDS-DMA (Infinity Cache)")) return 1;
    if (run<7>(src, 1ull << 21, out, reps, "+ fragment reads + LDS-DMA (Infinity Cache) + barrier")) return 1;
    if (run<7>(src, 1ull << 26, out, reps, "+ fragment reads + LDS-DMA (HBM) + barrier")) return 1;
    if (run<15>(src, 1ull << 21, out, reps, "+ NN fragment reads + LDS-DMA (Infinity Cache) + barrier")) return 1;
    if (run<0>(src, 1ull << 20, out, reps, "MFMA only (again)")) return 1;
    return 0;
}
