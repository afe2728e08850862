// L2 -> CU fill rate on gfx950: LDS-DMA (global_load_lds_dwordx4) against plain vector loads into registers, same addresses, data resident
// in L2 / Infinity Cache. Each workgroup (256 threads) fetches 16 KiB per iteration (what one half-step of the 128 x 128 f16 GEMM tile
// needs) from a window of `span` bytes it shares with the other workgroups. Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/cpp/fill_probe.hip -o gpurun_out/fill_probe && gpurun_out/fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE, int DEPTH> // MODE 0: LDS-DMA, 1: registers; DEPTH iterations in flight
__global__ __launch_bounds__(256) void fill(const char *base, uint64_t span, int iters, float *sink) {
    __shared__ __attribute__((aligned(16))) char smem[DEPTH * 16384];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    float acc = 0.f;
    uint64_t off = ((uint64_t)blockIdx.x * 7919u * 16384u) % span;
    for (int it = 0; it < iters; ++it) {
        const char *p = base + off + wave * 4096 + lane * 16;
        if (MODE == 0) {
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + (it % DEPTH) * 16384 + wave * 4096);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(dst));
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("global_load_lds_dwordx4 %0, off offset:0" ::"v"(p + q * 1024) : "memory");
            if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%c0)" ::"i"(4 * (DEPTH - 1)) : "memory");
        } else {
            float4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4 *>(p + q * 1024);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += v[q].x + v[q].w;
        }
        off += 16384u * 131u;
        if (off >= span) off -= span;
    }
    if (MODE == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); acc = ((float *)smem)[threadIdx.x]; }
    if (acc == 123.456f) sink[0] = acc;
}

template <int MODE, int DEPTH>
double run(const char *buf, uint64_t span, int wgs, float *sink) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((fill<MODE, DEPTH>), dim3(wgs), dim3(256), 0, 0, buf, span, 50, sink);
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill<MODE, DEPTH>), dim3(wgs), dim3(256), 0, 0, buf, span, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)wgs * iters * 16384.0 / (ms * 1e-3) / 1e12;
}

int main() {
    char *buf; float *sink;
    const uint64_t cap = 512ull << 20;
    CK(hipMalloc(&buf, cap + (1 << 20)));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, cap));
    const uint64_t spans[] = { 8ull << 20, 24ull << 20, 128ull << 20, 512ull << 20 };
    for (uint64_t span : spans)
        for (int wgs : { 256, 512, 1024 }) {
            std::printf("span %4llu MiB, %4d workgroups: LDS-DMA depth1 %5.1f depth4 %5.1f TB/s | registers %5.1f TB/s\n", (unsigned long long)(span >> 20), wgs,
                        run<0, 1>(buf, span, wgs, sink), run<0, 4>(buf, span, wgs, sink), run<1, 1>(buf, span, wgs, sink));
        }
    return 0;
}
