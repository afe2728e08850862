// Streaming a column-major M x K f32 matrix the way a skinny GEMM / GEMV wave does: per instruction either two 512-byte row segments (128
// rows at k and at k + 4: the operand layout of v_mfma_f32_32x32x2) or one 1 KiB segment (256 rows at one k). Which one does HBM like?
//   hipcc -O3 --offload-arch=gfx950 tools/cpp/stream_probe.hip -o gpurun_out/stream_probe && gpurun_out/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE, int DEPTH> // MODE 0: 2 x 512 B per instruction (128-row blocks); 1: 1 KiB per instruction (256-row blocks)
__global__ __launch_bounds__(256) void stream(const float *A, uint32_t M, uint32_t K, uint32_t kps, float *sink) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t rows = MODE == 0 ? 128u : 256u;
    const uint32_t r0 = blockIdx.x * rows;
    const uint32_t kq = kps / 4u, kb = blockIdx.y * kps + wave * kq, ke = min(kb + kq, K);
    const float *p = A + r0 + (MODE == 0 ? 4u * (lane & 31u) : 4u * lane);
    const uint32_t h = MODE == 0 ? lane >> 5 : 0u;
    float4 acc = make_float4(0, 0, 0, 0);
    // one "group" = 8 k: MODE 0: 4 instructions (k0 + 4 h + s); MODE 1: 8 instructions (k0 + s)
    constexpr int PER = MODE == 0 ? 4 : 8;
    float4 v[DEPTH][PER];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int s = 0; s < PER; ++s) v[d][s] = *(const float4 *)(p + (uint64_t)min(kb + 8u * d + (MODE == 0 ? 4u * h : 0u) + s, K - 1u) * M);
    for (uint32_t k = kb; k < ke; k += 8u * DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int s = 0; s < PER; ++s) { acc.x += v[d][s].x; acc.y += v[d][s].w; }
#pragma unroll
            for (int s = 0; s < PER; ++s) v[d][s] = *(const float4 *)(p + (uint64_t)min(k + 8u * (DEPTH + d) + (MODE == 0 ? 4u * h : 0u) + s, K - 1u) * M);
        }
    }
    if (acc.x == 123.456f) sink[0] = acc.y;
}

template <int MODE, int DEPTH>
double run(const float *A, uint32_t M, uint32_t K, uint32_t ns, float *sink) {
    const uint32_t rows = MODE == 0 ? 128u : 256u;
    const uint32_t kps = (((K + ns - 1) / ns) + 31u) & ~31u;
    dim3 grid(M / rows, (K + kps - 1) / kps);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream<MODE, DEPTH>), grid, dim3(256), 0, 0, A, M, K, kps, sink);
    (void)hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream<MODE, DEPTH>), grid, dim3(256), 0, 0, A, M, K, kps, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return (double)M * K * 4.0 * reps / (ms * 1e-3) / 1e12;
}

int main() {
    float *A, *sink;
    const uint64_t cap = 2048ull << 20;
    if (hipMalloc(&A, cap) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    (void)hipMemset(A, 0, cap);
    const uint32_t shapes[][2] = { { 4096, 4096 }, { 11008, 4096 }, { 32000, 4096 }, { 4096, 65536 }, { 16384, 16384 } };
    for (auto &sh : shapes) {
        const uint32_t M = sh[0] / 256 * 256, K = sh[1];
        for (uint32_t wgs_per_cu : { 1u, 2u, 4u }) {
            const uint32_t ns0 = (wgs_per_cu * 256 + M / 128 - 1) / (M / 128), ns1 = (wgs_per_cu * 256 + M / 256 - 1) / (M / 256);
            std::printf("%5u x %5u, ~%u workgroups per CU: 2x512B  depth4 %5.2f depth8 %5.2f TB/s | 1KiB depth2 %5.2f depth4 %5.2f TB/s\n", M, K, wgs_per_cu,
                        run<0, 4>(A, M, K, ns0, sink), run<0, 8>(A, M, K, ns0, sink), run<1, 2>(A, M, K, ns1, sink), run<1, 4>(A, M, K, ns1, sink));
        }
    }
    return 0;
}
