// Diagnostics that ride along with a measurement (wg_debug_*): what the chip was doing while a kernel ran. No reference counterpart -- the
// reference's only probe is the timestamp query set (crates/wgcore/src/timestamps.rs:226-230: GPU time between two points of a pass); these
// two add what a power-capped part needs next to a time: the shader clock the kernel actually got, and what the matrix cores sustain alone.
//
//   wg_debug_clock_begin / _end   One stamp kernel before and one after whatever the caller enqueues in between on the context's stream: every
//                                 workgroup records (XCC_ID, s_memtime, s_memrealtime). s_memtime ticks with the shader clock, s_memrealtime
//                                 at a constant 100 MHz, both free-running -- so (d memtime / d memrealtime) x 100 MHz over the bracketed
//                                 interval is the MEAN shader clock of that XCD while the bracketed work ran (idle chip: 2.4 GHz; the f16
//                                 Gemm on random operands: ~1.5). Nothing is added to the measured kernels, nothing runs beside them.
//   wg_debug_mfma_ceiling         v_mfma_f32_16x16x32_f16 only -- operands in registers, no LDS traffic, no loads, one 4-wave workgroup per CU
//                                 (what the f16 Gemm's workgroups are) -- on uniform random operands for a given time: the throughput the
//                                 package power cap leaves the matrix cores when NOTHING else draws power. Any f16 Gemm on this part sits
//                                 below it by what its data movement costs (profiles/r03_evidence.md section 9).
#include "wg_internal.hpp"

#include <new>

namespace {

constexpr uint32_t kStampBlocks = 64; // 8 per XCD if the dispatcher deals them round-robin; any XCD seen on both sides counts

struct Stamp {
    uint64_t memtime, realtime;
    uint32_t xcc, hw; // hw: HW_REG_HW_ID bits [15:8] = cu [11:8], sh [12], se [15:13] -- where on the XCD the stamp was taken
};

__global__ void clock_stamp_kernel(Stamp *out) {
    if (threadIdx.x != 0) return;
    Stamp s;
    s.xcc = __builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | ((4 - 1) << 11)) & 7u;
    s.hw = (__builtin_amdgcn_s_getreg((4 /* HW_REG_HW_ID */) | (0 << 6) | ((16 - 1) << 11)) >> 8) & 0xffu;
    s.memtime = __builtin_amdgcn_s_memtime();
    s.realtime = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x] = s;
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ half8 random_fragment(uint32_t seed) { // uniform in [-1, 1), different per lane and fragment
    half8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t r = mix(seed * 8u + i + 0x9e3779b9u);
        v[i] = (_Float16)((float)(r >> 8) * (2.f / 16777216.f) - 1.f);
    }
    return v;
}

// the Gemm's wave tile (8 x 8 fragments of 16 x 16, 256 accumulator registers), its instruction, nothing else
__global__ __launch_bounds__(256, 1) void mfma_only_kernel(float *out, int reps) {
    extern __shared__ char smem[]; // 160 KiB claimed: one workgroup per CU, like the Gemm's
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    half8 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = random_fragment(id * 16u + i);
        b[i] = random_fragment(id * 16u + 8u + i);
    }
    floatx4 acc[8][8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[t][u] = floatx4{ 0.f, 0.f, 0.f, 0.f };
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t], b[u], acc[t][u], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int u = 0; u < 8; ++u) s += acc[t][u][0] + acc[t][u][1] + acc[t][u][2] + acc[t][u][3];
    if (s == 123.456f) out[id] = s; // (keeps the accumulators alive)
    if (threadIdx.x == 0 && smem[0] == 77) out[0] = 1.f;
}

int ensure_stamps(wg_ctx *ctx) {
    if (!ctx->debug_stamps) WG_HIP_TRY(hipHostMalloc(&ctx->debug_stamps, 2 * kStampBlocks * sizeof(Stamp), hipHostMallocDefault));
    return WG_OK;
}

} // namespace

extern "C" {

int wg_debug_clock_begin(wg_ctx *ctx) {
    if (!ctx) return wg_set_error(WG_ERR_INVALID_ARG, "wg_debug_clock_begin: ctx is NULL");
    if (ctx->recording) return wg_set_error(WG_ERR_INVALID_ARG, "wg_debug_clock_begin: cannot be recorded");
    WG_HIP_TRY(hipSetDevice(ctx->device));
    if (int rc = ensure_stamps(ctx)) return rc;
    hipLaunchKernelGGL(clock_stamp_kernel, dim3(kStampBlocks), dim3(64), 0, ctx->stream, (Stamp *)ctx->debug_stamps);
    WG_HIP_TRY(hipGetLastError());
    ctx->debug_clock_open = true;
    return WG_OK;
}

int wg_debug_clock_end(wg_ctx *ctx, double *ghz_mean, double *ghz_min, double *ghz_max, double *seconds) {
    if (!ctx || !ghz_mean) return wg_set_error(WG_ERR_INVALID_ARG, "wg_debug_clock_end: NULL argument");
    if (!ctx->debug_clock_open) return wg_set_error(WG_ERR_INVALID_ARG, "wg_debug_clock_end: no wg_debug_clock_begin on this context");
    WG_HIP_TRY(hipSetDevice(ctx->device));
    Stamp *st = (Stamp *)ctx->debug_stamps;
    hipLaunchKernelGGL(clock_stamp_kernel, dim3(kStampBlocks), dim3(64), 0, ctx->stream, st + kStampBlocks);
    WG_HIP_TRY(hipGetLastError());
    WG_HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->debug_clock_open = false;
    // Per XCD: the MEDIAN of (d memtime / d memrealtime) over the stamp pairs taken at the same place (same CU of that XCD; any pair of the XCD
    // when no CU was hit on both sides), pairs outside 0.2 .. 3.5 GHz dropped -- a CU's shader-clock counter is not guaranteed to have run
    // through the interval (seen once on hardware: one pair 11 orders of magnitude off, the others fine).
    double sum = 0, lo = 1e30, hi = 0, secs = 0;
    int n = 0;
    for (uint32_t x = 0; x < 8; ++x) {
        double cand[2][kStampBlocks * 4];
        int nc[2] = { 0, 0 };
        double t = 0;
        for (uint32_t i = 0; i < kStampBlocks; ++i)
            for (uint32_t j = 0; j < kStampBlocks; ++j) {
                const Stamp &a = st[i], &b = st[kStampBlocks + j];
                if (a.xcc != x || b.xcc != x || b.realtime <= a.realtime || b.memtime <= a.memtime) continue;
                const double ghz = (double)(b.memtime - a.memtime) / (double)(b.realtime - a.realtime) * 0.1; // ticks per 10 ns
                if (!(ghz > 0.2 && ghz < 3.5)) continue;
                const int same = a.hw == b.hw ? 0 : 1;
                if (nc[same] < (int)(kStampBlocks * 4)) cand[same][nc[same]++] = ghz;
                t = (double)(b.realtime - a.realtime) * 1e-8;
            }
        const int w = nc[0] > 0 ? 0 : 1;
        if (nc[w] == 0) continue;
        for (int p = 1; p < nc[w]; ++p) // (insertion sort: a few dozen values)
            for (int q = p; q > 0 && cand[w][q] < cand[w][q - 1]; --q) { const double tmp = cand[w][q]; cand[w][q] = cand[w][q - 1]; cand[w][q - 1] = tmp; }
        const double ghz = cand[w][nc[w] / 2];
        sum += ghz; lo = ghz < lo ? ghz : lo; hi = ghz > hi ? ghz : hi;
        secs += t;
        ++n;
    }
    if (n == 0) return wg_set_error(WG_ERR_HIP, "wg_debug_clock_end: no XCD was stamped on both sides");
    *ghz_mean = sum / n;
    if (ghz_min) *ghz_min = lo;
    if (ghz_max) *ghz_max = hi;
    if (seconds) *seconds = secs / n;
    return WG_OK;
}

int wg_debug_mfma_ceiling(wg_ctx *ctx, double min_seconds, double *tflops, double *clock_ghz) {
    if (!ctx || !tflops) return wg_set_error(WG_ERR_INVALID_ARG, "wg_debug_mfma_ceiling: NULL argument");
    if (ctx->recording) return wg_set_error(WG_ERR_INVALID_ARG, "wg_debug_mfma_ceiling: cannot be recorded");
    if (!(min_seconds > 0.0) || min_seconds > 30.0) return wg_set_error(WG_ERR_INVALID_ARG, "wg_debug_mfma_ceiling: min_seconds must be in (0, 30]");
    WG_HIP_TRY(hipSetDevice(ctx->device));
    const int wgs = ctx->compute_units > 0 ? ctx->compute_units : 256;
    float *out = nullptr;
    void *ws = nullptr;
    if (int rc = wg_ctx_workspace(ctx, (size_t)wgs * 256 * sizeof(float), &ws)) return rc;
    out = (float *)ws;
    if (!(ctx->func_attr_bits & (1u << 8))) {
        WG_HIP_TRY(hipFuncSetAttribute((const void *)mfma_only_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->func_attr_bits |= 1u << 8;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    WG_HIP_TRY(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return wg_set_error(WG_ERR_HIP, "wg_debug_mfma_ceiling: event creation failed"); }
    auto done = [&](int rc) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return rc; };
    const int reps = 40000; // ~22 ms per launch at 1.9 PFLOP/s on 256 CUs: short enough to bound the run, long enough to sit at the power cap
    const double flops = (double)wgs * 4.0 * reps * 64.0 * 2.0 * 16 * 16 * 32;
    hipLaunchKernelGGL(mfma_only_kernel, dim3(wgs), dim3(256), 160 * 1024, ctx->stream, out, reps); // warm-up: clocks settle under load
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return done(wg_set_error(WG_ERR_HIP, "wg_debug_mfma_ceiling: launch failed"));
    const int launches = (int)(min_seconds / 0.022) + 1;
    if (int rc = wg_debug_clock_begin(ctx)) return done(rc);
    if (hipEventRecord(e0, ctx->stream) != hipSuccess) return done(wg_set_error(WG_ERR_HIP, "wg_debug_mfma_ceiling: hipEventRecord failed"));
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(mfma_only_kernel, dim3(wgs), dim3(256), 160 * 1024, ctx->stream, out, reps);
    if (hipEventRecord(e1, ctx->stream) != hipSuccess) return done(wg_set_error(WG_ERR_HIP, "wg_debug_mfma_ceiling: hipEventRecord failed"));
    double ghz = 0;
    if (int rc = wg_debug_clock_end(ctx, &ghz, nullptr, nullptr, nullptr)) return done(rc);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess || !(ms > 0.f)) return done(wg_set_error(WG_ERR_HIP, "wg_debug_mfma_ceiling: timing failed"));
    *tflops = flops * launches / ((double)ms * 1e-3) / 1e12;
    if (clock_ghz) *clock_ghz = ghz;
    return done(WG_OK);
}

} // extern "C"
