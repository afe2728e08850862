// Host-side cost of one eager dispatch through the C ABI (what a Rust shim would pay per wgebra dispatch), against the raw HIP
// launch of an empty kernel is not available from plain C++, so the yardstick is the GPU-side kernel time reported by the probes.
// Build: g++ -O2 -std=c++17 -Iinclude tools/cpp/dispatch_overhead.cpp -o gpurun_out/dispatch_overhead wgmath_amd/libwgebra_hip.so -Wl,-rpath,$PWD/wgmath_amd
#include "wgebra_hip.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { if (int rc_ = (x)) { std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, wg_last_error_string()); return 1; } } while (0)

int main() {
    wg_ctx *ctx = nullptr;
    CK(wg_ctx_create(0, &ctx));
    const uint32_t sizes[] = { 256, 512, 1024, 2048 };
    for (uint32_t n : sizes) {
        wg_buf *m = nullptr, *v = nullptr, *o = nullptr;
        std::vector<float> hm((size_t)n * n, 0.5f), hv(n, 1.0f);
        CK(wg_buf_create_init(ctx, hm.data(), hm.size() * 4, 128 | 4 | 8, &m));
        CK(wg_buf_create_init(ctx, hv.data(), hv.size() * 4, 128 | 4 | 8, &v));
        CK(wg_buf_create(ctx, n * 4, 128 | 4, &o));
        wg_view_shape sm = { { n, n, 1 }, n, n * n, 0 }, sv = { { n, 1, 1 }, n, n, 0 };
        for (int i = 0; i < 100; ++i) CK(wg_gemv(ctx, WG_GEMV, WG_F32, o, sv, m, sm, v, sv));
        CK(wg_ctx_sync(ctx));
        const int reps = 20000;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) CK(wg_gemv(ctx, WG_GEMV, WG_F32, o, sv, m, sm, v, sv));
        auto t1 = std::chrono::steady_clock::now();
        CK(wg_ctx_sync(ctx));
        auto t2 = std::chrono::steady_clock::now();
        const double enq = std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
        const double tot = std::chrono::duration<double, std::micro>(t2 - t0).count() / reps;
        std::printf("gemv %4u x %-4u: host enqueue %.2f us per dispatch, end-to-end %.2f us per dispatch\n", n, n, enq, tot);
        wg_buf_destroy(m); wg_buf_destroy(v); wg_buf_destroy(o);
    }
    wg_ctx_destroy(ctx);
    return 0;
}
