// Follow-up of sdma_probe.cpp: how many copy engines can one GPU drive at once, and does the f16 Gemm notice?
//   (1) hsa_amd_memory_copy_engine_status / _get_preferred_copy_engine: which SDMA engines exist for this agent;
//   (2) contiguous copies forced onto ONE named engine each (hsa_amd_memory_async_copy_on_engine, force_copy_on_sdma): rate per engine,
//       then 2 / 3 / 7 engines at once (what a rank of 2 / 4 / 8 needs: one engine per peer link);
//   (3) rect copies issued as hsaHostToDevice / hsaDeviceToHost / hsaDeviceToDevice at once: does the direction flag select
//       different engines (sdma_probe.cpp found ONE rect stream = 60 GB/s)?
//   (4) back-to-back 8192^3 f16 Gemms beside 3 and 7 continuously busy engines.
// Build: hipcc -O2 -std=c++17 -Iinclude tools/cpp/sdma_probe2.cpp -o /tmp/sdma_probe2 -Lwgmath_amd -lwgebra_hip -lhsa-runtime64 -Wl,-rpath,$PWD/wgmath_amd
#include "wgebra_hip.h"

#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { if (int rc_ = (x)) { std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, wg_last_error_string()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static const char *sstr(hsa_status_t s) { const char *m = nullptr; hsa_status_string(s, &m); return m ? m : "?"; }

static std::vector<hsa_agent_t> g_gpus;
static hsa_status_t agent_cb(hsa_agent_t a, void *) {
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) == HSA_STATUS_SUCCESS && t == HSA_DEVICE_TYPE_GPU) g_gpus.push_back(a);
    return HSA_STATUS_SUCCESS;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void wait_sig(hsa_signal_t s) { while (hsa_signal_wait_scacquire(s, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) >= 1) {} }

int main() {
    wg_ctx *ctx = nullptr;
    CK(wg_ctx_create(0, &ctx));
    if (hsa_init() != HSA_STATUS_SUCCESS || hsa_iterate_agents(agent_cb, nullptr) != HSA_STATUS_SUCCESS || g_gpus.empty()) { std::fprintf(stderr, "hsa\n"); return 1; }
    hsa_agent_t gpu = g_gpus[0];
    uint32_t avail = 0, pref = 0;
    hsa_status_t s1 = hsa_amd_memory_copy_engine_status(gpu, gpu, &avail);
    hsa_status_t s2 = hsa_amd_memory_get_preferred_copy_engine(gpu, gpu, &pref);
    std::printf("engine status (self -> self): %s mask 0x%x ; preferred: %s mask 0x%x\n", sstr(s1), avail, sstr(s2), pref);

    const size_t bytes = 256ull << 20;
    const int NB = 8;
    char *src[NB], *dst[NB];
    for (int i = 0; i < NB; ++i) { HK(hipMalloc(&src[i], bytes)); HK(hipMalloc(&dst[i], bytes)); HK(hipMemset(src[i], i + 1, bytes)); HK(hipMemset(dst[i], 0, bytes)); }
    HK(hipDeviceSynchronize());
    hsa_signal_t sig[NB];
    for (int i = 0; i < NB; ++i) if (hsa_signal_create(1, 0, nullptr, &sig[i]) != HSA_STATUS_SUCCESS) return 1;

    std::vector<int> engines;
    for (int e = 0; e < 16; ++e) {
        hsa_signal_store_relaxed(sig[0], 1);
        hsa_status_t st = hsa_amd_memory_async_copy_on_engine(dst[0], gpu, src[0], gpu, bytes, 0, nullptr, sig[0], (hsa_amd_sdma_engine_id_t)(1u << e), true);
        if (st != HSA_STATUS_SUCCESS) { std::printf("engine %2d: %s\n", e, sstr(st)); continue; }
        wait_sig(sig[0]);
        double t0 = now_ms();
        for (int r = 0; r < 3; ++r) {
            hsa_signal_store_relaxed(sig[0], 1);
            if (hsa_amd_memory_async_copy_on_engine(dst[0], gpu, src[0], gpu, bytes, 0, nullptr, sig[0], (hsa_amd_sdma_engine_id_t)(1u << e), true) != HSA_STATUS_SUCCESS) break;
            wait_sig(sig[0]);
        }
        double dt = (now_ms() - t0) / 3;
        std::printf("engine %2d: contiguous 256 MiB in %.3f ms = %.1f GB/s\n", e, dt, bytes / dt / 1e6);
        engines.push_back(e);
    }
    unsigned char probe = 0;
    HK(hipMemcpy(&probe, dst[0] + 12345, 1, hipMemcpyDeviceToHost));
    std::printf("data check: %s\n", probe == 1 ? "ok" : "WRONG");

    auto concurrent = [&](int n) -> double { // n engines at once, one 256 MiB copy each; returns aggregate GB/s
        if ((int)engines.size() < n) return 0;
        double t0 = now_ms();
        for (int r = 0; r < 3; ++r) {
            for (int i = 0; i < n; ++i) {
                hsa_signal_store_relaxed(sig[i], 1);
                hsa_amd_memory_async_copy_on_engine(dst[i], gpu, src[i], gpu, bytes, 0, nullptr, sig[i], (hsa_amd_sdma_engine_id_t)(1u << engines[i]), true);
            }
            for (int i = 0; i < n; ++i) wait_sig(sig[i]);
        }
        double dt = (now_ms() - t0) / 3;
        return n * bytes / dt / 1e6;
    };
    for (int n : { 1, 2, 3, 4, 7, 8 }) std::printf("%d engines at once: aggregate %.1f GB/s\n", n, concurrent(n));

    { // rect copies with the three direction flags at once
        const size_t width = 16384, rows = 4096, pitch = 65536; // 64 MiB payload each, 256 MiB span
        hsa_amd_copy_direction_t dirs[3] = { hsaHostToDevice, hsaDeviceToHost, hsaDeviceToDevice };
        const char *names[3] = { "hsaHostToDevice", "hsaDeviceToHost", "hsaDeviceToDevice" };
        auto rect = [&](int i, hsa_amd_copy_direction_t d) {
            hsa_pitched_ptr_t dp = { dst[i], pitch, pitch * rows }, sp = { src[i], pitch, pitch * rows };
            hsa_dim3_t off = { 0, 0, 0 }, range = { (uint32_t)width, (uint32_t)rows, 1 };
            hsa_signal_store_relaxed(sig[i], 1);
            return hsa_amd_memory_async_copy_rect(&dp, &off, &sp, &off, &range, gpu, d, 0, nullptr, sig[i]);
        };
        for (int i = 0; i < 3; ++i) {
            hsa_status_t st = rect(i, dirs[i]);
            if (st != HSA_STATUS_SUCCESS) { std::printf("rect %s: %s\n", names[i], sstr(st)); continue; }
            wait_sig(sig[i]);
            double t0 = now_ms();
            rect(i, dirs[i]); wait_sig(sig[i]);
            double dt = now_ms() - t0;
            std::printf("rect %-18s alone: %.1f GB/s\n", names[i], width * rows / dt / 1e6);
        }
        for (int n : { 2, 3 }) {
            double t0 = now_ms();
            for (int i = 0; i < n; ++i) rect(i, dirs[i]);
            for (int i = 0; i < n; ++i) wait_sig(sig[i]);
            double dt = now_ms() - t0;
            std::printf("rect, %d direction flags at once: aggregate %.1f GB/s\n", n, n * width * rows / dt / 1e6);
        }
    }

    // Gemm beside continuously busy engines
    const uint32_t G = 8192;
    wg_buf *a = nullptr, *b = nullptr, *c = nullptr;
    {
        std::vector<uint16_t> r((size_t)G * G);
        for (size_t i = 0; i < r.size(); ++i) r[i] = (uint16_t)(0x3000 + ((i * 2654435761u >> 12) & 0x0fff) + ((i & 1) << 15));
        CK(wg_buf_create_init(ctx, r.data(), r.size() * 2, 128 | 4 | 8, &a));
        CK(wg_buf_create_init(ctx, r.data(), r.size() * 2, 128 | 4 | 8, &b));
        CK(wg_buf_create(ctx, r.size() * 2, 128 | 4, &c));
    }
    wg_view_shape sg = { { G, G, 1 }, G, G * G, 0 };
    wg_timestamps *ts = nullptr;
    CK(wg_timestamps_create(ctx, 2, &ts));
    auto gemms = [&](int n, int nengines, double *ms, double *copy_gbs) -> int {
        CK(wg_ctx_sync(ctx));
        CK(wg_timestamps_clear(ts));
        CK(wg_timestamps_write(ctx, ts, nullptr));
        for (int i = 0; i < n; ++i) CK(wg_gemm(ctx, WG_GEMM, WG_F16, c, sg, a, sg, b, sg));
        CK(wg_timestamps_write(ctx, ts, nullptr));
        size_t copied = 0;
        double c0 = now_ms();
        if (nengines > 0) {
            const double budget = 0.76 * n * 0.9;
            for (int i = 0; i < nengines; ++i) {
                hsa_signal_store_relaxed(sig[i], 1);
                hsa_amd_memory_async_copy_on_engine(dst[i], gpu, src[i], gpu, bytes, 0, nullptr, sig[i], (hsa_amd_sdma_engine_id_t)(1u << engines[i]), true);
            }
            while (now_ms() - c0 < budget) {
                for (int i = 0; i < nengines; ++i)
                    if (hsa_signal_load_relaxed(sig[i]) < 1) {
                        copied += bytes;
                        hsa_signal_store_relaxed(sig[i], 1);
                        hsa_amd_memory_async_copy_on_engine(dst[i], gpu, src[i], gpu, bytes, 0, nullptr, sig[i], (hsa_amd_sdma_engine_id_t)(1u << engines[i]), true);
                    }
            }
            for (int i = 0; i < nengines; ++i) { wait_sig(sig[i]); copied += bytes; }
        }
        double cdt = now_ms() - c0;
        double t[2];
        CK(wg_timestamps_wait_for_results_ms(ts, t, 2));
        *ms = (t[1] - t[0]) / n;
        *copy_gbs = copied / cdt / 1e6;
        return 0;
    };
    double base = 0, g = 0, cg = 0;
    if (gemms(30, 0, &base, &cg)) return 1;
    for (int rep = 0; rep < 2; ++rep) {
        if (gemms(80, 0, &base, &cg)) return 1;
        std::printf("gemm_f16 8192^3 alone: %.4f ms = %.1f TFLOP/s\n", base, 2.0 * G * G * G / base / 1e9);
        for (int n : { 1, 3, 7 }) {
            if ((int)engines.size() < n) continue;
            if (gemms(80, n, &g, &cg)) return 1;
            std::printf("  beside %d busy SDMA engine(s) (%.0f GB/s aggregate): %.4f ms (%+.1f %%)\n", n, cg, g, (g / base - 1) * 100);
        }
    }
    return 0;
}
