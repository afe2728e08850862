// Which engine can gather row blocks of a column-major C next to a resident f16 GEMM without taking compute units?
// Candidates for the M-sharded Gemm's gather step: a strided (M/P rows x np columns) block copy
//   (1) hipMemcpy2DAsync device->device on a second stream       (the HIP runtime picks the engine: blit kernel or SDMA)
//   (2) hsa_amd_memory_async_copy_rect, hsaDeviceToDevice         (SDMA by contract: "API requires SDMA")
//   (3) hipMemcpyAsync of the same bytes, contiguous              (yardstick)
// Each is timed alone, then next to back-to-back 8192^3 f16 Gemms on the library's stream: the copy engine question is whether the
// Gemm keeps its rate. One GPU only: source and destination are two buffers of the same device (the peer case differs in the
// link, not in who issues the copy).
// Build: hipcc -O2 -std=c++17 -Iinclude tools/cpp/sdma_probe.cpp -o gpurun_out/sdma_probe wgmath_amd/libwgebra_hip.so -lhsa-runtime64 -Wl,-rpath,$PWD/wgmath_amd
#include "wgebra_hip.h"

#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { if (int rc_ = (x)) { std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, wg_last_error_string()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define SK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char *m_ = nullptr; hsa_status_string(s_, &m_); std::fprintf(stderr, "%s -> %s\n", #x, m_ ? m_ : "?"); return 1; } } while (0)

static std::vector<hsa_agent_t> g_gpus;
static hsa_status_t agent_cb(hsa_agent_t a, void *) {
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) == HSA_STATUS_SUCCESS && t == HSA_DEVICE_TYPE_GPU) g_gpus.push_back(a);
    return HSA_STATUS_SUCCESS;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
    const size_t M = 32768, P = 4, Mg = M / P, np = 4096; // the 4-GPU panel of config 5: 8192 rows x 4096 columns of f16 out of ld = 32768
    const size_t width = Mg * 2, spitch = M * 2, dpitch = M * 2, rows = np; // bytes; "rows" of the 2-D copy are matrix COLUMNS
    const size_t bytes = width * rows;
    wg_ctx *ctx = nullptr;
    CK(wg_ctx_create(0, &ctx));
    SK(hsa_init());
    SK(hsa_iterate_agents(agent_cb, nullptr));
    if (g_gpus.empty()) { std::fprintf(stderr, "no HSA GPU agent\n"); return 1; }
    hsa_agent_t gpu = g_gpus[0];

    char *src = nullptr, *dst = nullptr;
    HK(hipMalloc(&src, spitch * rows));
    HK(hipMalloc(&dst, dpitch * rows));
    std::vector<uint16_t> h(spitch * rows / 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint16_t)(i * 2654435761u >> 13);
    HK(hipMemcpy(src, h.data(), spitch * rows, hipMemcpyHostToDevice));
    HK(hipMemset(dst, 0, dpitch * rows));
    hipStream_t cs;
    HK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));

    // the Gemm next to it
    const uint32_t G = 8192;
    wg_buf *a = nullptr, *b = nullptr, *c = nullptr;
    {
        std::vector<uint16_t> r((size_t)G * G);
        for (size_t i = 0; i < r.size(); ++i) r[i] = (uint16_t)(0x3000 + ((i * 2654435761u >> 12) & 0x0fff) + ((i & 1) << 15)); // random-ish f16 in +-[0.125, 0.25)
        CK(wg_buf_create_init(ctx, r.data(), r.size() * 2, 128 | 4 | 8, &a));
        CK(wg_buf_create_init(ctx, r.data(), r.size() * 2, 128 | 4 | 8, &b));
        CK(wg_buf_create(ctx, r.size() * 2, 128 | 4, &c));
    }
    wg_view_shape sg = { { G, G, 1 }, G, G * G, 0 };
    wg_timestamps *ts = nullptr;
    CK(wg_timestamps_create(ctx, 2, &ts));
    auto gemms = [&](int n, double *ms) -> int {
        CK(wg_timestamps_clear(ts));
        CK(wg_timestamps_write(ctx, ts, nullptr));
        for (int i = 0; i < n; ++i) CK(wg_gemm(ctx, WG_GEMM, WG_F16, c, sg, a, sg, b, sg));
        CK(wg_timestamps_write(ctx, ts, nullptr));
        double t[2];
        CK(wg_timestamps_wait_for_results_ms(ts, t, 2));
        *ms = (t[1] - t[0]) / n;
        return 0;
    };
    double base = 0;
    if (gemms(20, &base)) return 1;
    if (gemms(60, &base)) return 1;
    std::printf("gemm_f16 8192^3 alone: %.4f ms = %.1f TFLOP/s\n", base, 2.0 * G * G * G / base / 1e9);

    hsa_signal_t sig;
    SK(hsa_signal_create(1, 0, nullptr, &sig));
    auto hsa_rect = [&]() -> int {
        hsa_pitched_ptr_t d = { dst, dpitch, dpitch * rows }, s = { src, spitch, spitch * rows };
        hsa_dim3_t off = { 0, 0, 0 }, range = { (uint32_t)width, (uint32_t)rows, 1 };
        hsa_signal_store_relaxed(sig, 1);
        SK(hsa_amd_memory_async_copy_rect(&d, &off, &s, &off, &range, gpu, hsaDeviceToDevice, 0, nullptr, sig));
        return 0;
    };
    auto hsa_wait = [&]() { while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) >= 1) {} };
    auto hsa_lin = [&]() -> int {
        hsa_signal_store_relaxed(sig, 1);
        SK(hsa_amd_memory_async_copy(dst, gpu, src, gpu, bytes, 0, nullptr, sig));
        return 0;
    };

    struct Mode { const char *name; int kind; };
    const Mode modes[] = { { "hipMemcpy2DAsync D2D (strided block)", 0 }, { "hsa_amd_memory_async_copy_rect D2D (SDMA, strided block)", 1 },
                           { "hipMemcpyAsync D2D (contiguous)", 2 }, { "hsa_amd_memory_async_copy D2D (SDMA, contiguous)", 3 } };
    for (const Mode &m : modes) {
        auto issue = [&]() -> int {
            switch (m.kind) {
            case 0: HK(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, hipMemcpyDeviceToDevice, cs)); return 0;
            case 1: return hsa_rect();
            case 2: HK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, cs)); return 0;
            default: return hsa_lin();
            }
        };
        auto wait = [&]() -> int {
            if (m.kind == 1 || m.kind == 3) hsa_wait(); else HK(hipStreamSynchronize(cs));
            return 0;
        };
        // alone
        if (issue() || wait()) { std::printf("%s: FAILED\n", m.name); continue; }
        double t0 = now_ms();
        const int reps = 10;
        for (int i = 0; i < reps; ++i) { if (issue() || wait()) return 1; }
        double alone = (now_ms() - t0) / reps;
        // correctness of the strided block (kinds 0, 1)
        if (m.kind < 2) {
            std::vector<uint16_t> back(dpitch * rows / 2);
            HK(hipMemcpy(back.data(), dst, dpitch * rows, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t r = 0; r < rows; r += 97)
                for (size_t x = 0; x < width / 2; x += 61) bad += back[r * dpitch / 2 + x] != h[r * spitch / 2 + x];
            if (bad) std::printf("  !! %zu mismatches\n", bad);
        }
        // next to the Gemms: keep a copy in flight for the whole timed region
        CK(wg_ctx_sync(ctx));
        CK(wg_timestamps_clear(ts));
        CK(wg_timestamps_write(ctx, ts, nullptr));
        const int ng = 60;
        for (int i = 0; i < ng; ++i) CK(wg_gemm(ctx, WG_GEMM, WG_F16, c, sg, a, sg, b, sg));
        CK(wg_timestamps_write(ctx, ts, nullptr));
        int copies = 0;
        double c0 = now_ms();
        while (now_ms() - c0 < base * ng * 0.9) { if (issue() || wait()) return 1; ++copies; }
        double cdt = now_ms() - c0;
        double t[2];
        CK(wg_timestamps_wait_for_results_ms(ts, t, 2));
        double with = (t[1] - t[0]) / ng;
        std::printf("%-58s alone %.3f ms (%.0f GB/s) | beside Gemm: %d copies at %.3f ms (%.0f GB/s); Gemm %.4f ms (%+.1f %%)\n", m.name, alone,
                    bytes / alone / 1e6, copies, cdt / copies, bytes / (cdt / copies) / 1e6, with, (with / base - 1) * 100);
    }
    (void)argc; (void)argv;
    return 0;
}
