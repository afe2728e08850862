// The multi-GPU entry points of the C ABI (include/wgebra_hip.h, "multi-GPU" section), driven from plain C++ with no torch in the
// process -- what a Rust caller of the boundary would do:
//   1. a 1-rank RCCL communicator: unique id -> wg_comm_create -> wg_all_gather round trip -> wg_gemm_sharded(WG_GATHER_RCCL), i.e.
//      staging cube + ncclAllGather + cube_to_matrix relayout, against a host f64 product;
//   2. two ranks in ONE process on one device (two contexts, two communicators without a collective library), WG_GATHER_PEER_STAGED:
//      each rank's Gemm writes its slot of its staging cube, one contiguous copy per peer carries it over, the receiver waits on the
//      flags and relayouts; both results must be the plain M x N product. (RCCL refuses two ranks on one device, so the 2-rank
//      collective itself needs a multi-GPU node.) Also: a peer that misses a step (time-out -> error in wg_ctx_sync -> clean retry),
//      and pipelined one-launch RCCL steps whose shapes alternate.
//   3. wg_cube_to_matrix on a hand-made cube (exact), wg_buf_ipc_export on a wrapped interior pointer (offset carried).
// f32 and f16, Gemm and GemmTr, ragged last panel. Exit code 0 and "ALL OK" on success.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "wgebra_hip.h"

static int failures = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { ++failures; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)
#define CK(x) do { int rc_ = (x); if (rc_ != WG_OK) { std::printf("FAIL %s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #x, rc_, wg_last_error_string()); ++failures; return; } } while (0)

static const uint32_t USAGE = WG_USAGE_STORAGE | WG_USAGE_COPY_SRC | WG_USAGE_COPY_DST;

template <typename T> struct dt;
template <> struct dt<float> { static constexpr wg_dtype v = WG_F32; };
template <> struct dt<_Float16> { static constexpr wg_dtype v = WG_F16; };

template <typename T> static std::vector<T> rnd(size_t n, uint32_t seed) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> d(-1.f, 1.f);
    std::vector<T> v(n);
    for (auto &x : v) x = (T)d(rng);
    return v;
}

// |got - truth| <= 2 sqrt(K) 2^-24 sum|a||b| (+ half an f16 ulp of the result for f16): tests/_util.py
template <typename T>
static void check_product(const char *what, const std::vector<T> &got, const std::vector<T> &A, bool tr, const std::vector<T> &B, uint32_t M, uint32_t N, uint32_t K) {
    // (round 5's form -- every element converted from f16 inside the k loop, A walked with stride M -- was where this test's two and a half minutes went: 130 s of the
    // 178 on the host. Operands to f64 once, both k-contiguous; every third column and, of the larger products, every second row -- the row phase alternating with the
    // column, so every rank's block of every panel is still hit)
    std::vector<double> Ad((size_t)M * K), Bd((size_t)K * N);
    for (uint32_t i = 0; i < M; ++i)
        for (uint32_t k = 0; k < K; ++k) Ad[(size_t)i * K + k] = tr ? (double)A[(size_t)i * K + k] : (double)A[(size_t)k * M + i];
    for (size_t x = 0; x < Bd.size(); ++x) Bd[x] = (double)B[x];
    double worst = 0;
    const uint32_t jstep = N > 512 ? 3 : 1, istep = (uint64_t)M * N > (1u << 20) ? 2 : 1;
    for (uint32_t j = 0; j < N; j += jstep)
        for (uint32_t i = (j / jstep) % istep; i < M; i += istep) {
            double t = 0, s = 0;
            const double *a = Ad.data() + (size_t)i * K, *b = Bd.data() + (size_t)j * K;
            for (uint32_t k = 0; k < K; ++k) {
                t += a[k] * b[k];
                s += std::fabs(a[k] * b[k]);
            }
            double tol = 2.0 * std::sqrt((double)K) * std::ldexp(1.0, -24) * s + 1e-30;
            if (sizeof(T) == 2) tol += std::ldexp(1.0, -11) * std::fabs(t) + std::ldexp(1.0, -25);
            const double e = std::fabs((double)got[(size_t)j * M + i] - t) / tol;
            if (e > worst) worst = e;
        }
    EXPECT(worst <= 1.0, "%s: worst err/tol %.3g", what, worst);
}

static wg_view_shape mat(uint32_t r, uint32_t c) { return wg_view_shape{ { r, c, 1 }, r, r * c, 0 }; }

// rank g's row block of op(A) as its own dense tensor
template <typename T> static std::vector<T> row_block(const std::vector<T> &A, bool tr, uint32_t M, uint32_t K, uint32_t g, uint32_t mg) {
    std::vector<T> out((size_t)mg * K);
    if (tr) { // stored K x M: columns g*mg .. are contiguous
        std::memcpy(out.data(), A.data() + (size_t)g * mg * K, out.size() * sizeof(T));
    } else {
        for (uint32_t k = 0; k < K; ++k)
            for (uint32_t i = 0; i < mg; ++i) out[(size_t)k * mg + i] = A[(size_t)k * M + g * mg + i];
    }
    return out;
}

template <typename T> static void rccl_one_rank(wg_ctx *ctx, wg_comm *comm, bool tr, uint32_t M, uint32_t N, uint32_t K, uint32_t panel) {
    auto A = rnd<T>((size_t)M * K, 11), B = rnd<T>((size_t)K * N, 12);
    wg_buf *a = nullptr, *b = nullptr, *c = nullptr;
    CK(wg_buf_create_init(ctx, A.data(), A.size() * sizeof(T), USAGE, &a));
    CK(wg_buf_create_init(ctx, B.data(), B.size() * sizeof(T), USAGE, &b));
    CK(wg_buf_create(ctx, (size_t)M * N * sizeof(T), USAGE, &c));
    CK(wg_buf_fill_zero(ctx, c));
    const wg_gemm_variant v = tr ? WG_GEMM_TR : WG_GEMM;
    CK(wg_gemm_sharded(comm, v, dt<T>::v, WG_GATHER_RCCL, panel, c, mat(M, N), a, tr ? mat(K, M) : mat(M, K), b, mat(K, N)));
    std::vector<T> got((size_t)M * N);
    CK(wg_buf_read(ctx, c, 0, got.data(), got.size() * sizeof(T)));
    char what[128];
    std::snprintf(what, sizeof what, "rccl 1-rank %s %s %ux%ux%u panel %u", sizeof(T) == 2 ? "f16" : "f32", tr ? "GemmTr" : "Gemm", M, N, K, panel);
    check_product(what, got, A, tr, B, M, N, K);
    wg_buf_destroy(a); wg_buf_destroy(b); wg_buf_destroy(c);
}

// wg_gemm_sharded_panels through the plain C ABI: a tapered tail (equal panels, then narrower ones, the last one the rest) in the one-launch form and a list that
// is not of that shape (panel by panel), against f64; bad lists are refused; the wait-timing diagnostics return one entry per panel of the stamped call
static void rccl_one_rank_tapered(wg_ctx *ctx, wg_comm *comm) {
    using T = _Float16;
    const uint32_t M = 4096, N = 4096 + 128, K = 512;
    auto A = rnd<T>((size_t)M * K, 71), B = rnd<T>((size_t)K * N, 72);
    wg_buf *a = nullptr, *b = nullptr, *c = nullptr;
    CK(wg_buf_create_init(ctx, A.data(), A.size() * sizeof(T), USAGE, &a));
    CK(wg_buf_create_init(ctx, B.data(), B.size() * sizeof(T), USAGE, &b));
    CK(wg_buf_create(ctx, (size_t)M * N * sizeof(T), USAGE, &c));
    const uint32_t taper[] = { 1024, 1024, 768, 512, 512, 256, 128 }, other[] = { 512, 1024, 2048, 640 }, short_[] = { 1024, 1024 }, odd[] = { 4222, 2 };
    for (int which = 0; which < 2; ++which) {
        CK(wg_buf_fill_zero(ctx, c));
        CK(wg_comm_set_wait_timing(comm, 1));
        CK(wg_gemm_sharded_panels(comm, WG_GEMM, WG_F16, WG_GATHER_RCCL, which ? other : taper, which ? 4u : 7u, c, mat(M, N), a, mat(M, K), b, mat(K, N)));
        CK(wg_comm_join(comm));
        uint32_t panels[64], n = 0;
        float ms[64];
        CK(wg_comm_wait_times(comm, panels, ms, 64, &n));
        CK(wg_comm_set_wait_timing(comm, 0));
        EXPECT(n == (which ? 4u : 7u), "wait-timing entries: %u", n);
        for (uint32_t i = 0; i < n; ++i) EXPECT(panels[i] == i && ms[i] >= 0.f && ms[i] < 1000.f, "wait %u: panel %u, %.3f ms", i, panels[i], ms[i]);
        std::vector<T> got((size_t)M * N);
        CK(wg_buf_read(ctx, c, 0, got.data(), got.size() * sizeof(T)));
        check_product(which ? "rccl 1-rank f16, panels 512 1024 2048 640" : "rccl 1-rank f16, tapered panels", got, A, false, B, M, N, K);
    }
    EXPECT(wg_gemm_sharded_panels(comm, WG_GEMM, WG_F16, WG_GATHER_RCCL, short_, 2, c, mat(M, N), a, mat(M, K), b, mat(K, N)) == WG_ERR_PRECONDITION, "widths that do not sum to N");
    EXPECT(wg_gemm_sharded_panels(comm, WG_GEMM, WG_F16, WG_GATHER_RCCL, odd, 2, c, mat(M, N), a, mat(M, K), b, mat(K, N)) == WG_ERR_PRECONDITION, "a width that is not a multiple of 4");
    EXPECT(wg_gemm_sharded_panels(comm, WG_GEMM, WG_F16, WG_GATHER_RCCL, nullptr, 0, c, mat(M, N), a, mat(M, K), b, mat(K, N)) == WG_ERR_INVALID_ARG, "no widths");
    uint64_t fr = 0, tot = 0;
    EXPECT(wg_ctx_mem_info(ctx, &fr, &tot) == WG_OK && fr > 0 && fr <= tot, "wg_ctx_mem_info: %llu of %llu", (unsigned long long)fr, (unsigned long long)tot);
    wg_buf_destroy(a); wg_buf_destroy(b); wg_buf_destroy(c);
}

// two ranks in one process, WG_GATHER_PEER_STAGED: staging cubes + contiguous per-peer copies + flags + wait kernel + relayout, three steps
// back to back WITHOUT any host synchronisation or barrier in between (the engine is stream-ordered and double-buffered by step parity)
template <typename T> static void staged_two_ranks(bool tr, uint32_t M, uint32_t N, uint32_t K, uint32_t panel, bool pipelined = false) {
    const uint32_t P = 2, mg = M / P;
    wg_ctx *ctx[2] = { nullptr, nullptr };
    wg_comm *comm[2] = { nullptr, nullptr };
    wg_buf *a[2] = {}, *b[2] = {}, *c[2] = {}, *st[2] = {}, *fl[2] = {};
    auto A = rnd<T>((size_t)M * K, 31), B = rnd<T>((size_t)K * N, 32);
    for (uint32_t g = 0; g < P; ++g) {
        CK(wg_ctx_create(0, &ctx[g]));
        CK(wg_comm_create(ctx[g], P, g, nullptr, &comm[g]));
        auto Ag = row_block(A, tr, M, K, g, mg);
        CK(wg_buf_create_init(ctx[g], Ag.data(), Ag.size() * sizeof(T), USAGE, &a[g]));
        CK(wg_buf_create_init(ctx[g], B.data(), B.size() * sizeof(T), USAGE, &b[g]));
        CK(wg_buf_create(ctx[g], (size_t)M * N * sizeof(T), USAGE, &c[g]));
        CK(wg_comm_stage_reserve(comm[g], 2 * (size_t)M * N * sizeof(T), &st[g], &fl[g]));
    }
    for (uint32_t g = 0; g < P; ++g) CK(wg_comm_set_peer_stages(comm[g], st, fl));
    for (uint32_t g = 0; g < P; ++g) CK(wg_comm_set_pipelined(comm[g], pipelined ? 1 : 0)); // last panel of a call deferred into the next call
    const wg_gemm_variant v = tr ? WG_GEMM_TR : WG_GEMM;
    for (int rep = 0; rep < 3; ++rep) {
        for (uint32_t g = 0; g < P; ++g) {
            CK(wg_buf_fill_zero(ctx[g], c[g]));
            CK(wg_gemm_sharded(comm[g], v, dt<T>::v, WG_GATHER_PEER_STAGED, panel, c[g], mat(M, N), a[g], tr ? mat(K, mg) : mat(mg, K), b[g], mat(K, N)));
        }
    }
    if (pipelined)
        for (uint32_t g = 0; g < P; ++g) CK(wg_comm_join(comm[g])); // completes the deferred last panel of the last call
    for (uint32_t g = 0; g < P; ++g) {
        std::vector<T> got((size_t)M * N);
        CK(wg_buf_read(ctx[g], c[g], 0, got.data(), got.size() * sizeof(T))); // stream order is all it takes
        char what[160];
        std::snprintf(what, sizeof what, "peer-staged rank %u of 2 %s %s %ux%ux%u panel %u", g, sizeof(T) == 2 ? "f16" : "f32", tr ? "GemmTr" : "Gemm", M, N, K, panel);
        check_product(what, got, A, tr, B, M, N, K);
        EXPECT(wg_comm_flush(comm[g]) == WG_OK, "flush: %s", wg_last_error_string());
        EXPECT(wg_comm_bytes_sent(comm[g]) == 3ull * mg * N * sizeof(T), "bytes_sent %llu", (unsigned long long)wg_comm_bytes_sent(comm[g]));
    }
    { // unregistered peers are an error, not a hang
        wg_comm *lone = nullptr;
        CK(wg_comm_create(ctx[0], P, 0, nullptr, &lone));
        EXPECT(wg_gemm_sharded(lone, v, dt<T>::v, WG_GATHER_PEER_STAGED, panel, c[0], mat(M, N), a[0], tr ? mat(K, mg) : mat(mg, K), b[0], mat(K, N)) == WG_ERR_INVALID_ARG,
               "PEER_STAGED without staging cubes must be rejected");
        wg_comm_destroy(lone);
    }
    for (uint32_t g = 0; g < P; ++g) {
        wg_comm_destroy(comm[g]);
        wg_buf_destroy(a[g]); wg_buf_destroy(b[g]); wg_buf_destroy(c[g]);
        wg_ctx_destroy(ctx[g]);
    }
}


// A peer that does not show up: rank 0 runs step 1 alone with a short time-out. Its wait kernel gives up, the panel is poisoned, and the
// error surfaces in wg_ctx_sync -- which must leave the communicator usable: rank 1 then runs its step 1 late (rank 0's slots are there),
// both run step 2, and both results are the plain product with no further error. (Round-3 review: only wg_gemm_sharded used to clear the
// device-side time-out word, so an error reported by wg_ctx_sync poisoned every later step silently.)
static void staged_timeout_and_retry() {
    typedef _Float16 T;
    const uint32_t P = 2, M = 1024, N = 1024, K = 256, mg = M / P, panel = 512;
    setenv("WG_COMM_TIMEOUT_MS", "300", 1); // read when a communicator is created
    wg_ctx *ctx[2] = { nullptr, nullptr };
    wg_comm *comm[2] = { nullptr, nullptr };
    wg_buf *a[2] = {}, *b[2] = {}, *c[2] = {}, *st[2] = {}, *fl[2] = {};
    auto A = rnd<T>((size_t)M * K, 41), B = rnd<T>((size_t)K * N, 42);
    for (uint32_t g = 0; g < P; ++g) {
        CK(wg_ctx_create(0, &ctx[g]));
        CK(wg_comm_create(ctx[g], P, g, nullptr, &comm[g]));
        auto Ag = row_block(A, false, M, K, g, mg);
        CK(wg_buf_create_init(ctx[g], Ag.data(), Ag.size() * sizeof(T), USAGE, &a[g]));
        CK(wg_buf_create_init(ctx[g], B.data(), B.size() * sizeof(T), USAGE, &b[g]));
        CK(wg_buf_create(ctx[g], (size_t)M * N * sizeof(T), USAGE, &c[g]));
        CK(wg_comm_stage_reserve(comm[g], 2 * (size_t)M * N * sizeof(T), &st[g], &fl[g]));
    }
    unsetenv("WG_COMM_TIMEOUT_MS");
    for (uint32_t g = 0; g < P; ++g) CK(wg_comm_set_peer_stages(comm[g], st, fl));
    auto step = [&](uint32_t g) { return wg_gemm_sharded(comm[g], WG_GEMM, WG_F16, WG_GATHER_PEER_STAGED, panel, c[g], mat(M, N), a[g], mat(mg, K), b[g], mat(K, N)); };
    EXPECT(step(0) == WG_OK, "step 1 of rank 0: %s", wg_last_error_string());
    const int rc = wg_ctx_sync(ctx[0]); // rank 1 never sent: the wait gives up after 300 ms
    EXPECT(rc == WG_ERR_HIP && std::strstr(wg_last_error_string(), "did not arrive") != nullptr, "time-out must surface in wg_ctx_sync: rc %d, %s", rc, wg_last_error_string());
    { // ... and what rank 0 holds for rank 1's rows is poison, not stale data
        std::vector<T> got((size_t)M * N);
        CK(wg_buf_read(ctx[0], c[0], 0, got.data(), got.size() * sizeof(T)));
        size_t nan = 0;
        for (uint32_t j = 0; j < N; ++j)
            for (uint32_t i = mg; i < M; ++i) nan += std::isnan((float)got[(size_t)j * M + i]) ? 1 : 0;
        EXPECT(nan == (size_t)mg * N, "poisoned rows: %zu of %zu are NaN", nan, (size_t)mg * N);
    }
    EXPECT(wg_ctx_sync(ctx[0]) == WG_OK, "a reported time-out is a cleared time-out: %s", wg_last_error_string());
    EXPECT(step(1) == WG_OK, "step 1 of rank 1 (late): %s", wg_last_error_string());
    for (int rep = 0; rep < 2; ++rep) // step 2 and 3 on both ranks
        for (uint32_t g = 0; g < P; ++g) EXPECT(step(g) == WG_OK, "step %d of rank %u: %s", rep + 2, g, wg_last_error_string());
    for (uint32_t g = 0; g < P; ++g) {
        EXPECT(wg_ctx_sync(ctx[g]) == WG_OK, "sync after the retry, rank %u: %s", g, wg_last_error_string());
        std::vector<T> got((size_t)M * N);
        CK(wg_buf_read(ctx[g], c[g], 0, got.data(), got.size() * sizeof(T)));
        char what[96];
        std::snprintf(what, sizeof what, "staged retry after a time-out, rank %u", g);
        check_product(what, got, A, false, B, M, N, K);
        EXPECT(wg_comm_flush(comm[g]) == WG_OK, "flush: %s", wg_last_error_string());
    }
    for (uint32_t g = 0; g < P; ++g) {
        wg_comm_destroy(comm[g]);
        wg_buf_destroy(a[g]); wg_buf_destroy(b[g]); wg_buf_destroy(c[g]);
        wg_ctx_destroy(ctx[g]);
    }
}

// Pipelined one-launch RCCL steps whose shapes alternate (small, big, small, big): a step's last panel is still waiting for its relayout in
// its own cube while the next step's kernel -- enqueued first -- writes the other cube. The cubes sit at fixed offsets, so a smaller step
// cannot land on the bigger previous step's cube (round-3 review: offsets used to follow the CURRENT step's M * N).
static void rccl_pipelined_alternating_shapes(wg_ctx *ctx, wg_comm *comm) {
    typedef _Float16 T;
    const uint32_t K = 256, shapes[4][3] = { { 4096, 4096, 1024 }, { 4096, 5120, 1024 }, { 4096, 4096, 2048 }, { 4096, 5120, 1280 } }; // M, N, panel_cols (>= 256 tiles each)
    auto A = rnd<T>((size_t)4096 * K, 51), B = rnd<T>((size_t)K * 5120, 52);
    wg_buf *a = nullptr, *b = nullptr, *c[4] = {};
    CK(wg_buf_create_init(ctx, A.data(), A.size() * sizeof(T), USAGE, &a));
    CK(wg_buf_create_init(ctx, B.data(), B.size() * sizeof(T), USAGE, &b));
    CK(wg_comm_set_pipelined(comm, 1));
    for (int s = 0; s < 4; ++s) {
        const uint32_t M = shapes[s][0], N = shapes[s][1];
        CK(wg_buf_create(ctx, (size_t)M * N * sizeof(T), USAGE, &c[s]));
        CK(wg_buf_fill_zero(ctx, c[s]));
        CK(wg_gemm_sharded(comm, WG_GEMM, WG_F16, WG_GATHER_RCCL, shapes[s][2], c[s], mat(M, N), a, mat(M, K), b, mat(K, N)));
    }
    CK(wg_comm_join(comm));
    for (int s = 0; s < 4; ++s) {
        const uint32_t M = shapes[s][0], N = shapes[s][1];
        std::vector<T> got((size_t)M * N), Bs(B.begin(), B.begin() + (size_t)K * N);
        CK(wg_buf_read(ctx, c[s], 0, got.data(), got.size() * sizeof(T)));
        char what[96];
        std::snprintf(what, sizeof what, "rccl pipelined, alternating shapes: step %d (%ux%u)", s, M, N);
        check_product(what, got, A, false, Bs, M, N, K);
        wg_buf_destroy(c[s]);
    }
    CK(wg_comm_set_pipelined(comm, 0));
    wg_buf_destroy(a); wg_buf_destroy(b);
}

static void cube_relayout(wg_ctx *ctx) {
    const uint32_t mg = 12, np = 8, P = 3, M = mg * P, ld = M + 4; // f16, 24-byte row blocks: the 8-byte vector path
    std::vector<_Float16> cube((size_t)mg * np * P), out((size_t)ld * np + 8, (_Float16)-1.f);
    for (size_t i = 0; i < cube.size(); ++i) cube[i] = (_Float16)(float)(i % 2048);
    wg_buf *bc = nullptr, *bo = nullptr;
    CK(wg_buf_create_init(ctx, cube.data(), cube.size() * 2, USAGE, &bc));
    CK(wg_buf_create_init(ctx, out.data(), out.size() * 2, USAGE, &bo));
    wg_view_shape cs = { { mg, np, P }, mg, mg * np, 0 }, os = { { M, np, 1 }, ld, ld * np, 4 };
    CK(wg_cube_to_matrix(ctx, WG_F16, bc, cs, bo, os));
    std::vector<_Float16> got(out.size());
    CK(wg_buf_read(ctx, bo, 0, got.data(), got.size() * 2));
    size_t bad = 0;
    for (uint32_t g = 0; g < P; ++g)
        for (uint32_t j = 0; j < np; ++j)
            for (uint32_t i = 0; i < mg; ++i) bad += got[4 + (size_t)j * ld + g * mg + i] != cube[((size_t)g * np + j) * mg + i];
    for (uint32_t j = 0; j < np; ++j)
        for (uint32_t i = M; i < ld; ++i) bad += got[4 + (size_t)j * ld + i] != (_Float16)-1.f; // the gap between columns is untouched
    EXPECT(bad == 0, "cube_to_matrix: %zu wrong elements", bad);
    wg_view_shape wrong = os;
    wrong.size[0] = M + 4;
    EXPECT(wg_cube_to_matrix(ctx, WG_F16, bc, cs, bo, wrong) == WG_ERR_DIM_MISMATCH, "cube_to_matrix accepted mismatching shapes");
    wg_buf_destroy(bc); wg_buf_destroy(bo);
}

// argument checking of the sharded entry point, and the IPC export of a buffer that wraps an INTERIOR pointer of an allocation
static void errors_and_export(wg_ctx *ctx) {
    wg_comm *comm = nullptr;
    CK(wg_comm_create(ctx, 2, 0, nullptr, &comm)); // two ranks, no collective library, no peers registered
    wg_buf *a = nullptr, *b = nullptr, *c = nullptr;
    CK(wg_buf_create(ctx, 256 * 64 * 4, USAGE, &a));
    CK(wg_buf_create(ctx, 64 * 128 * 4, USAGE, &b));
    CK(wg_buf_create(ctx, 512 * 128 * 4, USAGE, &c));
    // M = 2 * rows(a_rows) must hold; K and N must match (gemm.rs:91-95 on the sharded operands)
    EXPECT(wg_gemm_sharded(comm, WG_GEMM, WG_F32, WG_GATHER_NONE, 0, c, mat(500, 128), a, mat(256, 64), b, mat(64, 128)) == WG_ERR_DIM_MISMATCH, "M != P * rows");
    EXPECT(std::strstr(wg_last_error_string(), "Gemm: dimension mismatch.") != nullptr, "message: %s", wg_last_error_string());
    EXPECT(wg_gemm_sharded(comm, WG_GEMM, WG_F32, WG_GATHER_NONE, 0, c, mat(512, 128), a, mat(256, 64), b, mat(60, 128)) == WG_ERR_DIM_MISMATCH, "K mismatch");
    EXPECT(wg_gemm_sharded(comm, WG_GEMM, WG_F32, WG_GATHER_NONE, 6, c, mat(512, 128), a, mat(256, 64), b, mat(64, 128)) == WG_ERR_PRECONDITION, "panel_cols not a multiple of 4");
    EXPECT(wg_gemm_sharded(comm, WG_GEMM, WG_F32, WG_GATHER_RCCL, 0, c, mat(512, 128), a, mat(256, 64), b, mat(64, 128)) == WG_ERR_UNSUPPORTED, "RCCL mode without a unique id");
    EXPECT(wg_gemm_sharded(comm, WG_GEMM, WG_F32, (wg_gather_mode)1, 0, c, mat(512, 128), a, mat(256, 64), b, mat(64, 128)) == WG_ERR_INVALID_ARG, "gather mode 1 (the removed rect-copy engine)");
    EXPECT(wg_gemm_sharded(comm, WG_GEMM, WG_F32, (wg_gather_mode)9, 0, c, mat(512, 128), a, mat(256, 64), b, mat(64, 128)) == WG_ERR_INVALID_ARG, "unknown mode");
    EXPECT(wg_gemm_sharded(comm, WG_GEMM, WG_F32, WG_GATHER_NONE, 0, c, mat(512, 128), a, mat(256, 64), b, mat(64, 128)) == WG_OK, "local-only mode: %s", wg_last_error_string());
    EXPECT(wg_all_gather(comm, WG_F32, c, 0, 16) == WG_ERR_UNSUPPORTED && wg_comm_barrier(comm) == WG_ERR_UNSUPPORTED, "collectives without a library");
    // export of a wrapped interior pointer: the handle names the base allocation and carries the offset
    wg_buf *inner = nullptr;
    CK(wg_buf_wrap(ctx, (char *)wg_buf_device_ptr(c) + 4096, 8192, &inner));
    unsigned char h[WG_IPC_HANDLE_BYTES], h0[WG_IPC_HANDLE_BYTES];
    CK(wg_buf_ipc_export(inner, h));
    CK(wg_buf_ipc_export(c, h0));
    uint64_t off = 0, bytes = 0, off0 = 1;
    std::memcpy(&off, h + 64, 8); std::memcpy(&bytes, h + 72, 8); std::memcpy(&off0, h0 + 64, 8);
    EXPECT(off - off0 == 4096 && bytes == 8192 && std::memcmp(h, h0, 64) == 0, "interior export: offset %llu (+%llu), bytes %llu", (unsigned long long)off, (unsigned long long)off0, (unsigned long long)bytes);
    wg_buf_destroy(inner);
    CK(wg_ctx_sync(ctx));
    wg_comm_destroy(comm);
    wg_buf_destroy(a); wg_buf_destroy(b); wg_buf_destroy(c);
}

static double now_s() {
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
static double t_start = 0;
// a phase line per section on stderr (where this test's minutes go: the collective library's first communicator, see tests/test_cpp_facade.py)
#define PHASE(name) std::fprintf(stderr, "[comm_tests %7.2f s] %s\n", now_s() - t_start, name)

int main() {
    t_start = now_s();
    wg_ctx *ctx = nullptr;
    if (wg_ctx_create(0, &ctx) != WG_OK) { std::printf("no device: %s\n", wg_last_error_string()); return 2; }
    PHASE("context created");
    cube_relayout(ctx);
    errors_and_export(ctx);
    PHASE("relayout + error paths done; creating the RCCL communicator");

    unsigned char id[WG_COMM_ID_BYTES];
    wg_comm *comm = nullptr;
    if (wg_comm_unique_id(id) != WG_OK || wg_comm_create(ctx, 1, 0, id, &comm) != WG_OK) {
        std::printf("FAIL: RCCL communicator: %s\n", wg_last_error_string());
        return 1;
    }
    PHASE("RCCL communicator created");
    EXPECT(wg_comm_has_collectives(comm) == 1 && wg_comm_size(comm) == 1 && wg_comm_rank(comm) == 0, "communicator facts");
    { // all-gather round trip with one rank: in place, the range must survive; then a barrier
        std::vector<float> v(4096);
        for (size_t i = 0; i < v.size(); ++i) v[i] = (float)i;
        wg_buf *b = nullptr;
        if (wg_buf_create_init(ctx, v.data(), v.size() * 4, USAGE, &b) == WG_OK) {
            EXPECT(wg_all_gather(comm, WG_F32, b, 1024, 2048) == WG_OK, "wg_all_gather: %s", wg_last_error_string());
            EXPECT(wg_all_gather(comm, WG_F32, b, 1024, 4096) == WG_ERR_OUT_OF_BOUNDS, "wg_all_gather accepted a range past the buffer");
            EXPECT(wg_comm_join(comm) == WG_OK && wg_comm_barrier(comm) == WG_OK, "join/barrier: %s", wg_last_error_string());
            std::vector<float> back(v.size());
            EXPECT(wg_buf_read(ctx, b, 0, back.data(), back.size() * 4) == WG_OK && back == v, "all-gather round trip changed the data");
            wg_buf_destroy(b);
        }
    }
    PHASE("first collective + barrier done");
    rccl_one_rank<float>(ctx, comm, false, 512, 768, 256, 256);
    rccl_one_rank<float>(ctx, comm, true, 512, 768, 256, 512); // ragged last panel
    rccl_one_rank<_Float16>(ctx, comm, false, 1024, 1280, 512, 512);
    rccl_one_rank<_Float16>(ctx, comm, true, 1024, 1280, 512, 0);
    {
        int reported = -1;
        EXPECT(wg_comm_reported_size(comm, &reported) == WG_OK && reported == 1, "ncclCommCount: %d (%s)", reported, wg_last_error_string());
    }
    rccl_pipelined_alternating_shapes(ctx, comm);
    rccl_one_rank_tapered(ctx, comm);
    PHASE("RCCL engine cases done");
    wg_comm_destroy(comm);
    PHASE("RCCL communicator destroyed");

    staged_two_ranks<float>(false, 512, 768, 256, 256);
    staged_two_ranks<_Float16>(false, 1024, 1280, 512, 512);
    staged_two_ranks<_Float16>(true, 1024, 1280, 512, 768);
    staged_two_ranks<_Float16>(false, 1024, 1280, 512, 512, /*pipelined=*/true);
    staged_two_ranks<float>(true, 512, 768, 256, 256, /*pipelined=*/true);
    PHASE("staged engine cases done");
    staged_timeout_and_retry();
    PHASE("time-out + retry done");
    wg_ctx_destroy(ctx);
    if (failures == 0) std::printf("ALL OK\n");
    return failures ? 1 : 0;
}
