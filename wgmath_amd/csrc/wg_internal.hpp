// Internal declarations shared by the runtime, the operator front-end and the kernel launchers.
// Nothing here is part of the ABI (include/wgebra_hip.h is).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

#include "../../include/wgebra_hip.h"

// Destroy calls that arrived while a thread was recording (see wg_defer_if_capturing). The batch belongs to every recording that was open on that
// thread meanwhile and, after wg_encoder_finish, to their command buffers: the calls run when the LAST of those is destroyed (or at once if a
// recording is abandoned) -- a buffer a recorded dispatch used and the host dropped during the recording stays allocated for as long as something can
// still replay into it. Context destroys run last (the buffer destroys before them dereference buf->ctx).
struct wg_deferred_batch {
    std::vector<std::function<void()>> items, ctx_items;
    ~wg_deferred_batch() {
        for (auto &f : items) f();
        for (auto &f : ctx_items) f();
    }
};

struct wg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = true;
    bool recording = false;
    std::thread::id recording_thread;                 // the thread that called wg_encoder_begin (thread-local capture mode)
    std::shared_ptr<wg_deferred_batch> capture_batch; // while recording: the deferred-destroy batch of the recording thread
    void *workspace = nullptr; // scratch for split reductions (GEMV partials)
    size_t workspace_bytes = 0;
    void *tr_workspace = nullptr; // second scratch: transposed operand of the row-major GemmTr (lives across the GEMM that may use `workspace`)
    size_t tr_workspace_bytes = 0;
    void *stage_workspace = nullptr; // fourth scratch: dense zero-padded copies of operator views that are not vec4-aligned (api.hip)
    size_t stage_workspace_bytes = 0;
    void *pad_workspace = nullptr; // third scratch: zero-padded operand copies of f16 GEMMs whose shapes the MFMA kernels do not take as they are
    size_t pad_workspace_bytes = 0;
    int compute_units = 0;
    bool uneven_xcds = false; // CU-masked stream whose missing CUs all come from one XCD (runtime.hip): static tile -> XCD maps would wait for the short XCD
    unsigned *flags = nullptr;           // a few zeroed device words (arrival counters of fused epilogues), created on first use
    unsigned long long *tile_queues = nullptr; // f16 Gemm tile scheduler: 8 per-XCD queue words, 128 bytes apart (gemm_f16.hip), created on first use
    // f16 Gemm, calibrated shares across XCDs (gemm_f16.hip "balance"): measured relative time per stage of the workgroup slots b % 8,
    // the device block that holds the accumulators (and the prefix -> suffix flags), and the snapshot in flight on the side stream
    struct F16Balance {
        unsigned long long *dev = nullptr;  // [0, 16): calib accumulators (ticks, stages per slot); flags (u32) from byte 2048 on: 1024 of them
        unsigned long long *host = nullptr; // pinned: 16 words, the snapshot
        unsigned long long prev[16] = { 0 };
        hipStream_t side = nullptr;
        hipEvent_t ev = nullptr;
        bool inflight = false, valid = false, have_prev = false;
        double rel[8] = { 1, 1, 1, 1, 1, 1, 1, 1 };
        uint32_t epoch = 0, updates = 0;
        void *scratch = nullptr;            // raw f32 accumulator tiles of the prefix units (grow-only)
        size_t scratch_bytes = 0;
    } bal;
    int tuning[WG_TUNE_COUNT_] = { 0, -1, -1, -1, 0, -1, 0, 0, -1, -1 }; // wg_ctx_set_tuning; defaults read from the environment once, at creation
    void *debug_stamps = nullptr;        // pinned: the two stamp arrays of wg_debug_clock_begin / _end (debug.hip)
    bool debug_clock_open = false;
    uint32_t lds_attr_bits = 0;          // likewise for gemv_t_lds_kernel's instantiations (3 right-hand-side tiles x 5 workgroup shapes)
    uint32_t func_attr_bits = 0;         // hipFuncSetAttribute calls already made for this context's device (gemv.hip: GemvTr's 128 KiB dynamic LDS)
    int live_cmdbufs = 0;                // recorded command buffers not yet destroyed: their graphs hold scratch pointers
    std::vector<void *> retired_scratch; // outgrown scratch regions a live command buffer may still replay into
    // pinned host words that kernels of this context's stream raise when something went wrong asynchronously (a communicator's wait
    // kernel timing out on a peer): checked -- reported once, then cleared -- by wg_ctx_sync, wg_buf_read and the communicator's calls
    struct AsyncError { uint32_t *word; const char *what; uint32_t *dev_word; };
    std::vector<AsyncError> async_errors;
};
// A destroy call that arrives while THIS THREAD records a command buffer (hipStreamBeginCapture, thread-local mode) must not run now: hipFree /
// hipStreamSynchronize from the capturing thread are prohibited and invalidate the capture. It happens -- a garbage-collected host object
// (Python's cyclic GC, a Rust drop at scope end) owns a buffer or a command buffer -- and it is legal in the reference (wgpu keeps a dropped
// buffer alive until the submission using it retires). Returns true if `fn` was queued into the thread's wg_deferred_batch (it runs when the last
// command buffer recorded meanwhile is destroyed), otherwise the caller runs it itself. `is_ctx`: a context destroy (ordered after the others).
bool wg_defer_if_capturing(std::function<void()> fn, bool is_ctx = false);
void wg_ctx_register_async_error(wg_ctx *ctx, uint32_t *word, const char *what, uint32_t *dev_word = nullptr); // dev_word: a device-side twin cleared with it
void wg_ctx_unregister_async_error(wg_ctx *ctx, uint32_t *word);
int wg_ctx_check_async(wg_ctx *ctx); // WG_ERR_HIP + message "<what> <word - 1>" if a registered word is set (and clears it)

struct wg_buf {
    wg_ctx *ctx = nullptr;
    void *ptr = nullptr; // device pointer (or pinned-host pointer for MAP_* buffers; device-accessible)
    size_t bytes = 0;
    uint32_t usage = 0;
    bool owned = true;
    bool host_pinned = false;
    void *ipc_base = nullptr; // wg_buf_ipc_open: base of the mapped peer allocation (closed with the buffer)
    bool borrowed = false;    // a view owned by another object (a communicator's staging cube): wg_buf_destroy leaves it alone
};

struct wg_cmdbuf {
    wg_ctx *ctx = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::shared_ptr<wg_deferred_batch> deferred; // destroys that arrived during the recording: released (and, by the last holder, run) in wg_cmdbuf_destroy
};

struct wg_timestamps {
    wg_ctx *ctx = nullptr;
    std::vector<hipEvent_t> events;
    std::vector<uint8_t> written; // slot i holds a recorded event (a slot can be reserved -- next_query_indices -- and written later, or never)
    uint32_t len = 0;
};

// ---- error plumbing -------------------------------------------------------------------------------
int wg_set_error(int status, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void wg_clear_error();

#define WG_HIP_TRY(expr)                                                                              \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            return wg_set_error(WG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                                __FILE__, __LINE__);                                                  \
    } while (0)

// Grow-only scratch. Fails while recording if it would have to allocate.
int wg_ctx_workspace(wg_ctx *ctx, size_t bytes, void **out);
int wg_ctx_tr_workspace(wg_ctx *ctx, size_t bytes, void **out);
int wg_ctx_pad_workspace(wg_ctx *ctx, size_t bytes, void **out);
int wg_ctx_bal_workspace(wg_ctx *ctx, size_t bytes, void **out);
int wg_ctx_stage_workspace(wg_ctx *ctx, size_t bytes, void **out); // dense copies of views that are not vec4-aligned (api.hip)
int wgk_stage_copy(wg_ctx *ctx, wg_dtype dtype, void *dst, uint32_t ld_dst, uint64_t dst_batch, uint32_t rd, uint32_t cd, const void *src,
                   uint32_t ld_src, uint64_t src_batch, uint32_t rs, uint32_t cs, uint32_t nmats);
int wgk_transpose(wg_ctx *ctx, wg_dtype dtype, uint32_t rows, uint32_t cols, uint32_t nmats, const void *src, uint32_t ld_src,
                  uint64_t src_batch, void *dst, uint32_t ld_dst, uint64_t dst_batch);

// 16-byte streaming load with the non-temporal hint (data read exactly once: keep it out of the way in L2/MALL)
typedef float wg_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 wg_ld_nt(const float4 *p) {
    wg_f4 v = __builtin_nontemporal_load(reinterpret_cast<const wg_f4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
// ... at an address that is only element-aligned: the memory pipeline runs in unaligned-access mode (one global_load_dwordx4 either way, the aligned rate at
// 4-byte offsets: tools/cpp/unaligned_probe.hip, profiles/r06_unaligned_probe.txt); the pointer type states the alignment that is really there
typedef wg_f4 __attribute__((aligned(4))) wg_f4_u;
__device__ __forceinline__ float4 wg_ld_nt_u(const float *p) {
    wg_f4 v = __builtin_nontemporal_load(reinterpret_cast<const wg_f4_u *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 wg_ld_u(const float *p) {
    wg_f4 v = *reinterpret_cast<const wg_f4_u *>(p);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void wg_st_u(float *p, float4 v) { *reinterpret_cast<wg_f4_u *>(p) = wg_f4{ v.x, v.y, v.z, v.w }; }

static inline size_t wg_dtype_size(wg_dtype d) { return d == WG_F16 ? 2 : 4; }

// ---- kernel launchers (one per .hip file). All enqueue on ctx->stream and return a wg_status. -----
// Pointers are already offset to the first element of the view; ld* / batch strides in elements.

struct wgk_mat {
    const void *ptr; // first element of the view (matrix 0)
    uint32_t ld;     // column stride (elements)
    uint64_t batch;  // matrix stride (elements)
};

int wgk_op_assign(wg_ctx *ctx, int op, wg_dtype dtype, void *a, const void *b, uint32_t n, float alpha = 0.f); // op 5 = axpy

int wgk_reduce_fast(wg_ctx *ctx, int op, wg_dtype dtype, const void *base, uint32_t n, void *result);
int wgk_reduce(wg_ctx *ctx, int op, wg_dtype dtype, const void *base, uint32_t n, uint32_t ncols, uint32_t nmats,
               uint32_t stride, uint32_t stride_mat, void *results); // results: one element of `dtype` per vector

// out[rows_out, nrhs, nmats]; trans == false: out = m * v, m is (rows_out x k); trans: out = m^T v, m is (k x rows_out)
int wgk_gemv(wg_ctx *ctx, bool trans, wg_dtype dtype, uint32_t rows_out, uint32_t k, uint32_t nrhs, uint32_t nmats,
             void *out, uint32_t out_ld, uint64_t out_batch, wgk_mat m, wgk_mat v);

// out (M x N) = op(m1) (M x K) * m2 (K x N); trans: m1 stored K x M
// fused Gemv + Reduce (one launch) for launch-bound sizes; WG_ERR_UNSUPPORTED (no error message) = not that shape family
// Gemv / GemvTr on a matrix view of any alignment (gemv_any.hip): m is the R x C view as stored, v / out hold nrhs columns
int wgk_gemv_any(wg_ctx *ctx, bool trans, wg_dtype dtype, uint32_t R, uint32_t C, uint32_t nrhs, uint32_t nmats, void *out, uint32_t out_ld, uint64_t out_batch,
                 wgk_mat m, wgk_mat v);
int wgk_gemv_small_reduce(wg_ctx *ctx, int op, uint32_t rows_out, uint32_t k, float *y, wgk_mat m, wgk_mat v, unsigned *counter, float *result);

int wgk_gemm_f32(wg_ctx *ctx, bool trans, uint32_t M, uint32_t N, uint32_t K, uint32_t nmats,
                 float *out, uint32_t out_ld, uint64_t out_batch, wgk_mat m1, wgk_mat m2, float alpha = 1.f, float beta = 0.f);
// gemm_f32_mid.hip: bm x bn tiles (128 x 128, 128 x 64, 64 x 128 on 2 x 2 waves; 64 x 64, 64 x 32, 32 x 64 with K split over the waves), whole K per workgroup
// for nsplit = 1; nsplit >= 2 cuts K across workgroups too (f32 slabs in the context's workspace + a reduce launch: few tiles, long K);
// _ok: the shapes / strides it takes
bool wgk_gemm_f32_mid_ok(uint32_t M, uint32_t N, uint32_t K, uint32_t nmats, wgk_mat m1, wgk_mat m2);
int wgk_gemm_f32_mid(wg_ctx *ctx, bool trans, int bm, int bn, uint32_t M, uint32_t N, uint32_t K, uint32_t nmats, float *out, uint32_t out_ld, uint64_t out_batch,
                     wgk_mat m1, wgk_mat m2, float alpha, float beta, uint32_t nsplit = 1);
int wgk_gemm_f32_skinny(wg_ctx *ctx, bool trans, uint32_t M, uint32_t N, uint32_t K, uint32_t nmats, float *out, uint32_t out_ld, uint64_t out_batch,
                        wgk_mat m1, wgk_mat m2, float alpha, float beta, uint32_t out_row_stride = 1, bool m2_kmajor = false, uint32_t ns_force = 0);
// f16 GemmTr with N <= 16 on the few-column streaming kernel (gemm_f32_skinny.hip, T = _Float16): HBM-bound, m1 read once
int wgk_gemm_f16_skinny(wg_ctx *ctx, uint32_t M, uint32_t N, uint32_t K, uint32_t nmats, __half *out, uint32_t out_ld, uint64_t out_batch, wgk_mat m1, wgk_mat m2,
                        float alpha, float beta);
// N-panels with arrival counters (comm.hip's one-launch-per-step sharded Gemm; gemm_f16_common.hpp PanelArgs): `out` is where the product's column 0
// would sit if every panel were `cols` wide and the cube had one rank (i.e. panel 0's slot of this rank), leading dimension out_ld; panel p (first
// column c0, np columns) lives at out + c0 * col_stride + slot_rows * (np - cols) ... see m16_tile. n_main panels of `cols` columns, then n_tail
// (1 .. 8) panels of tail_cols[] columns (multiples of 256; the last one = whatever is left of N, may be ragged). counters[p]
// grows by wgk_panel_goal(...) per launch, complete once every tile of panel p is in memory (running totals, never reset by the kernel
// or its launcher: the caller keeps the totals and waits for them). wgk_gemm_f16 returns
// WG_ERR_UNSUPPORTED -- silently -- when the product does not take that path (shapes off the MFMA fast path, fewer tiles than CUs, ...):
// the caller then launches panel by panel.
struct wgk_panels {
    uint32_t cols = 0, n_main = 0, n_tail = 0;
    uint32_t tail_cols[8] = { 0 };
    uint64_t col_stride = 0, slot_rows = 0;
    uint32_t *counters = nullptr;
};
// what counters[p] reads when panel p (np columns of an M-row product) is complete: one arrival per wave, 4 waves per 256 x 256 tile
static inline uint32_t wgk_panel_goal(uint32_t M, uint32_t np) { return 4u * ((M + 255u) / 256u) * ((np + 255u) / 256u); }
int wgk_gemm_f16(wg_ctx *ctx, bool trans, uint32_t M, uint32_t N, uint32_t K, uint32_t nmats,
                 __half *out, uint32_t out_ld, uint64_t out_batch, wgk_mat m1, wgk_mat m2, float alpha = 1.f, float beta = 0.f,
                 const wgk_panels *panels = nullptr);
// gemm_f16_nt.hip: out (M x N, column-major) = a (M x K, m-contiguous: ld between k) * b (K x N, N-CONTIGUOUS: element (k, n) at n + k * ld) -- the row-major GemmTr in
// column-major terms. WG_ERR_UNSUPPORTED without a message: not a product that kernel takes (the caller transposes `b` and calls wgk_gemm_f16).
int wgk_gemm_f16_nt(wg_ctx *ctx, uint32_t M, uint32_t N, uint32_t K, uint32_t nmats, __half *out, uint32_t out_ld, uint64_t out_batch, wgk_mat a_mcontig, wgk_mat b_ncontig,
                    float alpha = 1.f, float beta = 0.f);
// gemm_f32.hip: the same for f32 (the 256 x 128 tile kernel with n-contiguous B tile bodies; whole K per workgroup)
int wgk_gemm_f32_nt(wg_ctx *ctx, uint32_t M, uint32_t N, uint32_t K, uint32_t nmats, float *out, uint32_t out_ld, uint64_t out_batch, wgk_mat a_mcontig, wgk_mat b_ncontig,
                    float alpha = 1.f, float beta = 0.f);

// The f16 product out_rows (M x N) = op(m1) m2 as ONE launch over N-panels with completion flags (api.hip; the operator front-end's checks
// on m1 / m2, then wgk_gemm_f16 with `panels`). WG_ERR_UNSUPPORTED without a message: not that kind of product, launch panel by panel.
int wg_gemm_f16_panels(wg_ctx *ctx, bool tr, void *out_panel0, uint32_t ldc, const wg_buf *m1, wg_view_shape m1_shape, const wg_buf *m2, wg_view_shape m2_shape,
                       const wgk_panels &panels);

// split-K (splitk.hip)
uint32_t wg_splitk_plan(uint64_t tiles, uint32_t slots, uint32_t k_units, uint32_t min_units, uint64_t out_elems, uint64_t max_ws_bytes);
// f32, beta = 0: out[z][r * row_stride + c * col_stride] = alpha * sum over splits (the transposed output of few-row products)
int wg_splitk_reduce_strided(wg_ctx *ctx, const float *part, uint32_t nsplit, uint32_t M, uint32_t N, uint32_t nmats, float *out,
                             uint32_t row_stride, uint32_t col_stride, uint64_t c_batch, float alpha);
int wg_splitk_reduce(wg_ctx *ctx, const float *part, uint32_t nsplit, uint32_t M, uint32_t N, uint32_t nmats, wg_dtype dtype, void *out,
                     uint32_t ldc, uint64_t c_batch, float alpha = 1.f, float beta = 0.f);
