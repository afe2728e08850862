// Runtime half of the boundary: contexts (device + in-order stream), buffers, record/replay, timestamps.
// Replaces wgcore's wgpu plumbing: gpu.rs:24-58 (GpuInstance), tensor.rs:115-186,227-264,300-384 (buffer
// create/copy/read), kernel.rs:15-27 (compute pass), timestamps.rs (GpuTimestamps).
#include "wg_internal.hpp"

#include <vector>

#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

namespace {
thread_local std::string g_last_error;
}

int wg_set_error(int status, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return status;
}
void wg_clear_error() { g_last_error.clear(); }

// Grow-only scratch regions. A recorded command buffer (hipGraphExec) has the scratch pointer of its kernels baked in and may be
// submitted many times (wgebra_hip.h, wg_queue_submit), so a region that a live command buffer may reference is never freed on regrow:
// it is retired and released when the last command buffer of the context is destroyed (or with the context).
static int grow_scratch(wg_ctx *ctx, void **region, size_t *region_bytes, size_t bytes, const char *what, void **out) {
    if (bytes > *region_bytes) {
        if (ctx->recording)
            return wg_set_error(WG_ERR_WORKSPACE, "%s of %zu bytes needed while recording: run the call once outside the recording first "
                                "(or wg_ctx_reserve_workspace)", what, bytes);
        WG_HIP_TRY(hipSetDevice(ctx->device));
        // in-order stream: earlier users of the old region must be done before it is freed
        WG_HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (*region) {
            if (ctx->live_cmdbufs > 0) ctx->retired_scratch.push_back(*region);
            else WG_HIP_TRY(hipFree(*region));
        }
        *region = nullptr;
        *region_bytes = 0;
        size_t want = (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
        WG_HIP_TRY(hipMalloc(region, want));
        *region_bytes = want;
    }
    *out = *region;
    return WG_OK;
}
int wg_ctx_workspace(wg_ctx *ctx, size_t bytes, void **out) { return grow_scratch(ctx, &ctx->workspace, &ctx->workspace_bytes, bytes, "workspace", out); }
int wg_ctx_tr_workspace(wg_ctx *ctx, size_t bytes, void **out) {
    return grow_scratch(ctx, &ctx->tr_workspace, &ctx->tr_workspace_bytes, bytes, "transpose workspace", out);
}
int wg_ctx_pad_workspace(wg_ctx *ctx, size_t bytes, void **out) {
    return grow_scratch(ctx, &ctx->pad_workspace, &ctx->pad_workspace_bytes, bytes, "padding workspace", out);
}
void wg_ctx_register_async_error(wg_ctx *ctx, uint32_t *word, const char *what, uint32_t *dev_word) { ctx->async_errors.push_back({ word, what, dev_word }); }
void wg_ctx_unregister_async_error(wg_ctx *ctx, uint32_t *word) {
    for (size_t i = 0; i < ctx->async_errors.size(); ++i)
        if (ctx->async_errors[i].word == word) { ctx->async_errors.erase(ctx->async_errors.begin() + (long)i); return; }
}
int wg_ctx_check_async(wg_ctx *ctx) {
    for (const wg_ctx::AsyncError &e : ctx->async_errors) {
        const uint32_t v = *(volatile uint32_t *)e.word;
        if (v) {
            *(volatile uint32_t *)e.word = 0;
            // the device-side twin (what the kernels themselves test) is cleared with it, in stream order: a reported error is a cleared error
            if (e.dev_word) (void)hipMemsetAsync(e.dev_word, 0, sizeof(uint32_t), ctx->stream);
            return wg_set_error(WG_ERR_HIP, "%s %u", e.what, v - 1u);
        }
    }
    return WG_OK;
}

int wg_ctx_stage_workspace(wg_ctx *ctx, size_t bytes, void **out) {
    return grow_scratch(ctx, &ctx->stage_workspace, &ctx->stage_workspace_bytes, bytes, "staging workspace", out);
}
int wg_ctx_bal_workspace(wg_ctx *ctx, size_t bytes, void **out) {
    return grow_scratch(ctx, &ctx->bal.scratch, &ctx->bal.scratch_bytes, bytes, "balance workspace", out);
}

namespace {
thread_local int t_capturing = 0;                          // recordings open on this thread
thread_local std::shared_ptr<wg_deferred_batch> t_batch;   // destroy calls that arrived meanwhile (shared with those recordings)
void capture_began(wg_ctx *ctx) {
    if (t_capturing++ == 0 || !t_batch) t_batch = std::make_shared<wg_deferred_batch>();
    ctx->capture_batch = t_batch;
    ctx->recording_thread = std::this_thread::get_id();
}
// The recording of `ctx` is over (on the thread that began it). Returns the batch for the command buffer to hold; dropping the returned pointer
// without storing it (an abandoned recording) runs the queued destroys as soon as no other recording of this thread shares them.
std::shared_ptr<wg_deferred_batch> capture_ended(wg_ctx *ctx) {
    std::shared_ptr<wg_deferred_batch> batch = std::move(ctx->capture_batch);
    ctx->capture_batch.reset();
    if (t_capturing > 0 && --t_capturing == 0) t_batch.reset();
    return batch;
}
} // namespace
bool wg_defer_if_capturing(std::function<void()> fn, bool is_ctx) {
    if (t_capturing == 0 || !t_batch) return false;
    (is_ctx ? t_batch->ctx_items : t_batch->items).push_back(std::move(fn));
    return true;
}

extern "C" {

int wg_abi_version(void) { return WGEBRA_HIP_ABI_VERSION; }

const char *wg_last_error_string(void) { return g_last_error.c_str(); }

int wg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

static int ctx_create_common(int device, hipStream_t stream, bool owns, wg_ctx **out) {
    if (!out) return wg_set_error(WG_ERR_INVALID_ARG, "wg_ctx_create: out is NULL");
    *out = nullptr;
    int n = wg_device_count();
    if (n <= 0) return wg_set_error(WG_ERR_NO_DEVICE, "Failed to initialize gpu adapter.: no HIP device visible");
    if (device < 0 || device >= n) return wg_set_error(WG_ERR_INVALID_ARG, "wg_ctx_create: device %d not in [0,%d)", device, n);
    WG_HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    WG_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return wg_set_error(WG_ERR_NO_DEVICE, "device %d is %s; this library ships gfx950 (MI355X) code objects only",
                            device, prop.gcnArchName);
    wg_ctx *ctx = new (std::nothrow) wg_ctx();
    if (!ctx) return wg_set_error(WG_ERR_HIP, "out of host memory");
    ctx->device = device;
    ctx->compute_units = prop.multiProcessorCount;
    if (owns) {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete ctx;
            return wg_set_error(WG_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
        }
    } else {
        ctx->stream = stream;
    }
    ctx->owns_stream = owns;
    // kernel-selection knobs: the environment is consulted here and nowhere on the dispatch path
    static const struct { const char *env; wg_tuning key; } knobs[] = {
        { "WG_F16_TILE", WG_TUNE_F16_TILE }, { "WG_F16_SCHED", WG_TUNE_F16_SCHED }, { "WG_F32_SKINNY", WG_TUNE_F32_SKINNY },
        { "WG_F32_PANELS", WG_TUNE_F32_PANELS }, { "WG_F16_BALANCE", WG_TUNE_F16_BALANCE }, { "WG_F32_MID", WG_TUNE_F32_MID }, { "WG_F32_MID_SPLIT", WG_TUNE_F32_MID_SPLIT },
        { "WG_GEMVT_LDS", WG_TUNE_GEMVT_LDS }, { "WG_F16_CONT", WG_TUNE_F16_CONT }, { "WG_RM_TR_NATIVE", WG_TUNE_RM_TR_NATIVE } };
    for (const auto &k : knobs)
        if (const char *v = getenv(k.env)) {
            // the same validation as wg_ctx_set_tuning: a value the knob does not take is ignored (with a note), never silently reinterpreted
            if (wg_ctx_set_tuning(ctx, k.key, atoi(v)) != WG_OK) fprintf(stderr, "libwgebra_hip: ignoring %s=%s (%s)\n", k.env, v, wg_last_error_string());
        }
    *out = ctx;
    return WG_OK;
}

int wg_ctx_create(int device, wg_ctx **out) { return ctx_create_common(device, nullptr, true, out); }
int wg_ctx_create_on_stream(int device, void *hip_stream, wg_ctx **out) {
    return ctx_create_common(device, (hipStream_t)hip_stream, false, out);
}

// A context whose stream may only use `cu_count` of the device's compute units (hipExtStreamCreateWithCUMask; which ones: below). For multi-GPU runs: the GEMM stream leaves a few CUs free, so that the collective library's copy kernels (a second
// queue) start at once instead of waiting for a GEMM workgroup -- which needs a whole CU -- to retire. The kernels' tile/split
// heuristics see `cu_count` CUs.
static int ctx_create_masked(int device, uint32_t cu_count, bool one_xcd, wg_ctx **out) {
    if (!out) return wg_set_error(WG_ERR_INVALID_ARG, "wg_ctx_create_with_cu_count: out is NULL");
    *out = nullptr;
    int n = wg_device_count();
    if (n <= 0) return wg_set_error(WG_ERR_NO_DEVICE, "Failed to initialize gpu adapter.: no HIP device visible");
    if (device < 0 || device >= n) return wg_set_error(WG_ERR_INVALID_ARG, "wg_ctx_create_with_cu_count: device %d not in [0,%d)", device, n);
    WG_HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    WG_HIP_TRY(hipGetDeviceProperties(&prop, device));
    const uint32_t total = (uint32_t)prop.multiProcessorCount;
    if (cu_count == 0 || cu_count > total) return wg_set_error(WG_ERR_INVALID_ARG, "wg_ctx_create_with_cu_count: %u not in [1,%u]", cu_count, total);
    std::vector<uint32_t> mask((total + 31) / 32, 0u);
    // Which CUs stay free: mask bit i belongs to XCD i % 8. Up to a whole XCD's worth, ALL of them come from the last XCD -- seven XCDs keep their
    // 32 CUs and with them the 4 x 8 tile patches of the f16 Gemm in their L2 (the tile scheduler lets the others take the short XCD's tiles):
    // 8192 x 32768 x 32768 on 248 CUs 13.80 -> 13.49 ms against one CU from every XCD (256 CUs: 13.17; profiles/r03_evidence.md section 3).
    // Only on request (wg_ctx_create_with_cu_count_one_xcd): kernels with a static tile -> XCD map (everything but the big f16 Gemm) would wait
    // for the short XCD. wg_ctx_create_with_cu_count spreads the missing CUs evenly, as does a request for a whole XCD's worth or more.
    const uint32_t missing = total - cu_count;
    const bool uneven = one_xcd && missing > 0 && missing < total / 8u; // (never a whole XCD)
    if (uneven) {
        for (auto &w : mask) w = 0xffffffffu;
        if (total % 32) mask.back() = (1u << (total % 32)) - 1u;
        for (uint32_t j = 0; j < missing; ++j) {
            const uint32_t bit = 7u + 8u * j; // XCD 7's CUs
            if (bit < total) mask[bit / 32] &= ~(1u << (bit % 32));
        }
    } else {
        for (uint32_t i = 0; i < cu_count; ++i) mask[i / 32] |= 1u << (i % 32);
    }
    hipStream_t stream = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&stream, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) return wg_set_error(WG_ERR_HIP, "hipExtStreamCreateWithCUMask failed: %s", hipGetErrorString(e));
    int rc = ctx_create_common(device, stream, false, out);
    if (rc) { (void)hipStreamDestroy(stream); return rc; }
    (*out)->owns_stream = true;
    (*out)->uneven_xcds = uneven;
    (*out)->compute_units = (int)cu_count;
    return WG_OK;
}
int wg_ctx_create_with_cu_count(int device, uint32_t cu_count, wg_ctx **out) { return ctx_create_masked(device, cu_count, false, out); }
int wg_ctx_create_with_cu_count_one_xcd(int device, uint32_t cu_count, wg_ctx **out) { return ctx_create_masked(device, cu_count, true, out); }

int wg_ctx_destroy(wg_ctx *ctx) {
    if (!ctx) return WG_OK;
    if (!ctx->recording && wg_defer_if_capturing([ctx] { (void)wg_ctx_destroy(ctx); }, true)) return WG_OK; // (another context of this thread is recording)
    if (ctx->recording) { // destroyed in the middle of its own recording: end the capture first
        if (ctx->recording_thread != std::this_thread::get_id()) // (thread-local capture: only the recording thread can end it, and its counters are its own)
            return wg_set_error(WG_ERR_INVALID_ARG, "wg_ctx_destroy: the context is recording on another thread; finish or destroy it there");
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(ctx->stream, &g);
        if (g) (void)hipGraphDestroy(g);
        ctx->recording = false;
        // the recording is abandoned: what was queued during it runs now -- unless another recording is still open on this thread, which then shares
        // the batch (t_batch holds it) and passes it on to its own command buffer
        std::shared_ptr<wg_deferred_batch> abandoned = capture_ended(ctx);
        abandoned.reset();
    }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->workspace) (void)hipFree(ctx->workspace);
    if (ctx->tr_workspace) (void)hipFree(ctx->tr_workspace);
    if (ctx->pad_workspace) (void)hipFree(ctx->pad_workspace);
    if (ctx->stage_workspace) (void)hipFree(ctx->stage_workspace);
    for (void *p : ctx->retired_scratch) (void)hipFree(p);
    if (ctx->flags) (void)hipFree(ctx->flags);
    if (ctx->tile_queues) (void)hipFree(ctx->tile_queues);
    if (ctx->bal.side) { (void)hipStreamSynchronize(ctx->bal.side); (void)hipStreamDestroy(ctx->bal.side); }
    if (ctx->bal.ev) (void)hipEventDestroy(ctx->bal.ev);
    if (ctx->bal.dev) (void)hipFree(ctx->bal.dev);
    if (ctx->bal.host) (void)hipHostFree(ctx->bal.host);
    if (ctx->bal.scratch) (void)hipFree(ctx->bal.scratch);
    if (ctx->debug_stamps) (void)hipHostFree(ctx->debug_stamps);
    if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return WG_OK;
}

int wg_ctx_sync(wg_ctx *ctx) {
    if (!ctx) return wg_set_error(WG_ERR_INVALID_ARG, "wg_ctx_sync: ctx is NULL");
    if (ctx->recording) return wg_set_error(WG_ERR_INVALID_ARG, "wg_ctx_sync: cannot synchronise while recording");
    WG_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return wg_ctx_check_async(ctx);
}
int wg_ctx_device(const wg_ctx *ctx) { return ctx ? ctx->device : -1; }
void *wg_ctx_stream(const wg_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int wg_ctx_device_info(const wg_ctx *ctx, char *name256, int *compute_units, int *clock_mhz, uint64_t *hbm_bytes) {
    if (!ctx) return wg_set_error(WG_ERR_INVALID_ARG, "wg_ctx_device_info: ctx is NULL");
    hipDeviceProp_t prop;
    WG_HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    if (name256) snprintf(name256, 256, "%s (%s)", prop.name[0] ? prop.name : "AMD Instinct (name not reported)", prop.gcnArchName);
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (clock_mhz) *clock_mhz = prop.clockRate / 1000;
    if (hbm_bytes) *hbm_bytes = (uint64_t)prop.totalGlobalMem;
    return WG_OK;
}

int wg_ctx_mem_info(const wg_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes) {
    if (!ctx) return wg_set_error(WG_ERR_INVALID_ARG, "wg_ctx_mem_info: ctx is NULL");
    WG_HIP_TRY(hipSetDevice(ctx->device));
    size_t f = 0, t = 0;
    WG_HIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (uint64_t)f;
    if (total_bytes) *total_bytes = (uint64_t)t;
    return WG_OK;
}

// Diagnostics (tools/overlap_probe.py): `blocks` workgroups of 256 threads spin for `usec` microseconds each and record the
// s_memrealtime tick (100 MHz) at which they started -- a stand-in for a communication library's copy kernel, to observe how the
// hardware interleaves a second queue's workgroups with a resident GEMM grid whose workgroups each need a whole CU.
__global__ void wg_debug_spin_kernel(uint64_t *start_ticks, uint32_t usec) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && start_ticks) start_ticks[blockIdx.x] = t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)usec * 100u) __builtin_amdgcn_s_sleep(16);
}
__global__ void wg_debug_now_kernel(uint64_t *tick) { *tick = __builtin_amdgcn_s_memrealtime(); }

int wg_debug_spin(wg_ctx *ctx, uint32_t blocks, uint32_t usec, wg_buf *start_ticks) {
    if (!ctx || blocks == 0) return wg_set_error(WG_ERR_INVALID_ARG, "wg_debug_spin: bad arguments");
    if (start_ticks && start_ticks->bytes < (size_t)(blocks + 1) * 8) return wg_set_error(WG_ERR_OUT_OF_BOUNDS, "wg_debug_spin: start_ticks too small");
    WG_HIP_TRY(hipSetDevice(ctx->device));
    uint64_t *p = start_ticks ? (uint64_t *)start_ticks->ptr : nullptr;
    if (p) hipLaunchKernelGGL(wg_debug_now_kernel, dim3(1), dim3(1), 0, ctx->stream, p + blocks); // tick at enqueue position
    hipLaunchKernelGGL(wg_debug_spin_kernel, dim3(blocks), dim3(256), 0, ctx->stream, p, usec);
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}

int wg_ctx_set_tuning(wg_ctx *ctx, wg_tuning key, int value) {
    if (!ctx || (int)key < 0 || (int)key >= (int)WG_TUNE_COUNT_) return wg_set_error(WG_ERR_INVALID_ARG, "wg_ctx_set_tuning: bad context or key %d", (int)key);
    if (key == WG_TUNE_F16_TILE && value != 0 && value != 128 && value != 256 && value != 256128)
        return wg_set_error(WG_ERR_INVALID_ARG, "wg_ctx_set_tuning: WG_TUNE_F16_TILE takes 0, 128, 256 or 256128, not %d", value);
    if (key == WG_TUNE_F32_MID && (value == 64064 || value == 64128 || value == 128064 || value == 128128 || value == 64032 || value == 32064 || value == 96096 || value == 96064 || value == 64096)) {
        ctx->tuning[key] = value;
        return WG_OK;
    }
    if (key == WG_TUNE_GEMVT_LDS) {
        if (value < 0 || value > 65536) return wg_set_error(WG_ERR_INVALID_ARG, "wg_ctx_set_tuning: WG_TUNE_GEMVT_LDS takes 0 or an output count per CU, not %d", value);
        ctx->tuning[key] = value;
        return WG_OK;
    }
    if (key == WG_TUNE_F32_MID_SPLIT) {
        if (value < 0 || value == 1 || value > 64) return wg_set_error(WG_ERR_INVALID_ARG, "wg_ctx_set_tuning: WG_TUNE_F32_MID_SPLIT takes 0 or 2..64, not %d", value);
        ctx->tuning[key] = value;
        return WG_OK;
    }
    if (key != WG_TUNE_F16_TILE && (value < -1 || value > 1))
        return wg_set_error(WG_ERR_INVALID_ARG, "wg_ctx_set_tuning: knob %d takes -1, 0 or 1, not %d", (int)key, value);
    ctx->tuning[key] = value;
    return WG_OK;
}
int wg_ctx_get_tuning(const wg_ctx *ctx, wg_tuning key, int *value) {
    if (!ctx || !value || (int)key < 0 || (int)key >= (int)WG_TUNE_COUNT_) return wg_set_error(WG_ERR_INVALID_ARG, "wg_ctx_get_tuning: bad argument");
    *value = ctx->tuning[key];
    return WG_OK;
}

int wg_ctx_reserve_workspace(wg_ctx *ctx, size_t bytes) {
    if (!ctx) return wg_set_error(WG_ERR_INVALID_ARG, "wg_ctx_reserve_workspace: ctx is NULL");
    void *p;
    return wg_ctx_workspace(ctx, bytes, &p);
}

// ---------------------------------------------------------------------------------------------------
// buffers
// ---------------------------------------------------------------------------------------------------
int wg_buf_create(wg_ctx *ctx, size_t bytes, uint32_t usage, wg_buf **out) {
    if (!ctx || !out) return wg_set_error(WG_ERR_INVALID_ARG, "wg_buf_create: NULL argument");
    *out = nullptr;
    if (ctx->recording) return wg_set_error(WG_ERR_INVALID_ARG, "wg_buf_create: cannot allocate while recording");
    WG_HIP_TRY(hipSetDevice(ctx->device));
    wg_buf *b = new (std::nothrow) wg_buf();
    if (!b) return wg_set_error(WG_ERR_HIP, "out of host memory");
    b->ctx = ctx;
    b->bytes = bytes;
    b->usage = usage;
    b->host_pinned = (usage & (WG_USAGE_MAP_READ | WG_USAGE_MAP_WRITE)) != 0;
    if (bytes > 0) {
        hipError_t e = b->host_pinned ? hipHostMalloc(&b->ptr, bytes, hipHostMallocDefault) : hipMalloc(&b->ptr, bytes);
        if (e != hipSuccess) {
            delete b;
            return wg_set_error(WG_ERR_HIP, "allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        }
    }
    *out = b;
    return WG_OK;
}

int wg_buf_create_init(wg_ctx *ctx, const void *data, size_t bytes, uint32_t usage, wg_buf **out) {
    if (bytes > 0 && !data) return wg_set_error(WG_ERR_INVALID_ARG, "wg_buf_create_init: data is NULL");
    int rc = wg_buf_create(ctx, bytes, usage, out);
    if (rc != WG_OK) return rc;
    if (bytes > 0) {
        // create_buffer_init is synchronous in the reference (mapped at creation): keep that contract.
        hipError_t e = hipMemcpy((*out)->ptr, data, bytes, (*out)->host_pinned ? hipMemcpyHostToHost : hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            wg_buf_destroy(*out);
            *out = nullptr;
            return wg_set_error(WG_ERR_HIP, "upload of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        }
    }
    return WG_OK;
}

int wg_buf_wrap(wg_ctx *ctx, void *device_ptr, size_t bytes, wg_buf **out) {
    if (!ctx || !out) return wg_set_error(WG_ERR_INVALID_ARG, "wg_buf_wrap: NULL argument");
    if (bytes > 0 && !device_ptr) return wg_set_error(WG_ERR_INVALID_ARG, "wg_buf_wrap: device_ptr is NULL");
    wg_buf *b = new (std::nothrow) wg_buf();
    if (!b) return wg_set_error(WG_ERR_HIP, "out of host memory");
    b->ctx = ctx;
    b->ptr = device_ptr;
    b->bytes = bytes;
    b->usage = WG_USAGE_STORAGE | WG_USAGE_COPY_SRC | WG_USAGE_COPY_DST;
    b->owned = false;
    *out = b;
    return WG_OK;
}

int wg_buf_destroy(wg_buf *buf) {
    if (!buf || buf->borrowed) return WG_OK;
    if (wg_defer_if_capturing([buf] { (void)wg_buf_destroy(buf); })) return WG_OK; // (this thread is recording: after the recording)
    int rc = WG_OK;
    if (buf->owned && buf->ptr) {
        (void)hipSetDevice(buf->ctx->device);
        // Pending stream work may still reference the memory (wgpu keeps a dropped buffer alive until the
        // submission that uses it retires): retire it first.
        (void)hipStreamSynchronize(buf->ctx->stream);
        hipError_t e = buf->host_pinned ? hipHostFree(buf->ptr) : hipFree(buf->ptr);
        if (e != hipSuccess) rc = wg_set_error(WG_ERR_HIP, "free failed: %s", hipGetErrorString(e));
    }
    if (buf->ipc_base) { // a peer's allocation mapped by wg_buf_ipc_open
        (void)hipSetDevice(buf->ctx->device);
        (void)hipStreamSynchronize(buf->ctx->stream);
        (void)hipIpcCloseMemHandle(buf->ipc_base);
    }
    delete buf;
    return rc;
}

size_t wg_buf_size(const wg_buf *buf) { return buf ? buf->bytes : 0; }
void *wg_buf_device_ptr(const wg_buf *buf) { return buf ? buf->ptr : nullptr; }

static int check_range(const char *what, const wg_buf *b, size_t offset, size_t bytes) {
    if (offset > b->bytes || bytes > b->bytes - offset)
        return wg_set_error(WG_ERR_OUT_OF_BOUNDS, "%s: range [%zu, %zu) exceeds the buffer's %zu bytes", what, offset,
                            offset + bytes, b->bytes);
    return WG_OK;
}

int wg_buf_write(wg_ctx *ctx, wg_buf *dst, size_t offset, const void *data, size_t bytes) {
    if (!ctx || !dst || (bytes && !data)) return wg_set_error(WG_ERR_INVALID_ARG, "wg_buf_write: NULL argument");
    if (int rc = check_range("wg_buf_write", dst, offset, bytes)) return rc;
    if (bytes == 0) return WG_OK;
    if (ctx->recording) return wg_set_error(WG_ERR_INVALID_ARG, "wg_buf_write: host uploads cannot be recorded");
    WG_HIP_TRY(hipSetDevice(ctx->device));
    WG_HIP_TRY(hipMemcpyAsync((char *)dst->ptr + offset, data, bytes, hipMemcpyDefault, ctx->stream));
    // Queue::write_buffer copies out of `data` before returning; pageable memcpyAsync already stages, but a
    // pinned source would be read later -- make the contract unconditional.
    WG_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return WG_OK;
}

int wg_buf_read(wg_ctx *ctx, const wg_buf *src, size_t offset, void *dst, size_t bytes) {
    if (!ctx || !src || (bytes && !dst)) return wg_set_error(WG_ERR_INVALID_ARG, "wg_buf_read: NULL argument");
    if (int rc = check_range("wg_buf_read", src, offset, bytes)) return rc;
    if (ctx->recording) return wg_set_error(WG_ERR_INVALID_ARG, "wg_buf_read: cannot read back while recording");
    WG_HIP_TRY(hipSetDevice(ctx->device));
    if (bytes) WG_HIP_TRY(hipMemcpyAsync(dst, (const char *)src->ptr + offset, bytes, hipMemcpyDefault, ctx->stream));
    WG_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return wg_ctx_check_async(ctx); // e.g. a sharded Gemm whose peer never delivered: the bytes just read are poison, say so
}

int wg_buf_copy(wg_ctx *ctx, const wg_buf *src, size_t src_offset, wg_buf *dst, size_t dst_offset, size_t bytes) {
    if (!ctx || !src || !dst) return wg_set_error(WG_ERR_INVALID_ARG, "wg_buf_copy: NULL argument");
    if (int rc = check_range("wg_buf_copy(src)", src, src_offset, bytes)) return rc;
    if (int rc = check_range("wg_buf_copy(dst)", dst, dst_offset, bytes)) return rc;
    if (bytes == 0) return WG_OK;
    WG_HIP_TRY(hipSetDevice(ctx->device));
    WG_HIP_TRY(hipMemcpyAsync((char *)dst->ptr + dst_offset, (const char *)src->ptr + src_offset, bytes, hipMemcpyDefault,
                              ctx->stream));
    return WG_OK;
}

int wg_buf_fill_zero(wg_ctx *ctx, wg_buf *buf) {
    if (!ctx || !buf) return wg_set_error(WG_ERR_INVALID_ARG, "wg_buf_fill_zero: NULL argument");
    if (buf->bytes == 0) return WG_OK;
    WG_HIP_TRY(hipSetDevice(ctx->device));
    WG_HIP_TRY(hipMemsetAsync(buf->ptr, 0, buf->bytes, ctx->stream));
    return WG_OK;
}

// ---------------------------------------------------------------------------------------------------
// record / replay
// ---------------------------------------------------------------------------------------------------
int wg_encoder_begin(wg_ctx *ctx) {
    if (!ctx) return wg_set_error(WG_ERR_INVALID_ARG, "wg_encoder_begin: ctx is NULL");
    if (ctx->recording) return wg_set_error(WG_ERR_INVALID_ARG, "wg_encoder_begin: already recording");
    WG_HIP_TRY(hipSetDevice(ctx->device));
    WG_HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    ctx->recording = true;
    capture_began(ctx);
    return WG_OK;
}

int wg_encoder_finish(wg_ctx *ctx, wg_cmdbuf **out) {
    if (!ctx || !out) return wg_set_error(WG_ERR_INVALID_ARG, "wg_encoder_finish: NULL argument");
    *out = nullptr;
    if (!ctx->recording) return wg_set_error(WG_ERR_INVALID_ARG, "wg_encoder_finish: not recording");
    if (ctx->recording_thread != std::this_thread::get_id())
        return wg_set_error(WG_ERR_INVALID_ARG, "wg_encoder_finish: called from another thread than wg_encoder_begin (thread-local capture)");
    ctx->recording = false;
    hipGraph_t graph = nullptr;
    const hipError_t ec = hipStreamEndCapture(ctx->stream, &graph);
    // Destroy calls that arrived while this thread was recording: the command buffer holds them until it is destroyed itself (a dropped buffer the
    // recording used must outlive every replay); on any failure below `deferred` goes out of scope and they run at once.
    std::shared_ptr<wg_deferred_batch> deferred = capture_ended(ctx);
    WG_HIP_TRY(ec);
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(graph);
        return wg_set_error(WG_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    }
    wg_cmdbuf *cb = new (std::nothrow) wg_cmdbuf();
    if (!cb) return wg_set_error(WG_ERR_HIP, "out of host memory");
    cb->ctx = ctx;
    cb->graph = graph;
    cb->exec = exec;
    cb->deferred = std::move(deferred);
    ctx->live_cmdbufs++;
    *out = cb;
    return WG_OK;
}

int wg_queue_submit(wg_ctx *ctx, wg_cmdbuf *cmdbuf) {
    if (!ctx || !cmdbuf) return wg_set_error(WG_ERR_INVALID_ARG, "wg_queue_submit: NULL argument");
    if (cmdbuf->ctx != ctx) return wg_set_error(WG_ERR_INVALID_ARG, "wg_queue_submit: command buffer belongs to another context");
    if (ctx->recording) return wg_set_error(WG_ERR_INVALID_ARG, "wg_queue_submit: cannot submit while recording");
    WG_HIP_TRY(hipSetDevice(ctx->device));
    WG_HIP_TRY(hipGraphLaunch(cmdbuf->exec, ctx->stream));
    return WG_OK;
}

int wg_cmdbuf_destroy(wg_cmdbuf *cmdbuf) {
    if (!cmdbuf) return WG_OK;
    if (t_capturing != 0 && t_batch) {
        // Queued into the batch of the recording(s) open on this thread. Should this command buffer itself hold that very batch (begin A, begin B, finish A -> X,
        // destroy X while B still records), the batch would hold the closure that destroys X and X the batch: a cycle nothing ever runs. So X's own reference moves
        // into the closure's capture list and is released there, after the destroy it guards.
        std::shared_ptr<wg_deferred_batch> own = std::move(cmdbuf->deferred);
        cmdbuf->deferred.reset();
        wg_deferred_batch *batch = t_batch.get();
        if (own.get() == batch) own.reset(); // (the batch it is queued into: its lifetime is that of the queue itself)
        batch->items.push_back([cmdbuf, own]() mutable { (void)wg_cmdbuf_destroy(cmdbuf); own.reset(); });
        return WG_OK;
    }
    (void)hipSetDevice(cmdbuf->ctx->device);
    (void)hipStreamSynchronize(cmdbuf->ctx->stream);
    if (cmdbuf->exec) (void)hipGraphExecDestroy(cmdbuf->exec);
    if (cmdbuf->graph) (void)hipGraphDestroy(cmdbuf->graph);
    wg_ctx *ctx = cmdbuf->ctx;
    if (ctx->live_cmdbufs > 0 && --ctx->live_cmdbufs == 0) { // nothing can replay the retired scratch regions any more
        for (void *p : ctx->retired_scratch) (void)hipFree(p);
        ctx->retired_scratch.clear();
    }
    std::shared_ptr<wg_deferred_batch> deferred = std::move(cmdbuf->deferred);
    delete cmdbuf;
    deferred.reset(); // the last command buffer of the recording(s) it was queued under: the buffers dropped meanwhile are freed now
    return WG_OK;
}

// ---------------------------------------------------------------------------------------------------
// timestamps
// ---------------------------------------------------------------------------------------------------
int wg_timestamps_create(wg_ctx *ctx, uint32_t capacity, wg_timestamps **out) {
    if (!ctx || !out) return wg_set_error(WG_ERR_INVALID_ARG, "wg_timestamps_create: NULL argument");
    *out = nullptr;
    WG_HIP_TRY(hipSetDevice(ctx->device));
    wg_timestamps *ts = new (std::nothrow) wg_timestamps();
    if (!ts) return wg_set_error(WG_ERR_HIP, "out of host memory");
    ts->ctx = ctx;
    ts->events.resize(capacity, nullptr);
    ts->written.assign(capacity, 0);
    for (uint32_t i = 0; i < capacity; ++i) {
        hipError_t e = hipEventCreate(&ts->events[i]);
        if (e != hipSuccess) {
            wg_timestamps_destroy(ts);
            return wg_set_error(WG_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(e));
        }
    }
    *out = ts;
    return WG_OK;
}

int wg_timestamps_destroy(wg_timestamps *ts) {
    if (!ts) return WG_OK;
    if (wg_defer_if_capturing([ts] { (void)wg_timestamps_destroy(ts); })) return WG_OK;
    (void)hipSetDevice(ts->ctx->device);
    for (hipEvent_t e : ts->events)
        if (e) (void)hipEventDestroy(e);
    delete ts;
    return WG_OK;
}

int wg_timestamps_clear(wg_timestamps *ts) {
    if (!ts) return wg_set_error(WG_ERR_INVALID_ARG, "wg_timestamps_clear: NULL argument");
    ts->len = 0;
    std::fill(ts->written.begin(), ts->written.end(), (uint8_t)0);
    return WG_OK;
}

// next_query_indices::<COUNT> (timestamps.rs:80-94): `count` consecutive slots, or none at all (*first = UINT32_MAX: the reference's None -- not an error)
int wg_timestamps_reserve(wg_timestamps *ts, uint32_t count, uint32_t *first) {
    if (!ts || !first) return wg_set_error(WG_ERR_INVALID_ARG, "wg_timestamps_reserve: NULL argument");
    if (count == 0) { *first = 0; return WG_OK; }
    if ((uint64_t)ts->len + count > ts->events.size()) { *first = UINT32_MAX; return WG_OK; }
    *first = ts->len;
    ts->len += count;
    return WG_OK;
}

// write_timestamp_at (timestamps.rs:108-115): the timestamp of slot `index` (any slot below the capacity) at the current point of the stream
int wg_timestamps_write_at(wg_ctx *ctx, wg_timestamps *ts, uint32_t index) {
    if (!ctx || !ts) return wg_set_error(WG_ERR_INVALID_ARG, "wg_timestamps_write_at: NULL argument");
    if (index >= ts->events.size()) return wg_set_error(WG_ERR_OUT_OF_BOUNDS, "wg_timestamps_write_at: slot %u of %zu", index, ts->events.size());
    if (ctx->recording) return wg_set_error(WG_ERR_INVALID_ARG, "wg_timestamps_write_at: not recordable");
    WG_HIP_TRY(hipSetDevice(ctx->device));
    WG_HIP_TRY(hipEventRecord(ts->events[index], ctx->stream));
    ts->written[index] = 1;
    return WG_OK;
}

int wg_timestamps_write(wg_ctx *ctx, wg_timestamps *ts, uint32_t *index) {
    if (!ctx || !ts) return wg_set_error(WG_ERR_INVALID_ARG, "wg_timestamps_write: NULL argument");
    if (ts->len >= ts->events.size())
        return wg_set_error(WG_ERR_OUT_OF_BOUNDS, "wg_timestamps_write: capacity %zu exhausted", ts->events.size());
    if (int rc = wg_timestamps_write_at(ctx, ts, ts->len)) return rc;
    if (index) *index = ts->len;
    ts->len++;
    return WG_OK;
}

uint32_t wg_timestamps_len(const wg_timestamps *ts) { return ts ? ts->len : 0; }

int wg_timestamps_wait_for_results_ms(wg_timestamps *ts, double *out_ms, uint32_t capacity) {
    if (!ts || (capacity && !out_ms)) return wg_set_error(WG_ERR_INVALID_ARG, "wg_timestamps_wait_for_results_ms: NULL argument");
    if (ts->len == 0) return WG_OK;
    WG_HIP_TRY(hipSetDevice(ts->ctx->device));
    const uint32_t n = ts->len < capacity ? ts->len : capacity;
    int first = -1; // times are relative to the first slot that was written (slot 0 in the usual begin / end use); a reserved slot never written reads 0
    for (uint32_t i = 0; i < ts->len; ++i)
        if (ts->written[i]) { WG_HIP_TRY(hipEventSynchronize(ts->events[i])); if (first < 0) first = (int)i; }
    for (uint32_t i = 0; i < n; ++i) {
        float ms = 0.f;
        if (ts->written[i] && (int)i != first) WG_HIP_TRY(hipEventElapsedTime(&ms, ts->events[first], ts->events[i]));
        out_ms[i] = (double)ms;
    }
    return WG_OK;
}

} // extern "C"
