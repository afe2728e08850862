// Multi-GPU half of the boundary: the north star's M-sharded Gemm. The reference has ONE wgpu::Device and one Queue
// (crates/wgcore/src/gpu.rs:7-12) and no multi-device path; what is kept from it is the tensor model -- every rank's operands and the
// gathered result are ordinary column-major GpuMatrix tensors (tensor.rs:47-48) addressed by ViewShape (shapes.rs:9-21), and the local
// product of every rank is exactly Gemm::dispatch_generic (gemm.rs:65-127) on views of them.
//
//   rank g of P owns A_g = A[g*M/P .. (g+1)*M/P, :] (its own dense tensor), B is replicated, every rank ends with the full M x N C.
//   N is cut into column panels; panel i's exchange runs while panel i+1 computes. Two exchange engines (a third, SDMA rect
//   copies straight into the peers' C through hsa_amd_memory_async_copy_rect, was removed in ABI 3: one rect-capable queue per direction, and a hang with two
//   processes on one GPU at 32768^3 that nothing in the HSA API let this library bound):
//
//   WG_GATHER_RCCL       the north star's primary. A row block of a column-major C is strided, a collective wants contiguous
//                        ranges: the panel Gemm writes slot g of a staging cube [M/P, np, P] (GpuCube, stride_mat = M/P*np), one
//                        in-place ncclAllGather per panel fills the other slots on the communicator's stream, and an HBM-bound
//                        relayout kernel (cube_to_matrix) scatters the cube into columns [c0, c0+np) of C. RCCL's copy kernels need
//                        compute units; every f16 Gemm workgroup needs a whole one (so the caller may give the Gemm a CU-masked
//                        context: wg_ctx_create_with_cu_count).
//   WG_GATHER_PEER_STAGED  no compute units for the transfer: contiguous peer copies go to an SDMA engine
//                        PER LINK (16 engines, 60.7 GB/s each, 7 at once 330 GB/s, Gemm beside 3 / 7 busy engines +2.3 % / +4.8 %:
//                        tools/cpp/sdma_probe2.cpp). So: the panel Gemm writes slot g of a staging cube (as in RCCL mode), one contiguous
//                        hipMemcpyAsync per peer (its own stream: the runtime's peer-to-peer path = that link's SDMA engine, no compute
//                        units) pushes the slot into the SAME slot of the peer's staging cube, followed by a 4-byte sequence-number copy
//                        into the peer's flag array; the receiver's context stream runs a one-wave wait kernel on those flags and then the
//                        relayout of that panel -- everything stream-ordered, no helper thread, no barrier: the staging cube is double-
//                        buffered by step parity, and a rank cannot be more than one step ahead of a peer whose data it needs.
//
// RCCL is resolved at run time (dlopen): the library links only libamdhip64, and a process that already
// loaded torch's bundled copies binds to those.
#include "wg_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h> // types and enums only

#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

namespace {

// ---------------------------------------------------------------------------------------------------------------
// run-time bindings
// ---------------------------------------------------------------------------------------------------------------
void *open_lib(const char *env, const char *const *names) {
    if (const char *p = getenv(env)) {
        if (void *h = dlopen(p, RTLD_NOW | RTLD_GLOBAL)) return h;
    }
    for (int pass = 0; pass < 2; ++pass) // first whatever this process already mapped (e.g. torch's bundled copy), then the system's
        for (const char *const *n = names; *n; ++n)
            if (void *h = dlopen(*n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0))) return h;
    return nullptr;
}

struct RcclApi {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr; // optional
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
RcclApi &rccl() {
    static RcclApi api = [] {
        RcclApi a;
        static const char *const names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", nullptr };
        a.h = open_lib("WG_RCCL_LIB", names);
        if (!a.h) return a;
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.h, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.h, "ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.h, "ncclCommDestroy");
        a.CommCount = (decltype(a.CommCount))dlsym(a.h, "ncclCommCount");
        a.AllGather = (decltype(a.AllGather))dlsym(a.h, "ncclAllGather");
        a.AllReduce = (decltype(a.AllReduce))dlsym(a.h, "ncclAllReduce");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.h, "ncclGetErrorString");
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather && a.AllReduce && a.GetErrorString;
        return a;
    }();
    return api;
}

} // namespace

struct wg_comm {
    wg_ctx *ctx = nullptr;
    int nranks = 1, rank = 0;
    ncclComm_t nccl = nullptr;
    hipStream_t stream = nullptr; // the collective library's stream
    hipEvent_t ev_ctx = nullptr, ev_comm = nullptr;
    std::vector<hipEvent_t> ev_panel; // per-panel events (grown on demand)
    void *stage = nullptr;            // RCCL mode: two staging cubes
    size_t stage_bytes = 0;
    float *token = nullptr; // 1 float for the barrier's all-reduce
    uint64_t bytes_sent = 0; // payload this rank pushed or contributed since creation (diagnostics: wg_comm_bytes_sent)
    // staged peer copies (WG_GATHER_PEER_STAGED)
    void *pstage = nullptr;          // two staging cubes of the whole step (step parity), caller-sized: wg_comm_stage_reserve
    size_t pstage_bytes = 0;
    uint32_t *pflags = nullptr;      // flags[sender][panel] = sequence number of the last step whose slot has landed (uncached memory)
    wg_buf stage_buf, flags_buf;     // non-owning views handed to the caller for export
    std::vector<wg_buf *> peer_stage, peer_flags; // the peers' staging cubes / flag arrays as addressable from here
    std::vector<hipStream_t> peer_stream;         // one copy stream per peer
    std::vector<hipEvent_t> sent_ev;              // [parity][panel][peer]: my slot of that panel has left for that peer
    uint32_t *seq_src = nullptr;     // 64 device words: word (seq % 64) holds step seq's sequence number (source of that step's flag copies; set_word_kernel)
    uint32_t *wait_err = nullptr;    // pinned: set by the wait kernel when a peer's slot did not arrive within the timeout (registered with the context)
    uint64_t timeout_ticks = 3000000000ull; // wait kernel's patience in 100 MHz ticks (WG_COMM_TIMEOUT_MS at creation, default 30 s)
    uint32_t step = 0;
    uint64_t staged_geom[4] = { 0, 0, 0, 0 }; // M, N, a hash of the panel widths, element size of the previous staged call: the slot layout its events refer to
    // one launch per step (f16): the rank's whole product is ONE kernel over all N-panels whose waves count themselves into panel_sync[p] as
    // their stores reach memory; the exchange of panel p waits for the full count (hipStreamWaitValue32) while the kernel works on
    bool can_wait_value = false;              // hipDeviceAttributeCanUseStreamWaitValue: without it every product launches panel by panel
    int one_launch = -1;                      // -1: by engine (RCCL: on -- what lets its Gemm run on 248 CUs; staged: off -- measured 1-2 % slower there), 0 / 1
    uint32_t *panel_sync = nullptr;           // [0, kMaxPanels): arrival counters of the panels: waves finished, RUNNING totals (a reset could overtake a
                                              // copy stream that has not evaluated its wait yet -- and then waits for the NEXT step's kernel, which waits for
                                              // the peer, which waits for this copy: found as a dead-lock with two processes on one GPU)
    std::vector<uint32_t> panel_total;        // what panel_sync[p] reads once every launch enqueued so far has finished panel p
    // pipelined steps (wg_comm_set_pipelined): the wait + relayout of a call's LAST panel is deferred until the next call has enqueued its
    // first Gemm (or wg_comm_join / flush / barrier), so that the one exchange nothing of its own step can hide runs under the next step
    bool pipelined = false;
    // diagnostics (wg_comm_set_wait_timing): a pair of timing events on the context's stream around every "wait for panel p's exchange" -- how long
    // the compute stream actually stood still for each panel (0 when the exchange hid under the Gemms; a slow link shows up as the first panels' waits)
    bool time_waits = false;
    struct WaitStamp { hipEvent_t before = nullptr, after = nullptr; uint32_t panel = 0; };
    std::vector<WaitStamp> wait_stamps;
    size_t wait_used = 0;
    struct Pending {
        bool on = false;
        hipEvent_t after = nullptr; // RCCL engine: the panel's gather (an event of the communicator's stream) instead of the peers' flags
        uint32_t seq = 0, panel = 0, mg = 0, np = 0;
        const char *src = nullptr;
        char *dst = nullptr;
        uint64_t ld = 0;
        size_t es = 0;
    } pending;
};

namespace {

int nccl_fail(const char *what, ncclResult_t r) {
    return wg_set_error(WG_ERR_HIP, "%s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(r) : "?");
}

// cube [mg, np, P] (contiguous slots) -> columns of a column-major matrix: c[(j)*ldc + g*mg + i] = stage[(g*np + j)*mg + i], in units of V
// `err` (may be null): the wait kernel's device-side time-out word. Set => a slot of this cube never arrived: write NaN bit patterns (all ones: a NaN in
// f16 and in f32) instead of whatever the slots hold, so that a result read before the error is reported cannot pass for data.
template <typename V>
__global__ __launch_bounds__(256) void cube_to_matrix_kernel(const V *__restrict__ stage, V *__restrict__ c, uint32_t mg_v, uint32_t np, uint64_t ldc_v,
                                                             const uint32_t *err) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x, g = blockIdx.z;
    if (i >= mg_v) return;
    const bool poison = err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; // (a device word: see wait_flags_kernel)
    for (uint32_t j = blockIdx.y; j < np; j += gridDim.y) {
        V v;
        if (poison) __builtin_memset(&v, 0xff, sizeof v);
        else v = __builtin_nontemporal_load(stage + ((uint64_t)g * np + j) * mg_v + i);
        __builtin_nontemporal_store(v, c + (uint64_t)j * ldc_v + (uint64_t)g * mg_v + i);
    }
}

typedef uint32_t wg_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t wg_u2 __attribute__((ext_vector_type(2)));

int launch_cube_to_matrix(wg_ctx *ctx, const void *stage, void *c_col0, uint32_t mg, uint32_t np, uint32_t nranks, uint64_t ldc, size_t es, const uint32_t *err = nullptr) {
    if (mg == 0 || np == 0) return WG_OK;
    const size_t row_bytes = (size_t)mg * es, ld_bytes = (size_t)ldc * es;
    const uintptr_t align = (uintptr_t)stage | (uintptr_t)c_col0 | row_bytes | ld_bytes;
    const uint32_t gy = np < 65535u ? np : 65535u;
    if (align % 16 == 0) {
        const uint32_t mv = (uint32_t)(row_bytes / 16);
        hipLaunchKernelGGL(cube_to_matrix_kernel<wg_u4>, dim3((mv + 255u) / 256u, gy, nranks), dim3(256), 0, ctx->stream, (const wg_u4 *)stage, (wg_u4 *)c_col0, mv,
                           np, (uint64_t)(ld_bytes / 16), err);
    } else if (align % 8 == 0) {
        const uint32_t mv = (uint32_t)(row_bytes / 8);
        hipLaunchKernelGGL(cube_to_matrix_kernel<wg_u2>, dim3((mv + 255u) / 256u, gy, nranks), dim3(256), 0, ctx->stream, (const wg_u2 *)stage, (wg_u2 *)c_col0, mv,
                           np, (uint64_t)(ld_bytes / 8), err);
    } else if (align % 4 == 0) {
        const uint32_t mv = (uint32_t)(row_bytes / 4);
        hipLaunchKernelGGL(cube_to_matrix_kernel<uint32_t>, dim3((mv + 255u) / 256u, gy, nranks), dim3(256), 0, ctx->stream, (const uint32_t *)stage,
                           (uint32_t *)c_col0, mv, np, (uint64_t)(ld_bytes / 4), err);
    } else {
        return wg_set_error(WG_ERR_PRECONDITION, "cube_to_matrix: row blocks are not 4-byte aligned");
    }
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}

constexpr uint32_t kMaxPanels = 1024, kMaxRanks = 16;
constexpr size_t kFlagBytes = (size_t)kMaxRanks * kMaxPanels * sizeof(uint32_t);

// One wave: lane r waits until rank r's slot of `panel` carries sequence number >= seq (wrap-safe). The flags live in uncached device
// memory written by the peers' copy engines; system-scope loads. Gives up after `timeout_ticks` (100 MHz; wg_comm's timeout_ms, default
// 30 s, WG_COMM_TIMEOUT_MS) and raises *err instead of hanging the queue. *err is sticky until the host has reported it: the relayout
// behind a failed wait poisons its output (NaN bit patterns), every later wait fails fast, and wg_ctx_sync / wg_buf_read on the
// context, wg_comm_flush / _join / _barrier and the next sharded call all return the error.
// (`err` is pinned HOST memory, for the host to report; `err_dev` is its device-side twin, which the relayout kernels read -- thousands of
// workgroups polling a host word over PCIe made a 0.4 ms relayout take 10 ms.)
__global__ void wait_flags_kernel(const uint32_t *flags, uint32_t nranks, uint32_t self, uint32_t panel, uint32_t seq, uint32_t *err, uint32_t *err_dev,
                                  uint64_t timeout_ticks) {
    const uint32_t r = threadIdx.x;
    if (r >= nranks || r == self) return;
    const uint32_t *f = flags + (size_t)r * kMaxPanels + panel;
    if (__hip_atomic_load(err_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return; // an earlier wait already gave up: fail fast
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while ((int32_t)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
        __builtin_amdgcn_s_sleep(64);
        if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
            __hip_atomic_store(err_dev, 1u + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(err, 1u + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
    }
}

// sequence number of a step -> the device word the flag copies of that step read (runs on the context's stream ahead of the step's Gemms;
// a kernel argument, so nothing the host may overwrite later is read at execution time)
__global__ void set_word_kernel(uint32_t *dst, uint32_t v) { *dst = v; }

int comm_flush(wg_comm *c) { // host-blocking: every peer copy enqueued so far has landed
    for (hipStream_t st : c->peer_stream)
        if (st) WG_HIP_TRY(hipStreamSynchronize(st));
    if (int rc = wg_ctx_check_async(c->ctx)) return rc; // a wait kernel gave up (reported once, then cleared)
    return WG_OK;
}

// wait timing (diagnostics): stamp the context's stream right before / right after the wait for `panel`'s exchange
int wait_begin(wg_comm *c, uint32_t panel) {
    if (!c->time_waits) return WG_OK;
    if (c->wait_used == c->wait_stamps.size()) {
        wg_comm::WaitStamp w;
        WG_HIP_TRY(hipEventCreate(&w.before));
        WG_HIP_TRY(hipEventCreate(&w.after));
        c->wait_stamps.push_back(w);
    }
    c->wait_stamps[c->wait_used].panel = panel;
    WG_HIP_TRY(hipEventRecord(c->wait_stamps[c->wait_used].before, c->ctx->stream));
    return WG_OK;
}
int wait_end(wg_comm *c) {
    if (!c->time_waits) return WG_OK;
    WG_HIP_TRY(hipEventRecord(c->wait_stamps[c->wait_used].after, c->ctx->stream));
    ++c->wait_used;
    return WG_OK;
}

int run_pending(wg_comm *c) { // the deferred last panel of the previous staged call: wait for its slots, relayout it
    if (!c->pending.on) return WG_OK;
    const wg_comm::Pending q = c->pending; // (a copy: cleared below once nothing can fail before the wait is enqueued)
    if (wait_begin(c, q.panel) != WG_OK) { // a diagnostic must not drop the deferred panel: timing off, the wait + relayout still run
        c->time_waits = false;
        c->wait_used = 0;
    }
    c->pending.on = false;
    if (q.after) { // RCCL engine
        WG_HIP_TRY(hipStreamWaitEvent(c->ctx->stream, q.after, 0));
        if (int rc = wait_end(c)) return rc;
        return launch_cube_to_matrix(c->ctx, q.src, q.dst, q.mg, q.np, (uint32_t)c->nranks, q.ld, q.es);
    }
    if (c->nranks > 1) {
        hipLaunchKernelGGL(wait_flags_kernel, dim3(1), dim3(64), 0, c->ctx->stream, c->pflags, (uint32_t)c->nranks, (uint32_t)c->rank, q.panel, q.seq, c->wait_err,
                           c->seq_src + 64, c->timeout_ticks);
        WG_HIP_TRY(hipGetLastError());
    }
    if (int rc = wait_end(c)) return rc;
    return launch_cube_to_matrix(c->ctx, q.src, q.dst, q.mg, q.np, (uint32_t)c->nranks, q.ld, q.es, c->nranks > 1 ? c->seq_src + 64 : nullptr);
}

int ensure_panel_sync(wg_comm *c) {
    if (!c->panel_sync) {
        WG_HIP_TRY(hipMalloc((void **)&c->panel_sync, kMaxPanels * sizeof(uint32_t)));
        WG_HIP_TRY(hipMemset(c->panel_sync, 0, kMaxPanels * sizeof(uint32_t)));
        c->panel_total.assign(kMaxPanels, 0u);
    }
    uint32_t mx = 0;
    for (uint32_t t : c->panel_total) mx = t > mx ? t : mx;
    if (mx > 0x60000000u) { // long before the 32-bit totals wrap: let everything that waits on them finish, start again from zero (hours apart)
        WG_HIP_TRY(hipStreamSynchronize(c->ctx->stream));
        WG_HIP_TRY(hipStreamSynchronize(c->stream));
        for (hipStream_t st : c->peer_stream)
            if (st) WG_HIP_TRY(hipStreamSynchronize(st));
        WG_HIP_TRY(hipMemset(c->panel_sync, 0, kMaxPanels * sizeof(uint32_t)));
        c->panel_total.assign(kMaxPanels, 0u);
    }
    return WG_OK;
}

struct IpcHandle { // WG_IPC_HANDLE_BYTES
    hipIpcMemHandle_t h;
    uint64_t offset, bytes;
};
static_assert(sizeof(IpcHandle) <= WG_IPC_HANDLE_BYTES, "WG_IPC_HANDLE_BYTES too small");

} // namespace

extern "C" {

int wg_comm_unique_id(void *id) {
    if (!id) return wg_set_error(WG_ERR_INVALID_ARG, "wg_comm_unique_id: id is NULL");
    static_assert(sizeof(ncclUniqueId) == WG_COMM_ID_BYTES, "WG_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
    RcclApi &R = rccl();
    if (!R.ok) return wg_set_error(WG_ERR_UNSUPPORTED, "wg_comm_unique_id: librccl.so.1 could not be loaded (%s)", dlerror() ? dlerror() : "symbols missing");
    ncclUniqueId u;
    ncclResult_t r = R.GetUniqueId(&u);
    if (r != ncclSuccess) return nccl_fail("ncclGetUniqueId", r);
    memcpy(id, &u, sizeof u);
    return WG_OK;
}

int wg_comm_create(wg_ctx *ctx, int nranks, int rank, const void *id, wg_comm **out) {
    if (!ctx || !out) return wg_set_error(WG_ERR_INVALID_ARG, "wg_comm_create: NULL argument");
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return wg_set_error(WG_ERR_INVALID_ARG, "wg_comm_create: rank %d not in [0,%d)", rank, nranks);
    if (ctx->recording) return wg_set_error(WG_ERR_INVALID_ARG, "wg_comm_create: cannot create a communicator while recording");
    WG_HIP_TRY(hipSetDevice(ctx->device));
    wg_comm *c = new (std::nothrow) wg_comm();
    if (!c) return wg_set_error(WG_ERR_HIP, "out of host memory");
    c->ctx = ctx;
    c->nranks = nranks;
    c->rank = rank;
    auto fail = [&](int rc) {
        wg_comm_destroy(c);
        return rc;
    };
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_ctx, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_comm, hipEventDisableTiming) != hipSuccess || hipMalloc(&c->token, 256) != hipSuccess)
        return fail(wg_set_error(WG_ERR_HIP, "wg_comm_create: stream / event creation failed"));
    if (id) {
        RcclApi &R = rccl();
        if (!R.ok) return fail(wg_set_error(WG_ERR_UNSUPPORTED, "wg_comm_create: librccl.so.1 could not be loaded"));
        ncclUniqueId u;
        memcpy(&u, id, sizeof u);
        ncclResult_t r = R.CommInitRank(&c->nccl, nranks, u, rank);
        if (r != ncclSuccess) return fail(nccl_fail("ncclCommInitRank", r));
    }
    {
        int can = 0;
        if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, ctx->device) != hipSuccess) { (void)hipGetLastError(); can = 0; }
        c->can_wait_value = can != 0;
    }
    if (const char *t = getenv("WG_COMM_TIMEOUT_MS")) { // read once, here: how long a receiver waits for a peer's slot before it reports the peer missing
        const long long ms = atoll(t);
        if (ms > 0) c->timeout_ticks = (uint64_t)ms * 100000ull;
    }
    *out = c;
    return WG_OK;
}

int wg_comm_destroy(wg_comm *c) {
    if (!c) return WG_OK;
    (void)hipSetDevice(c->ctx->device);
    for (hipStream_t st : c->peer_stream)
        if (st) (void)hipStreamSynchronize(st);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->ctx->stream);
    for (auto &w : c->wait_stamps) { // (after the streams they were recorded on have drained)
        if (w.before) (void)hipEventDestroy(w.before);
        if (w.after) (void)hipEventDestroy(w.after);
    }
    c->wait_stamps.clear();
    if (c->nccl) (void)rccl().CommDestroy(c->nccl);
    for (hipEvent_t e : c->ev_panel)
        if (e) (void)hipEventDestroy(e);
    if (c->ev_ctx) (void)hipEventDestroy(c->ev_ctx);
    if (c->ev_comm) (void)hipEventDestroy(c->ev_comm);
    for (hipStream_t st : c->peer_stream)
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (hipEvent_t e : c->sent_ev)
        if (e) (void)hipEventDestroy(e);
    if (c->pstage) (void)hipFree(c->pstage);
    if (c->pflags) (void)hipFree(c->pflags);
    if (c->panel_sync) (void)hipFree(c->panel_sync);
    if (c->seq_src) (void)hipFree(c->seq_src);
    if (c->wait_err) {
        wg_ctx_unregister_async_error(c->ctx, c->wait_err);
        (void)hipHostFree(c->wait_err);
    }
    if (c->stage) (void)hipFree(c->stage);
    if (c->token) (void)hipFree(c->token);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return WG_OK;
}

int wg_comm_rank(const wg_comm *c) { return c ? c->rank : -1; }
int wg_comm_size(const wg_comm *c) { return c ? c->nranks : 0; }
int wg_comm_has_collectives(const wg_comm *c) { return c && c->nccl ? 1 : 0; }
int wg_comm_reported_size(const wg_comm *c, int *count) {
    if (!c || !count) return wg_set_error(WG_ERR_INVALID_ARG, "wg_comm_reported_size: NULL argument");
    *count = 0;
    if (!c->nccl) return WG_OK;
    if (!rccl().CommCount) return wg_set_error(WG_ERR_UNSUPPORTED, "wg_comm_reported_size: this librccl has no ncclCommCount");
    ncclResult_t r = rccl().CommCount(c->nccl, count);
    if (r != ncclSuccess) return nccl_fail("ncclCommCount", r);
    return WG_OK;
}
uint64_t wg_comm_bytes_sent(const wg_comm *c) { return c ? c->bytes_sent : 0; }

int wg_comm_set_one_launch(wg_comm *c, int on) {
    if (!c) return wg_set_error(WG_ERR_INVALID_ARG, "wg_comm_set_one_launch: comm is NULL");
    c->one_launch = on < 0 ? -1 : (on != 0 ? 1 : 0);
    return WG_OK;
}

int wg_comm_set_wait_timing(wg_comm *c, int on) {
    if (!c) return wg_set_error(WG_ERR_INVALID_ARG, "wg_comm_set_wait_timing: comm is NULL");
    c->time_waits = on != 0;
    c->wait_used = 0;
    return WG_OK;
}

int wg_comm_wait_times(wg_comm *c, uint32_t *panels, float *ms, uint32_t capacity, uint32_t *count) {
    if (!c || !count) return wg_set_error(WG_ERR_INVALID_ARG, "wg_comm_wait_times: NULL argument");
    WG_HIP_TRY(hipSetDevice(c->ctx->device));
    WG_HIP_TRY(hipStreamSynchronize(c->ctx->stream));
    uint32_t n = 0;
    for (size_t i = 0; i < c->wait_used && n < capacity; ++i, ++n) {
        float t = 0.f;
        WG_HIP_TRY(hipEventElapsedTime(&t, c->wait_stamps[i].before, c->wait_stamps[i].after));
        if (panels) panels[n] = c->wait_stamps[i].panel;
        if (ms) ms[n] = t;
    }
    *count = n;
    c->wait_used = 0;
    return WG_OK;
}

int wg_comm_set_pipelined(wg_comm *c, int on) {
    if (!c) return wg_set_error(WG_ERR_INVALID_ARG, "wg_comm_set_pipelined: comm is NULL");
    c->pipelined = on != 0;
    if (!c->pipelined) {
        WG_HIP_TRY(hipSetDevice(c->ctx->device));
        return run_pending(c);
    }
    return WG_OK;
}

// In-place all-gather of the element range [first, first + nranks*per_rank) of `buf`: rank r contributes [first + r*per_rank, +per_rank).
// Runs on the communicator's stream, ordered after everything enqueued on the context's stream so far; the context's stream does NOT
// wait for it (wg_comm_join does), so later Gemms overlap it.
int wg_all_gather(wg_comm *c, wg_dtype dtype, wg_buf *buf, uint64_t first_elem, uint64_t elems_per_rank) {
    if (!c || !buf) return wg_set_error(WG_ERR_INVALID_ARG, "wg_all_gather: NULL argument");
    if (dtype != WG_F32 && dtype != WG_F16) return wg_set_error(WG_ERR_INVALID_ARG, "wg_all_gather: unknown dtype %d", (int)dtype);
    if (!c->nccl) return wg_set_error(WG_ERR_UNSUPPORTED, "wg_all_gather: this communicator was created without a collective library (id == NULL)");
    if (c->ctx->recording) return wg_set_error(WG_ERR_INVALID_ARG, "wg_all_gather: collectives cannot be recorded");
    const size_t es = wg_dtype_size(dtype);
    const uint64_t have = buf->bytes / es;
    if (first_elem > have || elems_per_rank * (uint64_t)c->nranks > have - first_elem)
        return wg_set_error(WG_ERR_OUT_OF_BOUNDS, "wg_all_gather: range [%llu, %llu) exceeds the buffer's %llu elements", (unsigned long long)first_elem,
                            (unsigned long long)(first_elem + elems_per_rank * c->nranks), (unsigned long long)have);
    if (elems_per_rank == 0) return WG_OK;
    WG_HIP_TRY(hipSetDevice(c->ctx->device));
    WG_HIP_TRY(hipEventRecord(c->ev_ctx, c->ctx->stream));
    WG_HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_ctx, 0));
    char *base = (char *)buf->ptr + first_elem * es;
    ncclResult_t r = rccl().AllGather(base + (size_t)c->rank * elems_per_rank * es, base, elems_per_rank, dtype == WG_F16 ? ncclFloat16 : ncclFloat32, c->nccl, c->stream);
    if (r != ncclSuccess) return nccl_fail("ncclAllGather", r);
    c->bytes_sent += elems_per_rank * es;
    return WG_OK;
}

// The context's stream waits for everything in flight on the communicator's stream (collectives).
int wg_comm_join(wg_comm *c) {
    if (!c) return wg_set_error(WG_ERR_INVALID_ARG, "wg_comm_join: comm is NULL");
    WG_HIP_TRY(hipSetDevice(c->ctx->device));
    if (int rc = run_pending(c)) return rc;
    WG_HIP_TRY(hipEventRecord(c->ev_comm, c->stream));
    WG_HIP_TRY(hipStreamWaitEvent(c->ctx->stream, c->ev_comm, 0));
    return WG_OK;
}

// Host-blocking: every peer copy this rank issued has landed in the peers' memory.
int wg_comm_flush(wg_comm *c) {
    if (!c) return wg_set_error(WG_ERR_INVALID_ARG, "wg_comm_flush: comm is NULL");
    WG_HIP_TRY(hipSetDevice(c->ctx->device));
    if (int rc = run_pending(c)) return rc;
    return comm_flush(c);
}

// Cross-rank barrier: flush this rank's peer copies, then a one-element all-reduce on the communicator's stream, joined into the
// context's stream: work enqueued afterwards on ANY rank's context sees every rank's earlier copies and collectives complete.
int wg_comm_barrier(wg_comm *c) {
    if (!c) return wg_set_error(WG_ERR_INVALID_ARG, "wg_comm_barrier: comm is NULL");
    if (!c->nccl) return wg_set_error(WG_ERR_UNSUPPORTED, "wg_comm_barrier: no collective library on this communicator; flush and use the caller's own barrier");
    WG_HIP_TRY(hipSetDevice(c->ctx->device));
    if (int rc = run_pending(c)) return rc;
    WG_HIP_TRY(hipStreamSynchronize(c->ctx->stream)); // the panel events the copies wait for have been recorded and will fire
    if (int rc = comm_flush(c)) return rc;
    ncclResult_t r = rccl().AllReduce(c->token, c->token, 1, ncclFloat32, ncclSum, c->nccl, c->stream);
    if (r != ncclSuccess) return nccl_fail("ncclAllReduce", r);
    return wg_comm_join(c);
}

// ---------------------------------------------------------------------------------------------------------------
// buffers across processes (one process per GPU): export / open a device allocation
// ---------------------------------------------------------------------------------------------------------------
int wg_buf_ipc_export(const wg_buf *buf, void *handle) {
    if (!buf || !handle) return wg_set_error(WG_ERR_INVALID_ARG, "wg_buf_ipc_export: NULL argument");
    if (buf->host_pinned || !buf->ptr) return wg_set_error(WG_ERR_INVALID_ARG, "wg_buf_ipc_export: only non-empty device buffers can be exported");
    WG_HIP_TRY(hipSetDevice(buf->ctx->device));
    IpcHandle h;
    memset(&h, 0, sizeof h);
    hipDeviceptr_t base = nullptr;
    size_t range = 0;
    WG_HIP_TRY(hipMemGetAddressRange(&base, &range, (hipDeviceptr_t)buf->ptr)); // a wrapped pointer may sit inside a larger allocation
    WG_HIP_TRY(hipIpcGetMemHandle(&h.h, base));
    h.offset = (uint64_t)((char *)buf->ptr - (char *)base);
    h.bytes = buf->bytes;
    memset(handle, 0, WG_IPC_HANDLE_BYTES);
    memcpy(handle, &h, sizeof h);
    return WG_OK;
}

int wg_buf_ipc_open(wg_ctx *ctx, const void *handle, wg_buf **out) {
    if (!ctx || !handle || !out) return wg_set_error(WG_ERR_INVALID_ARG, "wg_buf_ipc_open: NULL argument");
    *out = nullptr;
    WG_HIP_TRY(hipSetDevice(ctx->device));
    IpcHandle h;
    memcpy(&h, handle, sizeof h);
    void *base = nullptr;
    WG_HIP_TRY(hipIpcOpenMemHandle(&base, h.h, hipIpcMemLazyEnablePeerAccess));
    wg_buf *b = new (std::nothrow) wg_buf();
    if (!b) {
        (void)hipIpcCloseMemHandle(base);
        return wg_set_error(WG_ERR_HIP, "out of host memory");
    }
    b->ctx = ctx;
    b->ptr = (char *)base + h.offset;
    b->bytes = (size_t)h.bytes;
    b->usage = WG_USAGE_STORAGE | WG_USAGE_COPY_SRC | WG_USAGE_COPY_DST;
    b->owned = false;
    b->ipc_base = base;
    *out = b;
    return WG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// staged peer copies: the communicator's staging cube + flag array, to be exported to / imported from the peers by the caller
// ---------------------------------------------------------------------------------------------------------------
int wg_comm_stage_reserve(wg_comm *c, size_t bytes, wg_buf **stage, wg_buf **flags) {
    if (!c || !stage || !flags) return wg_set_error(WG_ERR_INVALID_ARG, "wg_comm_stage_reserve: NULL argument");
    if (c->ctx->recording) return wg_set_error(WG_ERR_INVALID_ARG, "wg_comm_stage_reserve: cannot allocate while recording");
    if (c->nranks > (int)kMaxRanks) return wg_set_error(WG_ERR_UNSUPPORTED, "staged peer copies support up to %u ranks", kMaxRanks);
    WG_HIP_TRY(hipSetDevice(c->ctx->device));
    if (bytes > c->pstage_bytes) {
        WG_HIP_TRY(hipStreamSynchronize(c->ctx->stream));
        if (int rc = comm_flush(c)) return rc;
        if (c->pstage) WG_HIP_TRY(hipFree(c->pstage));
        c->pstage = nullptr;
        c->pstage_bytes = 0;
        WG_HIP_TRY(hipMalloc(&c->pstage, bytes));
        c->pstage_bytes = bytes;
        c->peer_stage.clear(); // the peers' mappings of the old cube are stale for them too: re-exchange
    }
    if (!c->pflags) {
        // uncached device memory: written by the peers' copy engines, polled by the wait kernel
        hipError_t e = hipExtMallocWithFlags((void **)&c->pflags, kFlagBytes, hipDeviceMallocUncached);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            c->pflags = nullptr;
            WG_HIP_TRY(hipExtMallocWithFlags((void **)&c->pflags, kFlagBytes, hipDeviceMallocFinegrained));
        }
        WG_HIP_TRY(hipMemset(c->pflags, 0, kFlagBytes));
    }
    // each of these is tested on its own: a failed allocation leaves the others for the next attempt, never a half-built set in use
    if (!c->seq_src) {
        WG_HIP_TRY(hipMalloc((void **)&c->seq_src, 128 * sizeof(uint32_t))); // 64 sequence words + (word 64) the wait kernels' device-side time-out word
        WG_HIP_TRY(hipMemset(c->seq_src, 0, 128 * sizeof(uint32_t)));
    }
    if (!c->wait_err) {
        WG_HIP_TRY(hipHostMalloc((void **)&c->wait_err, 64, hipHostMallocDefault));
        *c->wait_err = 0;
        // (the device-side twin, seq_src + 64, is registered with it: whichever call reports the time-out -- wg_ctx_sync, wg_buf_read, this
        // communicator's calls -- clears BOTH words, the device one in context-stream order, so the next step runs clean instead of failing fast
        // and poisoning its output without ever raising the host word again)
        wg_ctx_register_async_error(c->ctx, c->wait_err, "Gemm (sharded): a peer's slot did not arrive within the communicator's time-out (peer gone, or its buffers not registered): rank",
                                    c->seq_src + 64);
    }
    c->stage_buf.ctx = c->ctx; c->stage_buf.ptr = c->pstage; c->stage_buf.bytes = c->pstage_bytes; c->stage_buf.usage = WG_USAGE_STORAGE; c->stage_buf.owned = false;
    c->flags_buf.ctx = c->ctx; c->flags_buf.ptr = c->pflags; c->flags_buf.bytes = kFlagBytes; c->flags_buf.usage = WG_USAGE_STORAGE; c->flags_buf.owned = false;
    c->stage_buf.borrowed = c->flags_buf.borrowed = true; // members of the communicator: wg_buf_destroy on them is a no-op
    *stage = &c->stage_buf;
    *flags = &c->flags_buf;
    return WG_OK;
}

int wg_comm_set_peer_stages(wg_comm *c, wg_buf *const *peer_stage, wg_buf *const *peer_flags) {
    if (!c || !peer_stage || !peer_flags) return wg_set_error(WG_ERR_INVALID_ARG, "wg_comm_set_peer_stages: NULL argument");
    if (!c->pstage) return wg_set_error(WG_ERR_INVALID_ARG, "wg_comm_set_peer_stages: call wg_comm_stage_reserve first");
    for (int r = 0; r < c->nranks; ++r) {
        if (r == c->rank) continue;
        // the two step-parity halves of a cube sit at offsets 0 and bytes / 2: every rank must have reserved the SAME size
        if (!peer_stage[r] || !peer_flags[r] || peer_stage[r]->bytes != c->pstage_bytes || peer_flags[r]->bytes < kFlagBytes)
            return wg_set_error(WG_ERR_INVALID_ARG, "wg_comm_set_peer_stages: rank %d's staging cube / flag array is missing, or its cube is not the size of this rank's (%zu bytes)", r, c->pstage_bytes);
    }
    WG_HIP_TRY(hipSetDevice(c->ctx->device));
    c->peer_stage.assign(peer_stage, peer_stage + c->nranks);
    c->peer_flags.assign(peer_flags, peer_flags + c->nranks);
    while ((int)c->peer_stream.size() < c->nranks) {
        hipStream_t st = nullptr;
        WG_HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        c->peer_stream.push_back(st);
    }
    return WG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// relayout: gathered cube -> columns of the M x N matrix (RCCL mode; also exported for callers that all-gather themselves)
// ---------------------------------------------------------------------------------------------------------------
int wg_cube_to_matrix(wg_ctx *ctx, wg_dtype dtype, const wg_buf *cube, wg_view_shape cube_shape, wg_buf *out, wg_view_shape out_shape) {
    if (!ctx || !cube || !out) return wg_set_error(WG_ERR_INVALID_ARG, "cube_to_matrix: NULL argument");
    if (dtype != WG_F32 && dtype != WG_F16) return wg_set_error(WG_ERR_INVALID_ARG, "cube_to_matrix: unknown dtype %d", (int)dtype);
    const uint32_t mg = cube_shape.size[0], np = cube_shape.size[1], P = cube_shape.size[2];
    if (out_shape.size[0] != mg * P || out_shape.size[1] != np || out_shape.size[2] != 1)
        return wg_set_error(WG_ERR_DIM_MISMATCH, "cube_to_matrix: dimension mismatch. (cube [%u,%u,%u] -> out [%u,%u,%u])", mg, np, P, out_shape.size[0],
                            out_shape.size[1], out_shape.size[2]);
    if (mg == 0 || np == 0 || P == 0) return WG_OK;
    if (cube_shape.stride != mg || (P > 1 && cube_shape.stride_mat != mg * np))
        return wg_set_error(WG_ERR_PRECONDITION, "cube_to_matrix: the cube must be dense (stride == rows, stride_mat == rows*cols)");
    const size_t es = wg_dtype_size(dtype);
    const uint64_t need_c = (uint64_t)cube_shape.offset + (uint64_t)mg * np * P;
    const uint64_t need_o = (uint64_t)out_shape.offset + (uint64_t)(np - 1) * out_shape.stride + (uint64_t)mg * P;
    if (need_c > cube->bytes / es || need_o > out->bytes / es) return wg_set_error(WG_ERR_OUT_OF_BOUNDS, "cube_to_matrix: a view exceeds its buffer");
    WG_HIP_TRY(hipSetDevice(ctx->device));
    return launch_cube_to_matrix(ctx, (const char *)cube->ptr + (size_t)cube_shape.offset * es, (char *)out->ptr + (size_t)out_shape.offset * es, mg, np, P,
                                 out_shape.stride, es);
}

// ---------------------------------------------------------------------------------------------------------------
// the M-sharded Gemm
// ---------------------------------------------------------------------------------------------------------------
// `widths` = the N-panels' column counts, left to right (multiples of 4 summing to N). The one-launch forms take lists of the shape "n equal panels
// of whole 256-column tiles, then 1 .. 8 other panels of whole tiles (the last one: whatever is left)" -- a uniform split and a tapered tail are
// both of that shape; any other list runs panel by panel.
static int gemm_sharded_impl(wg_comm *c, wg_gemm_variant variant, wg_dtype dtype, wg_gather_mode mode, const uint32_t *widths, uint32_t nwidths, wg_buf *out,
                             wg_view_shape out_shape, const wg_buf *a_rows, wg_view_shape a_shape, const wg_buf *b, wg_view_shape b_shape) {
    if (!c || !out || !a_rows || !b) return wg_set_error(WG_ERR_INVALID_ARG, "Gemm (sharded): NULL argument");
    if (dtype != WG_F32 && dtype != WG_F16) return wg_set_error(WG_ERR_INVALID_ARG, "Gemm (sharded): unknown dtype %d", (int)dtype);
    if ((int)variant < 0 || (int)variant > 3) return wg_set_error(WG_ERR_INVALID_ARG, "Gemm: unknown variant %d", (int)variant);
    if ((int)mode < 0 || (int)mode > 3 || (int)mode == 1) return wg_set_error(WG_ERR_INVALID_ARG, "Gemm (sharded): unknown gather mode %d", (int)mode);
    wg_ctx *ctx = c->ctx;
    if (ctx->recording) return wg_set_error(WG_ERR_INVALID_ARG, "Gemm (sharded): cannot be recorded");
    const bool tr = variant == WG_GEMM_TR || variant == WG_GEMM_TR_FAST;
    const uint32_t P = (uint32_t)c->nranks, g = (uint32_t)c->rank;
    const uint32_t M = out_shape.size[0], N = out_shape.size[1];
    const uint32_t mg = tr ? a_shape.size[1] : a_shape.size[0], K = tr ? a_shape.size[0] : a_shape.size[1];
    // gemm.rs:91-95 on the sharded operands: rank g's rows of op(A) times B is rank g's row block of C
    if ((uint64_t)mg * P != M || b_shape.size[0] != K || b_shape.size[1] != N || out_shape.size[2] != 1 || a_shape.size[2] != 1 || b_shape.size[2] != 1)
        return wg_set_error(WG_ERR_DIM_MISMATCH, "Gemm: dimension mismatch. (sharded over %u ranks: out [%u,%u,%u], m1 row block [%u,%u,%u]%s, m2 [%u,%u,%u])", P, M, N,
                            out_shape.size[2], a_shape.size[0], a_shape.size[1], a_shape.size[2], tr ? "^T" : "", b_shape.size[0], b_shape.size[1], b_shape.size[2]);
    if (M == 0 || N == 0) return WG_OK;
    if (mg % 4 || N % 4) return wg_set_error(WG_ERR_PRECONDITION, "Gemm (sharded): the row block (%u rows) and N=%u must be multiples of 4 (vec4 views, shape.wgsl:64-66)", mg, N);
    if (mode == WG_GATHER_RCCL && P > 1 && !c->nccl) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm (sharded): WG_GATHER_RCCL needs a communicator created with a unique id");
    // ---- the panel plan: first column and width of every panel ----
    std::vector<uint32_t> pc0, pnp;
    {
        uint64_t at = 0;
        for (uint32_t i = 0; i < nwidths; ++i) {
            if (widths[i] == 0 || widths[i] % 4) return wg_set_error(WG_ERR_PRECONDITION, "Gemm (sharded): panel %u is %u columns wide; widths must be positive multiples of 4", i, widths[i]);
            pc0.push_back((uint32_t)at);
            pnp.push_back(widths[i]);
            at += widths[i];
        }
        if (at != N) return wg_set_error(WG_ERR_PRECONDITION, "Gemm (sharded): the panel widths sum to %llu, N is %u", (unsigned long long)at, N);
    }
    const uint32_t npanels = (uint32_t)pnp.size();
    uint32_t panel_cols = 0; // the widest panel (sizes the panel-by-panel staging)
    for (uint32_t w : pnp) panel_cols = w > panel_cols ? w : panel_cols;
    // "n_main equal panels of whole tiles + 1 .. 8 tail panels of whole tiles (the last: the rest)": what the one-launch kernels take (wgk_panels)
    wgk_panels shape;
    bool one_launch_shape = false;
    if (npanels > 1 && pnp[0] % 256u == 0) {
        uint32_t n_main = 1;
        while (n_main < npanels - 1u && pnp[n_main] == pnp[0]) ++n_main;
        // (a uniform split whose last panel is as wide as the others: that one is the tail)
        const uint32_t n_tail = npanels - n_main;
        bool ok = n_tail >= 1 && n_tail <= 8;
        for (uint32_t q = 0; ok && q + 1u < n_tail; ++q) ok = pnp[n_main + q] % 256u == 0 && pnp[n_main + q] / 256u <= 255u;
        ok = ok && (pnp[npanels - 1u] + 255u) / 256u <= 255u;
        if (ok) {
            shape.cols = pnp[0]; shape.n_main = n_main; shape.n_tail = n_tail;
            for (uint32_t q = 0; q < n_tail; ++q) shape.tail_cols[q] = pnp[n_main + q];
            shape.col_stride = M; shape.slot_rows = (uint64_t)g * mg;
            one_launch_shape = true;
        }
    }
    const size_t es = wg_dtype_size(dtype);
    WG_HIP_TRY(hipSetDevice(ctx->device));

    // (the RCCL engine's one-launch form completes a deferred last panel itself, behind its kernel)
    const bool rccl_one = mode == WG_GATHER_RCCL && c->nccl != nullptr && c->one_launch != 0 && c->can_wait_value && dtype == WG_F16 && one_launch_shape &&
                          2ull * M * N < (1ull << 32);
    if (mode != WG_GATHER_PEER_STAGED && !rccl_one)
        if (int rc = run_pending(c)) return rc;
    if (mode == WG_GATHER_PEER_STAGED) {
        // ---- contiguous per-link copies into the peers' staging cubes + flag, wait kernel + relayout on the receiving side ----
        if (npanels > kMaxPanels) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm (sharded): more than %u panels", kMaxPanels);
        const size_t cube_bytes = (size_t)M * N * es;
        if (!c->pstage || ((c->pstage_bytes / 2) & ~(size_t)15) < cube_bytes)
            return wg_set_error(WG_ERR_INVALID_ARG, "Gemm (sharded): WG_GATHER_PEER_STAGED needs wg_comm_stage_reserve(>= %zu bytes) and, with > 1 rank, wg_comm_set_peer_stages", 2 * cube_bytes);
        if (P > 1 && (int)c->peer_stage.size() != c->nranks)
            return wg_set_error(WG_ERR_INVALID_ARG, "Gemm (sharded): WG_GATHER_PEER_STAGED: peers' staging cubes are not registered (wg_comm_set_peer_stages)");
        // The two step-parity halves of the cube sit at FIXED offsets (0 and half of the reserved size, the same on every rank:
        // wg_comm_set_peer_stages), whatever the shape of a step. A step's copies into a peer's half are issued after this rank has
        // seen the peer's flags of the previous step -- which the peer raises after its Gemms of that step, i.e. after it has
        // finished every relayout out of that half (two steps ago) -- so a change of M / N / panel_cols between steps needs no
        // cross-rank agreement beyond the one every step makes.
        const uint64_t half_elems = ((c->pstage_bytes / 2) & ~(size_t)15) / es;
        if (half_elems + (uint64_t)M * N >= (1ull << 32)) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm (sharded): the two staging cubes exceed u32 element indexing");
        if (int rc = wg_ctx_check_async(ctx)) return rc; // a wait of an earlier step gave up: reported here (both words cleared: the next call is a clean retry)
        uint64_t plan_hash = 1469598103934665603ull; // FNV-1a over the widths: another split of the same M x N is another slot layout
        for (uint32_t w : pnp) plan_hash = (plan_hash ^ w) * 1099511628211ull;
        const uint64_t geom[4] = { M, N, plan_hash, es };
        if (memcmp(geom, c->staged_geom, sizeof geom) != 0) {
            // another slot layout than the previous call's: its per-slot "sent" events no longer name these slots -- let everything of
            // THIS rank that is still leaving the old layout finish before any Gemm writes the cubes (host-blocking on this rank's own
            // copy streams only, first call of a new shape only)
            if (int rc = run_pending(c)) return rc;
            for (hipStream_t st : c->peer_stream)
                if (st) WG_HIP_TRY(hipStreamSynchronize(st));
            for (hipEvent_t &e : c->sent_ev)
                if (e) { (void)hipEventDestroy(e); e = nullptr; }
            memcpy(c->staged_geom, geom, sizeof geom);
        }
        const uint32_t seq = ++c->step, parity = seq & 1u;
        // the sequence number the flag copies carry: written into word (seq % 64) on the context's stream ahead of this step's Gemms, by
        // a kernel that takes it as an argument. (Word seq % 64 is next written for step seq + 64, on the context's stream behind Gemms
        // that waited for the copies of step seq + 62, which follow step seq's on every peer stream.)
        hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, ctx->stream, c->seq_src + (seq % 64u), seq);
        WG_HIP_TRY(hipGetLastError());
        while (c->sent_ev.size() < (size_t)2 * npanels * P) c->sent_ev.push_back(nullptr);
        wg_buf sbuf;
        sbuf.ctx = ctx; sbuf.ptr = c->pstage; sbuf.bytes = c->pstage_bytes; sbuf.usage = 0; sbuf.owned = false; sbuf.host_pinned = false;
        auto slot_elem = [&](uint32_t c0, uint32_t np, uint32_t r) { return (uint64_t)parity * half_elems + (uint64_t)c0 * M + (uint64_t)r * mg * np; };
        auto finish_panel = [&](uint32_t p) -> int { // wait for the peers' slots of panel p, then relayout it into columns of `out`
            const uint32_t c0 = pc0[p], np = pnp[p];
            if (int rc = wait_begin(c, p)) return rc;
            if (P > 1) {
                hipLaunchKernelGGL(wait_flags_kernel, dim3(1), dim3(64), 0, ctx->stream, c->pflags, P, g, p, seq, c->wait_err, c->seq_src + 64, c->timeout_ticks);
                WG_HIP_TRY(hipGetLastError());
            }
            if (int rc = wait_end(c)) return rc;
            const char *src = (const char *)c->pstage + slot_elem(c0, np, 0) * es;
            char *dst = (char *)out->ptr + ((size_t)out_shape.offset + (size_t)c0 * out_shape.stride) * es;
            return launch_cube_to_matrix(ctx, src, dst, mg, np, P, out_shape.stride, es, P > 1 ? c->seq_src + 64 : nullptr);
        };
        static const bool no_copy = getenv("WG_STAGED_NO_COPY") != nullptr; // tools/rank_emulation.py: one rank's compute + relayout alone
        // ---- ONE launch per step (f16 products of at least one round of tiles; on request for this engine: wg_comm_set_one_launch): the kernel
        // walks the panels left to right, its waves count themselves into a word per panel as their write-through stores reach memory; every
        // peer stream waits for panel p's full count (hipStreamWaitValue32) and pushes the slot; the relayouts follow the kernel on the
        // context's stream. The tile scheduler sees the rank's whole product.
        if (c->one_launch == 1 && c->can_wait_value && dtype == WG_F16 && one_launch_shape) {
            if (int rc = ensure_panel_sync(c)) return rc;
            wgk_panels pa = shape;
            pa.counters = c->panel_sync;
            // every slot of this parity is rewritten by the one kernel: the copies that left them two steps ago must be done
            for (uint32_t p = 0; p < npanels; ++p)
                for (uint32_t r = 0; r < P; ++r) {
                    hipEvent_t e = c->sent_ev[((size_t)parity * npanels + p) * P + r];
                    if (r != g && e) WG_HIP_TRY(hipStreamWaitEvent(ctx->stream, e, 0));
                }
            const int rc1 = wg_gemm_f16_panels(ctx, tr, (char *)c->pstage + slot_elem(0, pa.cols, g) * es, mg, a_rows, a_shape, b, b_shape, pa);
            if (rc1 == WG_OK) {
                for (uint32_t p = 0; p < npanels; ++p) // this launch's arrivals, on top of every earlier launch's
                    c->panel_total[p] += wgk_panel_goal(mg, pnp[p]);
                if (P > 1 && !no_copy) {
                    for (uint32_t p = 0; p < npanels; ++p) {
                        const uint32_t c0 = pc0[p], np = pnp[p];
                        const size_t off = slot_elem(c0, np, g) * es, bytes = (size_t)mg * np * es;
                        for (uint32_t i = 1; i < P; ++i) {
                            const uint32_t r = (g + i) % P;
                            hipStream_t st = c->peer_stream[r];
                            WG_HIP_TRY(hipStreamWaitValue32(st, pa.counters + p, c->panel_total[p], hipStreamWaitValueGte, 0xffffffffu));
                            WG_HIP_TRY(hipMemcpyAsync((char *)c->peer_stage[r]->ptr + off, (const char *)c->pstage + off, bytes, hipMemcpyDeviceToDevice, st));
                            WG_HIP_TRY(hipMemcpyAsync((char *)c->peer_flags[r]->ptr + ((size_t)g * kMaxPanels + p) * sizeof(uint32_t), c->seq_src + (seq % 64u),
                                                      sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
                            hipEvent_t &e = c->sent_ev[((size_t)parity * npanels + p) * P + r];
                            if (!e) WG_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                            WG_HIP_TRY(hipEventRecord(e, st));
                        }
                        c->bytes_sent += (uint64_t)(P - 1) * bytes;
                    }
                }
                if (int rc = run_pending(c)) return rc; // the previous call's deferred last panel: behind this call's kernel
                const uint32_t upto = c->pipelined ? npanels - 1u : npanels;
                for (uint32_t p = 0; p < upto; ++p)
                    if (int rc = finish_panel(p)) return rc;
                if (c->pipelined) {
                    const uint32_t p = npanels - 1, c0 = pc0[p], np = pnp[p];
                    c->pending.on = true; c->pending.after = nullptr;
                    c->pending.seq = seq; c->pending.panel = p; c->pending.mg = mg; c->pending.np = np;
                    c->pending.src = (const char *)c->pstage + slot_elem(c0, np, 0) * es;
                    c->pending.dst = (char *)out->ptr + ((size_t)out_shape.offset + (size_t)c0 * out_shape.stride) * es;
                    c->pending.ld = out_shape.stride; c->pending.es = es;
                }
                return WG_OK;
            }
            if (rc1 != WG_ERR_UNSUPPORTED) return rc1; // (unsupported: not that kind of product -- panel by panel below)
        }
        for (uint32_t p = 0; p < npanels; ++p) {
            const uint32_t c0 = pc0[p], np = pnp[p];
            wg_view_shape bs = b_shape;
            bs.size[1] = np;
            const uint64_t b_off = (uint64_t)b_shape.offset + (uint64_t)c0 * b_shape.stride; // 64-bit: K * N may reach 2^32 elements
            if (b_off >= (1ull << 32)) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm (sharded): panel %u of m2 starts beyond u32 element indexing", p);
            bs.offset = (uint32_t)b_off;
            wg_view_shape os;
            os.size[0] = mg; os.size[1] = np; os.size[2] = 1; os.stride = mg; os.stride_mat = mg * np;
            os.offset = (uint32_t)slot_elem(c0, np, g);
            // this slot last left for the peers two steps ago (same parity): those copies must be done before the Gemm overwrites it
            for (uint32_t r = 0; r < P; ++r) {
                hipEvent_t e = c->sent_ev[((size_t)parity * npanels + p) * P + r];
                if (r != g && e) WG_HIP_TRY(hipStreamWaitEvent(ctx->stream, e, 0));
            }
            if (int rc = wg_gemm_ex(ctx, variant, dtype, 1.f, 0.f, &sbuf, os, a_rows, a_shape, b, bs)) return rc;
            if (P > 1 && !no_copy) {
                WG_HIP_TRY(hipEventRecord(c->ev_ctx, ctx->stream));
                const size_t off = slot_elem(c0, np, g) * es, bytes = (size_t)mg * np * es;
                for (uint32_t i = 1; i < P; ++i) { // start with the next rank: every link is busy from the first panel on
                    const uint32_t r = (g + i) % P;
                    hipStream_t st = c->peer_stream[r];
                    WG_HIP_TRY(hipStreamWaitEvent(st, c->ev_ctx, 0));
                    WG_HIP_TRY(hipMemcpyAsync((char *)c->peer_stage[r]->ptr + off, (const char *)c->pstage + off, bytes, hipMemcpyDeviceToDevice, st));
                    WG_HIP_TRY(hipMemcpyAsync((char *)c->peer_flags[r]->ptr + ((size_t)g * kMaxPanels + p) * sizeof(uint32_t), c->seq_src + (seq % 64u),
                                              sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
                    hipEvent_t &e = c->sent_ev[((size_t)parity * npanels + p) * P + r];
                    if (!e) WG_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    WG_HIP_TRY(hipEventRecord(e, st));
                }
                c->bytes_sent += (uint64_t)(P - 1) * bytes;
            }
            if (p > 0) {
                if (int rc = finish_panel(p - 1)) return rc; // after this panel's Gemm: the previous panel's slots travelled beside it
            } else if (int rc = run_pending(c)) return rc;   // ... and the previous CALL's last panel beside this call's first Gemm (pipelined steps)
        }
        // Pipelined steps leave the last panel to the next call (or wg_comm_join): nothing in this call can hide its exchange. NOT with a
        // single panel: the next call would then enqueue its Gemm and its copy into the peers' other half before having waited for
        // anything of this step -- the "at most one step ahead of a peer" rule the parity halves rely on would not hold (the peer could
        // still be reading that half) -- so a one-panel step completes in the call.
        if (c->pipelined && npanels > 1) {
            const uint32_t p = npanels - 1, c0 = pc0[p], np = pnp[p];
            c->pending.on = true; c->pending.after = nullptr;
            c->pending.seq = seq; c->pending.panel = p; c->pending.mg = mg; c->pending.np = np;
            c->pending.src = (const char *)c->pstage + slot_elem(c0, np, 0) * es;
            c->pending.dst = (char *)out->ptr + ((size_t)out_shape.offset + (size_t)c0 * out_shape.stride) * es;
            c->pending.ld = out_shape.stride; c->pending.es = es;
            return WG_OK;
        }
        return finish_panel(npanels - 1);
    }

    const bool staged = mode == WG_GATHER_RCCL && c->nccl != nullptr; // a 1-rank communicator still runs the whole path (tests)
    // ---- RCCL engine, ONE launch per step (f16 products of at least one round of tiles): staging cubes for two whole steps (step parity, as
    // the staged engine's), the kernel's waves count themselves into a word per panel, the communicator's stream waits for the full count
    // (hipStreamWaitValue32) and all-gathers the panel while the kernel works on the next ones; the relayouts follow the kernel on the
    // context's stream, each behind its panel's gather. Pipelined steps (wg_comm_set_pipelined) leave the last panel's relayout -- the one
    // gather nothing of its own step hides -- to the next call, behind that call's kernel.
    if (rccl_one) {
        // The two step-parity cubes sit at FIXED offsets -- 0 and half of the reserved size -- whatever the shape of a step (as the staged engine's):
        // with pipelined steps the previous call's last panel may still be gathering into ITS cube, laid out for ITS M x N, while this call's kernel
        // (enqueued before that panel's relayout) writes the other one; halves computed from this call's M * N would overlap it when shapes alternate.
        const size_t cube_bytes = ((size_t)M * N * es + 255) & ~(size_t)255;
        if (2 * cube_bytes > c->stage_bytes) {
            if (int rc = run_pending(c)) return rc;
            WG_HIP_TRY(hipStreamSynchronize(ctx->stream));
            WG_HIP_TRY(hipStreamSynchronize(c->stream));
            if (c->stage) WG_HIP_TRY(hipFree(c->stage));
            c->stage = nullptr;
            c->stage_bytes = 0;
            WG_HIP_TRY(hipMalloc(&c->stage, 2 * cube_bytes));
            c->stage_bytes = 2 * cube_bytes;
        }
        const size_t half = (c->stage_bytes / 2) & ~(size_t)255;
        if (int rc = ensure_panel_sync(c)) return rc;
        if (npanels > kMaxPanels) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm (sharded): more than %u panels", kMaxPanels);
        const uint32_t parity = (++c->step) & 1u;
        // the events of a parity live at [kMaxPanels * (1 + parity), +npanels): also independent of the step's shape (a deferred panel keeps its own
        // event handle), and clear of [0, kMaxPanels), the panel-by-panel form's events
        if (c->ev_panel.size() < 3 * (size_t)kMaxPanels) c->ev_panel.resize(3 * (size_t)kMaxPanels, nullptr);
        for (uint32_t p = 0; p < npanels; ++p) {
            hipEvent_t &e = c->ev_panel[(size_t)kMaxPanels * (1u + parity) + p];
            if (!e) WG_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        char *cube = (char *)c->stage + (size_t)parity * half; // this step's cube; the other one may still be gathering its last panel
        wgk_panels pa = shape;
        pa.counters = c->panel_sync;
        // (this parity's cube is free: its last user's relayouts -- two steps ago, the deferred one included, see below -- precede this
        // kernel on the context's stream, and this step's gathers only start on counts this kernel produces)
        if (c->pending.on && c->pending.after && (c->pending.seq & 1u) == parity)
            if (int rc = run_pending(c)) return rc; // (cannot happen with alternating parities; kept as a guard)
        const int rc1 = wg_gemm_f16_panels(ctx, tr, cube + (size_t)g * mg * pa.cols * es, mg, a_rows, a_shape, b, b_shape, pa);
        if (rc1 == WG_OK) {
            hipEvent_t *ev = c->ev_panel.data() + (size_t)kMaxPanels * (1u + parity);
            for (uint32_t p = 0; p < npanels; ++p) {
                const uint32_t c0 = pc0[p], np = pnp[p];
                char *base = cube + (size_t)c0 * M * es;
                c->panel_total[p] += wgk_panel_goal(mg, np);
                WG_HIP_TRY(hipStreamWaitValue32(c->stream, pa.counters + p, c->panel_total[p], hipStreamWaitValueGte, 0xffffffffu));
                ncclResult_t r = rccl().AllGather(base + (size_t)g * mg * np * es, base, (size_t)mg * np, ncclFloat16, c->nccl, c->stream);
                if (r != ncclSuccess) return nccl_fail("ncclAllGather", r);
                c->bytes_sent += (uint64_t)mg * np * es;
                WG_HIP_TRY(hipEventRecord(ev[p], c->stream));
            }
            if (int rc = run_pending(c)) return rc; // the previous call's deferred last panel: behind this call's kernel
            const uint32_t upto = c->pipelined ? npanels - 1u : npanels;
            for (uint32_t p = 0; p < npanels; ++p) {
                const uint32_t c0 = pc0[p], np = pnp[p];
                const char *src = cube + (size_t)c0 * M * es;
                char *dst = (char *)out->ptr + ((size_t)out_shape.offset + (size_t)c0 * out_shape.stride) * es;
                if (p < upto) {
                    if (int rc = wait_begin(c, p)) return rc;
                    WG_HIP_TRY(hipStreamWaitEvent(ctx->stream, ev[p], 0));
                    if (int rc = wait_end(c)) return rc;
                    if (int rc = launch_cube_to_matrix(ctx, src, dst, mg, np, P, out_shape.stride, es)) return rc;
                } else {
                    c->pending.on = true; c->pending.after = ev[p];
                    c->pending.seq = c->step; c->pending.panel = p; c->pending.mg = mg; c->pending.np = np;
                    c->pending.src = src; c->pending.dst = dst; c->pending.ld = out_shape.stride; c->pending.es = es;
                }
            }
            return WG_OK;
        }
        if (rc1 != WG_ERR_UNSUPPORTED) return rc1;
        if (int rc = run_pending(c)) return rc; // panel by panel below
    }
    wg_buf stage_buf;
    if (staged) {
        const size_t need = 2 * (size_t)M * panel_cols * es;
        if (need > c->stage_bytes) {
            WG_HIP_TRY(hipStreamSynchronize(ctx->stream));
            WG_HIP_TRY(hipStreamSynchronize(c->stream));
            if (c->stage) WG_HIP_TRY(hipFree(c->stage));
            c->stage = nullptr;
            c->stage_bytes = 0;
            WG_HIP_TRY(hipMalloc(&c->stage, need));
            c->stage_bytes = need;
        }
        stage_buf.ctx = ctx; stage_buf.ptr = c->stage; stage_buf.bytes = c->stage_bytes; stage_buf.usage = 0; stage_buf.owned = false; stage_buf.host_pinned = false;
        if (npanels > kMaxPanels) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm (sharded): more than %u panels", kMaxPanels);
        if (c->ev_panel.size() < npanels) c->ev_panel.resize(npanels, nullptr);
        for (uint32_t p = 0; p < npanels; ++p)
            if (!c->ev_panel[p]) WG_HIP_TRY(hipEventCreateWithFlags(&c->ev_panel[p], hipEventDisableTiming));
    }

    auto relayout = [&](uint32_t p) -> int { // panel p's cube -> columns of `out`, once its all-gather is done
        const uint32_t c0 = pc0[p], np = pnp[p];
        if (int rc = wait_begin(c, p)) return rc;
        WG_HIP_TRY(hipStreamWaitEvent(ctx->stream, c->ev_panel[p], 0));
        if (int rc = wait_end(c)) return rc;
        const char *src = (const char *)c->stage + (size_t)(p & 1u) * M * panel_cols * es;
        char *dst = (char *)out->ptr + ((size_t)out_shape.offset + (size_t)c0 * out_shape.stride) * es;
        return launch_cube_to_matrix(ctx, src, dst, mg, np, P, out_shape.stride, es);
    };

    for (uint32_t p = 0; p < npanels; ++p) {
        const uint32_t c0 = pc0[p], np = pnp[p];
        wg_view_shape bs = b_shape;
        bs.size[1] = np;
        const uint64_t b_off = (uint64_t)b_shape.offset + (uint64_t)c0 * b_shape.stride; // 64-bit: K * N may reach 2^32 elements
        if (b_off >= (1ull << 32)) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm (sharded): panel %u of m2 starts beyond u32 element indexing", p);
        bs.offset = (uint32_t)b_off;
        wg_view_shape os;
        os.size[0] = mg; os.size[1] = np; os.size[2] = 1;
        if (staged) { // slot g of the staging cube of this panel
            const uint64_t slot0 = (uint64_t)(p & 1u) * M * panel_cols;
            os.stride = mg; os.stride_mat = mg * np; os.offset = (uint32_t)(slot0 + (uint64_t)g * mg * np);
            if (slot0 + (uint64_t)M * np >= (1ull << 32)) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm (sharded): staging cube exceeds u32 element indexing; use narrower panels");
            if (int rc = wg_gemm_ex(ctx, variant, dtype, 1.f, 0.f, &stage_buf, os, a_rows, a_shape, b, bs)) return rc;
            WG_HIP_TRY(hipEventRecord(c->ev_ctx, ctx->stream));
            WG_HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_ctx, 0));
            char *base = (char *)c->stage + slot0 * es;
            ncclResult_t r = rccl().AllGather(base + (size_t)g * mg * np * es, base, (size_t)mg * np, dtype == WG_F16 ? ncclFloat16 : ncclFloat32, c->nccl, c->stream);
            if (r != ncclSuccess) return nccl_fail("ncclAllGather", r);
            c->bytes_sent += (uint64_t)mg * np * es;
            WG_HIP_TRY(hipEventRecord(c->ev_panel[p], c->stream));
            if (p > 0)
                if (int rc = relayout(p - 1)) return rc; // after this panel's Gemm: the previous panel's gather ran beside it
        } else { // straight into this rank's rows of C
            os.stride = out_shape.stride; os.stride_mat = out_shape.stride_mat;
            const uint64_t off = (uint64_t)out_shape.offset + (uint64_t)c0 * out_shape.stride + (uint64_t)g * mg;
            if (off >= (1ull << 32)) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemm (sharded): output view exceeds u32 element indexing");
            os.offset = (uint32_t)off;
            if (int rc = wg_gemm_ex(ctx, variant, dtype, 1.f, 0.f, out, os, a_rows, a_shape, b, bs)) return rc;
        }
    }
    if (staged) return relayout(npanels - 1); // the only exposed exchange
    return WG_OK;
}

int wg_gemm_sharded(wg_comm *c, wg_gemm_variant variant, wg_dtype dtype, wg_gather_mode mode, uint32_t panel_cols, wg_buf *out, wg_view_shape out_shape,
                    const wg_buf *a_rows, wg_view_shape a_shape, const wg_buf *b, wg_view_shape b_shape) {
    const uint32_t N = out_shape.size[1];
    if (panel_cols == 0) // default: ~8 panels of whole 256-column tiles, so that all but the last exchange hides under a Gemm
        panel_cols = N >= 2048 ? ((N / 8 + 255u) / 256u) * 256u : N;
    if (panel_cols % 4) return wg_set_error(WG_ERR_PRECONDITION, "Gemm (sharded): panel_cols=%u must be a multiple of 4", panel_cols);
    if (panel_cols > N) panel_cols = N;
    std::vector<uint32_t> widths;
    for (uint32_t c0 = 0; c0 < N; c0 += panel_cols) widths.push_back(N - c0 < panel_cols ? N - c0 : panel_cols);
    return gemm_sharded_impl(c, variant, dtype, mode, widths.data(), (uint32_t)widths.size(), out, out_shape, a_rows, a_shape, b, b_shape);
}

int wg_gemm_sharded_panels(wg_comm *c, wg_gemm_variant variant, wg_dtype dtype, wg_gather_mode mode, const uint32_t *panel_widths, uint32_t npanels, wg_buf *out,
                           wg_view_shape out_shape, const wg_buf *a_rows, wg_view_shape a_shape, const wg_buf *b, wg_view_shape b_shape) {
    if (!panel_widths || npanels == 0) return wg_set_error(WG_ERR_INVALID_ARG, "Gemm (sharded): no panel widths");
    return gemm_sharded_impl(c, variant, dtype, mode, panel_widths, npanels, out, out_shape, a_rows, a_shape, b, b_shape);
}

} // extern "C"
