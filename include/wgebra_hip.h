/*
 * wgebra_hip.h -- C ABI of libwgebra_hip.so: the MI355X (gfx950) backend behind the wgebra operator
 * surface (Gemm / Gemv / Reduce / OpAssign on wgcore GpuTensor / GpuTensorView).
 *
 * It replaces exactly the slab of the reference that sits under that surface:
 *   wgpu::Buffer storage          (wgcore tensor.rs:117-122,150-154)      -> wg_buf_*      (hipMalloc / hipHostMalloc)
 *   CommandEncoder/ComputePass +
 *   Queue::submit + Device::poll  (wgcore kernel.rs:15-27,125-146;
 *                                  tensor.rs:300-325)                      -> wg_ctx_*      (one in-order hipStream_t)
 *   ViewShape uniform buffers     (wgcore shapes.rs:9-21,107-116)          -> wg_view_shape passed BY VALUE as kernel args
 *   the 10 WGSL entry points      (wgebra linalg/{gemm,gemv,reduce,
 *                                  op_assign}.wgsl)                        -> wg_gemm / wg_gemv / wg_reduce / wg_op_assign
 *   GpuTimestamps                 (wgcore timestamps.rs:9-248)             -> wg_timestamps_* (hipEvent pairs)
 *
 * Conventions
 *   - Plain C: opaque handles, plain pointers and sizes, `int` status returns (0 == WG_OK). No C++/torch types.
 *   - All tensors are COLUMN-MAJOR; a view is (buffer, wg_view_shape); element (i,j,t) lives at element index
 *     t*stride_mat + offset + i + j*stride of the buffer (shape.wgsl:45-47,60-62). Units: ELEMENTS of the dtype.
 *   - A context owns one in-order stream: operators enqueue asynchronously w.r.t. the host, in call order,
 *     exactly like dispatches recorded into one ComputePass; wg_ctx_sync / wg_buf_read block.
 *   - A context is not thread-safe; distinct contexts are independent (one per GPU for multi-GPU).
 *   - Errors: where the reference panics (assert_eq!) the call returns a status and records a message with the
 *     reference's text (wg_last_error_string); where the reference silently skips a dispatch (zero-sized buffer
 *     or grid, kernel.rs:111-123,144) the call returns WG_OK and launches nothing.
 *   - Where the reference would read or write out of bounds (its shaders run with bounds checks disabled,
 *     wgcore utils.rs:11-19) the call returns WG_ERR_OUT_OF_BOUNDS / WG_ERR_PRECONDITION instead.
 */
#ifndef WGEBRA_HIP_H
#define WGEBRA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WGEBRA_HIP_ABI_VERSION 4 /* 4: wg_gemm_sharded_panels (ragged N-panels), wg_ctx_mem_info, geometry ops 15-18; 3: the SDMA rect-copy exchange engine (gather mode 1, wg_comm_copy_engine, wg_gemm_sharded's peer_out) is gone;
                                    wg_comm_reported_size, wg_debug_*; non-vec4 views compute staged; async time-outs surface in wg_ctx_sync */

/* ------------------------------------------------------------------------------------------------ */
/* status codes                                                                                      */
/* ------------------------------------------------------------------------------------------------ */
typedef enum wg_status {
    WG_OK = 0,
    WG_ERR_DIM_MISMATCH = 1,  /* reference: assert_eq!(.., "Gemm: dimension mismatch.") & friends -> panic   */
    WG_ERR_PRECONDITION = 2,  /* an assertion of the reference that is not a dimension check: GemvFast rows % 4 (gemv.rs:122), ... */
    WG_ERR_INVALID_ARG = 3,   /* null handle, unknown enum value, foreign-context buffer                      */
    WG_ERR_OUT_OF_BOUNDS = 4, /* the view addresses elements past the end of its buffer                       */
    WG_ERR_HIP = 5,           /* a HIP runtime call failed; message carries hipGetErrorString                 */
    WG_ERR_UNSUPPORTED = 6,   /* dtype/variant combination not implemented                                    */
    WG_ERR_NO_DEVICE = 7,     /* no gfx950 device visible (GpuInstance::new() -> Err, gpu.rs:24-58)           */
    WG_ERR_WORKSPACE = 8      /* a context scratch region would have to grow while recording (allocation cannot be
                                 captured): run the call once eagerly, or wg_ctx_reserve_workspace, then record    */
} wg_status;

/* ------------------------------------------------------------------------------------------------ */
/* enums: same names and ORDER as the Rust enums                                                     */
/* ------------------------------------------------------------------------------------------------ */
typedef enum wg_dtype { WG_F32 = 0, WG_F16 = 1 } wg_dtype; /* reference kernels are f32 only; F16 is this build's extension */

typedef enum wg_gemm_variant { /* wgebra gemm.rs:26-35 */
    WG_GEMM = 0, WG_GEMM_FAST = 1, WG_GEMM_TR = 2, WG_GEMM_TR_FAST = 3
} wg_gemm_variant;

typedef enum wg_gemv_variant { /* wgebra gemv.rs:25-34 */
    WG_GEMV = 0, WG_GEMV_FAST = 1, WG_GEMV_TR = 2, WG_GEMV_TR_FAST = 3
} wg_gemv_variant;

typedef enum wg_reduce_op { /* wgebra reduce.rs:13-27 */
    WG_REDUCE_MIN = 0, WG_REDUCE_MAX = 1, WG_REDUCE_SUM = 2, WG_REDUCE_PROD = 3, WG_REDUCE_SQNORM = 4
} wg_reduce_op;

typedef enum wg_op_assign_variant { /* wgebra op_assign.rs:12-26 */
    WG_OP_ADD = 0, WG_OP_SUB = 1, WG_OP_MUL = 2, WG_OP_DIV = 3, WG_OP_COPY = 4
} wg_op_assign_variant;

/* wgpu::BufferUsages bit values (the flags TensorBuilder takes, tensor.rs:65-112). Only MAP_READ / MAP_WRITE
 * change behaviour here: a MAP_* buffer is pinned host memory (the "staging" tensor of gemm.rs:163-168). */
enum {
    WG_USAGE_MAP_READ = 1u << 0, WG_USAGE_MAP_WRITE = 1u << 1, WG_USAGE_COPY_SRC = 1u << 2,
    WG_USAGE_COPY_DST = 1u << 3, WG_USAGE_INDEX = 1u << 4, WG_USAGE_VERTEX = 1u << 5,
    WG_USAGE_UNIFORM = 1u << 6, WG_USAGE_STORAGE = 1u << 7, WG_USAGE_INDIRECT = 1u << 8,
    WG_USAGE_QUERY_RESOLVE = 1u << 9
};

/* ------------------------------------------------------------------------------------------------ */
/* wg_view_shape: byte-identical to wgcore::shapes::ViewShape (#[repr(C)], 24 B; shapes.rs:9-21)    */
/*                and to WGSL `Shape` (shape.wgsl:10-33).                                            */
/* ------------------------------------------------------------------------------------------------ */
typedef struct wg_view_shape {
    uint32_t size[3];    /* rows, cols, mats */
    uint32_t stride;     /* elements between two columns */
    uint32_t stride_mat; /* elements between two matrices */
    uint32_t offset;     /* index of the first element of the view in the buffer */
} wg_view_shape;

typedef struct wg_ctx wg_ctx;               /* GpuInstance (gpu.rs:7-12) + the encoder/pass/queue it feeds */
typedef struct wg_buf wg_buf;               /* wgpu::Buffer */
typedef struct wg_cmdbuf wg_cmdbuf;         /* wgpu::CommandBuffer (a recorded, replayable hipGraphExec) */
typedef struct wg_timestamps wg_timestamps; /* wgcore::timestamps::GpuTimestamps */

/* ------------------------------------------------------------------------------------------------ */
/* library / device                                                                                  */
/* ------------------------------------------------------------------------------------------------ */
int wg_abi_version(void);
/* Thread-local message of the last failing call on this thread ("" if none). Never NULL. */
const char *wg_last_error_string(void);
/* Number of visible HIP devices (0 without a GPU; never fails). */
int wg_device_count(void);

/* GpuInstance::new() (gpu.rs:24-58): bind `device`, create the in-order stream. */
int wg_ctx_create(int device, wg_ctx **out);
/* Same, but enqueue on an existing hipStream_t (e.g. the stream another runtime owns). Not destroyed with the ctx. */
int wg_ctx_create_on_stream(int device, void *hip_stream, wg_ctx **out);
/* A context whose stream may use only `cu_count` of the device's compute units (CU-masked stream; every XCD loses the same
 * share). For multi-GPU runs: the compute stream leaves a few CUs to the collective library's copy kernels. The tile / split
 * heuristics then plan for `cu_count` CUs. */
int wg_ctx_create_with_cu_count(int device, uint32_t cu_count, wg_ctx **out);
/* Same, but the missing CUs (fewer than one XCD's worth) all come from ONE XCD: the other seven keep their 32 CUs and the L2 locality of
 * the f16 Gemm's tile patches, and that kernel's tile scheduler lets them take the short XCD's tiles (248 CUs: 13.8 -> 13.5 ms per step of a
 * rank's 8192 x 32768 x 32768 product). For a context that runs the sharded f16 Gemm beside a collective; kernels with a static
 * tile -> XCD map run better on the evenly masked stream above. */
int wg_ctx_create_with_cu_count_one_xcd(int device, uint32_t cu_count, wg_ctx **out);
int wg_ctx_destroy(wg_ctx *ctx);
/* queue.submit(..) + device.poll(PollType::wait()) (tensor.rs:304-312): block until all enqueued work is done. Also reports (once) an
 * error a kernel of this context raised asynchronously -- a sharded Gemm whose peer never delivered (WG_ERR_HIP); so does wg_buf_read. */
int wg_ctx_sync(wg_ctx *ctx);
int wg_ctx_device(const wg_ctx *ctx);
void *wg_ctx_stream(const wg_ctx *ctx); /* the hipStream_t, for interop */
/* Device facts the bench prints next to every roofline: name (<=255 chars), CU count, clock MHz, HBM bytes. */
int wg_ctx_device_info(const wg_ctx *ctx, char *name256, int *compute_units, int *clock_mhz, uint64_t *hbm_bytes);
/* Free and total device memory in bytes right now (hipMemGetInfo on the context's device): what `wgpu::Device::limits` cannot say; the sharded
   GEMM sizes its panels from it and the tests check that dropped buffers really come back. */
int wg_ctx_mem_info(const wg_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes);
/*
 * Geometry: the reference's `wgebra::geometry` WGSL modules (the .wgsl files under crates/wgebra/src/geometry) are device
 * functions used inside other shaders; their HIP counterpart is the header include/wgebra_geometry.hpp (`__host__ __device__`).
 * This entry point applies one of them to `count` independent items, one per thread -- the same harness the reference's tests
 * use (a one-invocation-per-matrix test kernel). Item layouts: wgmath_amd/csrc/geometry.hip.
 */
typedef enum wg_geom_op {
    WG_GEOM_INV = 0, WG_GEOM_CHOLESKY = 1, WG_GEOM_LU = 2, WG_GEOM_QR = 3, WG_GEOM_SYM_EIGEN = 4, WG_GEOM_SVD = 5,
    WG_GEOM_ROT2 = 6, WG_GEOM_QUAT = 7, WG_GEOM_SIM2 = 8, WG_GEOM_SIM3 = 9,
    /* the transform functions one by one on raw coordinates (a non-unit quaternion, a (cos, sin) pair that is no rotation): the items the
       fixtures executed from the reference's WGSL text hold (tests/golden/wgsl_exec_geometry.npz); FROM = quat::fromScaledAxis + rot2::fromAngle */
    WG_GEOM_QUAT_RAW = 10, WG_GEOM_ROT2_RAW = 11, WG_GEOM_SIM2_RAW = 12, WG_GEOM_SIM3_RAW = 13, WG_GEOM_FROM = 14,
    /* UTILS = wgebra::trig (utils/trig.wgsl:12-38: stable_atan2, stable_tanh) + wgebra::min_max (utils/min_max.wgsl:4-51); ROT2_EXT = rot2.wgsl's
       angle / cancel_y / is_valid / rotate_rows3 / rotate_rows4 (:15-36, :51-53, :78-95); EIGVALS2 = eig2.wgsl:44-56 `eigenvalues`;
       SVD_RECOMPOSE (dim 2, 3) = svd2.wgsl:43-46 / svd3.wgsl:309-312 `recompose` on items in the layout WG_GEOM_SVD writes */
    WG_GEOM_UTILS = 15, WG_GEOM_ROT2_EXT = 16, WG_GEOM_EIGVALS2 = 17, WG_GEOM_SVD_RECOMPOSE = 18
} wg_geom_op;
int wg_geometry_apply(wg_ctx *ctx, wg_geom_op op, uint32_t dim, const wg_buf *in, wg_buf *out, uint32_t count);

/*
 * Diagnostics only (tools/overlap_probe.py): enqueue `blocks` workgroups of 256 threads that spin for `usec` microseconds and
 * record their start tick (s_memrealtime, 100 MHz) into start_ticks[0 .. blocks) (u64 each; start_ticks[blocks] = tick of a
 * 1-thread kernel enqueued just before). Stand-in for a collective library's copy kernel when studying queue interleaving.
 */
int wg_debug_spin(wg_ctx *ctx, uint32_t blocks, uint32_t usec, wg_buf *start_ticks);
/* What the chip did while a kernel ran (the role of GpuTimestamps, timestamps.rs:226-230, for a power-capped part: a time alone does not say
 * whether a slow run was a slow kernel or a slow clock).
 * wg_debug_clock_begin / _end bracket whatever is enqueued between them on the context's stream with two stamp kernels (s_memtime -- shader
 * clock ticks -- against s_memrealtime -- 100 MHz --, per XCD): _end synchronises the stream and returns the MEAN shader clock over the
 * bracketed interval in GHz (mean / min / max over the XCDs; any of the last three pointers may be NULL) and the interval's length. Nothing
 * is added to the bracketed kernels and nothing runs beside them.
 * wg_debug_mfma_ceiling runs v_mfma_f32_16x16x32_f16 alone (random operands in registers, no LDS or memory traffic, one 4-wave workgroup per
 * CU of the context's stream) for at least min_seconds (0 < s <= 30) and returns its TFLOP/s and mean shader clock: what the package power cap
 * leaves the matrix cores when nothing else draws power -- the ceiling of any f16 Gemm on this part. */
int wg_debug_clock_begin(wg_ctx *ctx);
int wg_debug_clock_end(wg_ctx *ctx, double *ghz_mean, double *ghz_min, double *ghz_max, double *seconds);
int wg_debug_mfma_ceiling(wg_ctx *ctx, double min_seconds, double *tflops, double *clock_ghz);

/*
 * Kernel-selection knobs of a context (tests and experiments; production code never needs them). The launchers choose between
 * kernel families by shape; a knob forces one choice for every later call on this context. Defaults come from the environment
 * ONCE, when the context is created (WG_F16_TILE, WG_F16_SCHED, WG_F32_SKINNY, WG_F32_PANELS, WG_F16_BALANCE): the dispatch path itself never
 * reads the environment.
 */
typedef enum wg_tuning {
    WG_TUNE_F16_TILE = 0,    /* 0 = by shape (default), 128 / 256 = force the 128 x 128 / 256 x 256 f16 kernel family, 256128 = the 256 x 128 tile (two
                                workgroups per CU: short K) whenever its kernel takes the shape */
    WG_TUNE_F16_SCHED = 1,   /* -1 = by size (default), 0 / 1 = static tile map / tile queues with stealing across XCDs */
    WG_TUNE_F32_SKINNY = 2,  /* -1 = by shape, 0 / 1 = never / whenever applicable: the few-column f32 kernel */
    WG_TUNE_F32_PANELS = 3,  /* -1 = by estimate, 0 / 1: 64-column panels of the few-column kernel for small square f32 products */
    WG_TUNE_F16_BALANCE = 4, /* calibrated per-XCD shares (K-prefix units) for f16 products of few rounds: 0 = off (default: measured not to pay on
                                MI355X, gemm_f16.hip), -1 = measure slot rates and let the planner decide, 1 = the tests' fixed pattern */
    WG_TUNE_F32_MID = 5,     /* the mid-size f32 tile family (gemm_f32_mid.hip: 128 x 128, 128 x 64, 64 x 128 tiles on 2 x 2 waves; 96 x 96, 96 x 64, 64 x 96, 64 x 64, 64 x 32,
                                32 x 64 with K split over the workgroup's waves; the whole K in one workgroup unless WG_TUNE_F32_MID_SPLIT / its estimate cuts K across workgroups
                                (few tiles, long K: slabs + a reduce launch); for outputs of <= 64 rows or columns it is tried BEFORE the few-column kernels unless
                                WG_TUNE_F32_SKINNY or _PANELS is forced to 1): -1 = by estimate (default), 0 = never, 1 = whenever
                                applicable with the estimate's tile, 128128 / 128064 / 64128 / 96096 / 96064 / 64096 / 64064 / 64032 / 32064 = that tile (tests) */
    WG_TUNE_F32_MID_SPLIT = 6, /* K cut of the mid family's k-split tiles across workgroups (few tiles, long K: 64 x 4096 x 4096): 0 = by estimate (default), n >= 2 = n
                                 splits whenever that family runs (tests) */
    WG_TUNE_GEMVT_LDS = 7,   /* GemvTr with 2 .. 8 right-hand sides on the vectors-in-LDS kernel (gemv.hip gemv_t_lds_kernel): 0 = where it measured ahead (default: from
                                128 outputs per CU on; two right-hand sides from 8), n >= 1 = from n outputs per CU on whatever the count of right-hand sides,
                                vectors longer than the LDS in up to 4 chunks (tests: every workgroup shape of the kernel) */
    WG_TUNE_F16_CONT = 8,    /* f16 Gemm / GemmTr on the continuous tile walk (gemm_f16.hip m16_cont: one workgroup per CU keeps its LDS-DMA stream going across its tiles):
                                -1 = by shape (default: K <= 4096, or <= 8192 below 16 rounds of tiles; more than one round of whole tiles), 0 = never,
                                1 = whenever applicable (tests) */
    WG_TUNE_RM_TR_NATIVE = 9, /* row-major GemmTr (wg_gemm_rm) on the kernels that take the second operand contiguous along N (gemm_f16_nt.hip, the B_NC instances of gemm_f16_t128.hip / gemm_f32.hip) instead of
                                transposing m1 into a scratch buffer first: -1 = from about half a round of tiles on (default: f16 128 x 128 tiles, f32 one round of 256 x 128 tiles), 0 = never
                                (the transposed copy: tests compare the two), 1 = whenever those kernels take the shape */
    WG_TUNE_COUNT_ = 10
} wg_tuning;
int wg_ctx_set_tuning(wg_ctx *ctx, wg_tuning key, int value);
/* Diagnostics / tests (no device needed): the f16 Gemm's calibrated-shares plan for `tiles` whole 256 x 256 tiles of `stages` stages (64 k
 * each) from eight relative slot rates (time per stage of the workgroup slots b % 8, mean 1), or the fixed test pattern (forced != 0),
 * decoded for every workgroup exactly as the kernel decodes it: units[5 i ..] = (tile, mode 0 whole / 1 prefix / 2 suffix, first stage,
 * stages, pair) of the i-th workgroup that has work; *nunits = how many there are (may exceed `capacity`), *nworkgroups = the grid. */
/* The per-slot rates the context has measured so far (time per stage of the workgroup slots b % 8 relative to their mean; all 1 until
 * `*valid`), how many snapshots went into them, and how many launches ran with calibrated shares. */
int wg_ctx_f16_balance_info(const wg_ctx *ctx, double *rel8, int *valid, uint32_t *updates, uint32_t *balanced_launches);
int wg_debug_f16_balance_plan(const double *rel8, uint32_t tiles, uint32_t stages, int forced, uint32_t *units, uint32_t capacity, uint32_t *nunits,
                              uint32_t *nworkgroups);
int wg_ctx_get_tuning(const wg_ctx *ctx, wg_tuning key, int *value);

/* Pre-size the context's scratch (GEMV split-K partials) so that no operator allocates while recording. An operator that would
 * have to grow a scratch region inside a recording returns WG_ERR_WORKSPACE. Scratch regions that a live command buffer may
 * replay into are never freed before that command buffer is destroyed. */
int wg_ctx_reserve_workspace(wg_ctx *ctx, size_t bytes);

/* ------------------------------------------------------------------------------------------------ */
/* buffers  (TensorBuilder::build / build_init / build_bytes, tensor.rs:115-186)                     */
/* ------------------------------------------------------------------------------------------------ */
/* Uninitialised buffer of `bytes` bytes (0 is legal: an empty tensor). */
int wg_buf_create(wg_ctx *ctx, size_t bytes, uint32_t usage, wg_buf **out);
/* create_buffer_init: allocate and upload synchronously (tensor.rs:149-161). */
int wg_buf_create_init(wg_ctx *ctx, const void *data, size_t bytes, uint32_t usage, wg_buf **out);
/* Non-owning wrapper around device memory someone else allocated (e.g. a torch tensor's data_ptr()). */
int wg_buf_wrap(wg_ctx *ctx, void *device_ptr, size_t bytes, wg_buf **out);
int wg_buf_destroy(wg_buf *buf); /* Drop for GpuTensor / Buffer */
size_t wg_buf_size(const wg_buf *buf);
void *wg_buf_device_ptr(const wg_buf *buf);
/* Queue::write_buffer: stream-ordered host->device copy of `bytes` at byte `offset`. Host memory may be pageable. */
int wg_buf_write(wg_ctx *ctx, wg_buf *dst, size_t offset, const void *data, size_t bytes);
/* GpuTensor::read / read_to (tensor.rs:300-384): device->host, BLOCKS until the data is in `dst`. */
int wg_buf_read(wg_ctx *ctx, const wg_buf *src, size_t offset, void *dst, size_t bytes);
/* CommandEncoder::copy_buffer_to_buffer (tensor.rs:227-264): stream-ordered device copy. */
int wg_buf_copy(wg_ctx *ctx, const wg_buf *src, size_t src_offset, wg_buf *dst, size_t dst_offset, size_t bytes);
int wg_buf_fill_zero(wg_ctx *ctx, wg_buf *buf);

/* ------------------------------------------------------------------------------------------------ */
/* operators                                                                                         */
/* ------------------------------------------------------------------------------------------------ */
/*
 * Gemm::dispatch_generic (wgebra gemm.rs:65-127): out = m1 * m2 (WG_GEMM, WG_GEMM_FAST) or m1^T * m2
 * (WG_GEMM_TR, WG_GEMM_TR_FAST), batched over size[2]; `out` is overwritten.
 *   DIM_MISMATCH  : the five assert_eq! of gemm.rs:91-95.
 *   Views that are not vec4-aligned (rows / stride / stride_mat / offset of a view, or M, N, K, not a multiple of 4 -- what
 *                   GpuMatrix::slice / rows / column hand out for odd offsets and lengths, tensor.rs:574-626): the reference's
 *                   kernels bind array<vec4<f32>> and address the wrong elements there (shape.wgsl:64-66). Here they compute
 *                   op(m1) m2 like any other view. Any offset / stride runs on the tuned kernels as it is (16-byte accesses and LDS-DMA take
 *                   element-aligned addresses on this target); lengths that are not multiples of 4 (of M and K; N too for f32 products with up to 128 rows -- otherwise any N runs as it is)
 *                   are staged into dense zero-padded copies of the operands that carry them (HBM-bound passes in a context scratch that cannot grow
 *                   inside a recording: WG_ERR_WORKSPACE); 1 .. 7 columns on otherwise aligned views run as a Gemv with that many right-hand
 *                   sides, without any copy. Only a view that exceeds its buffer is an error.
 *   *_FAST        : the reference requires K % 256 == 0 and reads out of bounds otherwise (gemm.wgsl:40,162);
 *                   here every K % 4 == 0 is accepted and all four variants run the same tuned kernel.
 * dtype WG_F16 (extension): f16 operands, f32 accumulation, result rounded once (RNE) to f16.
 */
int wg_gemm(wg_ctx *ctx, wg_gemm_variant variant, wg_dtype dtype,
            wg_buf *out, wg_view_shape out_shape,
            const wg_buf *m1, wg_view_shape m1_shape,
            const wg_buf *m2, wg_view_shape m2_shape);

/*
 * Extension: BLAS-style update  out = alpha * op(m1) * m2 + beta * out.  Same views, checks and variants as
 * wg_gemm; beta == 0 never reads `out` (NaN/Inf there are overwritten, like wg_gemm), and (alpha, beta) = (1, 0) is
 * bit-identical to wg_gemm. alpha, beta are f32 for both dtypes; f16: alpha*acc + beta*c is formed in f32 and rounded once.
 */
int wg_gemm_ex(wg_ctx *ctx, wg_gemm_variant variant, wg_dtype dtype, float alpha, float beta,
               wg_buf *out, wg_view_shape out_shape,
               const wg_buf *m1, wg_view_shape m1_shape,
               const wg_buf *m2, wg_view_shape m2_shape);

/*
 * Gemv::dispatch_generic (wgebra gemv.rs:64-137): out[:,y,z] = m[:,:,z] * v[:,y,z] (or m^T), for every RHS
 * column y < out.size[1] and matrix z < out.size[2]; `out` is overwritten.
 *   DIM_MISMATCH  : gemv.rs:89-90 (only m_cols == v_rows and m_rows == out_rows are checked there too).
 *   PRECONDITION  : WG_GEMV_FAST / WG_GEMV_TR_FAST with out rows % 4 != 0 (assert_eq! gemv.rs:122). Views that are not
 *                   vec4-aligned run in ONE pass on the matrix where it lies (gemv_any.hip: 16-byte loads at element-aligned addresses).
 *   WG_GEMV_TR_FAST with m rows % 128 != 0 silently runs as WG_GEMV_TR (gemv.rs:99-104) -- same kernel here.
 * dtype WG_F16 (extension): f16 elements, f32 accumulation, one rounding at the store -- the same HBM-bound kernels.
 * Several right-hand sides: one pass over the matrix for all of them; from 9 on -- and from 3 on when the matrix is past the launch-bound
 * sizes -- that pass runs on the Gemm kernels (same contract: f32 accumulation, results within the Gemv tolerance, deterministic).
 */
int wg_gemv(wg_ctx *ctx, wg_gemv_variant variant, wg_dtype dtype,
            wg_buf *out, wg_view_shape out_shape,
            const wg_buf *m, wg_view_shape m_shape,
            const wg_buf *v, wg_view_shape v_shape);

/*
 * ROW_MAJOR operator surface. Replaces composing the reference's shaders with
 * `row_major_shader_defs()` (wgebra linalg/shape.rs:11-15): every matrix view of the call is ROW-major,
 * index = t*stride_mat + offset + i*stride + j (shape.wgsl:49-52), `stride` = elements between consecutive ROWS;
 * the vec4 precondition becomes cols, stride, stride_mat, offset % 4 == 0 (shape.wgsl:54-56). Same dimension checks
 * and messages as wg_gemm / wg_gemv. A row-major view is the column-major view of the transpose, so WG_GEMM runs the
 * column-major kernel with the operands swapped (no copy); WG_GEMM_TR needs m1 transposed in memory first (one HBM-bound
 * pass into a context-owned scratch buffer). wg_gemv_rm with several right-hand-side columns (ncols % 4 == 0, the row-major
 * vec4 precondition) runs as the row-major Gemm / GemmTr it is.
 */
int wg_gemm_rm(wg_ctx *ctx, wg_gemm_variant variant, wg_dtype dtype,
               wg_buf *out, wg_view_shape out_shape,
               const wg_buf *m1, wg_view_shape m1_shape,
               const wg_buf *m2, wg_view_shape m2_shape);
int wg_gemv_rm(wg_ctx *ctx, wg_gemv_variant variant, wg_dtype dtype,
               wg_buf *out, wg_view_shape out_shape,
               const wg_buf *m, wg_view_shape m_shape,
               const wg_buf *v, wg_view_shape v_shape);

/*
 * Reduce::dispatch (wgebra reduce.rs:100-113): result[0] = reduce(op, value[offset .. offset+size[0])).
 * `result` is the GpuScalar's buffer (>= 4 bytes). The summation ORDER is the reference's (128 strided lanes,
 * then the 64..1 tree; reduce.wgsl:68-87), so Min/Max/Sum/Prod are bit-identical to it; n == 0 gives the init
 * value (0, 1, +3.4e38, -3.4e38).
 * dtype WG_F16 (extension): `value` and `result` are f16; elements are converted to f32 (exact), folded in the same order in f32,
 * and the result is rounded once (RNE) to f16. (wg_reduce_batched and wg_reduce_fast likewise.)
 */
int wg_reduce(wg_ctx *ctx, wg_reduce_op op, wg_dtype dtype,
              const wg_buf *value, wg_view_shape value_shape, wg_buf *result);

/*
 * Extension: result[0] = reduce(op, op(m) * v) in one call (e.g. SqNorm for |m v|^2, Max for the largest entry).
 * Launch-bound sizes (Gemv, rows * cols <= 4 Mi, rows >= 128) run as ONE kernel: the last workgroup to finish folds the product
 * vector in the reference order (reduce.wgsl:68-87); everything else is Gemv into a context-owned scratch vector, then Reduce, on the
 * same stream. Either way bit-identical to wg_gemv followed by wg_reduce; `m` is one matrix, `v` one vector.
 */
int wg_gemv_reduce(wg_ctx *ctx, wg_gemv_variant variant, wg_reduce_op op, wg_dtype dtype, wg_buf *result,
                   const wg_buf *m, wg_view_shape m_shape, const wg_buf *v, wg_view_shape v_shape);

/*
 * Extension: two-pass, multi-workgroup reduce of ONE long vector at HBM speed. Same arguments and
 * checks as wg_reduce, but NOT the reference's summation order (which serialises a vector onto one workgroup): Min/Max
 * are bit-identical to wg_reduce, Sum/Prod/SqNorm are re-associated -- deterministic (fixed chunking and tree, no
 * atomics) and within n * 2^-24 * sum|x| (sum x^2 for SqNorm) of the reference order.
 */
int wg_reduce_fast(wg_ctx *ctx, wg_reduce_op op, wg_dtype dtype,
                   const wg_buf *value, wg_view_shape value_shape, wg_buf *result);

/*
 * Extension (BASELINE config 4): one launch for many vectors. Column c of matrix t of the
 * column-major view is reduced exactly as wg_reduce would reduce the vector view at
 * offset + c*stride + t*stride_mat; results[c + t*size[1]] (one element of `dtype` each).
 */
int wg_reduce_batched(wg_ctx *ctx, wg_reduce_op op, wg_dtype dtype,
                      const wg_buf *values, wg_view_shape values_shape, wg_buf *results);

/*
 * OpAssign::dispatch (wgebra op_assign.rs:71-95): a[i] = a[i] (op) b[i], i < a.size[0]; WG_OP_COPY: a[i] = b[i].
 * Scalar indexing offset+i (shape.wgsl:36-38). DIM_MISMATCH: a.size[0] != b.size[0] (op_assign.rs:82-86).
 * IEEE-correct + - * / : bit-identical to the reference's CPU check.
 */
int wg_op_assign(wg_ctx *ctx, wg_op_assign_variant op, wg_dtype dtype,
                 wg_buf *a, wg_view_shape a_shape, const wg_buf *b, wg_view_shape b_shape);

/*
 * Extension (the north-star's "Axpy" -- the reference has no such operator, only OpAssign):
 * y[i] = fma(alpha, x[i], y[i]), i < y.size[0], one rounding per element. alpha = +1 / -1 reproduce WG_OP_ADD / WG_OP_SUB
 * (as `y += x` / `y -= x`) bit for bit. Same indexing, errors and skips as wg_op_assign. f16: computed in f32, rounded once.
 */
int wg_axpy(wg_ctx *ctx, float alpha, wg_dtype dtype, wg_buf *y, wg_view_shape y_shape, const wg_buf *x, wg_view_shape x_shape);

/*
 * Extension (strided / offset views): dst view = src view where the source has elements, 0 where it has none
 * (dst.size[0] x dst.size[1] per matrix; rows / columns of `src` beyond dst's are dropped). Any offset, stride and length on either side --
 * 16-byte accesses with a byte shift whatever the alignment (transpose.hip). It is the pass wg_gemm* / wg_gemv* run for operands that are
 * not vec4-aligned; a caller that multiplies the same odd view many times makes the aligned copy ONCE with this and passes that instead.
 * DIM_MISMATCH: dst.size[2] != src.size[2]. Zero-sized views: skipped. The views must not overlap.
 */
int wg_copy_view(wg_ctx *ctx, wg_dtype dtype, wg_buf *dst, wg_view_shape dst_shape, const wg_buf *src, wg_view_shape src_shape);

/* ------------------------------------------------------------------------------------------------ */
/* multi-GPU: the M-sharded Gemm of the north star. The reference has one wgpu::Device + Queue       */
/* (wgcore gpu.rs:7-12) and nothing to replace here; what is kept is its tensor model: every rank's   */
/* operands and the gathered result are plain column-major tensors addressed by ViewShape, and each   */
/* rank's local product is Gemm::dispatch_generic (gemm.rs:65-127) on views of them.                  */
/* ------------------------------------------------------------------------------------------------ */
typedef struct wg_comm wg_comm; /* one rank of a group of contexts (one context = one GPU); not thread-safe, like its context */
#define WG_COMM_ID_BYTES 128   /* == sizeof(ncclUniqueId) */
#define WG_IPC_HANDLE_BYTES 96
typedef enum wg_gather_mode {
    WG_GATHER_RCCL = 0,      /* per N-panel: Gemm into a staging cube [M/P, np, P], in-place ncclAllGather (RCCL over xGMI) on the
                                communicator's stream beside the next panel's Gemm, then an HBM-bound relayout into columns of C */
    /* 1 was WG_GATHER_PEER_COPY (Gemm straight into C + hsa_amd_memory_async_copy_rect pushes by a helper thread): removed in ABI 3 -- one
       rect-capable SDMA queue per direction made it a 2-rank engine at best, and it could hang two processes sharing a GPU at 32768^3 */
    WG_GATHER_NONE = 2,      /* this rank's rows of C only */
    WG_GATHER_PEER_STAGED = 3 /* per N-panel: Gemm into slot g of this rank's staging cube, one CONTIGUOUS copy per peer (the runtime's
                                peer-to-peer path: that link's own SDMA engine, no compute units) into the same slot of the peer's cube + a
                                sequence-number flag; the receiver's stream waits on the flags (one-wave kernel) and relayouts the panel into
                                C. Stream-ordered end to end, no barrier: the cubes are double-buffered by step parity. Needs
                                wg_comm_stage_reserve + wg_comm_set_peer_stages. The copy-engine exchange for any P.
                                The HIP runtime folds a process's streams onto 4 hardware queues unless GPU_MAX_HW_QUEUES says otherwise: give
                                a rank's process >= 3 + P of them, or its copies queue up behind its Gemms instead of running beside them
                                (correct either way; two ranks driven from ONE thread on one device need it for progress) */
} wg_gather_mode;

/* ncclGetUniqueId: call on one rank, ship the WG_COMM_ID_BYTES bytes to the others out of band (env, file, MPI, a torch store). */
int wg_comm_unique_id(void *id);
/* Rank `rank` of `nranks`, bound to `ctx` (its device, its stream). id != NULL: ncclCommInitRank (collective: every rank must call).
 * id == NULL: no collective library -- staged peer copies only, the caller brings its own barrier. librccl is bound at run time;
 * the library itself links only the HIP runtime. */
int wg_comm_create(wg_ctx *ctx, int nranks, int rank, const void *id, wg_comm **out);
int wg_comm_destroy(wg_comm *comm);
int wg_comm_rank(const wg_comm *comm);
int wg_comm_size(const wg_comm *comm);
int wg_comm_has_collectives(const wg_comm *comm);
int wg_comm_reported_size(const wg_comm *comm, int *count); /* ncclCommCount: the rank count the collective library itself reports (0 without one) */
uint64_t wg_comm_bytes_sent(const wg_comm *comm);     /* payload bytes this rank contributed / pushed so far */
/* In-place all-gather of elements [first, first + nranks*per_rank) of `buf` (rank r owns [first + r*per_rank, +per_rank)) on the
 * communicator's stream, ordered after the work already enqueued on the context; the context does not wait (wg_comm_join). */
int wg_all_gather(wg_comm *comm, wg_dtype dtype, wg_buf *buf, uint64_t first_elem, uint64_t elems_per_rank);
int wg_comm_join(wg_comm *comm);    /* the context's stream waits for the collectives in flight (and completes a deferred last panel) */
/* Pipelined steps (WG_GATHER_PEER_STAGED, and WG_GATHER_RCCL in its one-launch form): with on != 0 a wg_gemm_sharded call leaves the wait + relayout of its LAST panel -- the one
 * exchange nothing of its own call can hide -- to the next call on the communicator, which runs it right after enqueueing its first Gemm
 * (wg_comm_join / _flush / _barrier / a call in another mode complete it too). `out` is then complete in stream order only after that. */
/* One launch per step: an f16 wg_gemm_sharded of at least one round of 256 x 256 tiles (WG_GATHER_RCCL, WG_GATHER_PEER_STAGED, panels of whole
 * tiles) can run the rank's product as ONE kernel over all N-panels; the kernel writes each panel through to memory, its waves count
 * themselves into a per-panel word, the exchange of a panel waits for the full count (hipStreamWaitValue32) and the relayouts follow the
 * kernel. Bit for bit the result of the panel-by-panel launches (whenever those are not split along K). on = 1 / 0 force it / the panel
 * launches, on < 0 (default) decides by engine: RCCL on -- the scheduler-driven launch does not care how many CUs its stream has, so
 * RCCL's copy kernels need only 8 of them instead of 32 --, staged off (measured 1-2 % slower than 16 synchronised one-round launches). */
int wg_comm_set_one_launch(wg_comm *comm, int on);
/* Diagnostics: with on != 0 every later wg_gemm_sharded call stamps the context's stream right before and right after each "wait for panel p's
 * exchange" (a pair of timing events; a few microseconds of host time per panel -- leave it off in timed regions). wg_comm_wait_times synchronises
 * the context, returns (panel, milliseconds the compute stream stood still) for up to `capacity` waits since the last call, oldest first, and
 * starts over. An exchange that hid under the Gemms reads ~0; a slow link shows up as the waits of a step's FIRST relayouts, a slow rank as waits
 * on every panel of its peers, the un-hidden tail as the last panel's wait. */
int wg_comm_set_wait_timing(wg_comm *comm, int on);
int wg_comm_wait_times(wg_comm *comm, uint32_t *panels, float *ms, uint32_t capacity, uint32_t *count);
int wg_comm_set_pipelined(wg_comm *comm, int on); /* (a step with ONE panel always completes in its call: deferring it would let a rank run two steps ahead of a peer) */
int wg_comm_flush(wg_comm *comm);   /* host-blocking: every peer copy this rank issued has landed */
int wg_comm_barrier(wg_comm *comm); /* flush + a one-element all-reduce joined into the context: all ranks' earlier exchanges are complete */
/* One process per GPU: export a device buffer / map a peer's (hipIpcGetMemHandle / hipIpcOpenMemHandle; needs the dmabuf IPC mode,
 * HSA_ENABLE_IPC_MODE_LEGACY=0, on this platform). The mapped buffer is released by wg_buf_destroy. */
int wg_buf_ipc_export(const wg_buf *buf, void *handle /* WG_IPC_HANDLE_BYTES */);
int wg_buf_ipc_open(wg_ctx *ctx, const void *handle, wg_buf **out);
/* WG_GATHER_PEER_STAGED: make the communicator's staging cubes (>= 2 * sizeof(T) * M * N bytes: two steps in flight) and its flag array
 * exist and return them (owned by the communicator) so that the caller can export them (wg_buf_ipc_export) to the peers; then register
 * every peer's pair as addressable from here (wg_buf_ipc_open, or the buffers themselves when the ranks share a process). Growing the
 * cubes invalidates the registration on every rank. EVERY RANK MUST RESERVE THE SAME SIZE: the two step-parity halves sit at offsets 0
 * and bytes / 2 whatever the shape of a step (wg_comm_set_peer_stages checks it), which is what lets M / N / panel_cols change between
 * steps without a barrier. A receiver waits WG_COMM_TIMEOUT_MS (environment, read when the communicator is created; default 30000) for
 * a peer's slot; after that the panel's columns of `out` are filled with NaN bit patterns and the error is returned by the next
 * wg_ctx_sync / wg_buf_read on the context, wg_comm_flush / _join / _barrier, or wg_gemm_sharded call -- never a hung queue, never stale data. */
int wg_comm_stage_reserve(wg_comm *comm, size_t bytes, wg_buf **stage, wg_buf **flags);
int wg_comm_set_peer_stages(wg_comm *comm, wg_buf *const *peer_stage, wg_buf *const *peer_flags);
/* Relayout of a gathered GpuCube [M/P, np, P] (dense) into the (M x np) column-major view `out`: out[g*M/P + i, j] = cube[i, j, g].
 * HBM-bound: 2 * sizeof(T) * M * np bytes. (What WG_GATHER_RCCL runs per panel; exported for callers that gather themselves.) */
int wg_cube_to_matrix(wg_ctx *ctx, wg_dtype dtype, const wg_buf *cube, wg_view_shape cube_shape, wg_buf *out, wg_view_shape out_shape);
/*
 * out (M x N, one matrix, on EVERY rank) = op(A) * B with A sharded on M: `a_rows` is this rank's row block of op(A)
 * (M/P x K; WG_GEMM_TR*: stored K x M/P), `b` (K x N) is replicated. N is cut into panels of `panel_cols` columns (0 = default)
 * and panel i's exchange overlaps panel i+1's Gemm.
 * On return everything is enqueued: `out` is complete in context-stream order (WG_GATHER_NONE: this rank's rows only).
 * DIM_MISMATCH as Gemm (gemm.rs:91-95) with M = P * rows(a_rows).
 */
int wg_gemm_sharded(wg_comm *comm, wg_gemm_variant variant, wg_dtype dtype, wg_gather_mode mode, uint32_t panel_cols,
                    wg_buf *out, wg_view_shape out_shape,
                    const wg_buf *a_rows, wg_view_shape a_shape, const wg_buf *b, wg_view_shape b_shape);
/* The same with the N-panels' widths given one by one (`npanels` column counts, multiples of 4 summing to N): a TAPERED TAIL -- equal panels, then a
 * few narrower and narrower ones -- leaves only a narrow last panel's exchange exposed at the end of a step, and every panel's exchange still hides
 * under the next panel's Gemm as long as a panel is at least (exchange time / Gemm time) of the one before it. The one-launch forms
 * (wg_comm_set_one_launch) take lists of the shape "n equal panels of whole 256-column tiles, then 1 .. 8 other panels of whole tiles, the last one
 * whatever is left"; any other list runs panel by panel. Results are bit for bit those of wg_gemm_sharded.
 * PRECONDITION the library cannot check: EVERY rank passes the same widths (and the same panel_cols to wg_gemm_sharded) -- the list is the slot layout of the
 * staging cubes the peers copy into. Ranks with different lists would write each other's slots at the wrong offsets, silently; compare the plans across ranks once
 * before the first step (bench.py does, through its control plane). */
int wg_gemm_sharded_panels(wg_comm *comm, wg_gemm_variant variant, wg_dtype dtype, wg_gather_mode mode, const uint32_t *panel_widths, uint32_t npanels,
                           wg_buf *out, wg_view_shape out_shape,
                           const wg_buf *a_rows, wg_view_shape a_shape, const wg_buf *b, wg_view_shape b_shape);

/* ------------------------------------------------------------------------------------------------ */
/* record / replay: CommandEncoder -> finish() -> CommandBuffer -> Queue::submit, as a hipGraph      */
/* ------------------------------------------------------------------------------------------------ */
/* Start recording: operator and copy calls are captured instead of executed (no sync/read/create while recording). */
int wg_encoder_begin(wg_ctx *ctx);
/* encoder.finish(): stop recording and instantiate the replayable command buffer. */
int wg_encoder_finish(wg_ctx *ctx, wg_cmdbuf **out);
/* queue.submit(Some(cmdbuf)): enqueue one replay on the context's stream. May be submitted many times. */
int wg_queue_submit(wg_ctx *ctx, wg_cmdbuf *cmdbuf);
int wg_cmdbuf_destroy(wg_cmdbuf *cmdbuf);

/* ------------------------------------------------------------------------------------------------ */
/* GpuTimestamps (wgcore timestamps.rs): begin/end-of-pass timestamps on the context's stream        */
/* ------------------------------------------------------------------------------------------------ */
int wg_timestamps_create(wg_ctx *ctx, uint32_t capacity, wg_timestamps **out); /* GpuTimestamps::new */
int wg_timestamps_destroy(wg_timestamps *ts);
int wg_timestamps_clear(wg_timestamps *ts);
/* Record the next timestamp at the current point of the stream; *index (optional) receives its slot. (write_next_timestamp, timestamps.rs:98-102) */
int wg_timestamps_write(wg_ctx *ctx, wg_timestamps *ts, uint32_t *index);
/* next_query_indices::<COUNT> (timestamps.rs:80-94): `count` consecutive slots, all or none -- *first = UINT32_MAX when they do not fit (the reference's None; no error). */
int wg_timestamps_reserve(wg_timestamps *ts, uint32_t count, uint32_t *first);
/* write_timestamp_at (timestamps.rs:108-115): the timestamp of slot `index` (< capacity) at the current point of the stream. A reserved slot never written reads 0. */
int wg_timestamps_write_at(wg_ctx *ctx, wg_timestamps *ts, uint32_t index);
uint32_t wg_timestamps_len(const wg_timestamps *ts);
/* wait_for_results_ms (timestamps.rs:226-230): block, then out_ms[i] = time of slot i relative to the first written slot (slot 0 in the begin / end use). */
int wg_timestamps_wait_for_results_ms(wg_timestamps *ts, double *out_ms, uint32_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* WGEBRA_HIP_H */
