// Reduce: result = reduce(op, x[0..n))   (wgebra reduce.wgsl:68-96, op tables reduce.rs:30-58).
//
// The reference's order is fully specified: 128 lanes, lane t folds x[t], x[t+128], ... ascending into
// ws[t] (starting from `init`), then a tree ws[t] = f(ws[t], ws[t+s]) for s = 64,32,...,1.  This kernel keeps
// exactly that expression tree per vector -- so Min/Max/Sum/Prod are bit-identical to the reference order
// (SqNorm too, with the product rounded separately: no FMA contraction) -- but maps it onto CDNA4 differently:
//
//   * the 128 virtual lanes of one vector live in 32 physical lanes x float4 (virtual lane t = 4*p + c): every
//     load is a 16-byte-per-lane, 512-byte-contiguous row of the vector, kUnroll rows in flight per lane;
//   * a wave64 therefore carries TWO vectors, a 256-thread workgroup EIGHT, and one launch covers every column
//     of a matrix view (the reference needs one single-workgroup dispatch per vector: reduce.rs:110-112);
//   * the tree's cross-lane steps (strides 64..4) are wave shuffles inside each 32-lane half, strides 2 and 1
//     are in-register; no LDS, no barrier.
// HBM-bound: 4 bytes per element read, 4 bytes per vector written.
#include "wg_internal.hpp"
#include "reduce_ops.hpp"

namespace {

#ifndef RED_UNROLL
#define RED_UNROLL 16 // rows (float4 loads) in flight per lane: 8 -> 6.21 TB/s, 16 -> 6.60, 32 -> 6.60 (4096 x 65536)
#endif
#ifndef RED_NT
#define RED_NT 1 // non-temporal loads: +13 % on this read-once stream
#endif
constexpr int kThreads = 256;
constexpr int kUnroll = RED_UNROLL;

template <typename T>
__device__ __forceinline__ const T *vector_base(const T *base, uint32_t q, uint32_t ncols, uint32_t stride,
                                                    uint32_t stride_mat) {
    const uint32_t c = q % ncols, t = q / ncols;
    return base + (uint64_t)c * stride + (uint64_t)t * stride_mat;
}

// Element types: f32 (the reference's), and f16 as this build's extension -- f16 elements are converted to f32 (exact), folded in the
// reference's order in f32, and the result is rounded once (RNE) to f16. Four consecutive elements of a row, as floats:
// AL = false: at any element-aligned address (round 6: vectors whose base is not a multiple of 4 elements took a 128-lane element-wise twin of the kernel below before --
// 2.6x the time on one long vector; unaligned-access mode, wg_internal.hpp wg_ld_u). The aligned instance keeps its own typed load: the backend drops the non-temporal
// hint from a load whose type is under-aligned, and the hint is worth 13 % on config 4's read-once stream (6.59 -> 5.69 TB/s when it went missing for an afternoon).
template <bool AL>
__device__ __forceinline__ float4 load4(const float *p, bool nt) {
    if constexpr (AL) return nt ? wg_ld_nt(reinterpret_cast<const float4 *>(p)) : *reinterpret_cast<const float4 *>(p);
    else return wg_ld_u(p);
}
template <bool AL>
__device__ __forceinline__ float4 load4(const _Float16 *p, bool nt) {
    typedef _Float16 h4a __attribute__((ext_vector_type(4)));
    typedef h4a __attribute__((aligned(2))) h4u;
    h4a v;
    if constexpr (AL) v = nt ? __builtin_nontemporal_load(reinterpret_cast<const h4a *>(p)) : *reinterpret_cast<const h4a *>(p);
    else v = *reinterpret_cast<const h4u *>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}

// 32 physical lanes x 4 elements per vector, at any element-aligned base.
template <int OP, typename T, bool AL>
__global__ __launch_bounds__(kThreads) void reduce_rows4(const T *__restrict__ base, uint32_t n, uint32_t ncols,
                                                         uint32_t nvec, uint32_t stride, uint32_t stride_mat,
                                                         T *__restrict__ results) {
    const uint32_t q = blockIdx.x * (kThreads / 32) + (threadIdx.x >> 5);
    const uint32_t p = threadIdx.x & 31;
    if (q >= nvec) return; // whole 32-lane half leaves together; shuffles below never cross halves
    const T *x = vector_base(base, q, ncols, stride, stride_mat);

    float acc[4] = { r_init<OP>(), r_init<OP>(), r_init<OP>(), r_init<OP>() };
    const uint32_t full_rows = n / 128u;
    uint32_t r = 0;
    for (; r + kUnroll <= full_rows; r += kUnroll) {
        float4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = load4<AL>(x + ((uint64_t)(r + u) * 32u + p) * 4u, RED_NT);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) { // rows in ascending order: the per-lane chain of reduce.wgsl:71-74
            acc[0] = r_ws<OP>(acc[0], v[u].x);
            acc[1] = r_ws<OP>(acc[1], v[u].y);
            acc[2] = r_ws<OP>(acc[2], v[u].z);
            acc[3] = r_ws<OP>(acc[3], v[u].w);
        }
    }
    for (; r < full_rows; ++r) {
        float4 v = load4<AL>(x + ((uint64_t)r * 32u + p) * 4u, false);
        acc[0] = r_ws<OP>(acc[0], v.x);
        acc[1] = r_ws<OP>(acc[1], v.y);
        acc[2] = r_ws<OP>(acc[2], v.z);
        acc[3] = r_ws<OP>(acc[3], v.w);
    }
    { // ragged last row: i = 128*full_rows + 4p + c < n
        const uint32_t i0 = full_rows * 128u + 4u * p;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (i0 + c < n) acc[c] = r_ws<OP>(acc[c], (float)x[i0 + c]);
    }
    // tree, virtual strides 64,32,16,8,4 == physical 16,8,4,2,1 (lanes >= stride compute garbage nobody reads)
#pragma unroll
    for (int s = 16; s >= 1; s >>= 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = r_red<OP>(acc[c], __shfl_down(acc[c], s, 32));
    }
    // virtual strides 2 and 1 live inside lane 0
    acc[0] = r_red<OP>(acc[0], acc[2]);
    acc[1] = r_red<OP>(acc[1], acc[3]);
    acc[0] = r_red<OP>(acc[0], acc[1]);
    if (p == 0) results[q] = (T)acc[0];
}

// ---------------------------------------------------------------------------------------------------------------------
// Long vectors, few of them (the reference's own use: ONE vector per dispatch): the per-lane chain order pins the arithmetic
// to one half-wave, and a half-wave alone keeps only ~8 KiB of loads in flight (measured 10 GB/s from HBM). Here all 8 waves
// of a 512-thread workgroup stream the vector into a ring of 8 LDS slots (16 KiB = 32 rows of 128 floats each) with LDS-DMA,
// 7 slots ahead, awaited with a counted vmcnt; lanes 0-31 of wave 0 then fold the rows IN ORDER out of LDS -- same expression
// tree, same bits, but the memory system sees 100+ KiB in flight from the CU: 9.5 -> 37 GB/s on a 256 MiB vector.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kLongThreads = 512;
constexpr int kLongSlots = 8;
constexpr int kLongSlotBytes = 16384;                   // 32 rows of 128 floats (64 rows of 128 f16) per slot
constexpr int kLongPieces = kLongSlotBytes / 1024 / 8;  // DMA pieces per wave per slot (= 2)

__device__ __forceinline__ void red_dma16(const void *gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst));
}

template <int OP, typename T>
__global__ __launch_bounds__(kLongThreads) void reduce_long(const T *__restrict__ base, uint32_t n, uint32_t ncols, uint32_t nvec,
                                                            uint32_t stride, uint32_t stride_mat, T *__restrict__ results) {
    constexpr int kLongSlotRows = kLongSlotBytes / (128 * (int)sizeof(T)), kPieceElems = 1024 / (int)sizeof(T), kLaneElems = 16 / (int)sizeof(T);
    __shared__ __attribute__((aligned(16))) char ring[kLongSlots * kLongSlotBytes];
    const uint32_t q = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const T *x = vector_base(base, q, ncols, stride, stride_mat);
    const uint32_t full_rows = n / 128u;
    const uint32_t nslots = full_rows / kLongSlotRows; // whole slots; the rest is folded straight from global memory below
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)ring;

    auto issue = [&](uint32_t slot) { // this wave's 2 KiB of vector slot `slot` -> ring position slot % 8
        const T *src = x + (uint64_t)slot * (kLongSlotRows * 128u) + (uint32_t)wave * (kLongPieces * kPieceElems) + (uint32_t)(kLaneElems * lane);
        const uint32_t dst = lds_base + (slot % kLongSlots) * kLongSlotBytes + wave * (kLongPieces * 1024);
#pragma unroll
        for (int p = 0; p < kLongPieces; ++p) red_dma16(src + p * kPieceElems, __builtin_amdgcn_readfirstlane(dst + p * 1024));
    };

    // Consumers: waves 0 and 1, ONE chain per lane (chain t = 64 wave + lane folds x[t], x[t + 128], ... ascending -- reduce.wgsl:71-74): a slot's 32
    // rows cost a lane 32 dependent operations and 16 two-row LDS reads. (Round 2 folded on 32 lanes x float4 of wave 0: four chains per lane, 37 GB/s,
    // bound by what one wave pulls out of LDS; this form: 56 GB/s, ~19 cycles per element of a chain -- the dependent operation itself is ~10.)
    const bool consumer = wave < 2;
    const uint32_t t = 64u * (uint32_t)wave + (uint32_t)lane; // the chain (consumers only)
    float acc = r_init<OP>();
    for (uint32_t s = 0; s < kLongSlots - 1 && s < nslots; ++s) issue(s);
    for (uint32_t s = 0; s < nslots; ++s) {
        // slot s has landed once at most the (up to 6) younger slots' pieces of this wave are still in flight
        const uint32_t younger = min(nslots - 1 - s, (uint32_t)kLongSlots - 2);
        switch (younger) { // vmcnt takes an immediate
        case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
        __syncthreads(); // everyone's pieces of slot s are in LDS; the consumers have left slot s-1
        if (s + kLongSlots - 1 < nslots) issue(s + kLongSlots - 1); // refill the position slot s-1 occupied
        if (consumer) {
            const T *rows = reinterpret_cast<const T *>(ring + (s % kLongSlots) * kLongSlotBytes) + t;
            float v[kLongSlotRows];
#pragma unroll
            for (int r = 0; r < kLongSlotRows; ++r) v[r] = (float)rows[r * 128];
#pragma unroll
            for (int r = 0; r < kLongSlotRows; ++r) acc = r_ws<OP>(acc, v[r]); // ascending rows
            // (reading slot s while folding slot s - 1 out of registers measured slower: 56 -> 50 GB/s)
        }
    }
    if (consumer) {
        for (uint32_t r = nslots * kLongSlotRows; r < full_rows; ++r) acc = r_ws<OP>(acc, (float)x[(uint64_t)r * 128u + t]); // the left-over full rows, less than a slot
        const uint32_t i0 = full_rows * 128u + t;
        if (i0 < n) acc = r_ws<OP>(acc, (float)x[i0]); // ragged last row
    }
    // the 64 .. 1 tree of reduce.wgsl:76-87 over the 128 chains: stride 64 crosses the two consumer waves (LDS), the rest is inside wave 0
    __shared__ float upper[64];
    if (wave == 1) upper[lane] = acc;
    __syncthreads();
    if (wave != 0) return;
    acc = r_red<OP>(acc, upper[lane]);
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) acc = r_red<OP>(acc, __shfl_down(acc, sft, 64));
    if (lane == 0) results[q] = (T)acc;
}

template <int OP, typename T>
int launch(wg_ctx *ctx, const T *base, uint32_t n, uint32_t ncols, uint32_t nmats, uint32_t stride,
           uint32_t stride_mat, T *results) {
    const uint64_t nvec64 = (uint64_t)ncols * nmats;
    if (nvec64 == 0) return WG_OK;
    if (nvec64 > 0x7fffffffull) return wg_set_error(WG_ERR_UNSUPPORTED, "Reduce: more than 2^31 vectors in one call");
    const uint32_t nvec = (uint32_t)nvec64;
    if (n >= 65536u && nvec <= 256u) { // long vectors, fewer than there are CUs: one 8-wave workgroup each (f16 too since round 6: 2^22 elements 900 -> ~300 us)
        hipLaunchKernelGGL((reduce_long<OP, T>), dim3(nvec), dim3(kLongThreads), 0, ctx->stream, base, n, ncols, nvec, stride, stride_mat, results);
        WG_HIP_TRY(hipGetLastError());
        return WG_OK;
    }
    const uint32_t per_block = kThreads / 32;
    const bool aligned = ((uintptr_t)base % (4 * sizeof(T)) == 0) && (nvec == 1 || ((stride % 4 == 0) && (nmats == 1 || stride_mat % 4 == 0)));
    if (aligned) hipLaunchKernelGGL((reduce_rows4<OP, T, true>), dim3((nvec + per_block - 1) / per_block), dim3(kThreads), 0, ctx->stream, base, n, ncols, nvec, stride, stride_mat, results);
    else hipLaunchKernelGGL((reduce_rows4<OP, T, false>), dim3((nvec + per_block - 1) / per_block), dim3(kThreads), 0, ctx->stream, base, n, ncols, nvec, stride, stride_mat, results);
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}

// ------------------------------------------------------------------------------------------------------
// Two-pass, multi-workgroup reduce of ONE long vector. NOT the reference's summation order: the 128
// strided chains of reduce.wgsl serialise a vector onto one workgroup (37 GB/s above); this variant lets the whole chip
// stream it. Min/Max are the same bits as the reference order (min/max of non-NaN floats is associative and commutative);
// Sum/Prod/SqNorm are re-associated -- deterministic (fixed chunking and tree, no atomics), within n * 2^-24 * sum|x| of the
// reference order. Pass 1: workgroup b folds the contiguous chunk b (float4 non-temporal loads, 8 in flight per
// lane, per-lane accumulators, wave butterfly, LDS across the 4 waves) -> partial[b]; pass 2: one workgroup folds the partials.
// ------------------------------------------------------------------------------------------------------
template <int OP>
__device__ __forceinline__ float fast_block_fold(float acc) { // fold the 256 per-thread values of a workgroup; valid in thread 0
    __shared__ float red[kThreads / 64];
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) acc = r_red<OP>(acc, __shfl_xor(acc, s, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int w = 1; w < kThreads / 64; ++w) r = r_red<OP>(r, red[w]);
    return r;
}

template <int OP, typename T>
__global__ __launch_bounds__(kThreads) void reduce_fast_pass1(const T *__restrict__ x, uint32_t n, uint32_t chunk, float *__restrict__ partial) {
    const uint64_t begin = (uint64_t)blockIdx.x * chunk; // chunk is a multiple of 4; x + begin keeps x's alignment class
    const uint32_t len = (uint32_t)min((uint64_t)chunk, (uint64_t)n - begin);
    const T *p = x + begin;
    float acc = r_init<OP>();
    // scalar head up to 4-element alignment (16 bytes f32, 8 bytes f16), 4-element body, scalar tail
    const uint32_t mis = (uint32_t)(((uintptr_t)p / sizeof(T)) & 3u);
    const uint32_t head = mis ? min(4u - mis, len) : 0u;
    if (threadIdx.x < head) acc = r_ws<OP>(acc, (float)p[threadIdx.x]);
    const T *p4 = p + head; // indexed in units of 4 elements below
    const uint32_t n4 = (len - head) / 4u;
    uint32_t i = threadIdx.x;
    for (; (uint64_t)i + 7u * kThreads < n4; i += 8u * kThreads) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = load4<true>(p4 + 4ull * (i + u * kThreads), true);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc = r_ws<OP>(acc, v[u].x); acc = r_ws<OP>(acc, v[u].y); acc = r_ws<OP>(acc, v[u].z); acc = r_ws<OP>(acc, v[u].w);
        }
    }
    for (; i < n4; i += kThreads) {
        const float4 v = load4<true>(p4 + 4ull * i, true);
        acc = r_ws<OP>(acc, v.x); acc = r_ws<OP>(acc, v.y); acc = r_ws<OP>(acc, v.z); acc = r_ws<OP>(acc, v.w);
    }
    const uint32_t tail0 = head + n4 * 4u;
    if (tail0 + threadIdx.x < len) acc = r_ws<OP>(acc, (float)p[tail0 + threadIdx.x]);
    const float r = fast_block_fold<OP>(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

template <int OP, typename T>
__global__ __launch_bounds__(kThreads) void reduce_fast_pass2(const float *__restrict__ partial, uint32_t nparts, T *__restrict__ result) {
    // partials are already reduce_fn-domain values (sums of squares for SqNorm): fold with reduce_fn; the neutral start is the op's init
    float acc = r_init<OP>();
    for (uint32_t i = threadIdx.x; i < nparts; i += kThreads) acc = r_red<OP>(acc, partial[i]);
    const float r = fast_block_fold<OP>(acc);
    if (threadIdx.x == 0) result[0] = (T)r;
}

template <int OP, typename T>
int launch_fast(wg_ctx *ctx, const T *x, uint32_t n, T *result) {
    const int cus = ctx->compute_units > 0 ? ctx->compute_units : 256;
    // >= 16 Ki elements per workgroup, at most 4 workgroups per CU
    const uint64_t want = ((uint64_t)n + 16383u) / 16384u, cap = (uint64_t)cus * 4u;
    uint32_t nparts = (uint32_t)(want < cap ? want : cap);
    if (nparts == 0) nparts = 1;
    uint32_t chunk = (uint32_t)((((uint64_t)n + nparts - 1) / nparts + 3u) & ~3ull);
    if (chunk == 0) chunk = 4;
    nparts = (uint32_t)(((uint64_t)n + chunk - 1) / chunk);
    if (nparts == 0) nparts = 1; // n == 0: one workgroup writes the init value, like the reference (reduce.wgsl with an empty loop)
    void *ws = nullptr;
    if (int rc = wg_ctx_workspace(ctx, (size_t)nparts * sizeof(float), &ws)) return rc;
    hipLaunchKernelGGL((reduce_fast_pass1<OP, T>), dim3(nparts), dim3(kThreads), 0, ctx->stream, x, n, chunk, (float *)ws);
    WG_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL((reduce_fast_pass2<OP, T>), dim3(1), dim3(kThreads), 0, ctx->stream, (const float *)ws, nparts, result);
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}

} // namespace

template <typename T>
static int reduce_fast_dispatch(wg_ctx *ctx, int op, const T *b, uint32_t n, T *result) {
    switch (op) {
    case R_MIN: return launch_fast<R_MIN>(ctx, b, n, result);
    case R_MAX: return launch_fast<R_MAX>(ctx, b, n, result);
    case R_SUM: return launch_fast<R_SUM>(ctx, b, n, result);
    case R_PROD: return launch_fast<R_PROD>(ctx, b, n, result);
    case R_SQNORM: return launch_fast<R_SQNORM>(ctx, b, n, result);
    }
    return wg_set_error(WG_ERR_INVALID_ARG, "Reduce: unknown op %d", op);
}

int wgk_reduce_fast(wg_ctx *ctx, int op, wg_dtype dtype, const void *base, uint32_t n, void *result) {
    if (dtype == WG_F16) return reduce_fast_dispatch(ctx, op, (const _Float16 *)base, n, (_Float16 *)result);
    return reduce_fast_dispatch(ctx, op, (const float *)base, n, (float *)result);
}

template <typename T>
static int reduce_dispatch(wg_ctx *ctx, int op, const T *b, uint32_t n, uint32_t ncols, uint32_t nmats, uint32_t stride, uint32_t stride_mat, T *results) {
    switch (op) {
    case R_MIN: return launch<R_MIN>(ctx, b, n, ncols, nmats, stride, stride_mat, results);
    case R_MAX: return launch<R_MAX>(ctx, b, n, ncols, nmats, stride, stride_mat, results);
    case R_SUM: return launch<R_SUM>(ctx, b, n, ncols, nmats, stride, stride_mat, results);
    case R_PROD: return launch<R_PROD>(ctx, b, n, ncols, nmats, stride, stride_mat, results);
    case R_SQNORM: return launch<R_SQNORM>(ctx, b, n, ncols, nmats, stride, stride_mat, results);
    }
    return wg_set_error(WG_ERR_INVALID_ARG, "Reduce: unknown op %d", op);
}

// `results` has the element type of the input (Reduce::dispatch(value: GpuVectorView<T>, result: &GpuScalar<T>), reduce.rs:100-107)
int wgk_reduce(wg_ctx *ctx, int op, wg_dtype dtype, const void *base, uint32_t n, uint32_t ncols, uint32_t nmats,
               uint32_t stride, uint32_t stride_mat, void *results) {
    // Min / Max of ONE long vector: min and max of non-NaN floats do not depend on the order (v_min_f32 / v_max_f32 order -0 below +0 whatever the
    // operand order), so the whole chip may stream it -- the two-pass kernels give the bits the reference's 128 chains give (1 Mi elements: 100 -> 7 us).
    // Sum / Prod / SqNorm keep the reference's chains (wg_reduce_fast is the caller's explicit choice there).
    if ((op == R_MIN || op == R_MAX) && (uint64_t)ncols * nmats == 1 && n >= 65536u) return wgk_reduce_fast(ctx, op, dtype, base, n, results);
    if (dtype == WG_F16) return reduce_dispatch(ctx, op, (const _Float16 *)base, n, ncols, nmats, stride, stride_mat, (_Float16 *)results);
    return reduce_dispatch(ctx, op, (const float *)base, n, ncols, nmats, stride, stride_mat, (float *)results);
}
