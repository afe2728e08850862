// OpAssign: a[i] = a[i] (op) b[i]   (wgebra op_assign.wgsl:40-47, op table op_assign.rs:28-38).
//
// HBM-bound streaming kernel: 12 bytes per f32 element (8 for Copy, which never reads `a`).
// Layout: both views are contiguous runs of n elements starting at (buffer + offset).
// The reference launches ceil(n/64) workgroups of 64 threads doing one 4-byte access each; here each lane moves 16 bytes
// (float4 / 8 x half) with non-temporal loads and stores, ONE access per lane over a flat grid of n/4/256 workgroups --
// measured 6.3 TB/s (Add) / 6.6 TB/s (Copy) on 2^28 elements vs 4.9 TB/s for a capped grid-stride loop with 4-8 accesses
// in flight per lane: in dispatch order the flat grid sweeps the three streams sequentially, which is what HBM wants.
// A scalar head/tail (first lanes of block 0) makes any offset/length legal.
// IEEE-correct + - * / (no fast-math, correctly rounded division): results are bit-identical to the CPU.
#include "wg_internal.hpp"

namespace {

enum { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2, OP_DIV = 3, OP_COPY = 4, OP_AXPY = 5 /* a = fma(alpha, b, a): wg_axpy */ };

template <int OP>
__device__ __forceinline__ float apply(float a, float b, float alpha) {
    if constexpr (OP == OP_AXPY) return fmaf(alpha, b, a);
    else if constexpr (OP == OP_ADD) return __fadd_rn(a, b);
    else if constexpr (OP == OP_SUB) return __fsub_rn(a, b);
    else if constexpr (OP == OP_MUL) return __fmul_rn(a, b);
    else if constexpr (OP == OP_DIV) return __fdiv_rn(a, b);
    else return b;
}

template <int OP>
__device__ __forceinline__ float4 apply4(float4 a, float4 b, float alpha) {
    return make_float4(apply<OP>(a.x, b.x, alpha), apply<OP>(a.y, b.y, alpha), apply<OP>(a.z, b.z, alpha), apply<OP>(a.w, b.w, alpha));
}

#ifndef OPA_UNROLL
#define OPA_UNROLL 1
#endif
#ifndef OPA_NT
#define OPA_NT 2 // 1: non-temporal loads, 2: + non-temporal stores (measured best: 4.9 TB/s on a 2-read/1-write stream)
#endif
#ifndef OPA_WG_PER_CU
#define OPA_WG_PER_CU 8
#endif
#ifndef OPA_MODE
#define OPA_MODE 2 // 2: one workgroup per kUnroll*256 consecutive float4, flat uncapped grid (measured 6.3 TB/s Add, 6.6 TB/s Copy);
                   // 0: capped grid + grid-stride loop (4.9 TB/s: the streams interleave badly across the chip)
#endif
constexpr int kThreads = 256;
constexpr int kUnroll = OPA_UNROLL;

__device__ __forceinline__ float4 ld4(const float4 *p) {
    if (OPA_NT >= 1) return wg_ld_nt(p);
    return *p;
}
__device__ __forceinline__ void st4(float4 *p, float4 v) {
    if (OPA_NT >= 2) __builtin_nontemporal_store(wg_f4{ v.x, v.y, v.z, v.w }, reinterpret_cast<wg_f4 *>(p));
    else *p = v;
}

// `a`/`b` point at the first element; [head, head + 4*n4) is the 16-byte aligned body, the <= 3 elements before
// and after it are done by the first lanes of block 0.
// BU: `b` is not 16-byte aligned where `a` is (different offsets into their buffers): its float4 are loaded from element-aligned addresses (round 6; such pairs took
// the element-wise kernel before: Copy of 2^26 floats 116 us against 81)
template <int OP, bool BU>
__global__ __launch_bounds__(kThreads) void op_assign_f32_vec(float *a0, const float *b0, uint32_t head, uint32_t n4, uint32_t n, float alpha) {
    if (blockIdx.x == 0) {
        const uint32_t body_end = head + 4u * n4;
        const uint32_t edge = head + (n - body_end);
        if (threadIdx.x < edge) {
            const uint32_t i = threadIdx.x < head ? threadIdx.x : body_end + (threadIdx.x - head);
            float vb = b0[i], va = vb;
            if constexpr (OP != OP_COPY) va = a0[i];
            a0[i] = apply<OP>(va, vb, alpha);
        }
    }
    float4 *a = reinterpret_cast<float4 *>(a0 + head);
    const float *bs = b0 + head;
    auto ldb = [&](uint64_t i) -> float4 {
        if constexpr (BU) return OPA_NT >= 1 ? wg_ld_nt_u(bs + 4u * i) : wg_ld_u(bs + 4u * i);
        else return ld4(reinterpret_cast<const float4 *>(bs) + i);
    };
    if (OPA_MODE == 2) {
        const uint64_t base = (uint64_t)blockIdx.x * (kUnroll * kThreads) + threadIdx.x;
        float4 va[kUnroll], vb[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const uint64_t i = base + (uint64_t)u * kThreads;
            if (i < n4) {
                vb[u] = ldb(i);
                if constexpr (OP != OP_COPY) va[u] = ld4(&a[i]);
            }
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const uint64_t i = base + (uint64_t)u * kThreads;
            if (i < n4) st4(&a[i], apply4<OP>(va[u], vb[u], alpha));
        }
        return;
    }
    const uint32_t stride = gridDim.x * kThreads;
    uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    // main: kUnroll independent 16-byte accesses per lane per trip
    for (; (uint64_t)i + (uint64_t)(kUnroll - 1) * stride < n4; i += kUnroll * stride) {
        float4 va[kUnroll], vb[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            vb[u] = ldb(i + u * stride);
            if constexpr (OP != OP_COPY) va[u] = ld4(&a[i + u * stride]);
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) st4(&a[i + u * stride], apply4<OP>(va[u], vb[u], alpha));
    }
    for (; i < n4; i += stride) {
        float4 vb = ldb(i), va = vb;
        if constexpr (OP != OP_COPY) va = a[i];
        a[i] = apply4<OP>(va, vb, alpha);
    }
}

// f16 (extension): computed in f32 and rounded once; for + - * / that equals the correctly rounded f16 result
// (24 >= 2*11 + 2 significand bits, so the double rounding is innocuous).
template <int OP>
__device__ __forceinline__ __half apply_h(__half a, __half b, float alpha) {
    if constexpr (OP == OP_COPY) return b;
    else {
        float r = apply<OP>(__half2float(a), __half2float(b), alpha);
        // axpy: keep the f32 rounding of the fma (the stated contract, and what the float4-wide path does); without this
        // fence hipcc selects v_fma_mixlo_f16 on the scalar path, which rounds the exact result straight to f16.
        if constexpr (OP == OP_AXPY) asm volatile("" : "+v"(r));
        return __float2half_rn(r);
    }
}

struct alignas(16) half8 { __half h[8]; };
struct alignas(2) half8_u { __half h[8]; }; // (the same 16 bytes at an element-aligned address)

template <int OP, bool BU>
__global__ __launch_bounds__(kThreads) void op_assign_f16_vec(__half *a0, const __half *b0, uint32_t head, uint32_t n8, uint32_t n, float alpha) {
    if (blockIdx.x == 0) {
        const uint32_t body_end = head + 8u * n8;
        const uint32_t edge = head + (n - body_end);
        if (threadIdx.x < edge) {
            const uint32_t i = threadIdx.x < head ? threadIdx.x : body_end + (threadIdx.x - head);
            __half vb = b0[i], va = vb;
            if constexpr (OP != OP_COPY) va = a0[i];
            a0[i] = apply_h<OP>(va, vb, alpha);
        }
    }
    half8 *a = reinterpret_cast<half8 *>(a0 + head);
    const uint32_t stride = gridDim.x * kThreads; // flat grid: at most one trip
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < n8; i += stride) {
        half8 vb, va;
        if constexpr (BU) { const half8_u t = reinterpret_cast<const half8_u *>(b0 + head)[i]; __builtin_memcpy(&vb, &t, 16); }
        else vb = reinterpret_cast<const half8 *>(b0 + head)[i];
        va = vb;
        if constexpr (OP != OP_COPY) va = a[i];
        half8 r;
#pragma unroll
        for (int k = 0; k < 8; ++k) r.h[k] = apply_h<OP>(va.h[k], vb.h[k], alpha);
        a[i] = r;
    }
}
inline uint32_t grid_for(uint64_t work_items, int cus) {
    uint64_t blocks = (work_items + kThreads - 1) / kThreads;
    uint64_t cap = (uint64_t)(cus > 0 ? cus : 256) * OPA_WG_PER_CU; // workgroups of 4 waves per CU, grid-stride beyond
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (uint32_t)blocks;
}

template <int OP>
int launch_f32(wg_ctx *ctx, float *a, const float *b, uint32_t n, float alpha) {
    const uintptr_t pa = (uintptr_t)a, pb = (uintptr_t)b;
    // scalar head up to `a`'s 16-byte boundary, vector body, scalar tail; `b` at whatever alignment that leaves it
    uint32_t head = (uint32_t)(((16 - (pa & 15)) & 15) / 4);
    if (head > n) head = n;
    uint32_t n4 = (n - head) / 4;
    const uint32_t blocks = OPA_MODE == 2 ? (uint32_t)(((uint64_t)n4 + kUnroll * kThreads - 1) / (kUnroll * kThreads)) : grid_for(n4, ctx->compute_units);
    if ((pa & 15) == (pb & 15)) hipLaunchKernelGGL((op_assign_f32_vec<OP, false>), dim3(blocks ? blocks : 1), dim3(kThreads), 0, ctx->stream, a, b, head, n4, n, alpha);
    else hipLaunchKernelGGL((op_assign_f32_vec<OP, true>), dim3(blocks ? blocks : 1), dim3(kThreads), 0, ctx->stream, a, b, head, n4, n, alpha);
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}

template <int OP>
int launch_f16(wg_ctx *ctx, __half *a, const __half *b, uint32_t n, float alpha) {
    const uintptr_t pa = (uintptr_t)a, pb = (uintptr_t)b;
    uint32_t head = (uint32_t)(((16 - (pa & 15)) & 15) / 2);
    if (head > n) head = n;
    uint32_t n8 = (n - head) / 8;
    const uint32_t blocks = (n8 + kThreads - 1) / kThreads;
    if ((pa & 15) == (pb & 15)) hipLaunchKernelGGL((op_assign_f16_vec<OP, false>), dim3(blocks ? blocks : 1), dim3(kThreads), 0, ctx->stream, a, b, head, n8, n, alpha);
    else hipLaunchKernelGGL((op_assign_f16_vec<OP, true>), dim3(blocks ? blocks : 1), dim3(kThreads), 0, ctx->stream, a, b, head, n8, n, alpha);
    WG_HIP_TRY(hipGetLastError());
    return WG_OK;
}

} // namespace

int wgk_op_assign(wg_ctx *ctx, int op, wg_dtype dtype, void *a, const void *b, uint32_t n, float alpha) {
    if (n == 0) return WG_OK;
#define WG_CASE(OPV)                                                                              \
    case OPV:                                                                                     \
        return dtype == WG_F32 ? launch_f32<OPV>(ctx, (float *)a, (const float *)b, n, alpha)     \
                               : launch_f16<OPV>(ctx, (__half *)a, (const __half *)b, n, alpha);
    switch (op) {
        WG_CASE(OP_ADD)
        WG_CASE(OP_SUB)
        WG_CASE(OP_MUL)
        WG_CASE(OP_DIV)
        WG_CASE(OP_COPY)
        WG_CASE(OP_AXPY)
    }
#undef WG_CASE
    return wg_set_error(WG_ERR_INVALID_ARG, "OpAssign: unknown variant %d", op);
}
