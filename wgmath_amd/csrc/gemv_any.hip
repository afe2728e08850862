// Gemv / GemvTr on a matrix view of ANY alignment (odd offset, odd leading dimension, lengths that are not multiples of 4): one pass over the
// matrix where it lies. The reference's kernels cannot address such views at all (they bind array<vec4<f32>> and divide offsets and strides by
// 4: shape.wgsl:64-66) although its constructors hand them out (GpuMatrix::slice / rows / column, tensor.rs:574-626); until round 6 they were
// staged into an aligned copy first (three passes over the matrix: 2.7x the aligned kernel's time; 4-5x with the element-wise copy before that).
//
// What makes one pass possible: global_load_dwordx4 takes ANY element-aligned address on this target (the HSA ABI runs the memory pipeline in
// unaligned-access mode; tools/cpp/unaligned_probe.hip, profiles/r06_unaligned_probe.txt: 16-byte loads that start 4, 8 or 12 bytes past a 16-byte
// boundary stream at the aligned rate, 2-byte offsets at 0.88 of it). So a lane owns E consecutive rows (16 bytes of elements: 4 f32 / 8 f16) counted
// from the view's first row, exactly as in gemv.hip, and loads them with one instruction wherever they lie. Only a chunk that is not whole -- the last rows
// of a view whose length is not a multiple of E -- is read element by element (a 16-byte load there could run off the end of the buffer): row blocks /
// row steps that are whole take a loop without a single bounds check. Vectors and results are addressed element by element (any offset).
//   N: out[r] = sum_c m[r, c] v[c]: the four waves of a workgroup take column ranges, summed through LDS; column splits over grid.y write f32 partials.
//   T: out[c] = sum_r m[r, c] v[r]: a wave owns 4 columns and sweeps rows, butterfly at the end; row splits over grid.y write f32 partials.
// A small second kernel sums the partials in a fixed order. Order of the sums: per lane in sequence, then across lanes / waves / splits -- the
// tolerance contract of gemv.hip, not its bits.
#include "wg_internal.hpp"

#ifndef WG_ANY_NU
#define WG_ANY_NU 8 // N: columns in flight per lane
#endif
#ifndef WG_ANY_PER_CU
#define WG_ANY_PER_CU 0 // workgroups per CU the splits aim at; 0: 2, and 4 for N on matrices of 1 GiB and more (cliff sweep A/B, profiles/r06_gemv_any_ab.txt)
#endif

namespace {

constexpr int kThreads = 256, kWaves = 4;

template <typename T> struct Elt;
template <> struct Elt<float> { static constexpr int E = 4, ES = 4; };
template <> struct Elt<_Float16> { static constexpr int E = 8, ES = 2; };

typedef uint32_t wg_u32x4 __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ void to_floats(wg_u32x4 v, float (&f)[Elt<T>::E]) {
    if constexpr (Elt<T>::ES == 4) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    } else {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        const h8 h = __builtin_bit_cast(h8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (float)h[e];
    }
}
// 16 bytes at any element-aligned address (an integer address says nothing about the address space: without the cast the load is a flat_load; the pointer type
// states the alignment that is really there -- the backend keeps one global_load_dwordx4, the target runs in unaligned-access mode)
// NT: the non-temporal hint for a matrix that cannot stay in the Infinity Cache anyway (the launcher: from 512 MiB on -- 16384^2 f32 207 -> 199 us; below that it costs
// repeated products of one matrix their cache hits: 8192^2 f16 at offset 1 33.8 -> 39.3 us; profiles/r06_gemv_any_ab.txt)
template <int ES, bool NT>
__device__ __forceinline__ wg_u32x4 ld16(uintptr_t addr) {
    if constexpr (ES == 4) {
        const __attribute__((address_space(1), aligned(4))) wg_u32x4 *p = reinterpret_cast<const __attribute__((address_space(1), aligned(4))) wg_u32x4 *>(addr);
        return NT ? __builtin_nontemporal_load(p) : *p;
    } else {
        const __attribute__((address_space(1), aligned(2))) wg_u32x4 *p = reinterpret_cast<const __attribute__((address_space(1), aligned(2))) wg_u32x4 *>(addr);
        return NT ? __builtin_nontemporal_load(p) : *p;
    }
}
// the E elements from element `first` on of a run of `len` >= 1 elements at `base`, as floats; elements past the run are 0. WHOLE: the caller knows first + E <= len.
template <typename T, bool WHOLE, bool NT = false>
__device__ __forceinline__ void load_elems(uintptr_t base, uint32_t first, uint32_t len, float (&f)[Elt<T>::E]) {
    constexpr int E = Elt<T>::E, ES = Elt<T>::ES;
    if constexpr (WHOLE) to_floats<T>(ld16<ES, NT>(base + (uintptr_t)first * ES), f);
    else { // (no branch: every lane reads E elements at clamped positions and drops what lies past the run -- a branch per chunk serialises the columns' loads, and
           // the one ragged row block of an N launch then takes longer than all the whole ones together)
        const __attribute__((address_space(1))) T *p = reinterpret_cast<const __attribute__((address_space(1))) T *>(base);
        T raw[E];
#pragma unroll
        for (int e = 0; e < E; ++e) raw[e] = p[min(first + (uint32_t)e, len - 1u)];
#pragma unroll
        for (int e = 0; e < E; ++e) f[e] = first + e < len ? (float)raw[e] : 0.f;
    }
}

__device__ __forceinline__ float readlane_f(float x, uint32_t lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane)); }
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) x += __shfl_xor(x, s, 64);
    return x;
}

struct AnyArgs {
    uintptr_t m; uint32_t ldm; uint64_t m_batch; // elements
    uintptr_t v; uint32_t ldv; uint64_t v_batch;
    uintptr_t out; uint32_t ldo; uint64_t o_batch;
    float *part;       // f32 partials [matrix * rhs][split][out_len] when nsplit > 1
    uint32_t R, C;     // the matrix view: R rows (contiguous), C columns
    uint32_t nrhs, nsplit, per_split; // per_split: columns (N) / rows (T) of one split
};

template <typename T>
__device__ __forceinline__ void store_out(const AnyArgs &a, uint32_t z, uint32_t y, uint32_t split, uint32_t out_len, uint32_t i, float x) {
    if (a.part) a.part[((uint64_t)(z * a.nrhs + y) * a.nsplit + split) * out_len + i] = x;
    else reinterpret_cast<T *>(a.out)[z * a.o_batch + (uint64_t)y * a.ldo + i] = (T)x;
}

// grid = (row blocks of 64 E, column splits, matrices * right-hand sides)
template <typename T, bool NT>
__global__ __launch_bounds__(kThreads) void gemv_any_n_kernel(AnyArgs a) {
    constexpr int E = Elt<T>::E, ES = Elt<T>::ES, U = WG_ANY_NU;
    __shared__ float part[kWaves][E][64];
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t z = blockIdx.z / a.nrhs, y = blockIdx.z % a.nrhs;
    const uint32_t r0 = (blockIdx.x * 64u + lane) * E;
    const uint32_t c_begin = blockIdx.y * a.per_split, c_end = min(a.C, c_begin + a.per_split);
    const uint32_t per_wave = (c_end - c_begin + kWaves - 1u) / kWaves;
    const uint32_t w_begin = min(c_end, c_begin + wave * per_wave), w_end = min(c_end, w_begin + per_wave);
    const uintptr_t mb = a.m + z * a.m_batch * ES;
    const __attribute__((address_space(1))) T *vp = reinterpret_cast<const __attribute__((address_space(1))) T *>(a.v) + z * a.v_batch + (uint64_t)y * a.ldv;
    float acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = 0.f;
    auto sweep = [&](auto whole_c) {
        constexpr bool WHOLE = decltype(whole_c)::value;
        for (uint32_t cb = w_begin; cb < w_end; cb += 64u) {
            const float vv = cb + lane < w_end ? (float)vp[cb + lane] : 0.f;
            const uint32_t n = min(64u, w_end - cb);
            for (uint32_t u0 = 0; u0 < n; u0 += U) {
                float f[U][E];
#pragma unroll
                for (int u = 0; u < U; ++u) // (a slot past the end re-reads the last column; its values are dropped below, not multiplied by 0: they may be Inf / NaN)
                    load_elems<T, WHOLE, NT>(mb + (uint64_t)min(cb + u0 + u, w_end - 1u) * a.ldm * ES, r0, a.R, f[u]);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (u0 + u < n) {
                        const float x = readlane_f(vv, u0 + u);
#pragma unroll
                        for (int e = 0; e < E; ++e) acc[e] = fmaf(f[u][e], x, acc[e]);
                    }
                }
            }
        }
    };
    if ((blockIdx.x + 1u) * 64u * E <= a.R) sweep(std::true_type{}); else sweep(std::false_type{}); // (the last row block of a ragged view checks every chunk)
#pragma unroll
    for (int e = 0; e < E; ++e) part[wave][e][lane] = acc[e];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const float s = ((part[0][e][lane] + part[1][e][lane]) + part[2][e][lane]) + part[3][e][lane];
            if (r0 + e < a.R) store_out<T>(a, z, y, blockIdx.y, a.R, r0 + e, s);
        }
    }
}

// grid = (groups of 4 * kWaves columns, row splits, matrices * right-hand sides)
template <typename T, bool NT>
__global__ __launch_bounds__(kThreads) void gemv_any_t_kernel(AnyArgs a) {
    constexpr int E = Elt<T>::E, ES = Elt<T>::ES, CW = 4;
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t z = blockIdx.z / a.nrhs, y = blockIdx.z % a.nrhs;
    const uint32_t c0 = (blockIdx.x * kWaves + wave) * CW;
    if (c0 >= a.C) return;
    const uint32_t r_begin = blockIdx.y * a.per_split, r_end = min(a.R, r_begin + a.per_split);
    const uintptr_t mb = a.m + z * a.m_batch * ES;
    const uintptr_t vb = a.v + (z * a.v_batch + (uint64_t)y * a.ldv) * ES;
    float acc[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) acc[c] = 0.f;
    auto step = [&](uint32_t rb, auto whole_c) {
        constexpr bool WHOLE = decltype(whole_c)::value;
        const uint32_t r0 = rb + lane * E;
        float f[CW][E], x[E];
#pragma unroll
        for (int c = 0; c < CW; ++c) load_elems<T, WHOLE, NT>(mb + (uint64_t)min(c0 + c, a.C - 1u) * a.ldm * ES, r0, r_end, f[c]);
        load_elems<T, WHOLE>(vb, r0, r_end, x);
#pragma unroll
        for (int c = 0; c < CW; ++c)
#pragma unroll
            for (int e = 0; e < E; ++e) acc[c] = fmaf(f[c][e], x[e], acc[c]);
    };
    uint32_t rb = r_begin;
    for (; rb + 64u * E <= r_end; rb += 64u * E) step(rb, std::true_type{});
    if (rb < r_end) step(rb, std::false_type{}); // the ragged end of the range: rows past it are read as 0
#pragma unroll
    for (int c = 0; c < CW; ++c) {
        const float s = wave_sum(acc[c]);
        if (lane == 0 && c0 + c < a.C) store_out<T>(a, z, y, blockIdx.y, a.C, c0 + c, s);
    }
}

// out[i] = sum over the splits, in order. grid = (blocks of 256 outputs, matrices * right-hand sides)
template <typename T>
__global__ __launch_bounds__(kThreads) void gemv_any_combine_kernel(AnyArgs a, uint32_t out_len) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x, z = blockIdx.y / a.nrhs, y = blockIdx.y % a.nrhs;
    if (i >= out_len) return;
    const float *p = a.part + (uint64_t)blockIdx.y * a.nsplit * out_len + i;
    float s = p[0];
    for (uint32_t k = 1; k < a.nsplit; ++k) s += p[(uint64_t)k * out_len];
    reinterpret_cast<T *>(a.out)[z * a.o_batch + (uint64_t)y * a.ldo + i] = (T)s;
}

template <typename T>
int launch_any(wg_ctx *ctx, bool trans, uint32_t R, uint32_t C, uint32_t nrhs, uint32_t nmats, void *out, uint32_t out_ld, uint64_t out_batch, wgk_mat m, wgk_mat v) {
    constexpr uint32_t E = Elt<T>::E;
    const uint32_t cus = ctx->compute_units > 0 ? (uint32_t)ctx->compute_units : 256u;
    const uint64_t gz = (uint64_t)nmats * nrhs;
    if (gz > 65535u) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemv: matrices x right-hand sides = %llu exceeds 65535 on the path for views that are not vec4-aligned", (unsigned long long)gz);
    const uint32_t out_len = trans ? C : R;
    AnyArgs a;
    a.m = (uintptr_t)m.ptr; a.ldm = m.ld; a.m_batch = m.batch;
    a.v = (uintptr_t)v.ptr; a.ldv = v.ld; a.v_batch = v.batch;
    a.out = (uintptr_t)out; a.ldo = out_ld; a.o_batch = out_batch;
    a.R = R; a.C = C; a.nrhs = nrhs; a.part = nullptr;
    // ~per_cu workgroups per CU; a split is whole 64-column chunks (N) / whole 64 E-row steps (T)
    const uint32_t gx = trans ? (C + 4u * kWaves - 1u) / (4u * kWaves) : (R + 64u * E - 1u) / (64u * E);
    const uint32_t unit = trans ? 64u * E : 64u * kWaves, len = trans ? R : C, units = (len + unit - 1u) / unit;
    const uint64_t per_cu = WG_ANY_PER_CU ? WG_ANY_PER_CU : (!trans && (uint64_t)R * C * sizeof(T) >= (1ull << 30) ? 4u : 2u);
    uint32_t nsplit = (uint32_t)((per_cu * cus + (uint64_t)gx * gz - 1u) / ((uint64_t)gx * gz));
    if (nsplit > units) nsplit = units;
    if (nsplit < 1u) nsplit = 1u;
    a.per_split = ((units + nsplit - 1u) / nsplit) * unit;
    nsplit = (len + a.per_split - 1u) / a.per_split;
    if (nsplit > 65535u) return wg_set_error(WG_ERR_UNSUPPORTED, "Gemv: too many splits");
    a.nsplit = nsplit;
    if (nsplit > 1u) {
        void *ws = nullptr;
        if (int rc = wg_ctx_workspace(ctx, (size_t)gz * nsplit * out_len * sizeof(float), &ws)) return rc;
        a.part = (float *)ws;
    }
    const dim3 grid(gx, nsplit, (uint32_t)gz), block(kThreads);
    const bool nt = (uint64_t)R * C * sizeof(T) >= (512ull << 20);
    if (trans) {
        if (nt) hipLaunchKernelGGL((gemv_any_t_kernel<T, true>), grid, block, 0, ctx->stream, a);
        else hipLaunchKernelGGL((gemv_any_t_kernel<T, false>), grid, block, 0, ctx->stream, a);
    } else if (nt) hipLaunchKernelGGL((gemv_any_n_kernel<T, true>), grid, block, 0, ctx->stream, a);
    else hipLaunchKernelGGL((gemv_any_n_kernel<T, false>), grid, block, 0, ctx->stream, a);
    WG_HIP_TRY(hipGetLastError());
    if (nsplit > 1u) {
        hipLaunchKernelGGL(gemv_any_combine_kernel<T>, dim3((out_len + kThreads - 1u) / kThreads, (uint32_t)gz), block, 0, ctx->stream, a, out_len);
        WG_HIP_TRY(hipGetLastError());
    }
    return WG_OK;
}

} // namespace

// m: the R x C view as stored (GemvTr contracts its rows); v / out: nrhs columns, v.ld / out_ld apart.
int wgk_gemv_any(wg_ctx *ctx, bool trans, wg_dtype dtype, uint32_t R, uint32_t C, uint32_t nrhs, uint32_t nmats, void *out, uint32_t out_ld, uint64_t out_batch,
                 wgk_mat m, wgk_mat v) {
    if (R == 0 || C == 0 || nrhs == 0 || nmats == 0) return WG_OK;
    if (dtype == WG_F16) return launch_any<_Float16>(ctx, trans, R, C, nrhs, nmats, out, out_ld, out_batch, m, v);
    return launch_any<float>(ctx, trans, R, C, nrhs, nmats, out, out_ld, out_batch, m, v);
}
