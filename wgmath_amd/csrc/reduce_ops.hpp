// The reference's reduction operators (wgebra reduce.rs:30-58, reduce.wgsl:12-46): init value, per-lane fold (`workspace_fn`) and tree
// combine (`reduce_fn`), shared by reduce.hip and the fused Gemv+Reduce epilogue in gemv.hip. Every operation is a single correctly
// rounded IEEE operation (no contraction), which is what makes the kernels bit-identical to the reference order.
#pragma once

namespace {

enum { R_MIN = 0, R_MAX = 1, R_SUM = 2, R_PROD = 3, R_SQNORM = 4 };

template <int OP>
__device__ __forceinline__ float r_init() {
    if constexpr (OP == R_MIN) return 3.4e38f;        // init_max_f32 (reduce.wgsl:40-42) -- not FLT_MAX
    else if constexpr (OP == R_MAX) return -3.4e38f;  // init_min_f32 (reduce.wgsl:44-46)
    else if constexpr (OP == R_PROD) return 1.0f;
    else return 0.0f;
}
template <int OP>
__device__ __forceinline__ float r_ws(float acc, float x) { // workspace_fn
    if constexpr (OP == R_MIN) return fminf(acc, x);
    else if constexpr (OP == R_MAX) return fmaxf(acc, x);
    else if constexpr (OP == R_SUM) return __fadd_rn(acc, x);
    else if constexpr (OP == R_PROD) return __fmul_rn(acc, x);
    else {
        float sq = __fmul_rn(x, x);
        asm volatile("" : "+v"(sq)); // the product is rounded on its own (reduce_sqnorm_f32 as restated by the oracle): never an FMA
        return __fadd_rn(acc, sq);
    }
}
template <int OP>
__device__ __forceinline__ float r_red(float a, float b) { // reduce_fn
    if constexpr (OP == R_MIN) return fminf(a, b);
    else if constexpr (OP == R_MAX) return fmaxf(a, b);
    else if constexpr (OP == R_PROD) return __fmul_rn(a, b);
    else return __fadd_rn(a, b); // Sum and SqNorm (reduce.rs:55)
}


} // namespace
