// Probe: do global_load_dwordx4 / global_store_dwordx4 take addresses that are only element-aligned (2 or 4 bytes), and at what rate?
// A streaming sum over 1 GiB read as 16-byte pieces that start `off` bytes past a 16-byte boundary (off = 0, 2, 4, 6, 8, 12), checked against the host's sum of
// the same bytes; then a copy with misaligned stores.
// build: hipcc --offload-arch=gfx950 -O2 tools/cpp/unaligned_probe.hip -o tools/cpp/_bin/unaligned_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void sum_kernel(const char *p, uint64_t n16, unsigned long long *out) {
    unsigned long long s = 0;
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256ull) {
        u4 v;
        asm volatile("global_load_dwordx4 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p + 16 * i) : "memory");
        s += (unsigned long long)v.x + v.y + v.z + v.w;
    }
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}
__global__ __launch_bounds__(256) void sum4_kernel(const char *p, uint64_t n16, unsigned long long *out) { // 4 loads in flight, compiler-scheduled
    unsigned long long s = 0;
    const uint64_t stride = (uint64_t)gridDim.x * 256ull;
    uint64_t i = blockIdx.x * 256ull + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        u4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const __attribute__((address_space(1))) u4 *>((uintptr_t)(p + 16 * (i + k * stride)));
#pragma unroll
        for (int k = 0; k < 4; ++k) s += (unsigned long long)v[k].x + v[k].y + v[k].z + v[k].w;
    }
    for (; i < n16; i += stride) {
        const u4 v = *reinterpret_cast<const __attribute__((address_space(1))) u4 *>((uintptr_t)(p + 16 * i));
        s += (unsigned long long)v.x + v.y + v.z + v.w;
    }
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}
__global__ __launch_bounds__(256) void copy_kernel(char *d, const char *s, uint64_t n16) {
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256ull) {
        const u4 v = *reinterpret_cast<const __attribute__((address_space(1))) u4 *>((uintptr_t)(s + 16 * i));
        *reinterpret_cast<__attribute__((address_space(1))) u4 *>((uintptr_t)(d + 16 * i)) = v;
    }
}

int main() {
    const uint64_t bytes = 1ull << 30, n16 = bytes / 16 - 1;
    std::vector<uint16_t> h(bytes / 2);
    uint32_t x = 12345;
    for (auto &e : h) { x = x * 1664525u + 1013904223u; e = (uint16_t)(x >> 16); }
    char *d = nullptr, *d2 = nullptr;
    unsigned long long *out = nullptr;
    CK(hipMalloc((void **)&d, bytes)); CK(hipMalloc((void **)&d2, bytes)); CK(hipMalloc((void **)&out, 8));
    CK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int off : { 0, 2, 4, 6, 8, 12, 14 }) {
        unsigned long long want = 0;
        const uint16_t *hp = h.data() + off / 2;
        for (uint64_t i = 0; i < n16 * 8; i += 2) want += (unsigned long long)hp[i] + ((unsigned long long)hp[i + 1] << 16);
        for (int which = 0; which < 2; ++which) {
            unsigned long long got = 0;
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemset(out, 0, 8));
                CK(hipEventRecord(e0));
                if (which == 0) hipLaunchKernelGGL(sum_kernel, dim3(256 * 8), dim3(256), 0, 0, d + off, n16, out);
                else hipLaunchKernelGGL(sum4_kernel, dim3(256 * 8), dim3(256), 0, 0, d + off, n16, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
            }
            CK(hipMemcpy(&got, out, 8, hipMemcpyDeviceToHost));
            printf("load off %2d %s: %s  %.1f us  %.0f GB/s\n", off, which ? "x4 in flight" : "one at a time", got == want ? "sum OK " : "SUM WRONG", best * 1e3, n16 * 16 / (best * 1e-3) / 1e9);
        }
    }
    for (int off : { 0, 2, 4, 8 }) {
        CK(hipMemset(d2, 0, bytes));
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, 0, d2 + off, d, n16);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        std::vector<uint16_t> back(1 << 20);
        CK(hipMemcpy(back.data(), d2 + off + (bytes / 2), back.size() * 2, hipMemcpyDeviceToHost));
        bool ok = true;
        for (size_t i = 0; i < back.size(); ++i) ok &= back[i] == h[bytes / 4 + i];
        printf("store off %2d: %s  %.1f us  %.0f GB/s (read + write)\n", off, ok ? "copy OK " : "COPY WRONG", best * 1e3, 2.0 * n16 * 16 / (best * 1e-3) / 1e9);
    }
    return 0;
}
