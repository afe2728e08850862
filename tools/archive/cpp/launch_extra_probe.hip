// Probe: is the eager enqueue of a kernel with arguments (3.0 us here against 1.0 us for an empty kernel) the argument marshalling of
// hipLaunchKernelGGL, and does hipModuleLaunchKernel with a prepacked argument buffer (HIP_LAUNCH_PARAM_BUFFER_POINTER) avoid it?
// Build: hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/cpp/launch_extra_probe.hip -o tools/cpp/_bin/launch_extra_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
struct Args { const float *m; uint32_t ldm; uint64_t mb; const float *v; uint32_t ldv; uint64_t vb; float *out; float *part; uint32_t ld; uint64_t db, ds; uint32_t rows, k, nrhs, kps; };
__global__ void empty_kernel() {}
__global__ void struct_kernel(Args a) { if (a.rows == 0xdeadbeefu && threadIdx.x == 12345) a.out[0] = a.m[a.k] + a.v[a.nrhs]; }
static double us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s;
    hipSetDevice(0);
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    Args a{};
    a.rows = 1024; a.k = 1024;
    hipFunction_t f = nullptr;
    hipError_t e = hipGetFuncBySymbol(&f, (const void *)struct_kernel);
    printf("hipGetFuncBySymbol: %s\n", hipGetErrorString(e));
    const int R = 20000;
    for (int blocks : { 1, 128 }) {
        for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(struct_kernel, dim3(blocks), dim3(256), 0, s, a);
        hipStreamSynchronize(s);
        double t0 = us();
        for (int i = 0; i < R; ++i) hipLaunchKernelGGL(empty_kernel, dim3(blocks), dim3(256), 0, s);
        double t1 = us(); hipStreamSynchronize(s); double t2 = us();
        printf("%4d blocks  empty kernel:                          enqueue %.2f us, end-to-end %.2f us\n", blocks, (t1 - t0) / R, (t2 - t0) / R);
        t0 = us();
        for (int i = 0; i < R; ++i) hipLaunchKernelGGL(struct_kernel, dim3(blocks), dim3(256), 0, s, a);
        t1 = us(); hipStreamSynchronize(s); t2 = us();
        printf("%4d blocks  struct arg, hipLaunchKernelGGL:         enqueue %.2f us, end-to-end %.2f us\n", blocks, (t1 - t0) / R, (t2 - t0) / R);
        if (f) {
            size_t size = sizeof(a);
            void *extra[] = { HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END };
            for (int i = 0; i < 1000; ++i) hipModuleLaunchKernel(f, blocks, 1, 1, 256, 1, 1, 0, s, nullptr, extra);
            hipStreamSynchronize(s);
            t0 = us();
            for (int i = 0; i < R; ++i) hipModuleLaunchKernel(f, blocks, 1, 1, 256, 1, 1, 0, s, nullptr, extra);
            t1 = us(); hipStreamSynchronize(s); t2 = us();
            printf("%4d blocks  struct arg, hipModuleLaunchKernel extra: enqueue %.2f us, end-to-end %.2f us\n", blocks, (t1 - t0) / R, (t2 - t0) / R);
            void *params[] = { &a };
            t0 = us();
            for (int i = 0; i < R; ++i) hipModuleLaunchKernel(f, blocks, 1, 1, 256, 1, 1, 0, s, params, nullptr);
            t1 = us(); hipStreamSynchronize(s); t2 = us();
            printf("%4d blocks  struct arg, hipModuleLaunchKernel params: enqueue %.2f us, end-to-end %.2f us\n", blocks, (t1 - t0) / R, (t2 - t0) / R);
        }
    }
    return 0;
}
