// Floor of an eager dispatch on this runtime: host cost of hipSetDevice / hipGetDevice, and enqueue + end-to-end cost of an empty
// kernel and of a 1-block kernel with the GEMV's argument count -- the yardstick for tools/cpp/dispatch_overhead.cpp.
// Build: hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/cpp/launch_floor.hip -o /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void empty_kernel() {}
__global__ void args_kernel(float *o, const float *m, const float *v, unsigned a, unsigned b, unsigned c, unsigned d, unsigned e, unsigned long f, unsigned long g) {
    if (o == nullptr && threadIdx.x == 12345) o[0] = m[a] + v[b] + c + d + e + f + g;
}
static double us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s;
    hipSetDevice(0);
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int N = 1000000;
    double t0 = us();
    for (int i = 0; i < N; ++i) hipSetDevice(0);
    printf("hipSetDevice: %.3f us\n", (us() - t0) / N);
    int d;
    t0 = us();
    for (int i = 0; i < N; ++i) hipGetDevice(&d);
    printf("hipGetDevice: %.3f us\n", (us() - t0) / N);
    for (int blocks : { 1, 32, 1024 }) {
        for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(blocks), dim3(256), 0, s);
        hipStreamSynchronize(s);
        const int R = 20000;
        t0 = us();
        for (int i = 0; i < R; ++i) hipLaunchKernelGGL(empty_kernel, dim3(blocks), dim3(256), 0, s);
        double t1 = us();
        hipStreamSynchronize(s);
        double t2 = us();
        printf("empty kernel, %4d blocks: enqueue %.2f us, end-to-end %.2f us per launch\n", blocks, (t1 - t0) / R, (t2 - t0) / R);
        t0 = us();
        for (int i = 0; i < R; ++i) hipLaunchKernelGGL(args_kernel, dim3(blocks), dim3(256), 0, s, (float *)&d, (const float *)&d, (const float *)&d, 1u, 2u, 3u, 4u, 5u, 6ul, 7ul);
        t1 = us();
        hipStreamSynchronize(s);
        t2 = us();
        printf("10-arg kernel, %4d blocks: enqueue %.2f us, end-to-end %.2f us per launch\n", blocks, (t1 - t0) / R, (t2 - t0) / R);
    }
    return 0;
}
