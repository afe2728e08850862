// Floor of a REPLAYED dispatch on this runtime: a captured chain of N dependent kernel nodes, replayed; time per node for an empty kernel and
// for a kernel that reads 4 MiB (config 1's matrix) fully coalesced with one memory round trip -- the yardstick for bench.py --workload gemv_f32_1024_graph.
// Build: hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/cpp/graph_floor.hip -o tools/cpp/_bin/graph_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void empty_kernel() {}
__global__ __launch_bounds__(256) void read_kernel(const float4 *m, float *o, unsigned n4) { // every thread: 4 independent 16-byte loads, one store per block
    const unsigned i = blockIdx.x * 1024u + threadIdx.x;
    float4 a = m[i % n4], b = m[(i + 256u) % n4], c = m[(i + 512u) % n4], d = m[(i + 768u) % n4];
    const float s = a.x + b.y + c.z + d.w;
    if (s == 12345.678f) o[blockIdx.x] = s;
}
static double us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s;
    CK(hipSetDevice(0));
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float4 *m; float *o;
    CK(hipMalloc((void **)&m, 4u << 20)); CK(hipMalloc((void **)&o, 4096 * 4));
    CK(hipMemset(m, 0, 4u << 20));
    const int N = 200;
    for (int kind = 0; kind < 4; ++kind) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) {
            if (kind == 0) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s);
            else if (kind == 1) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s);
            else if (kind == 2) hipLaunchKernelGGL(read_kernel, dim3(256), dim3(256), 0, s, (const float4 *)m, o, (4u << 20) / 16u);
            else hipLaunchKernelGGL(read_kernel, dim3(1024), dim3(256), 0, s, (const float4 *)m, o, (4u << 20) / 16u);
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        const int R = 50;
        const double t0 = us();
        for (int i = 0; i < R; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        const double t = (us() - t0) / (R * N);
        const char *what[] = { "empty kernel, 1 block of 64", "empty kernel, 256 blocks of 256", "4 MiB read once, 256 blocks (16 KiB per CU)", "the 4 MiB read four times, 1024 blocks" };
        printf("replayed chain of %d nodes, %-45s: %.2f us per node\n", N, what[kind], t);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
