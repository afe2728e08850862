// Probe: can a stream wait on a device word that a RUNNING kernel of another stream sets (hipStreamWaitValue32)? Which memory kinds work?
// build: hipcc --offload-arch=gfx950 -O2 tools/cpp/wait_value_probe.hip -o /tmp/wait_value_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("  %s -> %s\n", #x, hipGetErrorString(e_)); (void)hipGetLastError(); } } while (0)
__global__ void setter(uint32_t *flag, uint32_t value, uint32_t delay_us, uint32_t *data) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)delay_us * 100u) __builtin_amdgcn_s_sleep(32);
    data[0] = 0x1234abcd; // payload written before the flag (system-scope release below)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // keep running: the waiter must be released while this kernel is still resident
    while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)(delay_us + 3000) * 100u) __builtin_amdgcn_s_sleep(32);
}
__global__ void stamp(uint64_t *t) { *t = __builtin_amdgcn_s_memrealtime(); }
int main() {
    int can = -1;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    uint64_t *stamps; CK(hipHostMalloc((void **)&stamps, 64, 0));
    uint32_t *data; CK(hipMalloc((void **)&data, 256));
    const char *kinds[] = { "hipMalloc", "hipExtMallocWithFlags(signal)", "hipExtMallocWithFlags(uncached)", "hipHostMalloc", "hipExtMallocWithFlags(finegrained)" };
    for (int kind = 0; kind < 5; ++kind) {
        uint32_t *flag = nullptr;
        hipError_t e = hipSuccess;
        if (kind == 0) e = hipMalloc((void **)&flag, 256);
        if (kind == 1) e = hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory);
        if (kind == 2) e = hipExtMallocWithFlags((void **)&flag, 256, hipDeviceMallocUncached);
        if (kind == 3) e = hipHostMalloc((void **)&flag, 256, 0);
        if (kind == 4) e = hipExtMallocWithFlags((void **)&flag, 256, hipDeviceMallocFinegrained);
        printf("%s: alloc %s\n", kinds[kind], hipGetErrorString(e));
        if (e != hipSuccess) { (void)hipGetLastError(); continue; }
        if (kind == 3) *flag = 0; else CK(hipMemset(flag, 0, 8));
        CK(hipDeviceSynchronize());
        stamps[0] = stamps[1] = stamps[2] = 0;
        stamp<<<1, 1, 0, s1>>>(stamps + 0);
        setter<<<1, 64, 0, s1>>>(flag, 7u, 2000u, data); // sets the flag after 2 ms, runs 5 ms in all
        e = hipStreamWaitValue32(s2, flag, 7u, hipStreamWaitValueGte, 0xffffffffu);
        printf("  hipStreamWaitValue32 -> %s\n", hipGetErrorString(e));
        if (e != hipSuccess) { (void)hipGetLastError(); CK(hipDeviceSynchronize()); continue; }
        stamp<<<1, 1, 0, s2>>>(stamps + 1);
        stamp<<<1, 1, 0, s1>>>(stamps + 2);
        auto t0 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(s2));
        double ms2 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        CK(hipStreamSynchronize(s1));
        printf("  waiter released %.2f ms after the setter started (flag set at 2.00, setter ends at 5.00); host saw s2 done after %.2f ms\n",
               (double)(stamps[1] - stamps[0]) / 1e5, ms2);
    }
    return 0;
}
