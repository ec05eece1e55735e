// How long do hipMalloc / hipFree / hipMallocAsync take on the host while a long kernel is running on another stream?
// (design input for the list-buffer growth policy of gev_library.hip)   build: hipcc --offload-arch=gfx950 -O2 tools/alloc_probe.hip -o tools/alloc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(unsigned long long* out, unsigned long long cycles)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = wall_clock64() - t0;
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    unsigned long long* d; CK(hipMalloc(&d, 8));
    for (int busy = 0; busy < 2; busy++) {
        for (size_t mb : {8, 64, 256}) {
            if (busy) hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, a, d, 100000000ull * 3);   // ~3 s at 100 MHz wall clock
            std::vector<void*> ps; double t0 = now();
            for (int i = 0; i < 8; i++) { void* p; CK(hipMalloc(&p, mb << 20)); ps.push_back(p); }
            double t1 = now();
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, b, d, 1000ull);                            // a launch on the other stream
            double t2 = now();
            void* q; CK(hipMallocAsync(&q, mb << 20, b)); double t3 = now();
            CK(hipFreeAsync(q, b)); double t4 = now();
            CK(hipFree(ps.back())); ps.pop_back(); double t5 = now();
            printf("gpu %s  %4zu MiB: hipMalloc %.3f ms each, launch %.3f ms, hipMallocAsync %.3f ms, hipFreeAsync %.3f ms, hipFree %.3f ms\n",
                   busy ? "BUSY" : "idle", mb, (t1 - t0) / 8, t2 - t1, t3 - t2, t4 - t3, t5 - t4);
            CK(hipDeviceSynchronize());
            for (void* p : ps) CK(hipFree(p));
        }
    }
    return 0;
}
