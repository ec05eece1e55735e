#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define GEV_M31 2147483647u
__device__ __forceinline__ uint32_t mulmod31_a(uint32_t a, uint32_t b)
{
    const uint32_t lo = a * b, hi = __umulhi(a, b);
    uint32_t r = (hi << 1) + (lo >> 31) + (lo & GEV_M31);
    r = (r & GEV_M31) + (r >> 31);
    return r >= GEV_M31 ? r - GEV_M31 : r;
}
__device__ __forceinline__ uint32_t mulmod31_b(uint32_t a, uint32_t b)
{
    const uint64_t p = (uint64_t)a * b;
    const uint32_t lo = (uint32_t)p, hi = (uint32_t)(p >> 32);
    uint32_t r = (hi << 1) + (lo >> 31) + (lo & GEV_M31);
    r = (r & GEV_M31) + (r >> 31);
    return r >= GEV_M31 ? r - GEV_M31 : r;
}
// double-precision variant: q = floor(a*b/M) via FMA, r = a*b - q*M exactly (two-FMA error-free product)
__device__ __forceinline__ double mulmod31_d(double a, double b)
{
    const double M = 2147483647.0, invM = 1.0 / 2147483647.0;
    const double h = a * b;                 // rounded product (< 2^62)
    const double l = fma(a, b, -h);         // exact low part
    double q = floor(h * invM);
    double r = fma(-q, M, h) + l;           // h - q*M is exact when |h - q*M| < 2^53 ... (q*M exact? q < 2^31, M < 2^31: product < 2^62: NOT exact) -> experiment only
    if (r < 0) r += M; if (r >= M) r -= M;
    return r;
}
template <int V> __global__ void k(uint32_t* x, uint32_t m, int n)
{
    uint32_t v0 = x[threadIdx.x] | 1, v1 = v0 + 2, v2 = v0 + 4, v3 = v0 + 6;
    if (V == 2) {
        double d0 = v0, d1 = v1, d2 = v2, d3 = v3; const double dm = m;
        for (int i = 0; i < n; i++) { d0 = mulmod31_d(d0, dm); d1 = mulmod31_d(d1, dm); d2 = mulmod31_d(d2, dm); d3 = mulmod31_d(d3, dm); }
        x[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(d0 + d1 + d2 + d3);
        return;
    }
    for (int i = 0; i < n; i++) {
        if (V == 0) { v0 = mulmod31_a(v0, m); v1 = mulmod31_a(v1, m); v2 = mulmod31_a(v2, m); v3 = mulmod31_a(v3, m); }
        else { v0 = mulmod31_b(v0, m); v1 = mulmod31_b(v1, m); v2 = mulmod31_b(v2, m); v3 = mulmod31_b(v3, m); }
    }
    x[blockIdx.x * blockDim.x + threadIdx.x] = v0 ^ v1 ^ v2 ^ v3;
}
int main()
{
    uint32_t* d; hipMalloc(&d, 4096 * 256 * 4); hipMemset(d, 7, 4096 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int n = 2000;
    for (int v = 0; v < 3; v++) for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(a);
        if (v == 0) hipLaunchKernelGGL(k<0>, dim3(4096), dim3(256), 0, 0, d, 1234567u, n);
        if (v == 1) hipLaunchKernelGGL(k<1>, dim3(4096), dim3(256), 0, 0, d, 1234567u, n);
        if (v == 2) hipLaunchKernelGGL(k<2>, dim3(4096), dim3(256), 0, 0, d, 1234567u, n);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("variant %d: %.3f ms  -> %.2f Tmulmod/s\n", v, ms, 4096.0 * 256 * 4 * n / ms / 1e9);
    }
    return 0;
}
