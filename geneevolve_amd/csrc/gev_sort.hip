// gev_sort.hip -- CommFunc::ras_rank (reference src/CommFunc.cpp:152-161) as a stable sort, gfx950.  Second translation unit of
// libgeneevolve_amd.so (kept apart from gev_library.hip because of rocPRIM's compile time); called by gev_rank_f64.
//
// The reference's pair loop (for i, for j < i: x[j] <= x[i] ? r[i]++ : r[j]++) leaves in r[k]
//     #{ j < k : x[j] <= x[k] }  +  #{ i > k : !(x[k] <= x[i]) }.
// Without NaNs that is #{x < x[k]} + #{j < k : x[j] == x[k]} = the position of k in a STABLE ascending sort (-0.0 == +0.0 tie like
// any equal pair).  A NaN compares false both ways: a NaN element gets n-1-k (every later element counts), a non-NaN element gets
// its stable position among the non-NaN elements plus the number of NaNs behind it.  The sort is rocPRIM's LSD radix sort of
// (order-preserving 64-bit key, index) pairs -- a plain library sort; everything specific to the reference's rule is here.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <stdint.h>

typedef unsigned long long ull;

__global__ void __launch_bounds__(256) k_rank_keys(const double* __restrict__ x, size_t n, uint64_t* __restrict__ key, uint32_t* __restrict__ idx, uint32_t* __restrict__ n_nan)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    uint64_t k;
    if (v != v) { k = ~0ull; atomicAdd(n_nan, 1u); }                  // NaNs behind everything (+inf maps to 0xfff0...)
    else {
        if (v == 0.0) v = 0.0;                                          // -0.0 and +0.0 are equal for <=: one key
        const uint64_t b = (uint64_t)__double_as_longlong(v);
        k = (b >> 63) ? ~b : (b | 0x8000000000000000ull);               // monotone map of the IEEE order onto unsigned order
    }
    key[i] = k; idx[i] = (uint32_t)i;
}
// sorted position -> rank of the element that landed there (no NaNs: rank = position)
__global__ void __launch_bounds__(256) k_rank_scatter(const uint32_t* __restrict__ idx_sorted, size_t n, ull* __restrict__ rank)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p < n) rank[idx_sorted[p]] = p;
}
// general case: nan_before[k] = #{j < k : isnan(x[j])} (exclusive scan of the flags)
__global__ void __launch_bounds__(256) k_rank_nan_flags(const double* __restrict__ x, size_t n, uint32_t* __restrict__ flag)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) flag[i] = x[i] != x[i] ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_rank_scatter_nan(const uint32_t* __restrict__ idx_sorted, const double* __restrict__ x, const uint32_t* __restrict__ nan_before,
                                                          size_t n, uint32_t n_nan, ull* __restrict__ rank)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t k = idx_sorted[p];
    const bool is_nan = x[k] != x[k];
    if (is_nan) rank[k] = (ull)(n - 1 - k);
    else rank[k] = (ull)p + (ull)(n_nan - nan_before[k]);             // stable position among the non-NaN elements (they sort first) + NaNs behind k
}

// d_x: n doubles on the device; d_rank: n results; d_tmp: caller-provided scratch of at least gev_rank_scratch_bytes(n) bytes.
// Returns a hipError_t as int.
// temporary storage of the library calls: the larger of the sort's and the (NaN path's) scan's
static size_t rank_lib_tmp_bytes(size_t n)
{
    size_t t_sort = 0, t_scan = 0;
    (void)rocprim::radix_sort_pairs((void*)nullptr, t_sort, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, n);
    (void)rocprim::exclusive_scan((void*)nullptr, t_scan, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, n, rocprim::plus<uint32_t>());
    return ((t_sort > t_scan ? t_sort : t_scan) + 255) & ~(size_t)255;
}
extern "C" size_t gev_rank_scratch_bytes(size_t n)
{
    const size_t a = (n * 8 + 255) & ~(size_t)255, b = (n * 4 + 255) & ~(size_t)255;
    return 2 * a + 3 * b + 256 + rank_lib_tmp_bytes(n) + 256;
}
extern "C" int gev_rank_device(const double* d_x, size_t n, ull* d_rank, void* d_tmp, hipStream_t st)
{
    const size_t a = (n * 8 + 255) & ~(size_t)255, b = (n * 4 + 255) & ~(size_t)255;
    uint8_t* p = (uint8_t*)d_tmp;
    uint64_t* key_in = (uint64_t*)p; p += a;
    uint64_t* key_out = (uint64_t*)p; p += a;
    uint32_t* idx_in = (uint32_t*)p; p += b;
    uint32_t* idx_out = (uint32_t*)p; p += b;
    uint32_t* flags = (uint32_t*)p; p += b;
    uint32_t* n_nan = (uint32_t*)p; p += 256;
    size_t tmp = rank_lib_tmp_bytes(n);                   // what gev_rank_scratch_bytes reserved behind the arrays
    hipError_t e = hipMemsetAsync(n_nan, 0, 4, st);
    if (e != hipSuccess) return (int)e;
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_rank_keys, dim3(nb), dim3(256), 0, st, d_x, n, key_in, idx_in, n_nan);
    e = rocprim::radix_sort_pairs((void*)p, tmp, key_in, key_out, idx_in, idx_out, n, 0, 64, st);     // stable LSD radix sort
    if (e != hipSuccess) return (int)e;
    uint32_t h_nan = 0;
    e = hipMemcpyAsync(&h_nan, n_nan, 4, hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) return (int)e;
    if (h_nan == 0) hipLaunchKernelGGL(k_rank_scatter, dim3(nb), dim3(256), 0, st, idx_out, n, d_rank);
    else {
        hipLaunchKernelGGL(k_rank_nan_flags, dim3(nb), dim3(256), 0, st, d_x, n, flags);
        e = rocprim::exclusive_scan((void*)p, tmp, flags, idx_in, 0u, n, rocprim::plus<uint32_t>(), st);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(k_rank_scatter_nan, dim3(nb), dim3(256), 0, st, idx_out, d_x, idx_in, n, h_nan, d_rank);
    }
    return (int)hipGetLastError();
}
