// tools/stitch_bench.hip -- A/B harness for variants of the dense stitch kernel (k_stitch_rows).
// Synthetic descriptors of config-2 shape; interleaved rounds in ONE process (guide rule 24).
//   hipcc --offload-arch=gfx950 -O3 -I. tools/stitch_bench.hip -o /tmp/stitch_bench && /tmp/stitch_bench [N] [L]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <vector>
#include "../geneevolve_amd/csrc/gev_kernels.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NTL, bool NTS, bool PERSIST, int THREADS>
__global__ void __launch_bounds__(THREADS) stitch_v(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, size_t stride, u32 chunks,
                                                    u32 nrows, const u64* __restrict__ pos, u32 L, SampleDev sd)
{
    __shared__ u32 s_idx[STITCH_KMAX];
    for (u32 row = blockIdx.x; row < nrows; row += PERSIST ? gridDim.x : nrows) {
        const u32 i = row >> 1, s = row & 1;
        const size_t G = 2 * (size_t)i + s;
        const u32 parent = s ? sd.mother[i] : sd.father[i];
        const u32 start = sd.start[G];
        const u32 k = min(sd.k[G], (u32)STITCH_KMAX);
        const u64* bk = sd.bk + sd.bk_off[G];
        const v4u* __restrict__ A = (const v4u*)(src + (size_t)(2 * parent + start) * stride);
        const v4u* __restrict__ B = (const v4u*)(src + (size_t)(2 * parent + (start ^ 1)) * stride);
        v4u* __restrict__ D = (v4u*)(dst + (size_t)row * stride);
        if (PERSIST) __syncthreads();
        for (u32 m = threadIdx.x; m < k; m += THREADS) s_idx[m] = lower_bound_u64(pos, L, bk[m]);
        __syncthreads();
        for (u32 q = threadIdx.x; q < chunks; q += THREADS * UNROLL) {
            v4u v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const u32 qq = q + u * THREADS;
                if (qq >= chunks) continue;
                const u32 bit0 = qq * 128u, bit1 = bit0 + 128u;
                u32 cnt = 0;
                for (u32 m = 0; m < k; m++) cnt += (s_idx[m] <= bit0);
                const u32 nxt = cnt < k ? s_idx[cnt] : 0xffffffffu;
                if (nxt >= bit1) {
                    const v4u* P = (cnt & 1) ? B : A;
                    v[u] = NTL ? __builtin_nontemporal_load(&P[qq]) : P[qq];
                } else {
                    const v4u a = A[qq], b = B[qq];
                    v4u mask = (cnt & 1) ? (v4u)(0xffffffffu) : (v4u)(0u);
                    for (u32 m = cnt; m < k; m++) {
                        const u32 id = s_idx[m];
                        if (id >= bit1) break;
                        const u32 rel = id - bit0;
                        v4u t;
                        t.x = rel <= 0 ? 0xffffffffu : (rel < 32 ? 0xffffffffu << rel : 0u);
                        t.y = rel <= 32 ? 0xffffffffu : (rel < 64 ? 0xffffffffu << (rel - 32) : 0u);
                        t.z = rel <= 64 ? 0xffffffffu : (rel < 96 ? 0xffffffffu << (rel - 64) : 0u);
                        t.w = rel <= 96 ? 0xffffffffu : (rel < 128 ? 0xffffffffu << (rel - 96) : 0u);
                        mask ^= t;
                    }
                    v[u] = (a & ~mask) | (b & mask);
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; u++) {
                const u32 qq = q + u * THREADS;
                if (qq >= chunks) continue;
                if (NTS) __builtin_nontemporal_store(v[u], &D[qq]); else D[qq] = v[u];
            }
        }
    }
}
// plain device copy for the ceiling: same bytes, no descriptors
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) copy_rows(v4u* __restrict__ dst, const v4u* __restrict__ src, size_t n16)
{
    for (size_t q = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; q < n16; q += (size_t)gridDim.x * 256 * UNROLL) {
        v4u v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) if (q + u * 256 < n16) v[u] = NT ? __builtin_nontemporal_load(&src[q + u * 256]) : src[q + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) if (q + u * 256 < n16) { if (NT) __builtin_nontemporal_store(v[u], &dst[q + u * 256]); else dst[q + u * 256] = v[u]; }
    }
}

int main(int argc, char** argv)
{
    const size_t N = argc > 1 ? atol(argv[1]) : 100000, L = argc > 2 ? atol(argv[2]) : 1000000;
    const size_t stride = ((L + 7) / 8 + 127) / 128 * 128, rows = 2 * N;
    const u32 chunks = (u32)(stride / 16);
    // one row pool as in the library: the parents are rows [0, rows), the offspring rows [rows, 2 rows)
    uint8_t* pool; CK(hipMalloc(&pool, 2 * rows * stride));
    uint8_t *src = pool, *dst = pool + rows * stride;
    CK(hipMemset(src, 0x5a, rows * stride)); CK(hipMemset(dst, 0, rows * stride));
    std::mt19937_64 rng(1);
    std::vector<u32> father(N), mother(N), k(rows), off(rows + 1); std::vector<uint8_t> start(rows); std::vector<u64> bk, pos(L);
    for (size_t i = 0; i < L; i++) pos[i] = i;
    std::poisson_distribution<int> pd(1.0);
    for (size_t i = 0; i < N; i++) { father[i] = rng() % N; mother[i] = rng() % N; }
    for (size_t g = 0; g < rows; g++) {
        k[g] = pd(rng); off[g] = (u32)bk.size(); start[g] = rng() & 1;
        std::vector<u64> b(k[g]); for (auto& x : b) x = rng() % L; std::sort(b.begin(), b.end()); bk.insert(bk.end(), b.begin(), b.end());
    }
    off[rows] = (u32)bk.size();
    SampleDev sd = {};
    u32 *dk, *doff, *df, *dm; u64 *dbk, *dpos; uint8_t* dst_;
    CK(hipMalloc(&dk, rows * 4)); CK(hipMalloc(&doff, (rows + 1) * 4)); CK(hipMalloc(&df, N * 4)); CK(hipMalloc(&dm, N * 4));
    CK(hipMalloc(&dbk, (bk.size() + 1) * 8)); CK(hipMalloc(&dpos, L * 8)); CK(hipMalloc(&dst_, rows));
    CK(hipMemcpy(dk, k.data(), rows * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(doff, off.data(), (rows + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(df, father.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dm, mother.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbk, bk.data(), bk.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dpos, pos.data(), L * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dst_, start.data(), rows, hipMemcpyHostToDevice));
    { std::vector<u32> bi(bk.begin(), bk.end()); bi.push_back(0); u32* dbi; CK(hipMalloc(&dbi, bi.size() * 4));   // pos[i] = i: index == base pair
      CK(hipMemcpy(dbi, bi.data(), bi.size() * 4, hipMemcpyHostToDevice)); sd.bk_idx = dbi; }
    sd.k = dk; sd.bk_off = doff; sd.bk = dbk; sd.start = dst_; sd.father = df; sd.mother = dm;
    // grouping by source individual for the parent-major kernel
    std::vector<u32> goff(N + 1, 0), glist(rows);
    for (size_t r = 0; r < rows; r++) goff[((r & 1) ? mother[r >> 1] : father[r >> 1]) + 1]++;
    for (size_t p = 0; p < N; p++) goff[p + 1] += goff[p];
    { std::vector<u32> cur(goff.begin(), goff.end() - 1); for (size_t r = 0; r < rows; r++) glist[cur[(r & 1) ? mother[r >> 1] : father[r >> 1]]++] = (u32)r; }
    u32 *dgoff, *dglist; CK(hipMalloc(&dgoff, (N + 1) * 4)); CK(hipMalloc(&dglist, rows * 4));
    CK(hipMemcpy(dgoff, goff.data(), (N + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dglist, glist.data(), rows * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double gbytes = (double)rows * stride * 2 / 1e9;   // algorithmic bytes (L/8 read + L/8 written per gamete)
    struct Var { const char* name; std::function<void()> run; std::vector<float> ms; };
    std::vector<Var> vars;
#define ADD(name, ...) vars.push_back({name, [&]() { __VA_ARGS__; }, {}})
    // one-entry work table (the library builds one entry per active chromosome)
    // slot -> row tables: parents identity; offspring either all fresh rows (every gamete copied) or, as the library does by
    // default, crossover-free gametes sharing the parental row (nothing copied for them)
    std::vector<u32> pc(rows), pa_copy(rows), pa_share(rows);
    size_t n_shared = 0;
    for (size_t r = 0; r < rows; r++) {
        pc[r] = (u32)r; pa_copy[r] = (u32)(rows + r);
        const u32 par = (r & 1) ? mother[r >> 1] : father[r >> 1];
        if (k[r] == 0) { pa_share[r] = 2 * par + start[r]; n_shared++; } else pa_share[r] = (u32)(rows + r);
    }
    u32 *dpc, *dpa_copy, *dpa_share;
    CK(hipMalloc(&dpc, rows * 4)); CK(hipMalloc(&dpa_copy, rows * 4)); CK(hipMalloc(&dpa_share, rows * 4));
    CK(hipMemcpy(dpc, pc.data(), rows * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dpa_copy, pa_copy.data(), rows * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dpa_share, pa_share.data(), rows * 4, hipMemcpyHostToDevice));
    ChrWork hw = {}; hw.pw.pool = pool; hw.pw.phys_cur = dpc; hw.pw.phys_alt = dpa_copy; hw.pw.alias = 0;
    hw.snp_pos = dpos; hw.stride = stride; hw.chunks = chunks; hw.bpr = 1; hw.L = (u32)L; hw.chr = 0;
    ChrWork* dw; CK(hipMalloc(&dw, sizeof hw)); CK(hipMemcpy(dw, &hw, sizeof hw, hipMemcpyHostToDevice));
    ChrWork hs = hw; hs.pw.phys_alt = dpa_share; hs.pw.alias = 1;
    ChrWork* dws; CK(hipMalloc(&dws, sizeof hs)); CK(hipMemcpy(dws, &hs, sizeof hs, hipMemcpyHostToDevice));
    ADD("rows (gamete-major)", hipLaunchKernelGGL(k_stitch_rows, dim3((unsigned)rows), dim3(256), 0, 0, dw, 1u, 1, sd));
#define PM(name, U, NT, occ) ADD(name, hipLaunchKernelGGL((k_stitch_parent<U, NT>), dim3((unsigned)N), dim3(256), (occ) >= 8 ? 0 : std::min(160 * 1024 / (occ) - 3 * 1024, 64 * 1024 - 2048), 0, dw, 1u, 1, dgoff, dglist, sd))
#define RG(name, U, NT, occ) ADD(name, hipLaunchKernelGGL((k_stitch_regions<U, NT>), dim3((unsigned)N), dim3(256), (occ) >= 8 ? 0 : std::min(160 * 1024 / (occ) - 6 * 1024, 64 * 1024 - 4096), 0, dw, 1u, 1, dgoff, dglist, sd, 0))
#define RGS(name, U) ADD(name, hipLaunchKernelGGL((k_stitch_regions<U, true>), dim3((unsigned)N), dim3(256), 0, 0, dws, 1u, 1, dgoff, dglist, sd, 0))
    RGS("regions U4 nt SHARED rows", 4);
    RGS("regions U2 nt SHARED rows", 2);
    PM("parent per-chunk U2 nt", 2, true, 8);
    RG("regions U2 nt", 2, true, 8);
    RG("regions U4 nt", 4, true, 8);
    RG("regions U4 nt occ6", 4, true, 6);
    RG("regions U4 nt occ4", 4, true, 4);
#define RGT(name, U, TH) ADD(name, hipLaunchKernelGGL((k_stitch_regions<U, true, TH>), dim3((unsigned)N), dim3(TH), 0, 0, dw, 1u, 1, dgoff, dglist, sd, 0))
    RGT("regions U2 nt 128thr", 2, 128);
    RGT("regions U4 nt 128thr", 4, 128);
    RGT("regions U8 nt 128thr", 8, 128);
    RGT("regions U4 nt 64thr", 4, 64);
    RGT("regions U8 nt 64thr", 8, 64);
    RG("regions U8 nt", 8, true, 8);
    RG("regions U8 nt occ4", 8, true, 4);
    RG("regions U4 plain", 4, false, 8);
    ADD("copy U4 nt g2048", hipLaunchKernelGGL((copy_rows<4, true>), dim3(2048), dim3(256), 0, 0, (v4u*)dst, (const v4u*)src, rows * stride / 16));
    ADD("copy U8 g8192", hipLaunchKernelGGL((copy_rows<8, false>), dim3(8192), dim3(256), 0, 0, (v4u*)dst, (const v4u*)src, rows * stride / 16));
    for (int round = 0; round < 6; round++)
        for (auto& v : vars) {
            CK(hipEventRecord(e0, 0)); v.run(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (round) v.ms.push_back(ms);
        }
    printf("N=%zu L=%zu rows=%zu stride=%zu  bytes moved per launch = %.2f GB (every gamete copied); SHARED: %zu of %zu gametes share the parental row, %.2f GB\n",
           N, L, rows, stride, gbytes, n_shared, rows, (double)(rows - n_shared) * stride * 2 / 1e9);
    for (auto& v : vars) {
        std::sort(v.ms.begin(), v.ms.end());
        printf("%-26s median %7.3f ms  min %7.3f ms  -> %7.1f GB/s (median)\n", v.name, v.ms[v.ms.size() / 2], v.ms[0], gbytes / (v.ms[v.ms.size() / 2] * 1e-3));
    }
    return 0;
}
