// tools/stitch_bench.hip -- A/B harness for the dense stitch (k_stitch_segments) on synthetic descriptors of config-2 shape,
// interleaved rounds in ONE process (guide rule 24), next to plain copy kernels of the same byte counts.
//   hipcc --offload-arch=gfx950 -O3 -I. tools/stitch_bench.hip -o tools/stitch_bench && tools/stitch_bench [N] [L] [seg_chunks]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <random>
#include <vector>
#include "../geneevolve_amd/csrc/gev_kernels.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

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
    const u32 SC = argc > 3 ? (u32)atoi(argv[3]) : 1024;
    u32 sh = 0; while ((1u << (sh + 1)) <= SC) sh++;
    const size_t stride = ((L + 7) / 8 + 127) / 128 * 128, rows = 2 * N;
    const u32 chunks = (u32)(stride / 16), S = (chunks + (1u << sh) - 1) >> sh;
    const size_t unit = (size_t)16 << sh, units_gen = rows * S;
    // one pool as in the library: the parents are units [0, units_gen) (row r = units r*S ..), fresh units follow
    uint8_t* pool; CK(hipMalloc(&pool, 2 * units_gen * unit));
    CK(hipMemset(pool, 0x5a, units_gen * unit)); CK(hipMemset(pool + units_gen * unit, 0, units_gen * unit));
    std::mt19937_64 rng(1);
    std::vector<u32> father(N), mother(N), k(rows), off(rows + 1), bidx; std::vector<uint8_t> start(rows);
    std::poisson_distribution<int> pd(1.0);
    for (size_t i = 0; i < N; i++) { father[i] = rng() % N; mother[i] = rng() % N; }
    for (size_t g = 0; g < rows; g++) {
        k[g] = pd(rng); off[g] = (u32)bidx.size(); start[g] = rng() & 1;
        std::vector<u32> b(k[g]); for (auto& x : b) x = (u32)(rng() % L); std::sort(b.begin(), b.end()); bidx.insert(bidx.end(), b.begin(), b.end());
    }
    off[rows] = (u32)bidx.size(); bidx.push_back(0);
    // tables: parents identity; offspring (a) every segment written, (b) only the segments that hold a boundary
    std::vector<u32> pc(units_gen), pa_all(units_gen), pa_sh(units_gen), it_all, it_sh;
    size_t fresh = 0;
    for (size_t e = 0; e < units_gen; e++) { pc[e] = (u32)e; pa_all[e] = (u32)(units_gen + e); it_all.push_back((u32)e); }
    for (size_t r = 0; r < rows; r++) {
        const u32 par = (r & 1) ? mother[r >> 1] : father[r >> 1];
        for (u32 g = 0; g < S; g++) {
            bool own = false; u32 cnt = 0;
            for (u32 m = 0; m < k[r]; m++) { const u32 id = bidx[off[r] + m]; if (((id >> 7) >> sh) == g) own = true; if (id <= ((g << sh) << 7)) cnt++; }
            if (own) { pa_sh[r * S + g] = (u32)(units_gen + fresh++); it_sh.push_back((u32)(r * S + g)); }
            else pa_sh[r * S + g] = (u32)((2 * (size_t)par + ((start[r] ^ cnt) & 1u)) * S + g);
        }
    }
    const size_t cap = units_gen;
    it_all.resize(cap + 1); it_all[cap] = (u32)units_gen; const size_t n_sh = it_sh.size(); it_sh.resize(cap + 1); it_sh[cap] = (u32)n_sh;
    auto up = [](const void* h, size_t bytes) { void* d; CK(hipMalloc(&d, bytes)); CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); return d; };
    SampleDev sd = {};
    sd.k = (u32*)up(k.data(), rows * 4); sd.bk_off = (u32*)up(off.data(), (rows + 1) * 4); sd.bk_idx = (u32*)up(bidx.data(), bidx.size() * 4);
    sd.start = (uint8_t*)up(start.data(), rows); sd.father = (u32*)up(father.data(), N * 4); sd.mother = (u32*)up(mother.data(), N * 4);
    ChrWork hw = {};
    hw.pw.pool = pool; hw.pw.phys_cur = (u32*)up(pc.data(), units_gen * 4); hw.pw.nseg = S; hw.pw.seg_shift = sh; hw.pw.items_cap = (u32)cap;
    hw.stride = stride; hw.chunks = chunks; hw.L = (u32)L; hw.chr = 0;
    ChrWork ha = hw; ha.pw.phys_alt = (u32*)up(pa_all.data(), units_gen * 4); ha.pw.items = (u32*)up(it_all.data(), (cap + 1) * 4); ha.pw.alias = 0;
    ChrWork hs = hw; hs.pw.phys_alt = (u32*)up(pa_sh.data(), units_gen * 4); hs.pw.items = (u32*)up(it_sh.data(), (cap + 1) * 4); hs.pw.alias = 1;
    ChrWork* dwa = (ChrWork*)up(&ha, sizeof ha); ChrWork* dws = (ChrWork*)up(&hs, sizeof hs);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Var { const char* name; std::function<void()> run; double gbytes; std::vector<float> ms; };
    std::vector<Var> vars;
    const double gb_all = (double)rows * stride * 2 / 1e9;
    double gb_sh = 0; for (size_t i = 0; i < n_sh; i++) { const u32 g = it_sh[i] % S; gb_sh += 2.0 * 16 * std::min<u32>(1u << sh, chunks - (g << sh)); } gb_sh /= 1e9;
#define ADD(name, gb, ...) vars.push_back({name, [&]() { __VA_ARGS__; }, gb, {}})
    ADD("segments with a boundary, grid 16384", gb_sh, hipLaunchKernelGGL((k_stitch_segments<true>), dim3(16384), dim3(256), 0, 0, dws, 1, sd));
    ADD("segments with a boundary, U8", gb_sh, hipLaunchKernelGGL((k_stitch_segments<true, 8>), dim3(16384), dim3(256), 0, 0, dws, 1, sd));
    ADD("segments with a boundary, U4", gb_sh, hipLaunchKernelGGL((k_stitch_segments<true, 4>), dim3(16384), dim3(256), 0, 0, dws, 1, sd));
    ADD("segments with a boundary, U1", gb_sh, hipLaunchKernelGGL((k_stitch_segments<true, 1>), dim3(16384), dim3(256), 0, 0, dws, 1, sd));
    ADD("segments with a boundary, U2 grid 32768", gb_sh, hipLaunchKernelGGL((k_stitch_segments<true, 2>), dim3(32768), dim3(256), 0, 0, dws, 1, sd));
    ADD("segments with a boundary, grid 4096", gb_sh, hipLaunchKernelGGL((k_stitch_segments<true>), dim3(4096), dim3(256), 0, 0, dws, 1, sd));
    ADD("segments with a boundary, plain ld/st", gb_sh, hipLaunchKernelGGL((k_stitch_segments<false>), dim3(16384), dim3(256), 0, 0, dws, 1, sd));
    ADD("gamete-major rows, own segments only", gb_sh, hipLaunchKernelGGL(k_stitch_rows, dim3((unsigned)rows), dim3(256), 0, 0, dws, 1, sd));
    ADD("every segment written", gb_all, hipLaunchKernelGGL((k_stitch_segments<true>), dim3(16384), dim3(256), 0, 0, dwa, 1, sd));
    ADD("copy U4 nt g2048 (whole rows)", gb_all, hipLaunchKernelGGL((copy_rows<4, true>), dim3(2048), dim3(256), 0, 0, (v4u*)(pool + units_gen * unit), (const v4u*)pool, rows * stride / 16));
    for (int round = 0; round < 6; round++)
        for (auto& v : vars) {
            CK(hipEventRecord(e0, 0)); v.run(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (round) v.ms.push_back(ms);
        }
    printf("N=%zu L=%zu rows=%zu stride=%zu, %u segments of %zu B per row: %zu of %zu segments hold a boundary (%.2f GB read+written; every segment: %.2f GB)\n",
           N, L, rows, stride, S, unit, n_sh, units_gen, gb_sh, gb_all);
    for (auto& v : vars) {
        std::sort(v.ms.begin(), v.ms.end());
        printf("%-40s median %7.3f ms  min %7.3f ms  -> %7.1f GB/s (median)\n", v.name, v.ms[v.ms.size() / 2], v.ms[0], v.gbytes / (v.ms[v.ms.size() / 2] * 1e-3));
    }
    return 0;
}
