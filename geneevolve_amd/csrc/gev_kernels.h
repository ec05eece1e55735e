// gev_kernels.h -- hand-written HIP kernels (gfx950 / CDNA4, wave64) of the reproduction hot path.
// Included once by gev_library.hip.  Every kernel names the reference code it replaces
// (MMesbahU/GeneEvolve, paths relative to the reference root).
//
// Data layout in HBM (one population, one chromosome):
//   genotype plane   uint8 [2*n_people][stride]   row = 2*individual + chromatid, locus ii = bit
//                    (ii&7) of byte ii>>3 (== bit ii&63 of LE word ii>>6); stride % 128 == 0.
//                    Holds FOUNDER alleles only; mutations are a sparse per-row overlay because
//                    the reference treats part::mutation_pos as a set (flip applied at read
//                    time, src/Simulation.cpp:1218-1222), not as an in-place toggle.
//   CV plane         same packing on the (sorted) CV grid, 1 + ceil(log2(n_pop)) sub-rows per
//                    row: the CV alleles AS ras_find_cv RESOLVES THEM (founder allele, !founder where
//                    the CV position is in the mutation set of the covering part, :2766-2775), then
//                    root-population id bits (a/d are looked up in the root population, :2778-2779).
//                    Both are inherited by the crossover pattern (recombine copies a part with its
//                    mutation_pos, modify_part_for_mutation_pos keeps the positions inside the
//                    trimmed part); a NEW mutation on a CV position flips the allele unless the
//                    position already is in the covering part's set (k_cv_newmut looks it up in the
//                    parental list).  A/D reads this plane only -- it never scans the mutation
//                    lists, whose length grows with the age of the run
//   mutation lists / interval lists: per (row, position range) pieces shared between parent and offspring (gev_lists.h);
//                    the CSR form -- off[2n+1] (u32) + pos[] (u64, ascending per row) / gev_part[] (the reference's own
//                    state, --out_interval) -- is made from the pieces when somebody asks for whole lists
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/geneevolve_amd.h"
#include "rng_device.h"

typedef uint32_t u32;
typedef uint64_t u64;

// static per-chromosome tables, resident on the device
struct ChrDev {
    const GevThr* rthr;   // [R]  recombination thresholds        (_recom_prob)
    const u64*    rbp;    // [R]  rMap::bp
    const GevThr* mthr;   // [M]  mutation thresholds             (MutationMap::mutation_rate)
    const u64*    mbp;    // [M]  MutationMap::bp
    u64 bp_dist;          // rMap::bp_dist_in_rmap
    u64 bp0, bp_end;      // rmap.bp[0], rmap.bp[R-1]: every haplotype covers [bp0, bp_end)
    u32 R, M;
    u32 r_amax, m_amax;   // largest a_hi of the recombination / mutation thresholds (scan prefilter)
    const u64* snp_pos;   // [L] Legend.pos of the genotype plane
    u32 L, active;        // active == 0: this context holds no genotype / CV state of the chromosome (locus-split populations)
    // coarse index over snp_pos: coarse[j] = first locus at or behind position coarse_base + (j << coarse_shift), j = 0 .. coarse_n:
    // a breakpoint becomes a locus index with one table read and a search over the few loci of its bucket (snp_lower_bound)
    const u32* coarse; u64 coarse_base; u32 coarse_shift, coarse_n;
};
// first locus index whose position is >= x (== lower_bound over the chromosome's SNP positions)
__device__ __forceinline__ u32 snp_lower_bound(const ChrDev& C, u64 x)
{
    if (!C.L || x <= C.coarse_base) return 0u;                    // coarse_base = position of the first locus
    const u64 j = (x - C.coarse_base) >> C.coarse_shift;
    if (j >= C.coarse_n) return C.L;                              // behind the last locus
    u32 lo = C.coarse[j], hi = C.coarse[j + 1];                   // the answer lies in [lo, hi]
    while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (C.snp_pos[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}

// per-generation sampling results (all chromosomes; task t = offspring*nchr + chr, gamete G = 2t+s)
struct SampleDev {
    u32* seed_pat;   // [T+1] srand seed of the paternal gamete of task t (src/Simulation.cpp:2447)
    u32* seed_mat;   // [T]   (:2453)
    u32* k;          // [2T]  crossover count of gamete G
    u32* bk_off;     // [2T+1] exclusive scan of k
    u64* bk;         // breakpoints, ascending per gamete (:2990)
    u32* bk_idx;     // same layout: first locus index of the genotype plane at or after each breakpoint
    uint8_t* start;  // [2T]  starting haplotype (:2449, :2455)
    u32* nmut;       // [T]   new mutations of task t (:2513)
    u32* nm_off;     // [T+1]
    u64* nm_pos;     // bp_mut (:2520)
    uint8_t* nm_side;// h01 (:2522)
    uint8_t* sex;    // [n_people] (:2472)
    const u32* father;  // [n_people] position of the father in the parent generation (:2438)
    const u32* mother;  // [n_people]
    // Records are written in ONE pass: gamete G owns slots [G*GEV_BK_CAP, +GEV_BK_CAP) of bk and task t
    // owns [t*GEV_NM_CAP, +GEV_NM_CAP) of nm_pos/nm_side; a gamete/task with more entries gets a range of the
    // overflow region (bump-allocated with one atomic), bk_off/nm_off point at whichever was used.
    u32 bk_ovf_base, bk_ovf_cap, nm_ovf_base, nm_ovf_cap;
    u32* status;     // GenStatus words (below)
};
// Per-generation work tables: one entry per ACTIVE chromosome (ChrWork) / per (phenotype, active chromosome) (CvWork, AdWork).
// Every per-chromosome kernel of a generation is ONE launch whose blockIdx.y (or .z) selects the entry, instead of one launch
// per chromosome (a 22-chromosome genome used to cost ~300 launches per generation).
// Genotype rows live in ONE pool of SEGMENTS per (population, chromosome).  A haplotype row (slot s = 2*individual + chromatid)
// is cut into nseg segments of 2^seg_shift 16-byte chunks (2 KiB by default); segment g of slot s is pool unit
// phys[s * nseg + g].  In a segment that contains none of its crossover boundaries an offspring gamete IS one parental haplotype
// (Simulation::recombine copies the parent's parts unchanged between two crossovers, src/Simulation.cpp:2939-2946; without any
// crossover it returns the parental Hap itself, :2910): that segment of the offspring slot then names the parent's unit and no
// byte is copied.  Only the segments that contain a boundary get a unit of their own -- one that no segment of the parents'
// generation names (free list, rebuilt from the parents' table at the start of every generation: mark, collect) -- and are
// written by the dense stitch.  Units are written once and never modified (mutations are a sparse overlay of the slot).
// Physical placement is arbitrary (atomics) and invisible: every consumer goes through the table (RowMap).
// One entry of the dense stitch's work list = one segment of one offspring row that contains a crossover boundary, complete: the
// stitch wave starts from ONE 32-byte uniform load (the first form read five dependent descriptor words per item).
struct __attribute__((aligned(32))) StitchItem {
    u32 src0, src1, dst;      // units of the parent's two haplotypes in this segment, unit to write
    u32 meta;                 // bit 0: haplotype copied at the segment's first locus; bits 1-3: boundaries inside (0..STITCH_ITEM_B), 7 = more: b[0] = output row, the gamete's list is read; bits 8-15: segment
    u32 b[4];                 // the boundaries inside, as bit offsets from the segment's first locus, ascending
};
#define STITCH_ITEM_B 4
struct PoolWork {
    uint8_t* pool; const u32* phys_cur; u32* phys_alt;      // units; (slot, segment) -> unit of the parents / of the offspring
    u32* live; u32* freel;                                  // [pool_units] marks, [pool_units] free units
    // pctr: {0: length of the free list, 1: cursor = units taken from it since it was built, 2: (pool_take) exhausted, 3: last segments
    // taken this generation, 4: cursor at the start of this generation}.  The free list is NOT rebuilt every generation: units
    // that no table named when it was built and that were not handed out since are still unnamed, so the list stays valid and
    // only shrinks; the host rebuilds it (mark + collect) when it runs low.
    u32* pctr;
    StitchItem* items;                                      // work list of the generation; ((u32*)&items[items_cap])[0] = its length
    u32 pool_units, alias, stamp, nseg, seg_shift, items_cap;
};
// read access to the rows of one generation
struct RowMap {
    const uint8_t* pool; const u32* phys; u32 nseg, seg_shift;
    __device__ __forceinline__ const uint4* chunk(size_t slot, u32 q) const
    {
        const u32 unit = phys[slot * nseg + (q >> seg_shift)];
        return (const uint4*)(pool + ((size_t)unit << (seg_shift + 4))) + (q & ((1u << seg_shift) - 1u));
    }
    __device__ __forceinline__ u32 word32(size_t slot, u32 w) const { return ((const u32*)chunk(slot, w >> 2))[w & 3u]; }
    __device__ __forceinline__ u64 word64(size_t slot, u32 w) const { return ((const u64*)chunk(slot, w >> 1))[w & 1u]; }
};
#include "gev_lists.h"
struct ChrWork {
    LpWork lp;                                                                              // mutation lists and ancestry intervals: shared pieces (gev_lists.h)
    PoolWork pw; const u64* snp_pos;                                                        // genotype rows
    size_t stride;                                                                          // bytes of a whole row (host-side layouts)
    u64 bp0, bp_end;
    u32 chunks, bpr, L;
    int chr;
};
struct CvWork {
    u32* cvp_alt; const u32* cvp_cur; const u64* pos_sorted; u32* counts;      // counts: allele count per CV column (k_cv_sum_partials / k_cv_count / k_cv_newmut add, k_cv_table / k_cv_freq consume and clear)
    uint16_t* partial;                                                        // [blocks of k_stitch_small][1024] allele counts of the block's rows per column
    u64 bp0, bp_end;
    u32 stride_w32, sub_w32, C;
    int chr;
    u32 cw;                                                                   // entry of the ChrWork table that holds this chromosome's lists
};
struct AdWork {
    const u32* cvp; const u64* pos_sorted; const u64* pos_file; const u32* col_of_icv;
    const double* a; const double* d; const double* const* aptr; const double* const* dptr;
    u32* counts; double* frq; double* tab; double* add_out; double* dom_out;
    double* add_tot; double* dom_tot;   // one chromosome: Human::additive / dominance of this phenotype written with the per-chromosome values (0 + x, the sum over chromosomes of :2729-2746); else null
    const uint16_t* partial;            // k_stitch_small's per-block allele counts of this CV grid, or null
    u64 bp0, bp_end;
    double vd;
    u32 stride_w32, sub_w32, C;
    int own_pop;
    u32 cols_sorted;      // the CV file is already in position order: CV j of the file is bit j of the rows
};
#define GEV_BK_CAP 8
#define GEV_NM_CAP 8
// status words written by the kernels of one generation, read back once at its end
enum { ST_BK_OVF_USED = 0, ST_NM_OVF_USED = 1, ST_FLAGS = 2, ST_SLOW_MUT = 3, ST_SLOW_REC = 4 /* tasks handed to the one-task-per-wave kernels */,
       ST_GLOB_STATE = 5 /* glob_generator behind the generation's ras_glob_seed() draws (gev_generation_begin) */, ST_NM_MATE = 6, ST_NF_MATE = 7 /* num_males_mate, num_females_mate */,
       ST_NEXT_STATE = 8 /* ... and behind the draws the host announced it makes before the next generation (gev_set_generation_chain) */,
       ST_TOTALS = 16 /* then per chr: mut_total, parts_total, segments the dense stitch writes, how many of them are last (partial) segments, free list length, free list cursor */ };
#define ST_PER_CHR 6
enum { FLAG_BK_OVF = 1, FLAG_NM_OVF = 2, FLAG_MUT_CAP = 4, FLAG_PARTS_CAP = 8, FLAG_POOL = 16,
       FLAG_RNG_SHORT = 32 /* a rejection stream ran out of candidates (internal) */, FLAG_NO_MATES = 64 /* "No one can marry", src/Simulation.cpp:2125 */ };
#define FLAG_REDO_MASK (FLAG_BK_OVF | FLAG_NM_OVF | FLAG_MUT_CAP | FLAG_PARTS_CAP)      // capacities the host grows before it enqueues the generation again

// The RNG tables (16 KB) are read 31 words at a time for every srand(); under a concurrently
// running HBM-saturating stitch a dependent global read costs microseconds, so every sampling
// workgroup stages them into LDS once and loops over many tasks (persistent grid).
__device__ __forceinline__ const GevRngTables* stage_tables(const GevRngTables* __restrict__ g, GevRngTables* s)
{
    const uint4* src = (const uint4*)g; uint4* dst = (uint4*)s;
    for (u32 i = threadIdx.x; i < sizeof(GevRngTables) / 16; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
    return s;
}
#define SAMPLE_GRID_MAX 1024      // persistent sampling workgroups (4 waves each): half the wave slots, the rest stays with the concurrent stitch

// ------------------------------------------------------------------------------------------
// exclusive scan of u32 counts (CSR offsets); n+1 outputs
// ------------------------------------------------------------------------------------------
#define SCAN_ITEMS 1024
__device__ __forceinline__ u32 block_exclusive_scan_256(u32 v, u32* lds /*>=8*/, u32& block_total)
{
    const u32 lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    u32 inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { u32 o = __shfl_up(inc, d); if (lane >= (u32)d) inc += o; }
    if (lane == 63) lds[wid] = inc;
    __syncthreads();
    u32 wave_off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { u32 s = lds[w]; if ((u32)w < wid) wave_off += s; tot += s; }
    __syncthreads();
    block_total = tot;
    return wave_off + inc - v;
}
// blockIdx.y = segment: `in` advances by in_stride, `sums` by sums_stride per segment (single scan: gridDim.y = 1)
__global__ void __launch_bounds__(256) k_scan_partial(const u32* __restrict__ in, size_t n, u32* __restrict__ sums, size_t in_stride = 0, size_t sums_stride = 0)
{
    __shared__ u32 lds[8];
    in += blockIdx.y * in_stride; sums += blockIdx.y * sums_stride;
    const size_t base = (size_t)blockIdx.x * SCAN_ITEMS + threadIdx.x * 4;
    u32 s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) if (base + j < n) s += in[base + j];
    u32 tot; block_exclusive_scan_256(s, lds, tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(256) k_scan_sums(u32* __restrict__ sums, size_t nb, size_t sums_stride = 0)
{
    __shared__ u32 lds[8];
    sums += blockIdx.y * sums_stride;
    u32 carry = 0;
    for (size_t base = 0; base < nb; base += 256) {
        const size_t i = base + threadIdx.x;
        u32 v = i < nb ? sums[i] : 0, tot;
        u32 ex = block_exclusive_scan_256(v, lds, tot);
        if (i < nb) sums[i] = carry + ex;
        carry += tot;
    }
}
// slots 0..n (slot n counts as 0 and receives the total): launch with ceil((n+1)/SCAN_ITEMS) blocks
__global__ void __launch_bounds__(256) k_scan_final(const u32* __restrict__ in, size_t n, const u32* __restrict__ sums, u32* __restrict__ out)
{
    __shared__ u32 lds[8];
    const size_t base = (size_t)blockIdx.x * SCAN_ITEMS + threadIdx.x * 4;
    u32 v[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { v[j] = base + j < n ? in[base + j] : 0; s += v[j]; }
    u32 tot; u32 ex = block_exclusive_scan_256(s, lds, tot) + sums[blockIdx.x];
#pragma unroll
    for (int j = 0; j < 4; j++) { if (base + j <= n) out[base + j] = ex; ex += v[j]; }
}

// ------------------------------------------------------------------------------------------
// synthetic founders (tests/synth.py is the specification) and gen-0 masking
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 mix64(u64 x)
{
    u64 z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void k_synth_thresholds(u32* __restrict__ thr, size_t L, u64 seed)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L) return;
    const u64 u = mix64(seed ^ 0xA5A5A5A5A5A5A5A5ull ^ ((u64)i * 0x9E3779B97F4A7C15ull)) >> 32;
    thr[i] = (u32)(214748365ull + ((u * 1932735283ull) >> 32));
}
// one thread = one 64-bit word of one row
__global__ void k_synth_rows(u64* __restrict__ plane, size_t stride_w64, size_t nrows, size_t L, const u32* __restrict__ thr, u64 seed)
{
    const size_t words = (L + 63) / 64;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nrows * words) return;
    const size_t r = q / words, w = q % words;
    const u64 off = seed * 0xD1342543DE82EF95ull;
    u64 bits = 0;
    for (u32 b = 0; b < 64; b++) {
        const size_t i = w * 64 + b;
        if (i >= L) break;
        const u64 ctr = (((u64)r << 32) | (u64)i) + off;
        if ((u32)(mix64(ctr) >> 32) < thr[i]) bits |= 1ull << b;
    }
    plane[r * stride_w64 + w] = bits;
}
// Gen-0 haplotypes cover [bp0, bp_end) only (src/Simulation.cpp:3029-3034): loci outside that
// range read 0 in every materialisation (:1210).  Zero them once; stitching preserves zeros.
__global__ void k_mask_rows(u32* __restrict__ plane, size_t stride_w32, size_t nrows, u32 lo, u32 hi)
{
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nrows * stride_w32) return;
    const size_t w = q % stride_w32;
    const u32 b0 = (u32)w * 32, b1 = b0 + 32;
    u32 m = 0xffffffffu;
    if (b1 <= lo || b0 >= hi) m = 0;
    else {
        if (b0 < lo) m &= 0xffffffffu << (lo - b0);
        if (b1 > hi) m &= 0xffffffffu >> (b1 - hi);
    }
    if (m != 0xffffffffu) plane[q] &= m;
}
__global__ void k_fill_u32(u32* p, size_t n, u32 v)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
// gen-0 interval state: one whole-chromosome part per haplotype (src/Simulation.cpp:3031-3034)
__global__ void k_init_parts(gev_part* __restrict__ parts, u32* __restrict__ off, size_t nrows, u64 bp0, u64 bp_end, int pop)
{
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nrows) return;
    off[r] = (u32)r;
    if (r < nrows) parts[r] = gev_part{bp0, bp_end, (u64)r, pop, 0};
}
// sex[i] = rand()%2+1 for i = 0..n-1 after srand(seed) (src/Simulation.cpp:3005, 3036). One wave.
__global__ void __launch_bounds__(64) k_sex_sequence(const GevRngTables* __restrict__ Tg, u32 seed, size_t n, uint8_t* __restrict__ sex)
{
    __shared__ __attribute__((aligned(16))) GevRngTables s_T;
    const GevRngTables* T = stage_tables(Tg, &s_T);
    GlibcWave g; g.seed(T, seed);
    const u32 lane = threadIdx.x;
    for (size_t base = 0; base < n; base += 64) {
        if (base) g.next_block(T);
        if (base + lane < n) sex[base + lane] = (uint8_t)(((g.x >> 1) & 1) + 1);
    }
}

// ------------------------------------------------------------------------------------------
// K2: mutation sampling  == Simulation::ras_add_mutation (src/Simulation.cpp:2497-2552)
// one wave per (offspring, chromosome) task
// ------------------------------------------------------------------------------------------
// one scan of the mutation map; the first `cap` hits are resolved (position, side) into `pos/side`
template <class G>
__device__ __forceinline__ u32 mut_scan_write(const GevRngTables* __restrict__ T, const ChrDev& C, u32 S, G& g, u64* __restrict__ pos, uint8_t* __restrict__ side, u32 cap)
{
    const u32 lane = threadIdx.x & 63;
    u32 xg = minstd_seed(S + 1u);                                // generator(seed+1), :2503
    u32 h = 0;
    if (C.M >= 2)
        wave_scan_hits(T, S + 2u, C.mthr, C.m_amax, 1, C.M - 1, [&](u32 row) {                       // generator_u(seed+2), i = 1..M-1
            if (h < cap) {
                const u64 bp_mut = uniform_int_fallback(xg, C.mbp[row - 1], C.mbp[row]);    // :2516-2520
                const u32 sd_ = g.out(T, h) & 1u;                                            // rand()%2, :2522
                if (lane == 0) { pos[h] = bp_mut; side[h] = (uint8_t)sd_; }
            }
            h++;
        });
    return h;
}
// `list` != NULL: only the tasks list[0 .. *n_list) (the ones the batched kernel k_mut_sample8 handed over); NULL: every task
__global__ void __launch_bounds__(256) k_mut_sample(const GevRngTables* __restrict__ Tg, const ChrDev* __restrict__ chrs, int nchr,
                                                    const u32* __restrict__ mut_seeds, size_t n_tasks, SampleDev sd,
                                                    const u32* __restrict__ list, const u32* __restrict__ n_list)
{
    __shared__ __attribute__((aligned(16))) GevRngTables s_T;
    const size_t n_iter = list ? (size_t)*n_list : n_tasks;
    if (n_iter == 0) return;
    const GevRngTables* T = stage_tables(Tg, &s_T);
    const u32 lane = threadIdx.x & 63;
    for (size_t it = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); it < n_iter; it += (size_t)gridDim.x * 4) {
    const size_t t = list ? (size_t)list[it] : it;
    asm volatile("" : "+v"(T));          // keep the 62 table words of this lane out of registers across tasks (occupancy)
    const int c = (int)(t % nchr);
    const ChrDev& C = chrs[c];
    const u32 S = mut_seeds[t];
    GlibcWave g; g.seed(T, S);                                   // srand(seed), :2501
    u32 off = (u32)t * GEV_NM_CAP;
    u32 n = mut_scan_write(T, C, S, g, sd.nm_pos + off, sd.nm_side + off, GEV_NM_CAP);
    u32 n_store = n;
    if (n > GEV_NM_CAP) {                                        // rare: take a range of the overflow region and redo the scan
        u32 o = 0;
        if (lane == 0) o = atomicAdd(&sd.status[ST_NM_OVF_USED], n);
        o = rl_u32(o, 0);
        if (o + n <= sd.nm_ovf_cap) {
            off = sd.nm_ovf_base + o;
            g.seed(T, S);
            mut_scan_write(T, C, S, g, sd.nm_pos + off, sd.nm_side + off, n);
        } else { if (lane == 0) atomicOr(&sd.status[ST_FLAGS], (u32)FLAG_NM_OVF); n_store = 0; }     // host grows the region and redoes the generation
    }
    if (lane == 0) { sd.nmut[t] = n_store; sd.nm_off[t] = off; }
    if (c == nchr - 1) {                                         // sex = rand()%2+1 after the last chromosome, :2472
        const u32 sx = (g.out(T, n) & 1u) + 1u;
        if (lane == 0) sd.sex[t / nchr] = (uint8_t)sx;
        n++;
    }
    const u32 nxt = g.out(T, n);                                 // seed_loc of the next task, :2447
    if (lane == 0) sd.seed_pat[t + 1] = nxt;
    }
}

// ------------------------------------------------------------------------------------------
// K1/K3: crossover sampling == Simulation::ras_sim_loc_rec (:2973-2995) and the rand() chain of
// Simulation::reproduce (:2447-2455)
// ------------------------------------------------------------------------------------------
// one scan of the recombination map for one gamete; the first `cap` breakpoints bp[j] + rand()%dist (:2990) go to `out`
__device__ __forceinline__ u32 rec_scan_write(const GevRngTables* __restrict__ T, const ChrDev& C, u32 seed, GlibcWave& g, u64* __restrict__ out, u32* __restrict__ out_idx, u32 cap)
{
    const u32 lane = threadIdx.x & 63;
    u32 h = 0;
    wave_scan_hits(T, seed + 1u, C.rthr, C.r_amax, 0, C.R, [&](u32 row) {
        if (h < cap) {
            const u64 v = C.rbp[row] + (u64)g.out(T, h) % C.bp_dist;
            if (lane == 0) { out[h] = v; out_idx[h] = snp_lower_bound(C, v); }     // the breakpoint and the first locus at or behind it
        }
        h++;
    });
    return h;
}
// ras_sim_loc_rec for gamete G: srand(seed), scan, breakpoints; returns k (the true crossover count) with g
// positioned so that g.out(k) is the rand() that follows the call (:2449 / :2455)
__device__ __forceinline__ u32 gamete_sample(const GevRngTables* __restrict__ T, const ChrDev& C, u32 seed, size_t G, GlibcWave& g, const SampleDev& sd)
{
    const u32 lane = threadIdx.x & 63;
    g.seed(T, seed);
    u32 off = (u32)G * GEV_BK_CAP;
    const u32 k = rec_scan_write(T, C, seed, g, sd.bk + off, sd.bk_idx + off, GEV_BK_CAP);
    u32 k_store = k;
    if (k > GEV_BK_CAP) {
        u32 o = 0;
        if (lane == 0) o = atomicAdd(&sd.status[ST_BK_OVF_USED], k);
        o = rl_u32(o, 0);
        if (o + k <= sd.bk_ovf_cap) {
            off = sd.bk_ovf_base + o;
            g.seed(T, seed);
            rec_scan_write(T, C, seed, g, sd.bk + off, sd.bk_idx + off, k);
        } else { if (lane == 0) atomicOr(&sd.status[ST_FLAGS], (u32)FLAG_BK_OVF); k_store = 0; }
    }
    if (lane == 0) { sd.k[G] = k_store; sd.bk_off[G] = off; }
    return k;
}
// one (offspring, chromosome) task: paternal then maternal gamete (:2447-2456); leaves g seeded with seed_mat,
// the next rand() of the reference is g.out(k_mat + 1)
__device__ __forceinline__ u32 task_sample(const GevRngTables* __restrict__ T, const ChrDev& C, u32 seed_pat, size_t t, GlibcWave& g, const SampleDev& sd)
{
    const u32 lane = threadIdx.x & 63;
    const u32 k_pat = gamete_sample(T, C, seed_pat, 2 * t, g, sd);
    const u32 start_pat = g.out(T, k_pat) & 1u;       // rand()%2 after k_pat position draws, :2449
    const u32 seed_mat = g.out(T, k_pat + 1);         // :2453
    const u32 k_mat = gamete_sample(T, C, seed_mat, 2 * t + 1, g, sd);
    const u32 start_mat = g.out(T, k_mat) & 1u;       // :2455
    if (lane == 0) {
        sd.seed_pat[t] = seed_pat; sd.seed_mat[t] = seed_mat;
        sd.start[2 * t] = (uint8_t)start_pat; sd.start[2 * t + 1] = (uint8_t)start_mat;
    }
    return k_mat;
}
// task-parallel form (a mutation map is loaded: every task's chain restarts at srand(S), see
// SURVEY.md section 7.2-1).  seed_pat[t] for t>0 was written by k_mut_sample.
__global__ void __launch_bounds__(256) k_rec_sample(const GevRngTables* __restrict__ Tg, const ChrDev* __restrict__ chrs, int nchr,
                                                    u32 seed_reproduce, const u32* __restrict__ seed_ptr /* device copy of the seed (gev_generation_begin), or null */, size_t n_tasks, SampleDev sd,
                                                    const u32* __restrict__ list, const u32* __restrict__ n_list)
{
    __shared__ __attribute__((aligned(16))) GevRngTables s_T;
    const size_t n_iter = list ? (size_t)*n_list : n_tasks;
    if (n_iter == 0) return;
    const GevRngTables* T = stage_tables(Tg, &s_T);
    for (size_t it = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); it < n_iter; it += (size_t)gridDim.x * 4) {
        const size_t t = list ? (size_t)list[it] : it;
        asm volatile("" : "+v"(T));      // keep the 62 table words of this lane out of registers across tasks (occupancy)
        GlibcWave g;
        u32 seed_pat;
        if (t == 0) { g.seed(T, seed_ptr ? *seed_ptr : seed_reproduce); seed_pat = g.out(T, 0); }     // srand(seed) :2400, first rand() :2447
        else seed_pat = sd.seed_pat[t];
        const ChrDev& C = chrs[t % nchr];
        if (!C.active) {                 // another context owns this chromosome: only the seed chain (k_mut_sample) is needed here
            if ((threadIdx.x & 63) == 0) {
                sd.k[2 * t] = 0; sd.k[2 * t + 1] = 0; sd.bk_off[2 * t] = (u32)(2 * t) * GEV_BK_CAP; sd.bk_off[2 * t + 1] = (u32)(2 * t + 1) * GEV_BK_CAP;
                sd.start[2 * t] = 0; sd.start[2 * t + 1] = 0; sd.seed_mat[t] = 0;
            }
            continue;
        }
        task_sample(T, C, seed_pat, t, g, sd);
    }
}
// serial form (no mutation map): every gamete's seed depends on the previous gamete's crossover
// count, so one wave walks the chain; each link is still a wave-parallel scan.
__global__ void __launch_bounds__(64) k_rec_chain(const GevRngTables* __restrict__ Tg, const ChrDev* __restrict__ chrs, int nchr,
                                                  u32 seed_reproduce, const u32* __restrict__ seed_ptr, size_t n_tasks, SampleDev sd)
{
    __shared__ __attribute__((aligned(16))) GevRngTables s_T;
    const GevRngTables* T = stage_tables(Tg, &s_T);
    const u32 lane = threadIdx.x;
    GlibcWave g; g.seed(T, seed_ptr ? *seed_ptr : seed_reproduce);
    u32 seed = g.out(T, 0);
    for (size_t t = 0; t < n_tasks; t++) {
        const int c = (int)(t % nchr);
        const u32 k_mat = task_sample(T, chrs[c], seed, t, g, sd);
        u32 n = k_mat + 1;
        if (c == nchr - 1) {                                       // :2472
            const u32 s = (g.out(T, n) & 1u) + 1u;
            if (lane == 0) sd.sex[t / nchr] = (uint8_t)s;
            n++;
        }
        seed = g.out(T, n);
    }
}

// The same chain with a WORKGROUP per link (no mutation map: the seed of every gamete is a rand() of the srand() of the gamete
// before it, src/Simulation.cpp:2447-2455, so the gametes of a generation form one serial chain).  One wave per link pays a
// 31-step srand, 32 scan steps for 2001 map rows and two or three dependent global loads -- about 10 us per gamete.  Here wave 0
// seeds glibc's generator (srand) WHILE waves 1..8 scan 256 map rows each (one scan step: 2048 rows) against thresholds staged in
// LDS; every hit is left in LDS with its map position already loaded; after one barrier wave 0 turns the hits into breakpoints
// (bp[row] + rand() % dist, in row order), reads the start haplotype and the next seed off the generator and publishes the seed
// for the next link behind a second barrier.  Same arithmetic, same records as k_rec_chain.
#define CHAIN_SCAN_WAVES 8
#define CHAIN_THREADS (64 * (CHAIN_SCAN_WAVES + 1))
#define CHAIN_HITS 64                       // hits a scanning wave can leave per link; more (hot maps): wave 0 samples the gamete alone, as k_rec_chain does
__global__ void __launch_bounds__(CHAIN_THREADS) k_rec_chain_wg(const GevRngTables* __restrict__ Tg, const ChrDev* __restrict__ chrs, int nchr,
                                                                u32 seed_reproduce, const u32* __restrict__ seed_ptr, size_t n_tasks, SampleDev sd, u32 thr_rows_lds)
{
    __shared__ __attribute__((aligned(16))) GevRngTables s_T;
    __shared__ u32 s_hit_row[CHAIN_SCAN_WAVES][CHAIN_HITS];
    __shared__ u64 s_hit_bp[CHAIN_SCAN_WAVES][CHAIN_HITS];
    __shared__ u32 s_nhit[CHAIN_SCAN_WAVES];
    __shared__ u32 s_seed, s_pw[64];
    extern __shared__ __attribute__((aligned(16))) GevThr s_thr[];          // [thr_rows_lds] thresholds of all chromosomes, back to back (0: read from global)
    const GevRngTables* T = stage_tables(Tg, &s_T);
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // 16807^(512 j): engine steps in front of draw 256 j
    if (threadIdx.x < 64) s_pw[threadIdx.x] = powmod31(16807u, 512ull * threadIdx.x);
    if (thr_rows_lds) {
        u32 base = 0;
        for (int c = 0; c < nchr; c++) { for (u32 e = threadIdx.x; e < chrs[c].R; e += CHAIN_THREADS) s_thr[base + e] = chrs[c].rthr[e]; base += chrs[c].R; }
    }
    GlibcWave g;
    if (wave == 0) { g.seed(T, seed_ptr ? *seed_ptr : seed_reproduce); const u32 first = g.out(T, 0); if (lane == 0) s_seed = first; }
    __syncthreads();
    for (size_t t = 0; t < n_tasks; t++) {
        const int c = (int)(t % nchr);
        const ChrDev& C = chrs[c];
        u32 thr_base = 0;
        if (thr_rows_lds) for (int q = 0; q < c; q++) thr_base += chrs[q].R;
        const GevThr* thr = thr_rows_lds ? s_thr + thr_base : C.rthr;
        u32 k_mat = 0;
        for (u32 gam = 0; gam < 2; gam++) {                               // paternal, then maternal gamete (:2447-2456)
            const u32 seed = s_seed;
            const size_t G = 2 * t + gam;
            u32 k = 0, off = 0;
            bool crowded = false;                                         // (wave 0 only)
            for (u32 row0 = 0; row0 < C.R; row0 += 256 * CHAIN_SCAN_WAVES) {      // one round for maps of up to 2048 rows
                if (wave == 0) { if (row0 == 0) g.seed(T, seed); }                  // srand(seed_loc), :2977
                else {
                    const u32 d0 = row0 + 256 * (wave - 1);
                    u32 n = 0;
                    if (d0 < C.R) {
                        const u32 s0 = mulmod31(d0 >> 8 < 64 ? s_pw[d0 >> 8] : powmod31(16807u, 2ull * d0), minstd_seed(seed + 1u));   // generator(seed+1) in front of draw d0, :2978
                        wave_scan_hits_state(T, s0, thr, C.r_amax, d0, min(256u, C.R - d0), [&](u32 row) {
                            if (lane == 0 && n < CHAIN_HITS) { s_hit_row[wave - 1][n] = row; s_hit_bp[wave - 1][n] = C.rbp[row]; }
                            n++;
                        });
                    }
                    if (lane == 0) s_nhit[wave - 1] = n;
                }
                __syncthreads();
                if (wave == 0) {
                    for (u32 w = 0; w < CHAIN_SCAN_WAVES; w++) {
                        const u32 n = s_nhit[w];
                        if (n > CHAIN_HITS) crowded = true;
                        for (u32 j = 0; j < n && !crowded; j++) {
                            if (k < GEV_BK_CAP) {
                                const u64 v = s_hit_bp[w][j] + (u64)g.out(T, k) % C.bp_dist;                              // :2990
                                if (lane == 0) { sd.bk[G * GEV_BK_CAP + k] = v; sd.bk_idx[G * GEV_BK_CAP + k] = snp_lower_bound(C, v); }
                            }
                            k++;
                        }
                    }
                }
                __syncthreads();                                                                                          // the hit lists are free again
            }
            if (wave == 0 && crowded) {                                   // more hits than a wave's list holds: the whole gamete by this wave alone
                k = gamete_sample(T, C, seed, G, g, sd);
                const u32 start = g.out(T, k) & 1u;
                if (lane == 0) { sd.start[G] = (uint8_t)start; if (gam == 0) sd.seed_pat[t] = seed; else sd.seed_mat[t] = seed; }
                if (gam == 0) { const u32 nxt = g.out(T, k + 1); if (lane == 0) s_seed = nxt; } else k_mat = k;
            } else if (wave == 0) {
                u32 k_store = k;
                off = (u32)G * GEV_BK_CAP;
                if (k > GEV_BK_CAP) {                                     // rare (hot maps): a range of the overflow region, the whole gamete once more by this wave alone
                    u32 o = 0;
                    if (lane == 0) o = atomicAdd(&sd.status[ST_BK_OVF_USED], k);
                    o = rl_u32(o, 0);
                    if (o + k <= sd.bk_ovf_cap) {
                        off = sd.bk_ovf_base + o;
                        g.seed(T, seed);
                        rec_scan_write(T, C, seed, g, sd.bk + off, sd.bk_idx + off, k);
                    } else { if (lane == 0) atomicOr(&sd.status[ST_FLAGS], (u32)FLAG_BK_OVF); k_store = 0; }
                }
                const u32 start = g.out(T, k) & 1u;                       // rand()%2 behind the k position draws, :2449 / :2455
                if (lane == 0) {
                    sd.k[G] = k_store; sd.bk_off[G] = off; sd.start[G] = (uint8_t)start;
                    if (gam == 0) sd.seed_pat[t] = seed; else sd.seed_mat[t] = seed;
                }
                if (gam == 0) { const u32 nxt = g.out(T, k + 1); if (lane == 0) s_seed = nxt; }                           // seed_loc of the maternal gamete, :2453
                else k_mat = k;
            }
            if (gam == 0) __syncthreads();
        }
        if (wave == 0) {
            u32 n = k_mat + 1;
            if (c == nchr - 1) {                                          // :2472
                const u32 sx = (g.out(T, n) & 1u) + 1u;
                if (lane == 0) sd.sex[t / nchr] = (uint8_t)sx;
                n++;
            }
            const u32 nxt = g.out(T, n);                                  // seed_loc of the next task's paternal gamete, :2447
            if (lane == 0) s_seed = nxt;
        }
        __syncthreads();
    }
}

#include "gev_sample8.h"

// ------------------------------------------------------------------------------------------
// K5: dense stitch -- the HBM-bound kernel.
// Dense equivalent of Simulation::recombine (:2903-2958) applied to the materialised haplotypes
// (ras_convert_interval_to_hap_matrix, :1186-1230): at locus position x the offspring gamete
// copies parental haplotype  start ^ parity(#{breakpoints c : c <= x}).  In locus-index space a
// gamete row is a concatenation of bit ranges of the parent's two rows AT THE SAME OFFSETS, so
// between two boundaries it IS one parental row.  Rows are stored as segments (PoolWork): a segment
// without a boundary is not copied at all -- the offspring's table entry names the parent's unit
// (k_pool_inherit) -- and the stitch writes only the segments that contain a boundary: about one
// 2 KiB segment per crossover instead of the whole row.  Per unit of work (one written segment):
// segment bytes read + segment bytes written.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 lower_bound_u64(const u64* __restrict__ a, u32 n, u64 v)   // #{a[i] < v}
{
    u32 lo = 0, hi = n;
    while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ __forceinline__ u32 count_le_u32(const u32* __restrict__ a, u32 n, u32 v)      // #{a[i] <= v}, a ascending
{
    if (n <= 8) { u32 c = 0; for (u32 m = 0; m < n; m++) c += (a[m] <= v); return c; }
    u32 lo = 0, hi = n;
    while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1; else hi = mid; }
    return lo;
}
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4u mask_from(u32 rel)           // ones at bit positions >= rel of a 128-bit chunk
{
    v4u t;
    t.x = rel < 32 ? 0xffffffffu << rel : 0u;
    t.y = rel <= 32 ? 0xffffffffu : (rel < 64 ? 0xffffffffu << (rel - 32) : 0u);
    t.z = rel <= 64 ? 0xffffffffu : (rel < 96 ? 0xffffffffu << (rel - 64) : 0u);
    t.w = rel <= 96 ? 0xffffffffu : (rel < 128 ? 0xffffffffu << (rel - 96) : 0u);
    return t;
}
// one 16-byte chunk of a gamete: `cnt` boundaries lie at or before its first locus bit0, `in[c], in[c+1], ...` are the following ones
__device__ __forceinline__ v4u blend_chunk(v4u a, v4u b, u32 sel, const u32* in, u32 c, u32 n, u32 bit0)
{
    v4u mask = sel ? (v4u)(0xffffffffu) : (v4u)(0u);       // 1 = take row 1
    for (u32 m = c; m < n; m++) { const u32 id = in[m]; if (id >= bit0 + 128u) break; mask ^= mask_from(id - bit0); }
    return (a & ~mask) | (b & mask);
}
// ---- unit pool (PoolWork): free units of a generation = units no (slot, segment) of the parents names -----------------
// live[unit] == stamp <=> an entry of the parents' table names the unit; the stamp changes with every rebuild, so the marks of
// earlier generations need no clearing (the buffer is zeroed when it is allocated, stamps start at 1)
__device__ __forceinline__ void pool_mark(const PoolWork& pw, size_t n_entries)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_entries; e += (size_t)gridDim.x * blockDim.x) pw.live[pw.phys_cur[e]] = pw.stamp;
    if (blockIdx.x == 0 && threadIdx.x < 8) pw.pctr[threadIdx.x] = 0;        // a new free list: length, cursor, ... (PoolWork)
}
// one atomic per 2048 units: the block counts its free units (8 per thread, coalesced), scans the counts, reserves a range
__device__ __forceinline__ void pool_collect(const PoolWork& pw)
{
    __shared__ u32 s_scan[8], s_base;
    const u32 n_chunks = (pw.pool_units + 2047u) / 2048u;
    for (u32 chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const size_t i0 = (size_t)chunk * 2048 + threadIdx.x;
        u32 mask = 0, c = 0;
#pragma unroll
        for (u32 j = 0; j < 8; j++) { const size_t i = i0 + j * 256; const u32 fr = (i < pw.pool_units && pw.live[i] != pw.stamp) ? 1u : 0u; mask |= fr << j; c += fr; }
        u32 tot;
        const u32 ex = block_exclusive_scan_256(c, s_scan, tot);
        if (threadIdx.x == 0) s_base = tot ? atomicAdd(&pw.pctr[0], tot) : 0u;
        __syncthreads();
        u32 at = s_base + ex;
#pragma unroll
        for (u32 j = 0; j < 8; j++) if ((mask >> j) & 1u) pw.freel[at++] = (u32)(i0 + j * 256);
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) k_pool_mark_tab(const ChrWork* __restrict__ Wt, size_t n_slots) { const PoolWork& pw = Wt[blockIdx.y].pw; pool_mark(pw, n_slots * pw.nseg); }
__global__ void __launch_bounds__(256) k_pool_collect_tab(const ChrWork* __restrict__ Wt) { pool_collect(Wt[blockIdx.y].pw); }
__global__ void __launch_bounds__(256) k_pool_mark(PoolWork pw, size_t n_slots) { pool_mark(pw, n_slots * pw.nseg); }
__global__ void __launch_bounds__(256) k_pool_collect(PoolWork pw) { pool_collect(pw); }
// fresh units for every segment of slots [slot0, slot0 + n) of pw.phys_alt (migration, order restoring); flag[0] set when the pool is exhausted
__global__ void __launch_bounds__(256) k_pool_take(PoolWork pw, size_t slot0, size_t n, u32* __restrict__ flag)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * pw.nseg) return;
    const u32 n_free = pw.pctr[0];
    const u32 at = pw.pctr[1] + (u32)e;
    u32* dst = pw.phys_alt + slot0 * pw.nseg + e;
    if (at >= n_free) { flag[0] = 1; *dst = n_free ? pw.freel[at % n_free] : 0u; return; }
    *dst = pw.freel[at];
}
// (pctr[4] = where the cursor stood when the generation in progress began: units taken outside a generation move both)
__global__ void k_pool_taken(PoolWork pw, u32 n_units) { if (threadIdx.x == 0 && blockIdx.x == 0) { pw.pctr[1] += n_units; pw.pctr[4] += n_units; } }
__global__ void __launch_bounds__(256) k_iota_u32(u32* __restrict__ a, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = (u32)i;
}
// Units of the offspring generation, in two kernels.  Which segments of an output row contain one of the gamete's boundaries (bk_idx
// = first locus index at or behind the breakpoint; its 16-byte chunk decides the segment)?  Those get a unit from the free list
// and a COMPLETE entry in the stitch's work list (StitchItem); every other segment names the unit of the parental haplotype that
// is being copied there: start ^ parity(#boundaries at or before the segment's first locus).
//  k_pool_inherit : ONE THREAD PER TABLE ENTRY (row, segment): the segments without a boundary copy the parental entry.  The S
//                   threads of a row read S consecutive words of one parental table row and write S consecutive words: coalesced,
//                   nothing serial; the row's few descriptor words are the same for all of them (served by the caches).
//  k_pool_fresh   : one thread per ROW, only for its segments WITH a boundary (about one): count them, reserve free units with
//                   one atomic per block of 256 rows, write the table entry and the work-list entry.
// (One thread walking all S entries of its row in global memory: 80 us at S = 16, 237 us at S = 32; the same through LDS: 197 us
// at S = 32; one wave per row, lane = segment: 333 us at S = 16 -- the walk is a latency chain, the entries are not.)
#define POOL_SEG_MAX 64          // segments per row handled with a 64-bit flag word (4 KiB segments: rows up to 256 KiB = 2M loci); longer rows use larger segments
#define POOL_INH 4              // table entries (consecutive segments of one row) per thread of k_pool_inherit: the row's descriptor words are read once for them
__global__ void __launch_bounds__(256) k_pool_inherit(const ChrWork* __restrict__ Wt, size_t n_rows_out, int nchr, SampleDev sd, u32 lg_tpr)
{
    const ChrWork& w = Wt[blockIdx.y]; const PoolWork& pw = w.pw;
    const u32 S = pw.nseg, sh = pw.seg_shift;
    const u32 g0 = (threadIdx.x & ((1u << lg_tpr) - 1u)) * POOL_INH;   // 2^lg_tpr threads per row (the launch's largest S / POOL_INH rounded up to a power of two): 256 >> lg_tpr rows per block
    const size_t row = (size_t)blockIdx.x * (256u >> lg_tpr) + (threadIdx.x >> lg_tpr);
    if (row >= n_rows_out || g0 >= S || !pw.alias) return;        // (rows are never shared: every segment is written, k_pool_fresh names them all)
    const size_t i = row >> 1; const u32 s = (u32)(row & 1);
    const size_t G = 2 * (i * nchr + w.chr) + s;
    const u32 k = sd.k[G];
    u32 cnt[POOL_INH] = {0, 0, 0, 0}; bool fl[POOL_INH] = {false, false, false, false};
    if (k) {
        const u32* __restrict__ idx = sd.bk_idx + sd.bk_off[G];
        for (u32 m = 0; m < k; m++) {
            const u32 id = idx[m], gs = (id >> 7) >> sh;
#pragma unroll
            for (int j = 0; j < POOL_INH; j++) { cnt[j] += id <= (((g0 + j) << sh) << 7); fl[j] |= gs == g0 + j; }
        }
    }
    const u32 parent = s ? sd.mother[i] : sd.father[i];
    const u32 start = sd.start[G];
    const u32* __restrict__ p0 = pw.phys_cur + (2 * (size_t)parent) * S;
    u32* __restrict__ out = pw.phys_alt + row * S;
    u32 v[POOL_INH];
#pragma unroll
    for (int j = 0; j < POOL_INH; j++) if (g0 + j < S && !fl[j]) v[j] = p0[((start ^ cnt[j]) & 1u) * S + g0 + j];
#pragma unroll
    for (int j = 0; j < POOL_INH; j++) if (g0 + j < S && !fl[j]) out[g0 + j] = v[j];
}
__global__ void __launch_bounds__(256) k_pool_fresh(const ChrWork* __restrict__ Wt, size_t n_rows_out, int nchr, SampleDev sd)
{
    __shared__ u32 s_scan[8], s_base;
    const ChrWork& w = Wt[blockIdx.y]; const PoolWork& pw = w.pw;
    const u32 S = pw.nseg, sh = pw.seg_shift;
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = row < n_rows_out;
    u64 flags = 0; u32 k = 0; const u32* idx = nullptr; size_t G = 0;
    if (live) {
        G = 2 * ((row >> 1) * nchr + w.chr) + (row & 1);
        k = sd.k[G]; idx = sd.bk_idx + sd.bk_off[G];
        if (!pw.alias) flags = S >= 64 ? ~0ull : ((1ull << S) - 1ull);
        else for (u32 m = 0; m < k; m++) { const u32 g = (idx[m] >> 7) >> sh; if (g < S) flags |= 1ull << g; }
    }
    const u32 c = (u32)__popcll(flags);
    u32 tot;
    const u32 ex = block_exclusive_scan_256(c, s_scan, tot);
    if (threadIdx.x == 0) s_base = tot ? atomicAdd(&pw.pctr[1], tot) : 0u;
    __syncthreads();
    if (!c) return;                                               // (no barrier below)
    u32 at = s_base + ex, n_last = 0;
    const u32 n_free = pw.pctr[0], gen_start = pw.pctr[4];
    const size_t i = row >> 1;
    const u32 parent = (row & 1) ? sd.mother[i] : sd.father[i];
    const u32 start = sd.start[G];
    const u32* p0 = pw.phys_cur + (2 * (size_t)parent) * S;      // the parent's two rows of the table are adjacent
    u32 m = 0, cnt = 0;                                           // boundaries with idx <= first locus of the segment: ascending list, one sweep
    for (u64 f = flags; f; f &= f - 1) {
        const u32 g = (u32)__ffsll((long long)f) - 1u;
        const u32 bit0 = (g << sh) << 7;
        while (m < k && idx[m] <= bit0) { m++; cnt++; }
        const u32 sel0 = (start ^ cnt) & 1u;
        u32 unit;
        if (at < n_free) unit = pw.freel[at];
        else { atomicOr(&sd.status[ST_FLAGS], (u32)FLAG_POOL); unit = n_free ? pw.freel[at % n_free] : 0u; }   // the free list ran out: the host rebuilds it and enqueues the generation again
        pw.phys_alt[row * S + g] = unit;
        const u32 item = at - gen_start;
        if (item < pw.items_cap) {
            // the boundaries inside the segment: behind its first locus (one exactly on it is part of cnt), in its chunks
            u32 b0 = 0xffffffffu, b1 = 0xffffffffu, b2 = 0xffffffffu, b3 = 0xffffffffu, nin = 0;
            for (u32 mm = m; mm < k && ((idx[mm] >> 7) >> sh) == g; mm++, nin++) {
                const u32 r = idx[mm] - bit0;
                if (nin == 0) b0 = r; else if (nin == 1) b1 = r; else if (nin == 2) b2 = r; else if (nin == 3) b3 = r;
            }
            const u32 meta = sel0 | ((nin <= STITCH_ITEM_B ? nin : 7u) << 1) | (g << 8);
            uint4* o = (uint4*)&pw.items[item];
            o[0] = make_uint4(p0[g], p0[S + g], unit, meta);
            o[1] = make_uint4(nin <= STITCH_ITEM_B ? b0 : (u32)row, b1, b2, b3);
        }
        n_last += (g == S - 1);
        at++;
    }
    if (n_last) atomicAdd(&pw.pctr[3], n_last);
}
// behind k_pool_assign: the length of the stitch's work list and the segment totals of the status block, per work entry; the
// counters of the next generation start here
__global__ void __launch_bounds__(64) k_pool_publish(const ChrWork* __restrict__ Wt, u32 n_work, u32* __restrict__ status)
{
    for (u32 y = threadIdx.x; y < n_work; y += 64) {
        const ChrWork& w = Wt[y]; u32* pctr = w.pw.pctr;
        const u32 n = pctr[1] - pctr[4];
        u32* st = status + ST_TOTALS + ST_PER_CHR * w.chr;
        st[2] = n;                                                                          // segments the dense stitch writes
        st[3] = pctr[3];                                                                    // ... of which last (partial) segments
        st[4] = pctr[0]; st[5] = pctr[1];                                                   // free list: length, cursor
        ((u32*)&w.pw.items[w.pw.items_cap])[0] = (status[ST_FLAGS] & FLAG_POOL) ? 0u : min(n, w.pw.items_cap);   // length of the work list (free list ran out: the stitch does nothing, the host rebuilds and repeats)
        pctr[4] = pctr[1]; pctr[3] = 0;
    }
}
// K5, production form: one WAVE per entry of the work list k_pool_assign wrote (one written segment of one gamete).  The entry is
// complete -- the two parental units, the unit to write, the haplotype copied at the segment's first locus and the (usually one)
// boundaries inside -- so a wave starts streaming after ONE uniform 32-byte load.  A chunk before / behind / between the
// boundaries is a plain 16-byte copy from ONE parental unit, the chunk that contains one is blended by mask; U chunks per lane in
// flight.  Grid: sized by the host from the previous generation's list length, the loop covers whatever the list holds.
template <bool NT, int U = 2>
__global__ void __launch_bounds__(256) k_stitch_segments(const ChrWork* __restrict__ Wt, int nchr, SampleDev sd)
{
    const ChrWork& w = Wt[blockIdx.y]; const PoolWork& pw = w.pw;
    const u32 sh = pw.seg_shift, SC = 1u << sh;
    const u32 n_items = ((const u32*)&pw.items[pw.items_cap])[0];
    const u32 lane = threadIdx.x & 63u;
    const u32 wave0 = (u32)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
    uint4 d0 = make_uint4(0u, 0u, 0u, 0u), d1 = d0;
    if (wave0 < n_items) { const uint4* dp = (const uint4*)&pw.items[wave0]; d0 = dp[0]; d1 = dp[1]; }
    for (u32 it = wave0; it < n_items; it += gridDim.x * 4u) {
        // the next entry's descriptor is loaded while this entry's chunks are copied (a wave that takes several entries does not
        // pay the descriptor's latency in front of each of them)
        const u32 nx = it + gridDim.x * 4u;
        uint4 e0 = d0, e1 = d1;
        if (nx < n_items) { const uint4* dq = (const uint4*)&pw.items[nx]; e0 = dq[0]; e1 = dq[1]; }
        const u32 g = (d0.w >> 8) & 0xffu, code = (d0.w >> 1) & 7u, sel0 = d0.w & 1u;
        const u32 q0 = g << sh, nq = min(SC, w.chunks - q0);
        const v4u* __restrict__ R0 = (const v4u*)(pw.pool + ((size_t)d0.x << (sh + 4)));
        const v4u* __restrict__ R1 = (const v4u*)(pw.pool + ((size_t)d0.y << (sh + 4)));
        v4u* __restrict__ D = (v4u*)(pw.pool + ((size_t)d0.z << (sh + 4)));
        if (code <= STITCH_ITEM_B) {
            const u32 in[STITCH_ITEM_B] = {d1.x, d1.y, d1.z, d1.w};          // bit offsets inside the segment; 0xffffffff = none
            for (u32 q = lane; q < nq; q += 64 * U) {
                v4u a[U], b[U]; u32 c[U]; bool mixed[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const u32 qq = q + u * 64;
                    c[u] = 0; mixed[u] = false;
                    if (qq >= nq) continue;
                    const u32 bit0 = qq << 7;
                    u32 cc = 0; bool mx = false;
#pragma unroll
                    for (int m = 0; m < STITCH_ITEM_B; m++) { cc += (in[m] <= bit0); mx |= (in[m] > bit0 && in[m] < bit0 + 128u); }
                    c[u] = cc; mixed[u] = mx;
                    const u32 sel = (sel0 ^ cc) & 1u;
                    if (mx || sel == 0u) a[u] = NT ? __builtin_nontemporal_load(&R0[qq]) : R0[qq];
                    if (mx || sel == 1u) b[u] = NT ? __builtin_nontemporal_load(&R1[qq]) : R1[qq];
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const u32 qq = q + u * 64;
                    if (qq >= nq) continue;
                    const u32 sel = (sel0 ^ c[u]) & 1u;
                    v4u o;
                    if (!mixed[u]) o = sel ? b[u] : a[u];
                    else {
                        const u32 bit0 = qq << 7;
                        v4u mask = sel ? (v4u)(0xffffffffu) : (v4u)(0u);       // 1 = take row 1
#pragma unroll
                        for (int m = 0; m < STITCH_ITEM_B; m++) if (in[m] > bit0 && in[m] < bit0 + 128u) mask ^= mask_from(in[m] - bit0);
                        o = (a[u] & ~mask) | (b[u] & mask);
                    }
                    if (NT) __builtin_nontemporal_store(o, &D[qq]); else D[qq] = o;
                }
            }
        } else {
            // many boundaries in one segment (hot maps): the gamete's own list, bisection per chunk
            const u32 row = d1.x, i = row >> 1, s = row & 1u;
            const size_t G = 2 * ((size_t)i * nchr + w.chr) + s;
            const u32 k = sd.k[G];
            const u32* __restrict__ idx = sd.bk_idx + sd.bk_off[G];
            const u32 bit_lo = q0 << 7, bit_hi = (q0 + nq) << 7;
            const u32 m0 = count_le_u32(idx, k, bit_lo), m1 = count_le_u32(idx, k, bit_hi - 1u);
            const u32* in = idx + m0; const u32 nin = m1 - m0;
            for (u32 qq = lane; qq < nq; qq += 64) {
                const u32 bit0 = bit_lo + (qq << 7);
                const u32 c = count_le_u32(in, nin, bit0);
                const u32 sel = (sel0 ^ c) & 1u;
                const bool mx = c < nin && in[c] < bit0 + 128u;
                D[qq] = mx ? blend_chunk(R0[qq], R1[qq], sel, in, c, nin, bit0) : (sel ? R1[qq] : R0[qq]);
            }
        }
        d0 = e0; d1 = e1;
    }
}
// K5, gamete-major form (cross-check, gev_set_stitch_mode(1)): one workgroup per OUTPUT ROW walks all its chunks and decides
// for itself which segments are its own (those that contain a boundary chunk; all of them when rows are never shared) and,
// per chunk, the source by bisection over the gamete's whole boundary list -- no work list, no per-segment staging.
#define STITCH_THREADS 256
#define STITCH_KMAX 256          // boundaries staged in LDS per row; more are read from global memory
__global__ void __launch_bounds__(STITCH_THREADS) k_stitch_rows(const ChrWork* __restrict__ Wt, int nchr, SampleDev sd)
{
    __shared__ u32 s_idx[STITCH_KMAX];
    const ChrWork& w = Wt[blockIdx.y]; const PoolWork& pw = w.pw;
    const u32 S = pw.nseg, sh = pw.seg_shift;
    const u32 row = blockIdx.x;                           // output row = 2*offspring + s
    const u32 i = row >> 1, s = row & 1u;
    const size_t G = 2 * ((size_t)i * nchr + w.chr) + s;
    const u32 parent = s ? sd.mother[i] : sd.father[i];
    const u32 start = sd.start[G], k = sd.k[G];
    const u32* __restrict__ gidx = sd.bk_idx + sd.bk_off[G];
    const bool in_lds = k <= STITCH_KMAX;
    if (in_lds) for (u32 m = threadIdx.x; m < k; m += STITCH_THREADS) s_idx[m] = gidx[m];
    __syncthreads();
    const u32* idx = in_lds ? s_idx : gidx;
    for (u32 q = threadIdx.x; q < w.chunks; q += STITCH_THREADS) {
        const u32 g = q >> sh, ql = q & ((1u << sh) - 1u);
        if (pw.alias) {                                     // is segment g one of the row's own?  (some boundary chunk lies in it)
            const u32 first_bit = (g << sh) << 7, last_chunk_end = ((g + 1u) << sh) << 7;
            const u32 c0 = first_bit ? count_le_u32(idx, k, first_bit - 1u) : 0u;                                                     // #{idx < first_bit}
            const u32 c1 = count_le_u32(idx, k, last_chunk_end - 1u);                                                                   // #{idx < end}
            if (c1 == c0) continue;                         // shared with the parent: nothing to write
        }
        const u32 bit0 = q << 7;
        const u32 cnt = count_le_u32(idx, k, bit0);
        const u32 sel = (start ^ cnt) & 1u;
        const uint4* A = (const uint4*)(pw.pool + ((size_t)pw.phys_cur[(2 * (size_t)parent + sel) * S + g] << (sh + 4))) + ql;
        uint4* Dq = (uint4*)(pw.pool + ((size_t)pw.phys_alt[(size_t)row * S + g] << (sh + 4))) + ql;
        if (cnt >= k || idx[cnt] >= bit0 + 128u) { *Dq = *A; continue; }
        const uint4* B = (const uint4*)(pw.pool + ((size_t)pw.phys_cur[(2 * (size_t)parent + (sel ^ 1u)) * S + g] << (sh + 4))) + ql;
        const uint4 av = *A, bv = *B;
        u32 aw[4] = {av.x, av.y, av.z, av.w}, bw[4] = {bv.x, bv.y, bv.z, bv.w}, ow[4], mask[4] = {0, 0, 0, 0};      // 1 = take B (the other haplotype)
        for (u32 m = cnt; m < k; m++) {
            const u32 id = idx[m];
            if (id >= bit0 + 128u) break;
            const u32 rel = id - bit0;                       // toggle every bit >= rel
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const u32 wb = x * 32u;
                u32 t = 0;
                if (rel <= wb) t = 0xffffffffu; else if (rel < wb + 32u) t = 0xffffffffu << (rel - wb);
                mask[x] ^= t;
            }
        }
#pragma unroll
        for (int x = 0; x < 4; x++) ow[x] = (aw[x] & ~mask[x]) | (bw[x] & mask[x]);
        *Dq = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    }
}

// small planes (CV grid: ~125 B rows): one HALF-WAVE (32 lanes) per output row.  The lanes first turn the row's breakpoints into
// CV-column indices in parallel (one binary search per lane, 32 breakpoints per round) and share them with shuffles, then
// every lane blends its own 32-bit words of the two parental rows -- every sub-row (alleles, root-population bits) by the same
// pattern; loads and stores are 128-byte coalesced per half-wave.  Fused into the same pass: the allele count of
// every CV column (:2647-2655), when `count_cols` (every CV grid of the launch has <= 1024 columns): each lane adds its allele
// word into bit-sliced counters over the block's rounds, the block adds them up in LDS and writes its 1024 partial counts with
// plain stores (k_cv_sum_partials adds the blocks up: one global atomic per column and block was measured three times as slow
// as the whole rest of the kernel).  The kernel is bound by the latency of the per-row descriptor chain (parent -> rows, breakpoints -> columns):
// more threads per block = more rows in flight per block, fewer rounds (512: 16 rows, 16 rounds).
#define SMALL_ROWS_PER_BLOCK 256         // the block stages the CV position grid in LDS once for this many rows
#define SMALL_POS_LDS 4096               // most CV positions held in LDS (32 KiB); longer grids are searched in global memory
template <int SMALL_THREADS>
__global__ void __launch_bounds__(SMALL_THREADS) k_stitch_small(const CvWork* __restrict__ Vt, u32 nsub, size_t n_rows_out, int nchr, SampleDev sd, u32 pos_lds, int count_cols)
{
    extern __shared__ u64 s_pos[];                                        // min(largest CV grid of the launch, SMALL_POS_LDS) entries
    __shared__ u32 s_cnt[1024];                                           // [bit][word]: column 32 * word + bit
    const CvWork& v = Vt[blockIdx.y];
    u32* __restrict__ dst = v.cvp_alt; const u32* __restrict__ src = v.cvp_cur;
    const u32 stride_w32 = v.stride_w32, sub_w32 = v.sub_w32, Cn = v.C;
    const int chr = v.chr;
    const bool in_lds = Cn <= pos_lds;
    if (in_lds) for (u32 e = threadIdx.x; e < Cn; e += SMALL_THREADS) s_pos[e] = v.pos_sorted[e];
    if (count_cols) for (u32 e = threadIdx.x; e < 1024; e += SMALL_THREADS) s_cnt[e] = 0;
    __syncthreads();
    const u64* __restrict__ pos = in_lds ? s_pos : v.pos_sorted;
    const u32 hl = threadIdx.x & 31u, half0 = threadIdx.x & 32u;          // lane inside the half-wave / first wave lane of the half
    const u32 HW = SMALL_THREADS / 32;                                    // rows per round
    u32 p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0;                           // bit-sliced counters (up to 31 >= rounds) of this lane's allele word over the block's rounds
    for (u32 round = 0; round < SMALL_ROWS_PER_BLOCK / HW; round++) {
    const size_t row = (size_t)blockIdx.x * SMALL_ROWS_PER_BLOCK + round * HW + (threadIdx.x >> 5);
    if ((size_t)blockIdx.x * SMALL_ROWS_PER_BLOCK + round * HW >= n_rows_out) break;      // block-uniform
    const bool live = row < n_rows_out;                                   // dead halves keep running: their lanes take part in the shuffles
    const u32 i = live ? (u32)(row >> 1) : 0u, s = (u32)(row & 1);
    const size_t G = 2 * ((size_t)i * nchr + chr) + s;
    const u32 parent = s ? sd.mother[i] : sd.father[i];
    const u32 start = sd.start[G];
    const u32 k = live ? sd.k[G] : 0u;
    const u64* bk = sd.bk + sd.bk_off[G];
    const u32 k_other = (u32)__shfl((int)k, (int)(half0 ^ 32u));
    const u32 kmax = k > k_other ? k : k_other;                           // wave-uniform trip count
    const u32* __restrict__ A = src + (size_t)(2 * parent + start) * stride_w32;
    const u32* __restrict__ B = src + (size_t)(2 * parent + (start ^ 1u)) * stride_w32;
    u32* __restrict__ D = dst + row * stride_w32;
    for (u32 w0 = 0; w0 < sub_w32; w0 += 32) {
        const u32 w = w0 + hl, bit0 = w * 32u;
        u32 mask = 0;
        for (u32 m0 = 0; m0 < kmax; m0 += 32) {
            u32 id = 0xffffffffu;                                         // "never": contributes nothing below
            if (m0 + hl < k) id = lower_bound_u64(pos, Cn, bk[m0 + hl]);
            const u32 nm = min(32u, kmax - m0);
            for (u32 m = 0; m < nm; m++) {
                const u32 x = (u32)__shfl((int)id, (int)(half0 + m));
                u32 tt = 0;
                if (x <= bit0) tt = 0xffffffffu; else if (x < bit0 + 32u) tt = 0xffffffffu << (x - bit0);
                mask ^= tt;
            }
        }
        if (live && w < sub_w32) {
            u32 av = 0;
            for (u32 sub = 0; sub < nsub; sub++) {
                const u32 wq = sub * sub_w32 + w;
                const u32 val = (A[wq] & ~mask) | (B[wq] & mask);
                D[wq] = val;
                if (sub == 0) av = val;
            }
            if (count_cols) {                                             // p += av, bit-sliced (ripple carry through the planes)
                u32 c = av, t1;
                t1 = p0 & c; p0 ^= c; c = t1; t1 = p1 & c; p1 ^= c; c = t1; t1 = p2 & c; p2 ^= c; c = t1; t1 = p3 & c; p3 ^= c; c = t1; p4 ^= c;
            }
        }
    }
    }
    if (count_cols) {                                                     // (sub_w32 <= 32: the lane's word is hl in every round)
        if (hl < sub_w32)
            for (u32 b = 0; b < 32; b++) {
                const u32 n = ((p0 >> b) & 1u) | (((p1 >> b) & 1u) << 1) | (((p2 >> b) & 1u) << 2) | (((p3 >> b) & 1u) << 3) | (((p4 >> b) & 1u) << 4);
                if (n) atomicAdd(&s_cnt[b * 32u + hl], n);
            }
        __syncthreads();
        uint16_t* __restrict__ out = v.partial + (size_t)blockIdx.x * 1024;
        for (u32 col = threadIdx.x; col < 1024; col += SMALL_THREADS) out[col] = (uint16_t)s_cnt[(col & 31u) * 32u + (col >> 5)];   // (<= 256 rows per block)
    }
}
// The generation's NEW mutations (ras_add_mutation, :2497-2552) that fall on a CV position, applied to the offspring plane behind
// k_stitch_small: ras_find_cv reads !founder where the position is in the covering part's mutation_pos (:2770-2775), so the
// resolved allele flips -- unless the position already was in that set (a set: a second hit changes nothing).  "Already" means:
// in the mutation list of the parental haplotype the offspring part was cut from (start ^ parity of the breakpoints at or before
// the position; the list the offspring inherits), or an earlier new mutation of the same task and side.  One thread per
// (offspring, chromosome) task; a position that cannot be a CV position is dropped with one LDS read (64-Kbit hash bitmap of the
// CV grid), so the lookups run for a handful of mutations per generation.  The column's allele count moves with the flip.
__device__ __forceinline__ u32 cv_hash16(u64 x) { return (u32)((x * 0x9E3779B97F4A7C15ull) >> 48); }
__global__ void __launch_bounds__(256) k_cv_newmut(const CvWork* __restrict__ Vt, const ChrWork* __restrict__ Wt, size_t n_people, int nchr, SampleDev sd, int count_cols)
{
    __shared__ u32 s_bits[2048];
    const CvWork& v = Vt[blockIdx.y];
    for (u32 q = threadIdx.x; q < 2048; q += 256) s_bits[q] = 0;
    __syncthreads();
    for (u32 c = threadIdx.x; c < v.C; c += 256) { const u32 h = cv_hash16(v.pos_sorted[c]); atomicOr(&s_bits[h >> 5], 1u << (h & 31)); }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_people) return;
    const size_t t = i * nchr + v.chr;
    const u32 n = sd.nmut[t];
    if (!n) return;
    const u32 off = sd.nm_off[t];
    const ChrWork& w = Wt[v.cw];
    for (u32 m = 0; m < n; m++) {
        const u64 x = sd.nm_pos[off + m];
        if (x < v.bp0 || x >= v.bp_end) continue;                         // no part contains it: ras_add_mutation appends nothing (:2526-2545)
        const u32 h = cv_hash16(x);
        if (!((s_bits[h >> 5] >> (h & 31)) & 1u)) continue;
        u32 c = lower_bound_u64(v.pos_sorted, v.C, x);
        if (c >= v.C || v.pos_sorted[c] != x) continue;
        const u32 side = sd.nm_side[off + m];
        bool seen = false;                                                // an earlier new mutation of this task on the same haplotype and position
        for (u32 m2 = 0; m2 < m; m2++) seen |= sd.nm_pos[off + m2] == x && sd.nm_side[off + m2] == side;
        if (!seen) {                                                      // inherited: in the list of the parental haplotype that covers x
            const size_t G = 2 * t + side;
            const u32 parent = side ? sd.mother[i] : sd.father[i];
            const u32 k = sd.k[G]; const u64* bk = sd.bk + sd.bk_off[G];
            u32 cnt = 0;
            for (u32 j = 0; j < k; j++) cnt += bk[j] <= x;
            const u32 hap = (sd.start[G] ^ cnt) & 1u;
            seen = lp_has_mutation(w.lp, (size_t)2 * parent + hap, x, w.bp0);
        }
        if (seen) continue;
        u32* row = v.cvp_alt + (2 * i + side) * v.stride_w32;
        for (; c < v.C && v.pos_sorted[c] == x; c++) {                    // every CV column at this position
            const u32 bit = 1u << (c & 31);
            const u32 a_old = atomicXor(&row[c >> 5], bit);
            if (count_cols) atomicAdd(&v.counts[c], (a_old & bit) ? 0xffffffffu : 1u);
        }
    }
}

// ------------------------------------------------------------------------------------------
// K6/K7: ras_find_cv + ras_compute_AD (src/Simulation.cpp:2624-2815)
// ------------------------------------------------------------------------------------------
// allele sub-rows of a CV plane, contiguous (gev_download_cv: the --debug .cvval dump, :2665-2683)
__global__ void __launch_bounds__(256) k_cv_resolve(const u32* __restrict__ plane, u32 stride_w32, u32 sub_w32, u32* __restrict__ out, size_t n_rows)
{
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_rows * sub_w32) return;
    const size_t r = q / sub_w32; const u32 w = (u32)(q % sub_w32);
    out[q] = plane[r * stride_w32 + w];
}
// allele counts per CV column (sorted order): f = sum_ih cv0+cv1 (:2647-2655), exact integers
__global__ void __launch_bounds__(256) k_cv_count(const AdWork* __restrict__ At, size_t n_rows)
{
    const AdWork& a = At[blockIdx.z];
    const u32 stride = a.stride_w32, Cn = a.C; u32* __restrict__ counts = a.counts;
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t rows_per = (n_rows + gridDim.y - 1) / gridDim.y;
    const size_t r0 = (size_t)blockIdx.y * rows_per, r1 = min(r0 + rows_per, n_rows);
    if (c >= Cn) return;
    const u32* col = a.cvp + (c >> 5);
    const u32 sh = c & 31;
    u32 n = 0;
    size_t r = r0;
    for (; r + 8 <= r1; r += 8) {                      // 16 independent loads in flight
        u32 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = col[(r + j) * stride];
#pragma unroll
        for (int j = 0; j < 8; j++) n += (v[j] >> sh) & 1u;
    }
    for (; r < r1; r++) n += (col[r * stride] >> sh) & 1u;
    if (n) atomicAdd(&counts[c], n);
}
// Column counts complete -> allele frequencies and the per-CV term table, in ONE launch: every block first adds its slice of
// k_stitch_small's per-block partial counts to the column counters (n_blocks == 0: k_cv_count has filled them); the block that
// finishes last (a counter per work entry) then turns the counters into frq[] and the table of the three possible A- and D-terms
// per CV, computed with EXACTLY the expressions of the reference's per-individual loop (:2686-2712: same operations, same order,
// so the sums stay bit-identical): tab[icv] = { (0-2p)alpha, (1-2p)alpha, (2-2p)alpha, -2pp*d, 2pq*d, -2qq*d }, and clears the
// counters for the next generation.  (As separate launches behind one another these were two to three kernels of a few
// microseconds each -- 75 us each next to the dense stitch.)
__global__ void __launch_bounds__(256) k_cv_finish(const AdWork* __restrict__ At, u32 n_blocks, size_t n_human, u32* __restrict__ done, u32* __restrict__ nan_flag)
{
    __shared__ bool s_last;
    const AdWork& aw = At[blockIdx.z];
    const u32 col = blockIdx.x * 256 + threadIdx.x;
    if (n_blocks && col < aw.C) {
        const u32 per = (n_blocks + gridDim.y - 1) / gridDim.y, b0 = blockIdx.y * per, b1 = min(b0 + per, n_blocks);
        u32 n = 0;
        u32 b = b0;
        for (; b + 8 <= b1; b += 8) {                       // eight independent loads in flight (next to the dense stitch a dependent load costs microseconds)
            u32 v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = aw.partial[(size_t)(b + j) * 1024 + col];
#pragma unroll
            for (int j = 0; j < 8; j++) n += v[j];
        }
        for (; b < b1; b++) n += aw.partial[(size_t)b * 1024 + col];
        if (n) atomicAdd(&aw.counts[col], n);
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&done[blockIdx.z], 1u) == gridDim.x * gridDim.y - 1u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (threadIdx.x == 0) { done[blockIdx.z] = 0; if (blockIdx.z == 0) *nan_flag = 0xffffffffu; }     // (the A/D kernels of this launch sequence report the first NaN individual with atomicMin)
    // four CVs per thread and round: the loads of a round are issued together (a dependent global access costs microseconds next
    // to the dense stitch, and this block is alone on the generation's critical path)
    for (u32 icv0 = threadIdx.x; icv0 < aw.C; icv0 += 1024) {
        u32 cc[4]; u32 cnt[4]; u64 xs[4]; double as_[4], ds_[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const u32 icv = icv0 + 256u * u; cc[u] = icv < aw.C ? aw.col_of_icv[icv] : 0u; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const u32 icv = icv0 + 256u * u;
            cnt[u] = icv < aw.C ? atomicExch(&aw.counts[cc[u]], 0u) : 0u;   // every column is read by exactly one thread; read at the L2, where the atomics landed
            xs[u] = icv < aw.C ? aw.pos_file[icv] : 0ull;
            as_[u] = icv < aw.C ? aw.a[icv] : 0.0; ds_[u] = icv < aw.C ? aw.d[icv] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const u32 icv = icv0 + 256u * u;
            if (icv >= aw.C) continue;
            const double f = (double)cnt[u];
            const double p = f / (double)(2 * n_human);
            aw.frq[icv] = p;
            if (!aw.tab) continue;                                      // (several root populations: k_ad_accumulate looks a and d up per haplotype)
            const u64 x = xs[u];
            const bool covered = (x >= aw.bp0 && x < aw.bp_end);
            const double a0 = covered ? as_[u] : 0.0, d0 = covered ? ds_[u] : 0.0;
            const double a = (a0 + a0) / 2;
            double d = (d0 + d0) / 2;
            if (aw.vd == 0) d = 0;
            const double q = 1 - p;
            const double alpha = a + d * (q - p);
            double* t = aw.tab + 6 * (size_t)icv;
            t[0] = ((double)0u - 2 * p) * alpha;
            t[1] = ((double)1u - 2 * p) * alpha;
            t[2] = ((double)2u - 2 * p) * alpha;
            t[3] = (-2 * p * p) * d;
            t[4] = (2 * p * q) * d;
            t[5] = (-2 * q * q) * d;
        }
    }
}
// per individual, CVs in FILE order, sequential FP64 (no contraction: built with -ffp-contract=off):
//   A += (t - 2p)(a + d(q-p));  D += {-2pp, 2pq, -2qq}[t] * d      (:2686-2712)
// a, d are the mean of the two haplotypes' root-population values (:2695-2696)
__global__ void __launch_bounds__(256) k_ad_accumulate(const AdWork* __restrict__ At, u32 rp_bits, size_t n_human, size_t out_stride, size_t tot_stride, u32* __restrict__ nan_flag)
{
    const AdWork& aw = At[blockIdx.y];
    const u32 sub_w32 = aw.sub_w32; const u32* __restrict__ plane = aw.cvp; const u32 stride_w32 = aw.stride_w32;
    const u32* __restrict__ col_of_icv = aw.col_of_icv; const double* __restrict__ frq = aw.frq;
    const double* const* __restrict__ a_of_pop = aw.aptr; const double* const* __restrict__ d_of_pop = aw.dptr; const int own_pop = aw.own_pop;
    const u64* __restrict__ cvpos_file = aw.pos_file; const u64 bp0 = aw.bp0, bp_end = aw.bp_end; const double vd = aw.vd; const u32 Cn = aw.C;
    double* __restrict__ add_out = aw.add_out; double* __restrict__ dom_out = aw.dom_out;
    const size_t ih = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ih >= n_human) return;
    const u32* p0 = plane + (2 * ih) * stride_w32; const u32* p1 = plane + (2 * ih + 1) * stride_w32;
    double A_chr = 0, D_chr = 0;
    for (u32 icv = 0; icv < Cn; icv++) {
        const u32 c = col_of_icv[icv];
        const u32 t = ((p0[c >> 5] >> (c & 31)) & 1u) + ((p1[c >> 5] >> (c & 31)) & 1u);
        u32 rp0 = own_pop, rp1 = own_pop;
        if (rp_bits) {
            rp0 = 0; rp1 = 0;
            for (u32 b = 0; b < rp_bits; b++) {
                rp0 |= ((p0[(1 + b) * sub_w32 + (c >> 5)] >> (c & 31)) & 1u) << b;
                rp1 |= ((p1[(1 + b) * sub_w32 + (c >> 5)] >> (c & 31)) & 1u) << b;
            }
        }
        // a CV outside [bp0,bp_end) is covered by no part: its a, d stay 0 (Human_CV ctor, Population.h:99-108)
        const u64 x = cvpos_file[icv];
        const bool covered = (x >= bp0 && x < bp_end);
        const double a0 = covered ? a_of_pop[rp0][icv] : 0.0, a1 = covered ? a_of_pop[rp1][icv] : 0.0;
        const double d0 = covered ? d_of_pop[rp0][icv] : 0.0, d1 = covered ? d_of_pop[rp1][icv] : 0.0;
        const double a = (a0 + a1) / 2;
        double d = (d0 + d1) / 2;
        if (vd == 0) d = 0;
        const double p = frq[icv];
        const double q = 1 - p;
        const double alpha = a + d * (q - p);
        A_chr += ((double)t - 2 * p) * alpha;
        const double ct = (t == 0) ? (-2 * p * p) : (t == 1 ? (2 * p * q) : (-2 * q * q));
        D_chr += ct * d;
    }
    add_out[ih * out_stride] = A_chr;
    dom_out[ih * out_stride] = D_chr;
    if (aw.add_tot) { aw.add_tot[ih * tot_stride] = 0.0 + A_chr; aw.dom_tot[ih * tot_stride] = 0.0 + D_chr; }   // one chromosome: the sum over chromosomes (:2729-2746) is 0 + x
    if (A_chr != A_chr || D_chr != D_chr) atomicMin(nan_flag, (u32)(ih < 0xffffffffull ? ih : 0xfffffffeull));
}
// One thread per individual; the block's 2*IPB haplotype rows are staged through LDS with a
// coalesced read ([hap][individual][S+1] layout: row stride S+1 words is odd -> conflict-free
// column reads).  col_of_icv / tab are wave-uniform (scalar loads); t selects per lane.
#define AD_CHUNK 128          // CVs whose table entries sit in LDS at a time
template <int IPB>
__global__ void __launch_bounds__(IPB) k_ad_accumulate_tab(const AdWork* __restrict__ At, u32 s1_max, size_t n_human, size_t out_stride, size_t tot_stride, u32* __restrict__ nan_flag, int direct)
{
    // direct (every CV file of the launch is in position order): a thread walks its two rows word by word, 32 CVs per word, so
    // it reads them straight from global memory (256 contiguous bytes per individual) and the launch passes s1_max = 0: only the
    // 6.6 KiB table chunk sits in LDS, which lets the block run in the LDS a concurrent stitch leaves over
    extern __shared__ u32 s_rows[];                              // [2 * IPB * s1_max] rows | [AD_CHUNK] columns | [AD_CHUNK * 6] table (8-byte aligned)
    const AdWork& aw = At[blockIdx.y];
    const u32 sub_w32 = aw.sub_w32, stride = aw.stride_w32; const u32* __restrict__ col_of_icv = aw.col_of_icv;
    const double* __restrict__ tab = aw.tab; const u32 Cn = aw.C; double* __restrict__ add_out = aw.add_out; double* __restrict__ dom_out = aw.dom_out;
    u32* s_col = s_rows + (((size_t)2 * IPB * s1_max + 1) & ~(size_t)1);
    double* s_tab = (double*)(s_col + AD_CHUNK);
    const u32 S1 = sub_w32 | 1u;                                 // odd row stride
    const size_t ih0 = (size_t)blockIdx.x * IPB;
    const size_t n_here = min((size_t)IPB, n_human - ih0);
    const u32 words = (u32)(2 * n_here) * sub_w32;
    const u32* src = aw.cvp + 2 * ih0 * stride;                  // the block's first CV-plane row
    if (!direct)
        for (u32 e = threadIdx.x; e < words; e += IPB) {
            const u32 row = e / sub_w32, w = e - row * sub_w32;     // row = 2*local_individual + hap
            s_rows[((row & 1u) * IPB + (row >> 1)) * S1 + w] = src[(size_t)row * stride + w];
        }
    const bool live = threadIdx.x < n_here;
    // direct: the thread reads the allele words of its two plane rows itself; staged: they sit in LDS
    const u32* r0 = direct ? src + (live ? 2 * (size_t)threadIdx.x * stride : 0) : s_rows + (size_t)threadIdx.x * S1;
    const u32* r1 = direct ? r0 + stride : s_rows + ((size_t)IPB + threadIdx.x) * S1;
    // vd == 0: the reference zeroes d (:2698-2699), every D-term is (+-0) * ... = +-0 and the running sum stays +0.0
    const bool skip_d = aw.vd == 0;
    double A_chr = 0, D_chr = 0;
    for (u32 base = 0; base < Cn; base += AD_CHUNK) {
        const u32 nj = min((u32)AD_CHUNK, Cn - base);
        __syncthreads();                                         // rows staged / previous chunk consumed
        if (!aw.cols_sorted) for (u32 e = threadIdx.x; e < nj; e += IPB) s_col[e] = col_of_icv[base + e];
        for (u32 e = threadIdx.x; e < nj * 6; e += IPB) s_tab[e] = tab[6 * (size_t)base + e];
        __syncthreads();
        // CVs in FILE order, sequential FP64 adds; the per-CV term is picked from the LDS table by the genotype t (0, 1, 2)
        if (aw.cols_sorted) {
            // file order == column order (the usual case: CV files are written by position): one word of each row serves 32 CVs
            // the chunk's (up to) four words of each row: one 16-byte load per row when the rows are 16-byte aligned (a 4-byte
            // walk refetches each 128-byte line of the rows from L2 up to 32 times once the working set of the CU exceeds its L1)
            u32 c0[4] = {0, 0, 0, 0}, c1[4] = {0, 0, 0, 0};
            const u32 wq = base >> 5;
            if (direct && (sub_w32 & 3u) == 0 && wq + 4 <= sub_w32) {
                const uint4 v0 = *(const uint4*)(r0 + wq), v1 = *(const uint4*)(r1 + wq);
                c0[0] = v0.x; c0[1] = v0.y; c0[2] = v0.z; c0[3] = v0.w; c1[0] = v1.x; c1[1] = v1.y; c1[2] = v1.z; c1[3] = v1.w;
            } else {
#pragma unroll
                for (u32 q = 0; q < 4; q++) if (wq + q < sub_w32) { c0[q] = r0[wq + q]; c1[q] = r1[wq + q]; }
            }
#pragma unroll
            for (u32 j0 = 0; j0 < AD_CHUNK; j0 += 32) {
                if (j0 >= nj) break;
                u32 w0 = c0[j0 >> 5], w1 = c1[j0 >> 5];
                const u32 m = min(32u, nj - j0);
                const double* tj = s_tab + 6 * j0;
                if (skip_d) {
#pragma unroll 4
                    for (u32 b = 0; b < m; b++) { const u32 t = (w0 & 1u) + (w1 & 1u); w0 >>= 1; w1 >>= 1; A_chr += tj[6 * b + t]; }
                } else {
#pragma unroll 4
                    for (u32 b = 0; b < m; b++) { const u32 t = (w0 & 1u) + (w1 & 1u); w0 >>= 1; w1 >>= 1; A_chr += tj[6 * b + t]; D_chr += tj[6 * b + 3 + t]; }
                }
            }
        } else if (skip_d) {
#pragma unroll 4
            for (u32 j = 0; j < nj; j++) {
                const u32 c = s_col[j];
                const u32 t = ((r0[c >> 5] >> (c & 31)) & 1u) + ((r1[c >> 5] >> (c & 31)) & 1u);
                A_chr += s_tab[6 * j + t];
            }
        } else {
#pragma unroll 4
            for (u32 j = 0; j < nj; j++) {
                const u32 c = s_col[j];
                const u32 t = ((r0[c >> 5] >> (c & 31)) & 1u) + ((r1[c >> 5] >> (c & 31)) & 1u);
                A_chr += s_tab[6 * j + t];
                D_chr += s_tab[6 * j + 3 + t];
            }
        }
    }
    if (!live) return;
    const size_t ih = ih0 + threadIdx.x;
    add_out[ih * out_stride] = A_chr;
    dom_out[ih * out_stride] = D_chr;
    if (aw.add_tot) { aw.add_tot[ih * tot_stride] = 0.0 + A_chr; aw.dom_tot[ih * tot_stride] = 0.0 + D_chr; }   // one chromosome: the sum over chromosomes (:2729-2746) is 0 + x
    if (A_chr != A_chr || D_chr != D_chr) atomicMin(nan_flag, (u32)(ih < 0xffffffffull ? ih : 0xfffffffeull));
}
// The same sums when every CV file of the launch is in position order and the allele sub-row is a whole number of 16-byte
// chunks: the term table goes through LDS in FEW large pieces (12 KiB: 512 CVs when only the A-terms are needed -- vd == 0, the
// D-sum stays +0.0 -- else 256 CVs), and a thread reads the piece's words of its two plane rows (up to 4 x 16 bytes each) in one
// go before the barrier: two to four global round trips per individual instead of eight (next to the dense stitch every one of
// them costs microseconds).  Same operations in the same order per individual as k_ad_accumulate_tab.
template <bool SKIP_D>
__global__ void __launch_bounds__(256) k_ad_accumulate_wide(const AdWork* __restrict__ At, size_t n_human, size_t out_stride, size_t tot_stride, u32* __restrict__ nan_flag)
{
    constexpr u32 PIECE = SKIP_D ? 512u : 256u, TERMS = SKIP_D ? 3u : 6u, WORDS = PIECE / 32u;
    __shared__ double s_tab[PIECE * TERMS];                           // 12 KiB
    const AdWork& aw = At[blockIdx.y];
    const double* __restrict__ tab = aw.tab;
    const u32 Cn = aw.C, stride = aw.stride_w32;
    const size_t ih = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = ih < n_human;
    const u32* r0 = aw.cvp + (live ? 2 * ih * stride : 0);
    const u32* r1 = r0 + stride;
    double A_chr = 0, D_chr = 0;
    for (u32 base = 0; base < Cn; base += PIECE) {
        const u32 nj = min(PIECE, Cn - base);
        u32 w0[WORDS], w1[WORDS];
#pragma unroll
        for (u32 q = 0; q < WORDS / 4; q++) {
            uint4 a = make_uint4(0u, 0u, 0u, 0u), b = a;
            if (q * 128 < nj) { a = *(const uint4*)(r0 + (base >> 5) + 4 * q); b = *(const uint4*)(r1 + (base >> 5) + 4 * q); }
            w0[4 * q] = a.x; w0[4 * q + 1] = a.y; w0[4 * q + 2] = a.z; w0[4 * q + 3] = a.w;
            w1[4 * q] = b.x; w1[4 * q + 1] = b.y; w1[4 * q + 2] = b.z; w1[4 * q + 3] = b.w;
        }
        __syncthreads();                                              // the previous piece is consumed
        for (u32 e = threadIdx.x; e < nj * TERMS; e += 256) {
            const u32 cv = e / TERMS, k = e - cv * TERMS;
            s_tab[e] = tab[6 * (size_t)(base + cv) + k];
        }
        __syncthreads();
#pragma unroll
        for (u32 q = 0; q < WORDS; q++) {
            if (q * 32 >= nj) break;
            u32 a = w0[q], b = w1[q];
            const u32 m = min(32u, nj - q * 32);
            const double* tj = s_tab + TERMS * (q * 32);
#pragma unroll 4
            for (u32 j = 0; j < m; j++) {
                const u32 t = (a & 1u) + (b & 1u); a >>= 1; b >>= 1;
                A_chr += tj[TERMS * j + t];
                if (!SKIP_D) D_chr += tj[TERMS * j + 3 + t];
            }
        }
    }
    if (!live) return;
    aw.add_out[ih * out_stride] = A_chr;
    aw.dom_out[ih * out_stride] = D_chr;
    if (aw.add_tot) { aw.add_tot[ih * tot_stride] = 0.0 + A_chr; aw.dom_tot[ih * tot_stride] = 0.0 + D_chr; }   // one chromosome: the sum over chromosomes (:2729-2746) is 0 + x
    if (A_chr != A_chr || D_chr != D_chr) atomicMin(nan_flag, (u32)(ih < 0xffffffffull ? ih : 0xfffffffeull));
}
// Several root populations with DIFFERENT CV effects (after migration a haplotype's a, d are its root population's, :2695-2696), CV
// files in position order: the arithmetic of k_ad_accumulate, statement by statement, with its loads arranged as in
// k_ad_accumulate_wide -- a thread reads the words of its two rows (alleles and the RPB root-population bit sub-rows) once per 128
// CVs, and the piece's a / d values of every root population, its 2p, q - p and the three dominance coefficients sit in LDS
// ([cv][population]: the lanes of a wave read neighbouring words).  k_ad_accumulate did eight dependent global loads per CV and lane.
#define ADRP_PIECE 128u
template <u32 RPB>
__global__ void __launch_bounds__(256) k_ad_accumulate_rp(const AdWork* __restrict__ At, u32 n_pop, size_t n_human, size_t out_stride, size_t tot_stride, u32* __restrict__ nan_flag)
{
    extern __shared__ double s_ad[];                                  // [PIECE][n_pop] a | [PIECE][n_pop] d | [PIECE][5]: 2p, q - p, -2pp, 2pq, -2qq
    const AdWork& aw = At[blockIdx.y];
    double* s_a = s_ad; double* s_d = s_a + ADRP_PIECE * n_pop; double* s_c = s_d + ADRP_PIECE * n_pop;
    const u32 Cn = aw.C, stride = aw.stride_w32, sub = aw.sub_w32;
    const double vd = aw.vd; const u64 bp0 = aw.bp0, bp_end = aw.bp_end;
    const size_t ih = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = ih < n_human;
    const u32* r0 = aw.cvp + (live ? 2 * ih * stride : 0);
    const u32* r1 = r0 + stride;
    const bool vec = (sub & 3u) == 0 && (stride & 3u) == 0;           // whole 16-byte chunks of every sub-row
    double A_chr = 0, D_chr = 0;
    for (u32 base = 0; base < Cn; base += ADRP_PIECE) {
        const u32 nj = min(ADRP_PIECE, Cn - base);
        u32 w0[1 + RPB][4], w1[1 + RPB][4];                           // [alleles, root-population bit 0 ..][word of the piece]
#pragma unroll
        for (u32 sr = 0; sr <= RPB; sr++) {
            const u32 wq = sr * sub + (base >> 5);
            if (vec) {
                const uint4 a = *(const uint4*)(r0 + wq), b = *(const uint4*)(r1 + wq);
                w0[sr][0] = a.x; w0[sr][1] = a.y; w0[sr][2] = a.z; w0[sr][3] = a.w; w1[sr][0] = b.x; w1[sr][1] = b.y; w1[sr][2] = b.z; w1[sr][3] = b.w;
            } else {
#pragma unroll
                for (u32 q = 0; q < 4; q++) { const bool in = (base >> 5) + q < sub; w0[sr][q] = in ? r0[wq + q] : 0u; w1[sr][q] = in ? r1[wq + q] : 0u; }
            }
        }
        __syncthreads();                                              // the previous piece is consumed
        for (u32 e = threadIdx.x; e < nj * n_pop; e += 256) {
            const u32 cv = e / n_pop, r = e - cv * n_pop; const u32 icv = base + cv;
            // a CV outside [bp0,bp_end) is covered by no part: its a, d stay 0 (Human_CV ctor, Population.h:99-108)
            const u64 x = aw.pos_file[icv];
            const bool covered = (x >= bp0 && x < bp_end);
            s_a[e] = covered ? aw.aptr[r][icv] : 0.0; s_d[e] = covered ? aw.dptr[r][icv] : 0.0;
        }
        for (u32 e = threadIdx.x; e < nj; e += 256) {
            const double p = aw.frq[base + e];
            const double q = 1 - p;
            s_c[5 * e] = 2 * p; s_c[5 * e + 1] = q - p; s_c[5 * e + 2] = -2 * p * p; s_c[5 * e + 3] = 2 * p * q; s_c[5 * e + 4] = -2 * q * q;
        }
        __syncthreads();
#pragma unroll
        for (u32 q = 0; q < 4; q++) {
            if (q * 32 >= nj) break;
            const u32 m = min(32u, nj - q * 32);
            for (u32 j = 0; j < m; j++) {
                const u32 t = ((w0[0][q] >> j) & 1u) + ((w1[0][q] >> j) & 1u);
                u32 rp0 = 0, rp1 = 0;
#pragma unroll
                for (u32 b = 0; b < RPB; b++) { rp0 |= ((w0[1 + b][q] >> j) & 1u) << b; rp1 |= ((w1[1 + b][q] >> j) & 1u) << b; }
                rp0 = min(rp0, n_pop - 1u); rp1 = min(rp1, n_pop - 1u);
                const u32 cv = q * 32 + j;
                const double a0 = s_a[cv * n_pop + rp0], a1 = s_a[cv * n_pop + rp1];
                const double d0 = s_d[cv * n_pop + rp0], d1 = s_d[cv * n_pop + rp1];
                const double a = (a0 + a1) / 2;
                double d = (d0 + d1) / 2;
                if (vd == 0) d = 0;
                const double* cc = s_c + 5 * cv;
                const double alpha = a + d * cc[1];
                A_chr += ((double)t - cc[0]) * alpha;
                const double ct = (t == 0) ? cc[2] : (t == 1 ? cc[3] : cc[4]);
                D_chr += ct * d;
            }
        }
    }
    if (!live) return;
    aw.add_out[ih * out_stride] = A_chr;
    aw.dom_out[ih * out_stride] = D_chr;
    if (aw.add_tot) { aw.add_tot[ih * tot_stride] = 0.0 + A_chr; aw.dom_tot[ih * tot_stride] = 0.0 + D_chr; }   // one chromosome: the sum over chromosomes (:2729-2746) is 0 + x
    if (A_chr != A_chr || D_chr != D_chr) atomicMin(nan_flag, (u32)(ih < 0xffffffffull ? ih : 0xfffffffeull));
}
// sum over chromosomes in order (:2729-2746), additive and dominance values in one launch
__global__ void k_ad_sum_chr(const double* __restrict__ add_chr /*[n][nchr][nphen]*/, const double* __restrict__ dom_chr, double* __restrict__ add /*[n][nphen]*/, double* __restrict__ dom, size_t n, int nchr, int nphen)
{
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n * nphen) return;
    const size_t ih = q / nphen; const int p = (int)(q % nphen);
    double sa = 0, sd = 0;
    for (int c = 0; c < nchr; c++) { sa += add_chr[(ih * nchr + c) * nphen + p]; sd += dom_chr[(ih * nchr + c) * nphen + p]; }
    add[q] = sa; dom[q] = sd;
}

// ------------------------------------------------------------------------------------------
// output materialisation == ras_convert_interval_to_hap_matrix (:1186-1230): plane XOR overlay
// ------------------------------------------------------------------------------------------
// rows [row0, row0+n) already copied into `out` (stride out_w32 words); flip loci whose position is in the row's set
__global__ void __launch_bounds__(256) k_snp_apply_mut(
    RowMap rm, u32* __restrict__ out, size_t out_w32, size_t row0, size_t n_rows,
    const u32* __restrict__ m_off, const u64* __restrict__ m_pos, const u64* __restrict__ pos, u32 L)
{
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    u32* o = out + r * out_w32;
    for (u32 j = m_off[row0 + r]; j < m_off[row0 + r + 1]; j++) {
        const u64 x = m_pos[j];
        u32 c = lower_bound_u64(pos, L, x);
        for (; c < L && pos[c] == x; c++) {
            const u32 f = (rm.word32(row0 + r, c >> 5) >> (c & 31)) & 1u;
            if (f) o[c >> 5] &= ~(1u << (c & 31)); else o[c >> 5] |= (1u << (c & 31));
        }
    }
}
// K8: genotype tile from the ancestry intervals == Simulation::ras_convert_interval_to_hap_matrix (src/Simulation.cpp:1186-1230)
// restricted to haplotype rows [row0, row0+n_rows) x loci [s0, s0+ns): out[r][ii] = founder[part.hap_index][ii] for the part
// that contains pos[ii]; loci outside every part stay 0.  One thread per 32-bit word of the tile; the parts of a row are
// disjoint and ascending, so the first candidate is found by bisection on `en` and usually covers the whole word.
// `founder` holds the same loci range of every founder haplotype (bit j = locus s0 + j).  Mutations: k_tile_apply_mut.
__global__ void __launch_bounds__(256) k_materialize_tile(const u32* __restrict__ p_off, const gev_part* __restrict__ parts, size_t row0, size_t n_rows,
                                                          const u64* __restrict__ pos, u32 s0, u32 ns, const u32* __restrict__ founder, size_t founder_w32,
                                                          const u64* __restrict__ founder_row0 /* [n_pop+1]: rows of root population p are [row0[p], row0[p+1]) of `founder` */,
                                                          int n_pop, u32* __restrict__ out, size_t out_w32, u32* __restrict__ status)
{
    const u32 words = (ns + 31) / 32;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_rows * words) return;
    const size_t r = q / words; const u32 w = (u32)(q % words);
    const u32 nb = min(32u, ns - 32u * w);
    const u64* wp = pos + s0 + 32u * w;                       // positions of this word's loci
    const u64 x0 = wp[0], x1 = wp[nb - 1];
    u32 lo = p_off[row0 + r], hi = p_off[row0 + r + 1];
    const u32 end = hi;
    while (lo < hi) { const u32 m = (lo + hi) >> 1; if (parts[m].en <= x0) lo = m + 1; else hi = m; }    // first part with en > x0
    u32 acc = 0;
    for (u32 i = lo; i < end && parts[i].st <= x1; i++) {
        const u64 st = parts[i].st, en = parts[i].en;
        u32 a = 0, b = 0;
        for (u32 t = 0; t < nb; t++) { a += wp[t] < st ? 1u : 0u; b += wp[t] < en ? 1u : 0u; }        // loci [a, b) lie in [st, en)
        if (b > a) {
            const u32 mask = (b - a == 32u) ? 0xffffffffu : (((1u << (b - a)) - 1u) << a);
            const u64 h = parts[i].hap_index;
            const int rp = parts[i].root_population;                                                  // founder panel of the part's ROOT population (:1204)
            if (rp < 0 || rp >= n_pop || h >= founder_row0[rp + 1] - founder_row0[rp]) { atomicOr(status, 1u); continue; }   // :1205-1209 "hap_index is not in range"
            acc |= founder[(founder_row0[rp] + h) * founder_w32 + w] & mask;
        }
    }
    out[r * out_w32 + w] = acc;
}
// Diagnostic (gev_dbg_verify_planes): EVERY 32-bit word of the resident genotype plane (written by the dense stitch K5 from
// breakpoints) against the reference's materialisation rule applied to the ancestry intervals (written by K4, k_parts) --
// two independent paths.  The founder panel is the synthetic one (k_synth_rows), recomputed on the fly, so no founder copy
// has to be resident.  Mutations are not in the plane (sparse overlay), so none are applied here.
__global__ void __launch_bounds__(256) k_verify_plane(const u32* __restrict__ p_off, const gev_part* __restrict__ parts, size_t n_rows,
                                                      const u64* __restrict__ pos, u32 L, RowMap rm,
                                                      const u32* __restrict__ thr /* [n_pop][L] */, const u64* __restrict__ seeds /* [n_pop] */, int n_pop, const u64* __restrict__ n_founder_rows /* [n_pop] */,
                                                      unsigned long long* __restrict__ n_bad /* [0] mismatching words, [1] parts with an unknown / out-of-range founder */)
{
    const u32 words = (L + 31) / 32;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_rows * words) return;
    const size_t r = q / words; const u32 w = (u32)(q % words);
    const u32 nb = min(32u, L - 32u * w);
    const u64* wp = pos + 32u * w;
    const u64 x0 = wp[0], x1 = wp[nb - 1];
    u32 lo = p_off[r], hi = p_off[r + 1];
    const u32 end = hi;
    while (lo < hi) { const u32 m = (lo + hi) >> 1; if (parts[m].en <= x0) lo = m + 1; else hi = m; }
    u32 acc = 0;
    for (u32 i = lo; i < end && parts[i].st <= x1; i++) {
        const u64 st = parts[i].st, en = parts[i].en;
        u32 a = 0, b = 0;
        for (u32 t = 0; t < nb; t++) { a += wp[t] < st ? 1u : 0u; b += wp[t] < en ? 1u : 0u; }
        if (b <= a) continue;
        const u64 h = parts[i].hap_index;
        const int rp = parts[i].root_population;                                   // founder panel of the part's ROOT population (:1204)
        if (rp < 0 || rp >= n_pop || h >= n_founder_rows[rp]) { atomicAdd(&n_bad[1], 1ull); continue; }
        const u64 off = seeds[rp] * 0xD1342543DE82EF95ull;
        const u32* __restrict__ th = thr + (size_t)rp * L;
        for (u32 t = a; t < b; t++) {
            const size_t ii = (size_t)32 * w + t;
            const u64 ctr = (((u64)h << 32) | (u64)ii) + off;
            if ((u32)(mix64(ctr) >> 32) < th[ii]) acc |= 1u << t;
        }
    }
    if (acc != rm.word32(r, w)) atomicAdd(&n_bad[0], 1ull);
}
// mutation overlay of a tile: out bit = !unmutated bit at every tile locus whose position is in the row's mutation list (:1212-1216)
__global__ void __launch_bounds__(256) k_tile_apply_mut(const u32* __restrict__ plain, u32* __restrict__ out, size_t w32, size_t row0, size_t n_rows,
                                                        const u32* __restrict__ m_off, const u64* __restrict__ m_pos, const u64* __restrict__ pos, u32 s0, u32 ns)
{
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const u32* in = plain + r * w32; u32* o = out + r * w32;
    for (u32 j = m_off[row0 + r]; j < m_off[row0 + r + 1]; j++) {
        const u64 x = m_pos[j];
        u32 c = lower_bound_u64(pos + s0, ns, x);
        for (; c < ns && pos[s0 + c] == x; c++) {
            const u32 f = (in[c >> 5] >> (c & 31)) & 1u;
            if (f) o[c >> 5] &= ~(1u << (c & 31)); else o[c >> 5] |= (1u << (c & 31));
        }
    }
}
// generic row gather (migration, capacity growth, download staging): dst row r <- src row map[r]
__global__ void __launch_bounds__(256) k_gather_rows16(uint4* __restrict__ dst, size_t dst_stride16, const uint4* __restrict__ src, size_t src_stride16,
                                                       const u32* __restrict__ map, size_t n_rows, u32 chunks)
{
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_rows * chunks) return;
    const size_t r = q / chunks; const u32 c = (u32)(q % chunks);
    dst[r * dst_stride16 + c] = src[(size_t)map[r] * src_stride16 + c];
}
// rows between the segment pool and flat buffers (migration, order restoring, download staging): one side or both can be a
// flat array of whole rows (phys == null: row r at base + r * stride16 chunks) or a pool addressed through a unit table
struct RowRef {
    uint8_t* base; const u32* phys; size_t stride16; u32 nseg, seg_shift;
    __device__ __forceinline__ uint4* chunk(size_t r, u32 q) const
    {
        if (!phys) return (uint4*)base + r * stride16 + q;
        const u32 unit = phys[r * nseg + (q >> seg_shift)];
        return (uint4*)(base + ((size_t)unit << (seg_shift + 4))) + (q & ((1u << seg_shift) - 1u));
    }
};
// dst row r <- src row (map ? map[r] : base + r)
__global__ void __launch_bounds__(256) k_copy_rows16(RowRef dst, RowRef src, const u32* __restrict__ map, size_t base, size_t n_rows, u32 chunks)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_rows * chunks) return;
    const size_t r = e / chunks; const u32 q = (u32)(e % chunks);
    const size_t l = map ? (size_t)map[r] : base + r;
    *dst.chunk(r, q) = *src.chunk(l, q);
}
// Migrants that travel as lists (gev_set_migrant_rows(ctx, 0)): the genotype row of an immigrant is rebuilt where it arrives from
// its ancestry intervals and the founder panels kept there (gev_upload_founder_panel / gev_synth_founder_panel, one per ROOT
// population) -- Simulation::ras_convert_interval_to_hap_matrix's rule (src/Simulation.cpp:1198-1211) for the rows of the
// resident plane, which holds the founder mosaic without the mutation overlay (the mutation list travels and stays sparse).
// The parts of the records are turned into locus ranges once (k_parts_locus_range), the rows are then assembled from the ranges alone.
struct PanelRef { const u32* base; u64 w32; u64 rows; };        // flat rows of w32 words; rows == 0: no panel of this root population here
// locus range of every part: locus ii lies in [st, en) exactly when lower_bound(pos, st) <= ii < lower_bound(pos, en) (pos ascends)
__global__ void __launch_bounds__(256) k_parts_locus_range(const gev_part* __restrict__ parts, size_t n_parts, const ChrDev* __restrict__ chrs, int chr, uint2* __restrict__ range)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_parts) return;
    const ChrDev& C = chrs[chr];
    const u32 a = snp_lower_bound(C, parts[i].st), b = snp_lower_bound(C, parts[i].en);                // (one table read + a search inside the bucket)
    range[i] = make_uint2(a, max(a, b));
}
#define REBUILD_CPW 256u        // chunks per wave of k_rebuild_rows: four steps of 64 lanes behind one bisection
__global__ void __launch_bounds__(256) k_rebuild_rows(const u32* __restrict__ p_off /* [n_rows + 1], into `parts` */, const gev_part* __restrict__ parts,
                                                      const uint2* __restrict__ range /* k_parts_locus_range */, u32 L, const PanelRef* __restrict__ panels, int n_pop,
                                                      RowRef dst, u32 chunks, u32* __restrict__ status /* |1: hap_index out of range, |2: no panel of a root population */)
{
    // grid: x = row, y = blocks of 4 waves x REBUILD_CPW chunks of that row -- a wave works on 1024 consecutive words of one row.
    // The kernel is a chain of dependent loads (bisection -> range -> part -> panel -> founder words): one bisection per wave
    // (the same addresses on every lane: broadcast loads), then every word walks on from where the lane's previous word stood
    // (the parts of a row are disjoint and ascending), and a lane's four chunks keep four founder loads in flight.
    const size_t r = blockIdx.x;
    const u32 qbase = (blockIdx.y * 4u + (threadIdx.x >> 6)) * REBUILD_CPW;
    if (qbase >= chunks) return;
    const u32 words = (L + 31) / 32;
    const u32 begin = p_off[r], end = p_off[r + 1];
    const u32 l0 = 128u * qbase;                                                                       // first locus of the wave
    u32 lo = begin, hi = end;
    while (lo < hi) { const u32 m = (lo + hi) >> 1; if (range[m].y <= l0) lo = m + 1; else hi = m; }    // first part that ends behind it
    for (u32 q = qbase + (threadIdx.x & 63u); q < min(qbase + REBUILD_CPW, chunks); q += 64u) {
        const u32 c0 = 128u * q, c1 = min(c0 + 128u, L);                                              // the chunk's loci [c0, c1)
        if (c0 >= L) { *dst.chunk(r, q) = make_uint4(0u, 0u, 0u, 0u); continue; }                      // padding behind the last locus
        while (lo < end && range[lo].y <= c0) lo++;
        if (lo < end) {
            // the usual case: one part covers the whole chunk -- one 16-byte load of the founder row (its pad bits are zero)
            const uint2 g = range[lo];
            if (g.x <= c0 && g.y >= c1) {
                const u64 h = parts[lo].hap_index;
                const int rp = parts[lo].root_population;
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (rp < 0 || rp >= n_pop || !panels[rp].rows) atomicOr(status, 2u);
                else if (h >= panels[rp].rows) atomicOr(status, 1u);
                else v = *(const uint4*)(panels[rp].base + h * panels[rp].w32 + 4u * q);
                *dst.chunk(r, q) = v;
                continue;
            }
        }
        u32 o[4] = {0u, 0u, 0u, 0u};
        for (u32 t4 = 0; t4 < 4; t4++) {
            const u32 w = 4 * q + t4;
            if (w >= words) break;                                                                    // pad words of the row stay 0
            const u32 i0 = 32u * w, i1 = min(i0 + 32u, L);                                            // the word's loci [i0, i1)
            while (lo < end && range[lo].y <= i0) lo++;
            u32 acc = 0;
            for (u32 i = lo; i < end; i++) {
                const uint2 g = range[i];
                if (g.x >= i1) break;
                const u32 a = max(g.x, i0) - i0, b = min(g.y, i1) - i0;                                // bits [a, b) of the word
                if (b > a) {
                    const u32 mask = (b - a == 32u) ? 0xffffffffu : (((1u << (b - a)) - 1u) << a);
                    const u64 h = parts[i].hap_index;
                    const int rp = parts[i].root_population;
                    if (rp < 0 || rp >= n_pop || !panels[rp].rows) { atomicOr(status, 2u); continue; }
                    if (h >= panels[rp].rows) { atomicOr(status, 1u); continue; }                      // :1205-1209 "hap_index is not in range"
                    acc |= panels[rp].base[h * panels[rp].w32 + w] & mask;
                }
            }
            o[t4] = acc;
        }
        *dst.chunk(r, q) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}
// dst[i] = src[map[i << shift] >> shift]: bytes of selected individuals (shift = 1: `map` holds haplotype rows 2*individual, 2*individual+1)
__global__ void k_gather_u8(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, const u32* __restrict__ map, u32 shift, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[map[i << shift] >> shift];
}
// CSR gather: count / fill of rows selected by map (element size templated)
__global__ void k_csr_gather_count(const u32* __restrict__ s_off, const u32* __restrict__ map, size_t n_rows, u32* __restrict__ cnt)
{
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_rows) cnt[r] = s_off[map[r] + 1] - s_off[map[r]];
}
template <class E>
__global__ void k_csr_gather_fill(const u32* __restrict__ s_off, const E* __restrict__ s_val, const u32* __restrict__ map, size_t n_rows,
                                  const u32* __restrict__ d_off, E* __restrict__ d_val)
{
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const u32 a = s_off[map[r]], n = s_off[map[r] + 1] - a, o = d_off[r];
    for (u32 j = 0; j < n; j++) d_val[o + j] = s_val[a + j];
}

// ------------------------------------------------------------------------------------------
// Simulation::ras_scale_AD_compute_GEF (src/Simulation.cpp:3075-3206)
// std::normal_distribution<double> on minstd_rand0 = Marsaglia polar method with rejection
// (libstdc++ bits/random.tcc:1802-1835): candidate pair j consumes engine outputs 4j+1..4j+4, the
// k-th ACCEPTED pair yields normals 2k (= y*mult) and 2k+1 (= x*mult).  Candidates are evaluated in
// parallel (LCG jump-ahead) and compacted with a scan.  u = generate_canonical is computed in FP64
// exactly as libstdc++ does (no contraction); log() is the device libm (<= 1 ulp from glibc's).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double canonical_f64(u32 x1, u32 x2)
{
    const double R = 2147483646.0, R2 = 0x1.fffffff000000p+61;
    const double t = (double)(x2 - 1) * R;
    const double sum = (double)(x1 - 1) + t;
    double ret = sum / R2;
    if (ret >= 1.0) ret = 0x1.fffffffffffffp-1;       // nextafter(1, 0)
    return ret;
}
__global__ void __launch_bounds__(256) k_polar_candidates(u32 engine_seed, size_t n_cand, u32* __restrict__ flag, double* __restrict__ first, double* __restrict__ second)
{
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_cand) return;
    u32 x = mulmod31(powmod31(16807u, 4 * j + 1), minstd_seed(engine_seed));
    const u32 x1 = x; x = mulmod31(x, 16807u); const u32 x2 = x; x = mulmod31(x, 16807u); const u32 x3 = x; x = mulmod31(x, 16807u); const u32 x4 = x;
    const double xx = 2.0 * canonical_f64(x1, x2) - 1.0;
    const double yy = 2.0 * canonical_f64(x3, x4) - 1.0;
    const double r2 = xx * xx + yy * yy;
    const bool ok = !(r2 > 1.0 || r2 == 0.0);
    flag[j] = ok ? 1u : 0u;
    if (ok) { const double mult = sqrt(-2 * log(r2) / r2); first[j] = yy * mult; second[j] = xx * mult; }
}
__global__ void __launch_bounds__(256) k_polar_emit(const u32* __restrict__ flag, const u32* __restrict__ off, const double* __restrict__ first, const double* __restrict__ second,
                                                    size_t n_cand, size_t n, double sd, double* __restrict__ out)
{
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_cand || !flag[j]) return;
    const size_t k = off[j];
    if (2 * k < n) out[2 * k] = first[j] * sd + 0.0;          // ret * stddev + mean
    if (2 * k + 1 < n) out[2 * k + 1] = second[j] * sd + 0.0;
}
// deterministic two-level sum of (x[i]-shift)^pw, pw in {1,2}
__global__ void __launch_bounds__(256) k_sum_partial(const double* __restrict__ x, size_t n, double shift, int pw, double* __restrict__ partial)
{
    __shared__ double s[256];
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const double v = x[i] - shift; acc += pw == 2 ? v * v : v; }
    s[threadIdx.x] = acc; __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0];
}
__global__ void __launch_bounds__(256) k_sum_final(const double* __restrict__ partial, int nb, double* __restrict__ out)
{
    __shared__ double s[256];
    s[threadIdx.x] = (int)threadIdx.x < nb ? partial[threadIdx.x] : 0.0; __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0) *out = s[0];
}
// :3174-3203; raw a, d strided by nphen
__global__ void __launch_bounds__(256) k_gef_apply(const double* __restrict__ a, const double* __restrict__ d, size_t stride, const double* __restrict__ e, const double* __restrict__ par_eff,
                                                   const double* __restrict__ common, size_t n, double s_a, double s_d, double s_ev, double vf,
                                                   double* __restrict__ o_add, double* __restrict__ o_dom, double* __restrict__ o_bv, double* __restrict__ o_e, double* __restrict__ o_par, double* __restrict__ o_phen)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double en = s_ev > 0 ? e[i] / s_ev : 0;
    const double ad = a[i * stride] / s_a;
    const double dm = s_d > 0 ? d[i * stride] / s_d : 0;
    const double pe = vf > 0 ? par_eff[i] : 0;
    const double cs = common ? common[i] : 0.0;
    o_e[i] = en; o_add[i] = ad; o_dom[i] = dm; o_bv[i] = ad + dm; o_par[i] = pe;
    o_phen[i] = ad + dm + cs + en + pe;
}
__global__ void __launch_bounds__(256) k_par_eff(const double* __restrict__ ff, const double* __restrict__ fm, size_t n, double beta, double* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = beta * ((ff ? ff[i] : 0.0) + (fm ? fm[i] : 0.0));
}

// ------------------------------------------------------------------------------------------
// K9: output packing.  The reference's .hap files are SNP-major text
// (format_hap::write_hap, src/format_hap.cpp:6-30); PLINK .bed is SNP-major 2-bit.  The resident
// plane is haplotype-major, so output = 64x64 bit-tile transposes (64 ballots per tile: lane b ends
// up with SNP 64*sw+b across 64 haplotypes), then the sparse mutation overlay, then formatting.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_transpose_tiles(const u64* __restrict__ plane /* flat rows, or null: */, RowMap rm /* segment pool */, size_t stride_w64, size_t n_rows, u32 L,
                                                         u32 snp_begin, u32 n_snps, u64* __restrict__ out, size_t out_stride_w64, u32 words_per_wave)
{
    const u32 lane = threadIdx.x & 63;
    const size_t hb = (size_t)blockIdx.x;                                   // block of 64 haplotype rows
    const u32 wave = blockIdx.y * 4 + (threadIdx.x >> 6);
    const u32 sw_first = snp_begin >> 6, sw_last = (snp_begin + n_snps - 1) >> 6;
    const u32 sw0 = sw_first + wave * words_per_wave;
    const size_t row = hb * 64 + lane;
    for (u32 sw = sw0; sw < sw0 + words_per_wave && sw <= sw_last; sw++) {
        const u64 v = row < n_rows ? (plane ? plane[row * stride_w64 + sw] : rm.word64(row, sw)) : 0ull;
        u64 mine = 0;
#pragma unroll 8
        for (u32 b = 0; b < 64; b++) {
            const u64 m = __ballot((v >> b) & 1ull);
            if (lane == b) mine = m;
        }
        const u32 snp = sw * 64 + lane;
        if (snp >= snp_begin && snp < snp_begin + n_snps && snp < L) out[(size_t)(snp - snp_begin) * out_stride_w64 + hb] = mine;
    }
}
// flip (snp, hap) where the SNP position is in the haplotype's mutation set: value = !founder (idempotent)
__global__ void __launch_bounds__(256) k_snpmajor_apply_mut(RowMap rm, size_t n_rows,
                                                            const u32* __restrict__ m_off, const u64* __restrict__ m_pos, const u64* __restrict__ pos, u32 L,
                                                            u32 snp_begin, u32 n_snps, unsigned long long* __restrict__ out, size_t out_stride_w64)
{
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    for (u32 j = m_off[r]; j < m_off[r + 1]; j++) {
        const u64 x = m_pos[j];
        u32 c = lower_bound_u64(pos, L, x);
        for (; c < L && pos[c] == x; c++) {
            if (c < snp_begin || c >= snp_begin + n_snps) continue;
            const u32 f = (rm.word32(r, c >> 5) >> (c & 31)) & 1u;
            unsigned long long* w = out + (size_t)(c - snp_begin) * out_stride_w64 + (r >> 6);
            const unsigned long long bit = 1ull << (r & 63);
            if (f) atomicAnd(w, ~bit); else atomicOr(w, bit);
        }
    }
}
// one .hap line per SNP: "b b b ... b \n" (digit + space per haplotype, then newline)
// The text is produced as a flat byte stream: thread t owns the aligned 16 bytes [16t, 16t+16) of the output (lines have odd
// lengths, so line-relative ownership would make every store misaligned), one 16-byte store per thread.
__global__ void __launch_bounds__(256) k_format_hap_text(const u64* __restrict__ snpmajor, size_t stride_w64, size_t n_rows, u32 n_snps, char* __restrict__ out)
{
    const size_t line_len = 2 * n_rows + 1, total = (size_t)n_snps * line_len;
    const size_t o = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (o >= total) return;
    size_t j = o / line_len, p = o % line_len;
    union { char ch[16]; uint4 v; } u;
    u64 word = 0; size_t word_at = ~(size_t)0;
#pragma unroll
    for (int b = 0; b < 16; b++) {
        char ch = ' ';
        if (o + b >= total) ch = 0;
        else if (p == 2 * n_rows) ch = '\n';
        else if (!(p & 1)) {
            const size_t h = p >> 1, at = j * stride_w64 + (h >> 6);
            if (at != word_at) { word = snpmajor[at]; word_at = at; }
            ch = (char)('0' + (int)((word >> (h & 63)) & 1ull));
        }
        u.ch[b] = ch;
        if (++p == line_len) { p = 0; j++; }
    }
    if (o + 16 <= total) *reinterpret_cast<uint4*>(out + o) = u.v;
    else for (int b = 0; b < 16 && o + b < total; b++) out[o + b] = u.ch[b];
}
// CommFunc::ras_rank (src/CommFunc.cpp:152-161): r[k] = #{j : x[j] < x[k]} + #{j < k : x[j] == x[k]}  -- what the reference's
// O(n^2) pair loop leaves in r[k] (for j < i: x[j] <= x[i] ? r[i]++ : r[j]++).  Evaluated as the same all-pairs count, one
// element per thread against LDS tiles of 1024 values; NaNs follow the reference's comparisons literally.
__global__ void __launch_bounds__(256) k_rank_f64(const double* __restrict__ x, size_t n, unsigned long long* __restrict__ rank)
{
    __shared__ double tile[1024];
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    const double xk = k < n ? x[k] : 0.0;
    unsigned long long r = 0;
    for (size_t base = 0; base < n; base += 1024) {
        __syncthreads();
        for (u32 t = threadIdx.x; t < 1024; t += 256) tile[t] = base + t < n ? x[base + t] : 0.0;
        __syncthreads();
        const u32 m = (u32)min((size_t)1024, n - base);
        if (k < n) {
            u32 cnt = 0;
            if (base + m <= k) {                               // every j < k:  x[j] <= x[k] -> r[k]++
#pragma unroll 8
                for (u32 t = 0; t < m; t++) cnt += tile[t] <= xk ? 1u : 0u;
            } else if (base > k) {                             // every j > k:  !(x[k] <= x[j]) -> r[k]++
#pragma unroll 8
                for (u32 t = 0; t < m; t++) cnt += !(xk <= tile[t]) ? 1u : 0u;
            } else {
                for (u32 t = 0; t < m; t++) {
                    const size_t j = base + t;
                    if (j < k) cnt += tile[t] <= xk ? 1u : 0u; else if (j > k) cnt += !(xk <= tile[t]) ? 1u : 0u;
                }
            }
            r += cnt;
        }
    }
    if (k < n) rank[k] = r;
}
// VCF genotype columns of one data line (format_vcf::write_vcf_file, src/format_vcf.cpp:55-59): per individual "\ta|b"
// (a, b = haplotype rows 2i, 2i+1 at this SNP), then '\n'; flat-stream layout as above
__global__ void __launch_bounds__(256) k_format_vcf_gt(const u64* __restrict__ snpmajor, size_t stride_w64, size_t n_ind, u32 n_snps, char* __restrict__ out)
{
    const size_t line_len = 4 * n_ind + 1, total = (size_t)n_snps * line_len;
    const size_t o = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (o >= total) return;
    size_t j = o / line_len, p = o % line_len;
    union { char ch[16]; uint4 v; } u;
    u64 word = 0; size_t word_at = ~(size_t)0;
#pragma unroll
    for (int b = 0; b < 16; b++) {
        char ch;
        if (o + b >= total) ch = 0;
        else if (p == 4 * n_ind) ch = '\n';
        else if ((p & 3) == 0) ch = '\t';
        else if ((p & 3) == 2) ch = '|';
        else {
            const size_t h = p >> 1, at = j * stride_w64 + (h >> 6);      // p = 4i+1 -> row 2i, p = 4i+3 -> row 2i+1
            if (at != word_at) { word = snpmajor[at]; word_at = at; }
            ch = (char)('0' + (int)((word >> (h & 63)) & 1ull));
        }
        u.ch[b] = ch;
        if (++p == line_len) { p = 0; j++; }
    }
    if (o + 16 <= total) *reinterpret_cast<uint4*>(out + o) = u.v;
    else for (int b = 0; b < 16 && o + b < total; b++) out[o + b] = u.ch[b];
}
// PLINK .ped genotype columns (format_plink::write_ped_map / write_ped01_map, src/format_plink.cpp:42-49 / :114-121):
// per individual  L x " a b"  then '\n', a/b = allele letters of haplotype 0/1 (al1 if the bit is set else al0; "1"/"0"
// when al0 == NULL).  `rows` = staged hap-major rows (mutations applied) of individuals [0, n_ind); same flat-stream layout.
__global__ void __launch_bounds__(256) k_format_ped_text(const u32* __restrict__ rows, size_t stride_w32, size_t n_ind, u32 L,
                                                         const char* __restrict__ al0, const char* __restrict__ al1, char* __restrict__ out)
{
    const size_t line_len = 4 * (size_t)L + 1, total = n_ind * line_len;
    const size_t o = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (o >= total) return;
    size_t i = o / line_len, p = o % line_len;
    union { char ch[16]; uint4 v; } u;
    u32 w0 = 0, w1 = 0; size_t word_at = ~(size_t)0;
#pragma unroll
    for (int b = 0; b < 16; b++) {
        char ch = ' ';
        if (o + b >= total) ch = 0;
        else if (p == 4 * (size_t)L) ch = '\n';
        else if (p & 1) {
            const u32 s = (u32)(p >> 2), hap = (u32)(p >> 1) & 1u;
            const size_t at = 2 * i * stride_w32 + (s >> 5);
            if (at != word_at) { w0 = rows[at]; w1 = rows[at + stride_w32]; word_at = at; }
            const u32 bit = ((hap ? w1 : w0) >> (s & 31)) & 1u;
            ch = al0 ? (bit ? al1[s] : al0[s]) : (char)('0' + bit);
        }
        u.ch[b] = ch;
        if (++p == line_len) { p = 0; i++; }
    }
    if (o + 16 <= total) *reinterpret_cast<uint4*>(out + o) = u.v;
    else for (int b = 0; b < 16 && o + b < total; b++) out[o + b] = u.ch[b];
}
// matrix_plink_ped of ras_convert_interval_to_format_plink (src/Simulation.cpp:1308-1362): bit 2*ii+ihap of row ih
__device__ __forceinline__ u64 spread_bits(u32 x)
{
    u64 v = x;
    v = (v | (v << 16)) & 0x0000ffff0000ffffull; v = (v | (v << 8)) & 0x00ff00ff00ff00ffull;
    v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0full;  v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}
__global__ void __launch_bounds__(256) k_interleave_haps(const u32* __restrict__ rows, size_t stride_w32, size_t n_ind, u32 words, u64* __restrict__ out, size_t out_stride_w64)
{
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_ind * words) return;
    const size_t i = q / words, w = q % words;
    out[i * out_stride_w64 + w] = spread_bits(rows[2 * i * stride_w32 + w]) | (spread_bits(rows[(2 * i + 1) * stride_w32 + w]) << 1);
}
// PLINK .bed body, SNP-major: 2 bits per individual, A1 = allele 1: 00 = 1/1, 10 = heterozygous, 11 = 0/0, pad = 00
__global__ void __launch_bounds__(256) k_format_bed(const u64* __restrict__ snpmajor, size_t stride_w64, size_t n_people, u32 n_snps, uint8_t* __restrict__ out)
{
    const size_t bpl = (n_people + 3) / 4;
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (size_t)n_snps * bpl) return;
    const size_t j = q / bpl, by = q % bpl;
    const u32 bits = (u32)((snpmajor[j * stride_w64 + (by >> 3)] >> ((by & 7) * 8)) & 0xffull);     // haplotypes 8*by .. 8*by+7
    u32 o = 0;
    for (u32 i = 0; i < 4; i++) {
        if (by * 4 + i >= n_people) break;
        const u32 b0 = (bits >> (2 * i)) & 1u, b1 = (bits >> (2 * i + 1)) & 1u;
        const u32 code = (b0 & b1) ? 0u : ((b0 ^ b1) ? 2u : 3u);
        o |= code << (2 * i);
    }
    out[q] = (uint8_t)o;
}

#define GEV_LISTS_KERNELS
#include "gev_lists.h"
#include "gev_mate.h"
