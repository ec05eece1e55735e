// gev_sample8.h -- batched crossover / mutation sampling (K1-K3), gfx950 wave64.  Included by gev_kernels.h.
//
// Same arithmetic as the one-task-per-wave kernels of gev_kernels.h (k_mut_sample, k_rec_sample: bit-exact restatements of
// Simulation::ras_add_mutation, src/Simulation.cpp:2497-2552, and Simulation::ras_sim_loc_rec, :2973-2995, with the rand()
// chain of Simulation::reproduce, :2447-2455), reorganised so that the per-task fixed costs are shared by EIGHT tasks:
//
//   * srand(): the glibc TYPE_3 generator is linear, output j = sum_i W[i][j] * r_i (mod 2^32).  The old form evaluates all 64
//     outputs of one generator per wave (31 multiply steps) and uses ~3 of them.  Here lane (g, j) = (lane >> 3, lane & 7) holds
//     output j of generator g: ONE pass of 31 multiply steps yields the first 8 outputs of 8 generators.
//   * map scan: unchanged inner loop (lane l owns draws l, l+64, ...; only the high digit of generate_canonical is produced per
//     draw and compared with the map's largest a_hi), but a candidate is only RECORDED (row, high digit, task) in an LDS list.
//   * resolve: the candidates of all 8 scans are tested against their threshold rows lane-parallel (one 16-byte load per
//     candidate instead of a divergent load inside the scan loop), hits are ranked per task with ballots, and the per-hit work
//     (breakpoint = bp[row] + rand() % dist; mutation position by uniform_int_distribution, side = rand() % 2) runs one hit per
//     lane.
//
// A task that needs more than the 8 precomputed rand() outputs (k + 2 > 8 crossovers / mutations), or whose mutation
// position draw hits the rejection branch of uniform_int_distribution, is appended to a list and redone by the one-task-
// per-wave kernels (expected: ~1e-4 of the tasks at one event per gamete).
#pragma once

#define SB_TASKS 8          // tasks per wave batch
#define SB_OUT 8            // rand() outputs per generator held in the lane grid
#define SB_CAND 320         // candidate slots per wave: a flush is forced at >= 64, one scan step adds <= 256

#define SB_J 32             // draws per lane that are prefiltered from ONE exact engine state (64 * SB_J = 2048 draws per block)
struct __attribute__((aligned(32))) SmpTabs {   // constant tables, one copy per workgroup (LDS)
    double kd[SB_J];        // c_j / M with c_j = 16807^(128 j) mod M: draw (lane + 64 j)'s high digit = x_lane * c_j mod M
    u32 ci[SB_J];           // c_j
    u32 pow_even[64];       // 16807^(2l+2): lane l's first high-digit engine output
    u32 w8[31 * SB_OUT];    // w_init[i][j], j < 8
    u32 pow_lcg[32];        // 16807^(i-1), i = 1..30
    u32 pow128, pow_blk /* 16807^(128 SB_J): next block of draws */, inv16807, pad;
};
struct SmpWave {            // scratch of one wave (LDS)
    u32 r[SB_TASKS * 33];   // seed words r_0..r_30 of the 8 generators, odd stride
    u32 cand_row[SB_CAND];
    u32 cand_x2[SB_CAND];
    uint8_t cand_tag[SB_CAND];
};

__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ u32 mbcnt64(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u)); }

__device__ __forceinline__ void stage_smp_tabs(const GevRngTables* __restrict__ g, SmpTabs* s)
{
    for (u32 i = threadIdx.x; i < 64; i += blockDim.x) s->pow_even[i] = g->pow_even[i];
    for (u32 i = threadIdx.x; i < 31 * SB_OUT; i += blockDim.x) s->w8[i] = g->w_init[(i / SB_OUT) * 64 + (i % SB_OUT)];
    for (u32 i = threadIdx.x; i < 31; i += blockDim.x) s->pow_lcg[i] = g->pow_lcg[i];
    if (threadIdx.x < SB_J) {
        const u32 c = powmod31(g->pow128, threadIdx.x);              // (16807^128)^j
        s->ci[threadIdx.x] = c; s->kd[threadIdx.x] = (double)c / 2147483647.0;
    }
    if (threadIdx.x == 0) { s->pow_lcg[31] = 0; s->pow128 = g->pow128; s->pow_blk = powmod31(g->pow128, SB_J); s->inv16807 = g->inv16807; s->pad = 0; }
    __syncthreads();
}

// srand(seed) for eight generators at once: every lane of group g passes generator g's seed; returns raw output word j = lane & 7
// of generator g = lane >> 3 (rand() = word >> 1).  Same seeding arithmetic as GlibcWave::seed.
__device__ __forceinline__ u32 srand8(const SmpTabs* __restrict__ T, volatile u32* __restrict__ Wr, u32 seed)
{
    const u32 lane = threadIdx.x & 63;
#pragma unroll 1
    for (int p = 0; p < 4; p++) {
        const u32 e = p * 64 + lane;                       // word e % 31 of generator e / 31
        const u32 ge = e / 31u, ie = e - ge * 31u;
        u32 sv = (u32)__shfl((int)seed, (int)((ge & 7u) * 8u));
        if (sv == 0) sv = 1;
        const int32_t w0 = (int32_t)sv;                    // r[1] by the signed Schrage step of random_r.c
        const long hi = w0 / 127773, lo = w0 % 127773;
        long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        const u32 val = ie == 0 ? sv : mulmod31(T->pow_lcg[ie], (u32)w);
        if (e < 31u * SB_TASKS) Wr[ge * 33u + ie] = val;
    }
    wave_fence();
    const u32 g = lane >> 3, j = lane & 7u;
    u32 acc = 0;
#pragma unroll 8
    for (int i = 0; i < 31; i++) acc += T->w8[i * SB_OUT + j] * Wr[g * 33u + i];
    wave_fence();
    return acc;
}

// One Bernoulli scan (engine = minstd_rand0(engine_seed), draw d tests map row first_row + d) that only RECORDS candidates:
// draws whose high digit x2 - 1 is below `amax` (the largest a_hi of the map) go to the wave's candidate list tagged `tag`.
// `count` (wave-uniform) is the list length; `flush()` must consume the list and reset count when it reaches 64.
//
// FP64 prefilter.  Lane l holds ONE exact engine state x (the high digit of draw l of the current block of 64 * SB_J draws); the
// high digit of its draw l + 64 j is x2 = x * c_j mod M, i.e. M * frac(T) with T = x * c_j / M < 2^31.  One FMA evaluates
//     s = x * fl(c_j / M) + (1.5 * 2^32 + 4 u),   u = 2^-20 = ulp(s),
// whose 20 low mantissa bits are  L = (frac(T) * 2^20 + e + 4) mod 2^20  with |e| < 1 (2^-22 from the rounded constant, 2^-21 from
// the rounding of the FMA, in units of u = 2^-20 less than one unit together).  A draw with x2 <= amax has frac(T) * 2^20 in
// (0, A], A = amax * 2^20 / M, hence L in [4, A + 5): it passes  L < ceil(A) + 6.  The test is a SUPERSET of the exact one
// (a few thousand x2 values wide: ~0.4 % extra candidates at 5e-4 per row); every candidate's exact x2 is then computed with
// the integer multiply and the exact threshold test runs in the resolve step, so the result is bit-identical.  Cost per 64 draws:
// one v_fma_f64, one v_and, one v_cmp (the exact form needs a 10-instruction modular multiply per 64 draws), plus one uniform
// 32-byte LDS read of the constants per 256 draws (broadcasting them from lanes with v_readlane measured 33 % slower).
// gev_dbg_prefilter_sweep checks the bound exhaustively over all 2^31 - 2 engine states x 32 multipliers.
template <class Flush>
__device__ __forceinline__ void scan_candidates(const SmpTabs* __restrict__ T, SmpWave* __restrict__ W, u32 engine_seed, u32 amax,
                                                u32 first_row, u32 n_draws, u32 tag, u32& count, Flush&& flush)
{
    const u32 lane = threadIdx.x & 63;
    if (amax == 0) return;                                 // every row has probability 0: no draw can hit
    const u32 s0 = minstd_seed(engine_seed);
    u32 x = mulmod31(T->pow_even[lane], s0);               // exact high digit (engine output 2 lane + 2) of draw `lane`
    const double A = (double)amax * (1048576.0 / 2147483647.0);
    const bool all_cand = A + 8.0 >= 1048576.0;            // thresholds near 1: every draw is a candidate
    const u32 thr20 = all_cand ? 0xffffffffu : (u32)A + 7u;   // >= ceil(A) + 6
    const double C = 6442450944.0 + 4.0 / 1048576.0;       // 1.5 * 2^32 + 4 u
    for (u32 base = 0; base < n_draws; base += 64 * SB_J) {
        const u32 n_here = min(64u * SB_J, n_draws - base);
        const u32 jfull = n_here >> 6, jn = (n_here + 63u) >> 6;   // full groups of 64 draws / groups incl. a partial last one
        const double xd = (double)x;
        auto testk = [&](double k) -> bool { return ((u32)__double2loint(__fma_rn(xd, k, C)) & 0xfffffu) < thr20; };
        auto push = [&](unsigned long long m, bool c, u32 j) {
            if (c) {
                const u32 idx = count + mbcnt64(m);
                W->cand_row[idx] = first_row + base + 64u * j + lane;
                W->cand_x2[idx] = mulmod31(x, T->ci[j]);  // the exact high digit of this draw
                W->cand_tag[idx] = (uint8_t)tag;
            }
            count += (u32)__popcll(m);
        };
        // four groups (256 draws) per step; the constants c_j / M of a step are one uniform 32-byte LDS read
        u32 j = 0;
        for (; j + 4 <= jfull; j += 4) {                   // full groups: no bounds to check
            const double4 kk = *reinterpret_cast<const double4*>(&T->kd[j]);
            const bool c0 = testk(kk.x), c1 = testk(kk.y), c2 = testk(kk.z), c3 = testk(kk.w);
            const unsigned long long m0 = __ballot(c0), m1 = __ballot(c1), m2 = __ballot(c2), m3 = __ballot(c3);
            if (m0 | m1 | m2 | m3) {
                push(m0, c0, j); push(m1, c1, j + 1); push(m2, c2, j + 2); push(m3, c3, j + 3);
                if (count >= 64) flush();
            }
        }
        for (; j < jn; j++) {                              // the last (up to three full + one partial) groups of the block
            const bool c0 = 64u * j + lane < n_here && testk(T->kd[j]);
            const unsigned long long m0 = __ballot(c0);
            if (m0) { push(m0, c0, j); if (count >= 64) flush(); }
        }
        if (base + 64 * SB_J < n_draws) x = mulmod31(x, T->pow_blk);
    }
}

// rank of a hit among the hits of its own task: hits so far (lane-held per group in `hc`) + earlier hit lanes with the same tag.
// Updates hc.  `valid`/`tg`/`hit` are per lane; all lanes must call.
__device__ __forceinline__ u32 rank_hits(bool valid, u32 tg, bool hit, u32& hc)
{
    const u32 lane = threadIdx.x & 63, g = lane >> 3;
    u32 h = 0;
#pragma unroll
    for (u32 q = 0; q < SB_TASKS; q++) {
        const unsigned long long mq = __ballot(valid && hit && tg == q);
        const u32 base = rl_u32(hc, q * 8u);
        if (tg == q) h = base + mbcnt64(mq);
        if (g == q) hc += (u32)__popcll(mq);
    }
    return h;
}

// ------------------------------------------------------------------------------------------
// K2 batched: Simulation::ras_add_mutation for 8 (offspring, chromosome) tasks per wave
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 4) k_mut_sample8(const GevRngTables* __restrict__ Tg, const ChrDev* __restrict__ chrs, int nchr,
                                                     const u32* __restrict__ mut_seeds, u32 seed_reproduce, const u32* __restrict__ seed_ptr, size_t n_tasks, SampleDev sd,
                                                     u32* __restrict__ slow_list)
{
    __shared__ SmpTabs s_T;
    __shared__ SmpWave s_W[4];
    stage_smp_tabs(Tg, &s_T);
    const SmpTabs* T = &s_T;
    SmpWave* W = &s_W[threadIdx.x >> 6];
    const u32 lane = threadIdx.x & 63, g = lane >> 3, j = lane & 7u;
    const size_t n_batches = (n_tasks + SB_TASKS - 1) / SB_TASKS;
    for (size_t b = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < n_batches; b += (size_t)gridDim.x * 4) {
        const size_t t = b * SB_TASKS + g;
        const bool valid = t < n_tasks;
        const u32 cidx = valid ? (u32)(t % (size_t)nchr) : 0u;
        const u32 S = valid ? mut_seeds[t] : 1u;
        if (b == 0) {                                       // first rand() after srand(seed) of reproduce (:2400, :2447) = seed_loc of task 0
            const u32 x0 = srand8(T, W->r, seed_ptr ? *seed_ptr : seed_reproduce);
            if (lane == 0) sd.seed_pat[0] = x0 >> 1;
        }
        const u32 xout = srand8(T, W->r, S);               // srand(seed), :2501
        u32 hc = 0; bool rej = false;
        u32 count = 0;
        auto resolve = [&]() {
            wave_fence();
            for (u32 c0 = 0; c0 < count; c0 += 64) {
                const u32 idx = c0 + lane; const bool cv = idx < count;
                const u32 row = cv ? W->cand_row[idx] : 1u, x2 = cv ? W->cand_x2[idx] : 1u, tg = cv ? (u32)W->cand_tag[idx] : 0u;
                const u32 cq = (u32)__shfl((int)cidx, (int)(tg * 8u));
                const u32 Sq = (u32)__shfl((int)S, (int)(tg * 8u));
                const ChrDev& C = chrs[cq];
                bool hit = false;
                if (cv) { const GevThr th = C.mthr[row]; hit = thr_hit(th, mulmod31(x2, T->inv16807), x2); }
                const u32 h = rank_hits(cv, tg, hit, hc);
                const u32 o = (u32)__shfl((int)xout, (int)(tg * 8u + (h < SB_OUT ? h : SB_OUT - 1u)));
                bool rj = false;
                if (hit && h < GEV_NM_CAP) {
                    // uniform_int_distribution<unsigned long>(bp[i-1], bp[i]) on generator(seed+1), :2503, :2516-2520: hit h of a task
                    // takes engine output h+1 as long as no earlier draw of the task was rejected
                    const u32 x = mulmod31(T->pow_lcg[h + 2], minstd_seed(Sq + 1u));
                    const u64 lo = C.mbp[row - 1], hi = C.mbp[row];
                    const u32 uerange = (u32)(hi - lo) + 1u;                   // < 2147483645 (checked by gev_set_mutmap)
                    const u32 scaling = 2147483645u / uerange, past = uerange * scaling, ret = x - 1u;
                    rj = ret >= past;
                    const size_t slot = (size_t)(b * SB_TASKS + tg) * GEV_NM_CAP + h;
                    sd.nm_pos[slot] = (u64)(ret / scaling) + lo;
                    sd.nm_side[slot] = (uint8_t)((o >> 1) & 1u);                // rand()%2, :2522
                }
#pragma unroll
                for (u32 q = 0; q < SB_TASKS; q++) { const unsigned long long mr = __ballot(rj && tg == q); if (g == q && mr) rej = true; }
            }
            count = 0;
            wave_fence();
        };
#pragma unroll 1
        for (u32 q = 0; q < SB_TASKS; q++) {
            if (b * SB_TASKS + q >= n_tasks) break;
            const ChrDev& C = chrs[rl_u32(cidx, q * 8u)];
            if (C.M >= 2) scan_candidates(T, W, rl_u32(S, q * 8u) + 2u, C.m_amax, 1u, C.M - 1u, q, count, resolve);   // generator_u(seed+2), i = 1..M-1
        }
        if (count) resolve();
        const u32 n = hc;
        const bool last = cidx == (u32)(nchr - 1);
        const bool slow = valid && (n + 2u > SB_OUT || rej);
        const u32 o_sex = (u32)__shfl((int)xout, (int)(g * 8u + (n < SB_OUT ? n : SB_OUT - 1u)));
        const u32 n2 = n + (last ? 1u : 0u);
        const u32 o_nxt = (u32)__shfl((int)xout, (int)(g * 8u + (n2 < SB_OUT ? n2 : SB_OUT - 1u)));
        if (valid && j == 0) {
            if (!slow) {
                sd.nmut[t] = n; sd.nm_off[t] = (u32)t * GEV_NM_CAP;
                if (last) sd.sex[t / (size_t)nchr] = (uint8_t)(((o_sex >> 1) & 1u) + 1u);      // :2472
                sd.seed_pat[t + 1] = o_nxt >> 1;                                                // seed_loc of the next task, :2447
            } else slow_list[atomicAdd(&sd.status[ST_SLOW_MUT], 1u)] = (u32)t;
        }
    }
}

// ------------------------------------------------------------------------------------------
// K1/K3 batched: both gametes of 8 (offspring, chromosome) tasks per wave
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 4) k_rec_sample8(const GevRngTables* __restrict__ Tg, const ChrDev* __restrict__ chrs, int nchr,
                                                     size_t n_tasks, SampleDev sd, u32* __restrict__ slow_list)
{
    __shared__ SmpTabs s_T;
    __shared__ SmpWave s_W[4];
    stage_smp_tabs(Tg, &s_T);
    const SmpTabs* T = &s_T;
    SmpWave* W = &s_W[threadIdx.x >> 6];
    const u32 lane = threadIdx.x & 63, g = lane >> 3, j = lane & 7u;
    const size_t n_batches = (n_tasks + SB_TASKS - 1) / SB_TASKS;
    for (size_t b = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < n_batches; b += (size_t)gridDim.x * 4) {
        const size_t t = b * SB_TASKS + g;
        const bool valid = t < n_tasks;
        const u32 cidx = valid ? (u32)(t % (size_t)nchr) : 0u;
        const bool active = valid && chrs[cidx].active != 0;       // inactive: another context owns this chromosome (only the seed chain is needed)
        const u32 seed_pat = valid ? sd.seed_pat[t] : 1u;
        u32 seed_cur = seed_pat, xout = 0, hc = 0, count = 0, side = 0;
        bool run = active;
        auto resolve = [&]() {
            wave_fence();
            for (u32 c0 = 0; c0 < count; c0 += 64) {
                const u32 idx = c0 + lane; const bool cv = idx < count;
                const u32 row = cv ? W->cand_row[idx] : 0u, x2 = cv ? W->cand_x2[idx] : 1u, tg = cv ? (u32)W->cand_tag[idx] : 0u;
                const u32 cq = (u32)__shfl((int)cidx, (int)(tg * 8u));
                const ChrDev& C = chrs[cq];
                bool hit = false;
                if (cv) { const GevThr th = C.rthr[row]; hit = thr_hit(th, mulmod31(x2, T->inv16807), x2); }
                const u32 h = rank_hits(cv, tg, hit, hc);
                const u32 o = (u32)__shfl((int)xout, (int)(tg * 8u + (h < SB_OUT ? h : SB_OUT - 1u)));
                if (hit && h < GEV_BK_CAP) {
                    const u32 rv = o >> 1;                                                        // rand(), :2990
                    const u64 dist = C.bp_dist;
                    const u64 v = C.rbp[row] + (dist > 0x7fffffffull ? (u64)rv : (u64)(rv % (u32)dist));
                    const size_t G = 2 * (b * SB_TASKS + tg) + side;
                    sd.bk[G * GEV_BK_CAP + h] = v;
                    sd.bk_idx[G * GEV_BK_CAP + h] = snp_lower_bound(C, v);     // first locus at or behind the breakpoint (what the dense stitch and the unit table work with)
                }
            }
            count = 0;
            wave_fence();
        };
        auto phase = [&]() {                               // ras_sim_loc_rec for the current gamete of every task that is still on the fast path
            xout = srand8(T, W->r, seed_cur);              // srand(seed), :2977
            hc = 0; count = 0;
#pragma unroll 1
            for (u32 q = 0; q < SB_TASKS; q++) {
                if (!rl_u32((u32)run, q * 8u)) continue;
                const ChrDev& C = chrs[rl_u32(cidx, q * 8u)];
                scan_candidates(T, W, rl_u32(seed_cur, q * 8u) + 1u, C.r_amax, 0u, C.R, q, count, resolve);   // generator(seed+1), :2978
            }
            if (count) resolve();
        };
        // paternal gamete (:2447-2449)
        side = 0; phase();
        const u32 k_pat = hc;
        bool slow = run && k_pat + 2u > SB_OUT;
        const u32 o_sp = (u32)__shfl((int)xout, (int)(g * 8u + (k_pat < SB_OUT ? k_pat : SB_OUT - 1u)));
        const u32 o_sm = (u32)__shfl((int)xout, (int)(g * 8u + (k_pat + 1u < SB_OUT ? k_pat + 1u : SB_OUT - 1u)));
        const u32 start_pat = (o_sp >> 1) & 1u;            // rand()%2 after k_pat position draws, :2449
        const u32 seed_mat = o_sm >> 1;                    // :2453
        // maternal gamete (:2453-2455)
        run = run && !slow; seed_cur = seed_mat;
        side = 1; phase();
        const u32 k_mat = hc;
        slow = slow || (run && k_mat + 1u > SB_OUT);
        const u32 o_st = (u32)__shfl((int)xout, (int)(g * 8u + (k_mat < SB_OUT ? k_mat : SB_OUT - 1u)));
        const u32 start_mat = (o_st >> 1) & 1u;            // :2455
        if (valid && j == 0) {
            if (!active) {
                sd.k[2 * t] = 0; sd.k[2 * t + 1] = 0; sd.bk_off[2 * t] = (u32)(2 * t) * GEV_BK_CAP; sd.bk_off[2 * t + 1] = (u32)(2 * t + 1) * GEV_BK_CAP;
                sd.start[2 * t] = 0; sd.start[2 * t + 1] = 0; sd.seed_mat[t] = 0;
            } else if (!slow) {
                sd.k[2 * t] = k_pat; sd.k[2 * t + 1] = k_mat; sd.bk_off[2 * t] = (u32)(2 * t) * GEV_BK_CAP; sd.bk_off[2 * t + 1] = (u32)(2 * t + 1) * GEV_BK_CAP;
                sd.start[2 * t] = (uint8_t)start_pat; sd.start[2 * t + 1] = (uint8_t)start_mat; sd.seed_mat[t] = seed_mat;
            } else slow_list[atomicAdd(&sd.status[ST_SLOW_REC], 1u)] = (u32)t;
        }
    }
}
