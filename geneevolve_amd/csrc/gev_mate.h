// gev_mate.h -- the step in front of the reproduction path on the device (gfx950, wave64).  Included by gev_kernels.h.
//
//   Simulation::ras_glob_seed   (reference src/Simulation.cpp:17-21)     k_glob_count / k_glob_emit
//   Simulation::random_mate     (reference src/Simulation.cpp:2090-2157) k_mate_flags / k_mate_compact / k_pick_count / k_pick_emit
//
// Both are streams of libstdc++ uniform_int_distribution draws on minstd_rand0 (bits/uniform_int_dist.h, down-scaling branch:
// ret = engine() - 1 is rejected while ret >= past = uerange * scaling, scaling = 2147483645 / uerange; value = ret / scaling + lo).
// Engine output number j is 16807^j * state (mod 2^31 - 1), so every candidate is evaluated independently by jump-ahead; what makes
// the stream sequential is only that the i-th VALUE is the i-th ACCEPTED candidate.  Rejections are rare (2.3e-4 per draw for
// ras_glob_seed, < n / 2^31 for the mate picks), so the rank of a candidate is its index minus the rejections in front of it:
// pass 1 counts the rejections of every block of REJ_CHUNK candidates, pass 2 sums the counts of the blocks in front (a strided
// read of at most n / REJ_CHUNK words), ranks its own candidates with a block scan and stores the values.
#pragma once

#define REJ_PER_THREAD 16
#define REJ_CHUNK (256 * REJ_PER_THREAD)      // candidates per block
enum { MATE_NM = 0, MATE_NF = 1, MATE_WORDS = 4 };

// engine outputs number j0+1 .. j0+N of the engine whose state is x0 (candidates j0 .. j0+N-1); returns the mask of the candidates
// below n_cand that uniform_int_distribution rejects
template <int N>
__device__ __forceinline__ u32 minstd_run(u32 x0, u64 j0, u64 n_cand, u32 past, u32 (&xs)[N])
{
    u32 x = mulmod31(powmod31(16807u, j0 + 1), x0), rej = 0;
#pragma unroll
    for (int q = 0; q < N; q++) {
        if (q) x = mulmod31(x, 16807u);
        xs[q] = x;
        if (j0 + q < n_cand && x - 1u >= past) rej |= 1u << q;
    }
    return rej;
}
__device__ __forceinline__ u32 block_sum_256(u32 v, u32* lds /*>=8*/)
{
    u32 tot; block_exclusive_scan_256(v, lds, tot);
    return tot;
}
// pass 1: rejections among this block's candidates
__device__ __forceinline__ void rej_count_block(u32 x0, u32 past, u64 n_cand, u32* __restrict__ blk, u32* lds)
{
    const u64 j0 = (u64)blockIdx.x * REJ_CHUNK + (u64)threadIdx.x * REJ_PER_THREAD;
    u32 rej = 0;
    if (j0 < n_cand) { u32 xs[REJ_PER_THREAD]; rej = minstd_run<REJ_PER_THREAD>(x0, j0, n_cand, past, xs); }
    const u32 tot = block_sum_256((u32)__popc(rej), lds);
    if (threadIdx.x == 0) blk[blockIdx.x] = tot;
}
// pass 2: emit(i, ret, x) for the i-th accepted candidate (i < n), ret = x - 1; *short_flag |= flag_bit when the n_cand candidates
// hold fewer than n accepted ones (cannot happen with the candidate counts the host computes; reported, never silently wrong)
template <class Emit>
__device__ __forceinline__ void rej_emit_block(u32 x0, u32 past, u64 n, u64 n_cand, const u32* __restrict__ blk, u32* lds, u32* short_flag, u32 flag_bit, Emit&& emit)
{
    u32 part = 0;
    for (u32 b = threadIdx.x; b < blockIdx.x; b += 256) part += blk[b];
    const u32 before_blocks = block_sum_256(part, lds);
    const u64 j0 = (u64)blockIdx.x * REJ_CHUNK + (u64)threadIdx.x * REJ_PER_THREAD;
    u32 xs[REJ_PER_THREAD]; u32 rej = 0;
    if (j0 < n_cand) rej = minstd_run<REJ_PER_THREAD>(x0, j0, n_cand, past, xs);
    u32 tot;
    const u32 before = before_blocks + block_exclusive_scan_256((u32)__popc(rej), lds, tot);
    if (j0 >= n_cand) return;
#pragma unroll
    for (int q = 0; q < REJ_PER_THREAD; q++) {
        const u64 j = j0 + q;
        if (j >= n_cand) break;
        const u32 r_before = before + (u32)__popc(rej & ((1u << q) - 1u));
        const bool ok = !((rej >> q) & 1u);
        const u64 i = j - r_before;
        if (ok && i < n) emit(i, xs[q] - 1u, xs[q]);
        if (j == n_cand - 1 && (j + 1 - r_before - (ok ? 0u : 1u)) < n) atomicOr(short_flag, flag_bit);
    }
}

// ---- Simulation::ras_glob_seed: n values of uniform_int_distribution<unsigned>(1, 1000000) on glob_generator -----------------
#define GLOB_SCALING 2147u                    // 2147483645 / 1000000
#define GLOB_PAST 2147000000u                 // 1000000 * 2147
__global__ void __launch_bounds__(256) k_glob_count(u32 state_val, const u32* __restrict__ state_ptr, u64 n_cand, u32* __restrict__ blk)
{
    __shared__ u32 lds[8];
    rej_count_block(state_ptr ? *state_ptr : state_val, GLOB_PAST, n_cand, blk, lds);
}
// vals[i] = the i-th ras_glob_seed() value; *state_out = the engine state behind the n-th value (what the host's glob_generator
// holds after the same n calls)
__global__ void __launch_bounds__(256) k_glob_emit(u32 state_val, const u32* __restrict__ state_ptr, u64 n, u64 n_cand, const u32* __restrict__ blk,
                                                   u32* __restrict__ vals, u32* __restrict__ state_out, u32* __restrict__ flags)
{
    __shared__ u32 lds[8];
    rej_emit_block(state_ptr ? *state_ptr : state_val, GLOB_PAST, n, n_cand, blk, lds, flags, (u32)FLAG_RNG_SHORT,
                   [&](u64 i, u32 ret, u32 x) { vals[i] = ret / GLOB_SCALING + 1u; if (i == n - 1) *state_out = x; });
}

// glob_generator after n more ras_glob_seed() calls (a handful: the draws the host makes itself between two generations)
__global__ void __launch_bounds__(64) k_glob_skip(const u32* __restrict__ state_in, u32 n, u32* __restrict__ state_out)
{
    if (threadIdx.x || blockIdx.x) return;
    u32 x = *state_in;
    for (u32 i = 0; i < n; i++) { do x = mulmod31(x, 16807u); while (x - 1u >= GLOB_PAST); }
    *state_out = x;
}

// ---- Simulation::random_mate (src/Simulation.cpp:2090-2157) --------------------------------------------------------------------
// :2109-2118  one uniform_real draw per individual from generator(seed) (drawn whether or not it matters), marriageable when
// r < selection_value_func; males and females keep their order.  svf == null: every selection_value_func is 1 (selection function
// "none" / generation 0, :3388) and r < 1 always holds (generate_canonical clamps below 1), so the draws cannot change anything and
// are skipped.  mflag[i] = 1 male / 2 female / 0 not marriageable; blk[b] = males | females << 16 of block b (1024 individuals).
#define MATE_PER_THREAD 4
#define MATE_CHUNK (256 * MATE_PER_THREAD)
__global__ void __launch_bounds__(256) k_mate_flags(const uint8_t* __restrict__ sex, const u32* __restrict__ logical, const double* __restrict__ svf, size_t n_h,
                                                    u32 seed_val, const u32* __restrict__ seed_ptr, uint8_t* __restrict__ mflag, u32* __restrict__ blk)
{
    __shared__ u32 lds[8];
    const size_t i0 = (size_t)blockIdx.x * MATE_CHUNK + (size_t)threadIdx.x * MATE_PER_THREAD;
    u32 cnt = 0;
    if (i0 < n_h) {
        u32 x = 0;
        if (svf) x = mulmod31(powmod31(16807u, 2 * (u64)i0), minstd_seed(seed_ptr ? *seed_ptr : seed_val));   // engine state in front of individual i0's draw
#pragma unroll
        for (int q = 0; q < MATE_PER_THREAD; q++) {
            const size_t i = i0 + q;
            if (i >= n_h) break;
            bool ok = true;
            if (svf) {
                const u32 x1 = mulmod31(x, 16807u), x2 = mulmod31(x1, 16807u);        // generate_canonical: two engine calls, low digit first
                x = x2;
                ok = canonical_f64(x1, x2) < svf[i];
            }
            const u32 s = sex[logical ? logical[i] : i];
            const u32 f = ok ? (s == 1u ? 1u : (s == 2u ? 2u : 0u)) : 0u;
            mflag[i] = (uint8_t)f;
            cnt += (f == 1u ? 1u : 0u) + (f == 2u ? 0x10000u : 0u);
        }
    }
    const u32 tot = block_sum_256(cnt, lds);
    if (threadIdx.x == 0) blk[blockIdx.x] = tot;
}
// pos_male / pos_female (:2107-2108) in ascending order; meta = {num_males_mate, num_females_mate}; "No one can marry" (:2125) is a flag
__global__ void __launch_bounds__(256) k_mate_compact(const uint8_t* __restrict__ mflag, size_t n_h, const u32* __restrict__ blk, u32 n_blk,
                                                      u32* __restrict__ pos_m, u32* __restrict__ pos_f, u32* __restrict__ meta, u32* __restrict__ flags)
{
    __shared__ u32 lds[8];
    u32 pm = 0, pf = 0;
    for (u32 b = threadIdx.x; b < blockIdx.x; b += 256) { const u32 v = blk[b]; pm += v & 0xffffu; pf += v >> 16; }
    const u32 base_m = block_sum_256(pm, lds), base_f = block_sum_256(pf, lds);
    const size_t i0 = (size_t)blockIdx.x * MATE_CHUNK + (size_t)threadIdx.x * MATE_PER_THREAD;
    u32 f[MATE_PER_THREAD], cnt = 0;
#pragma unroll
    for (int q = 0; q < MATE_PER_THREAD; q++) { f[q] = i0 + q < n_h ? (u32)mflag[i0 + q] : 0u; cnt += (f[q] == 1u ? 1u : 0u) + (f[q] == 2u ? 0x10000u : 0u); }
    u32 tot;
    const u32 ex = block_exclusive_scan_256(cnt, lds, tot);
    u32 am = base_m + (ex & 0xffffu), af = base_f + (ex >> 16);
#pragma unroll
    for (int q = 0; q < MATE_PER_THREAD; q++) {
        if (f[q] == 1u) pos_m[am++] = (u32)(i0 + q);
        else if (f[q] == 2u) pos_f[af++] = (u32)(i0 + q);
    }
    if (blockIdx.x == n_blk - 1 && threadIdx.x == 0) {
        const u32 nm = base_m + (tot & 0xffffu), nf = base_f + (tot >> 16);
        meta[MATE_NM] = nm; meta[MATE_NF] = nf;
        if (nm == 0 || nf == 0) atomicOr(flags, (u32)FLAG_NO_MATES);
    }
}
// :2132-2147  i_f from g_uint_f(seed+1) on (0, num_males_mate-1), i_m from g_uint_m(seed+2) on (0, num_females_mate-1); blockIdx.y = 0 fathers, 1 mothers
struct PickStream { u32 x0, past, scaling; };
__device__ __forceinline__ PickStream pick_stream(u32 seed_val, const u32* __restrict__ seed_ptr, const u32* __restrict__ meta)
{
    const u32 y = blockIdx.y;
    const u32 uer = max(meta[y ? MATE_NF : MATE_NM], 1u);          // (0 is reported by FLAG_NO_MATES; keeps the arithmetic defined)
    PickStream s;
    s.x0 = minstd_seed((seed_ptr ? *seed_ptr : seed_val) + 1u + y);
    s.scaling = 2147483645u / uer; s.past = uer * s.scaling;
    return s;
}
__global__ void __launch_bounds__(256) k_pick_count(u32 seed_val, const u32* __restrict__ seed_ptr, const u32* __restrict__ meta, u64 n_cand, u32* __restrict__ blk)
{
    __shared__ u32 lds[8];
    const PickStream s = pick_stream(seed_val, seed_ptr, meta);
    rej_count_block(s.x0, s.past, n_cand, blk + (size_t)blockIdx.y * gridDim.x, lds);
}
// couple i = (pos_male[i_f], pos_female[i_m]), one child, never inbred (:2146-2149).  father / mother = the parents' PHYSICAL row
// indices for the kernels of gev_reproduce (`logical` maps positions to rows after a cross-GPU migration; null = identity);
// couples (optional) = the reference's Couples_Info records, positions as the host knows them.
__global__ void __launch_bounds__(256) k_pick_emit(u32 seed_val, const u32* __restrict__ seed_ptr, const u32* __restrict__ meta, u64 n, u64 n_cand, const u32* __restrict__ blk,
                                                   const u32* __restrict__ pos_m, const u32* __restrict__ pos_f, const u32* __restrict__ logical,
                                                   u32* __restrict__ father, u32* __restrict__ mother, gev_couple* __restrict__ couples, u32* __restrict__ flags)
{
    __shared__ u32 lds[8];
    const PickStream s = pick_stream(seed_val, seed_ptr, meta);
    const u32 y = blockIdx.y;
    const u32 n_list = meta[y ? MATE_NF : MATE_NM];
    const u32* __restrict__ list = y ? pos_f : pos_m;
    u32* __restrict__ parent = y ? mother : father;
    rej_emit_block(s.x0, s.past, n, n_cand, blk + (size_t)y * gridDim.x, lds, flags, (u32)FLAG_RNG_SHORT, [&](u64 i, u32 ret, u32) {
        const u32 pick = ret / s.scaling;
        const u32 pos = pick < n_list ? list[pick] : 0u;
        parent[i] = logical ? logical[pos] : pos;
        if (couples) {
            if (y) couples[i].pos_female = pos;
            else { couples[i].pos_male = pos; couples[i].inbreed = 0; couples[i].num_offspring = 1; }
        }
    });
}
