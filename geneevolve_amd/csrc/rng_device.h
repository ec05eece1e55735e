// rng_device.h -- bit-exact device restatements of the RNG arithmetic on the reference hot path,
// reshaped for 64-wide wavefronts (gfx950).  No FP anywhere: every Bernoulli draw is an integer
// compare against host-precomputed thresholds.
//
//   reference call site                                what runs here
//   -------------------------------------------------- -----------------------------------------
//   std::default_random_engine g(seed+1)               minstd jump-ahead: lane l of a wave owns
//   uniform_real_distribution(0,1)(g) < p              draws l, l+64, ... (x_n = 16807^n * s0)
//     src/Simulation.cpp:2978-2989, 2503-2513          and tests (a,b) against GevThr (below)
//   std::srand(s); std::rand() ...                     GlibcWave: the TYPE_3 additive generator is
//     src/Simulation.cpp:2400,2447-2455,2472,          linear over Z/2^32, so 64 consecutive outputs
//     2501,2522,2977,2990                              are 64 dot products with a constant matrix
//   uniform_int_distribution<ulong>(lo,hi)(g)          uniform_int_fallback (one lane, rejection)
//     src/Simulation.cpp:2519-2520
//
// Algorithms: libstdc++ 11 bits/random.h, bits/random.tcc, bits/uniform_int_dist.h; glibc 2.35
// stdlib/random_r.c.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GEV_M31 2147483647u

// Two-level integer threshold of one map row (16 B, one dwordx4 load per draw).
// With a = x2-1 (HIGH digit), b = x1-1 (LOW digit) of generate_canonical<double,53>:
//   a <  a_lo            -> r < p for every b
//   a >= a_hi            -> r >= p for every b
//   a_lo <= a < a_hi     -> r < p  <=>  b < bthr[a - a_lo]      (a_hi - a_lo <= 2, checked on host)
struct GevThr { uint32_t a_lo, a_hi, b0, b1; };

struct GevRngTables {
    uint32_t pow_even[64];    // 16807^(2l+2) mod M31 : lane l's first HIGH-digit engine output (index 2l+2)
    uint32_t pow128;          // 16807^128 mod M31    : 64 draws = 128 engine steps
    uint32_t pow512;          // 16807^512 mod M31    : 256 draws
    uint32_t inv16807;        // 16807^-1 mod M31     : low-digit output x1 = x2 * inv (only needed for candidates)
    uint32_t pow_lcg[31];     // 16807^(i-1) mod M31, i = 1..30 (glibc seeding), [0] unused
    uint32_t w_init[31 * 64]; // x_{344+k} = sum_i w_init[i*64+k] * r_i  (mod 2^32); lane k reads a coalesced row
    uint32_t w_next[31 * 64]; // x_{n+31+k} = sum_i w_next[i*64+k] * x_{n+i}
    uint32_t pad[2];          // sizeof % 16 == 0: the tables are staged into LDS with 16-byte copies
};
static_assert(sizeof(GevRngTables) % 16 == 0, "GevRngTables must be a multiple of 16 bytes");

// a*b mod (2^31-1) for a, b < 2^31, in 32-bit operations: p = hi*2^32 + lo, 2^32 = 2 and 2^31 = 1 (mod M)
__device__ __forceinline__ uint32_t mulmod31(uint32_t a, uint32_t b)
{
    const uint32_t lo = a * b, hi = __umulhi(a, b);                  // hi < 2^30
    uint32_t r = (hi << 1) + (lo >> 31) + (lo & GEV_M31);            // <= 2^32 - 2
    r = (r & GEV_M31) + (r >> 31);                                   // <= M + 1
    return r >= GEV_M31 ? r - GEV_M31 : r;
}

// The same product for CANONICAL operands (a, b in [0, M-1]): then p = a*b < M^2, r = floor(p / 2^31) + (p mod 2^31) <= 2M,
// and the folded value (r & M) + (r >> 31) could only leave [0, M-1] for r = M or r = 2M, i.e. p = 0 (mod M) with p != 0, which a
// prime modulus rules out: the conditional subtraction of mulmod31 is never taken.  Used by the scan loops, whose chains
// multiply canonical engine states by canonical constants.
__device__ __forceinline__ uint32_t mulmod31_canon(uint32_t a, uint32_t b)
{
    const uint32_t lo = a * b, hi = __umulhi(a, b);
    const uint32_t r = (hi << 1) + (lo >> 31) + (lo & GEV_M31);      // <= 2M
    return (r & GEV_M31) + (r >> 31);
}

__device__ __forceinline__ uint32_t powmod31(uint32_t a, uint64_t e)
{
    uint32_t r = 1;
    while (e) { if (e & 1) r = mulmod31(r, a); a = mulmod31(a, a); e >>= 1; }
    return r;
}

// std::minstd_rand0::seed(s): state = s mod (2^31-1), 0 -> 1.  `s` is the unsigned expression the
// reference constructs the engine from (seed+1 / seed+2, wrapped mod 2^32).
__device__ __forceinline__ uint32_t minstd_seed(uint32_t s)
{
    uint32_t x = s % GEV_M31;
    return x == 0 ? 1u : x;
}

__device__ __forceinline__ bool thr_hit(const GevThr t, uint32_t x1, uint32_t x2)
{
    const uint32_t a = x2 - 1, b = x1 - 1;
    if (a < t.a_lo) return true;
    if (a >= t.a_hi) return false;
    return b < (a == t.a_lo ? t.b0 : t.b1);
}

// Wave-cooperative Bernoulli scan: draw d (0-based) tests map row first_row+d and consumes engine
// outputs 2d+1 (low digit x1), 2d+2 (high digit x2).  Only the high digit is generated per draw
// (one modular multiply by 16807^128); a draw can hit only if a = x2-1 is below `amax`, the
// largest a_hi of the map, so the threshold row and the low digit (x1 = x2 * 16807^-1) are
// fetched for those rare candidates alone.  Calls on_hit(row) wave-uniformly for every hit, in
// row order.  Returns the number of hits.  All 64 lanes must call it.
// (s0 = the engine state in front of the first draw: minstd_seed(seed), or that state advanced by 2 d0 steps for draw d0)
template <class F>
__device__ __forceinline__ uint32_t wave_scan_hits_state(const GevRngTables* __restrict__ T, uint32_t s0,
                                                         const GevThr* __restrict__ thr, uint32_t amax, uint32_t first_row,
                                                         uint32_t n_draws, F on_hit)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t p128 = T->pow128, inv = T->inv16807;
    // four independent LCG chains per lane (draws d, d+64, d+128, d+192; step 16807^512) hide the
    // latency of the dependent modular multiply
    uint32_t xa = mulmod31(T->pow_even[lane], s0);
    uint32_t xb = mulmod31(xa, p128), xc = mulmod31(xb, p128), xd = mulmod31(xc, p128);
    const uint32_t p512 = T->pow512;
    uint32_t n_hits = 0;
    auto test = [&](uint32_t x2, uint32_t d) -> bool {
        if (d < n_draws && x2 - 1 < amax) {
            const uint32_t x1 = mulmod31(x2, inv);
            const GevThr t = thr[first_row + d];
            return thr_hit(t, x1, x2);
        }
        return false;
    };
    auto drain = [&](unsigned long long m, uint32_t base) {
        n_hits += __popcll(m);
        while (m) {
            const uint32_t l = __ffsll((long long)m) - 1;
            on_hit(first_row + base + l);
            m &= m - 1;
        }
    };
    for (uint32_t base = 0; base < n_draws; base += 256) {
        const unsigned long long ma = __ballot(test(xa, base + lane));
        const unsigned long long mb = __ballot(test(xb, base + 64 + lane));
        const unsigned long long mc = __ballot(test(xc, base + 128 + lane));
        const unsigned long long md = __ballot(test(xd, base + 192 + lane));
        if (ma | mb | mc | md) { drain(ma, base); drain(mb, base + 64); drain(mc, base + 128); drain(md, base + 192); }
        xa = mulmod31(xa, p512); xb = mulmod31(xb, p512); xc = mulmod31(xc, p512); xd = mulmod31(xd, p512);
    }
    return n_hits;
}

template <class F>
__device__ __forceinline__ uint32_t wave_scan_hits(const GevRngTables* __restrict__ T, uint32_t engine_seed,
                                                   const GevThr* __restrict__ thr, uint32_t amax, uint32_t first_row,
                                                   uint32_t n_draws, F on_hit)
{
    return wave_scan_hits_state(T, minstd_seed(engine_seed), thr, amax, first_row, n_draws, on_hit);
}

// wave-uniform lane index -> v_readlane_b32 (SGPR broadcast) instead of the LDS crossbar (ds_bpermute) of __shfl
__device__ __forceinline__ uint32_t rl_u32(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
__device__ __forceinline__ double rl_f64(double v, uint32_t lane)
{
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = rl_u32((uint32_t)b, lane), hi = rl_u32((uint32_t)(b >> 32), lane);
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// glibc srand()/rand(): lane k of the wave holds raw word x of output (64*block + k); rand() = x >> 1.
struct GlibcWave {
    uint32_t x;        // this lane's raw output word of the current block
    uint32_t block;    // index of the current block of 64 outputs

    // srand(seed)
    __device__ __forceinline__ void seed(const GevRngTables* __restrict__ T, uint32_t seed)
    {
        const uint32_t lane = threadIdx.x & 63;
        if (seed == 0) seed = 1;
        // r[1] by the exact signed Schrage step of random_r.c (seeds >= 2^31 enter as negative int32)
        const int32_t w0 = (int32_t)seed;
        const long hi = w0 / 127773, lo = w0 % 127773;
        long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        const uint32_t r1 = (uint32_t)w;
        uint32_t r = 0;
        if (lane == 0) r = seed;
        else if (lane < 31) r = mulmod31(T->pow_lcg[lane], r1);      // 16807^(lane-1) * r1
        uint32_t acc = 0;
        const uint32_t* wcol = T->w_init + lane;
#pragma unroll 4
        for (int i = 0; i < 31; i++) acc += wcol[i * 64] * rl_u32(r, i);
        x = acc; block = 0;
    }
    __device__ __forceinline__ void next_block(const GevRngTables* __restrict__ T)
    {
        const uint32_t lane = threadIdx.x & 63;
        uint32_t acc = 0;
        const uint32_t* wcol = T->w_next + lane;
#pragma unroll 4
        for (int i = 0; i < 31; i++) acc += wcol[i * 64] * rl_u32(x, 33 + i);
        x = acc; block++;
    }
    // n-th output (0-based) of the stream; n must not decrease between calls.  Wave-uniform.
    __device__ __forceinline__ uint32_t out(const GevRngTables* __restrict__ T, uint32_t n)
    {
        while ((n >> 6) > block) next_block(T);
        return rl_u32(x, n & 63) >> 1;
    }
};

// std::uniform_int_distribution<unsigned long>(lo,hi) on minstd_rand0, down-scaling "fallback case
// (2 divisions)" branch; requires hi-lo < 2147483645 (checked on the host).  `x` = engine state.
__device__ __forceinline__ uint64_t uniform_int_fallback(uint32_t& x, uint64_t lo, uint64_t hi)
{
    const uint64_t uerange = hi - lo + 1;
    const uint64_t scaling = 2147483645ull / uerange;
    const uint64_t past = uerange * scaling;
    uint64_t ret;
    do { x = mulmod31(x, 16807u); ret = (uint64_t)x - 1; } while (ret >= past);
    return ret / scaling + lo;
}

// ------------------------------------------------------------------------------------------
// host side: constant tables and exact thresholds
// ------------------------------------------------------------------------------------------
#include <cmath>
#include <vector>

static inline uint32_t h_mulmod31(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) % GEV_M31); }
static inline uint32_t h_powmod31(uint32_t a, uint64_t e)
{
    uint32_t r = 1;
    while (e) { if (e & 1) r = h_mulmod31(r, a); a = h_mulmod31(a, a); e >>= 1; }
    return r;
}

static inline void gev_build_rng_tables(GevRngTables& T)
{
    for (int l = 0; l < 64; l++) T.pow_even[l] = h_powmod31(16807u, 2 * l + 2);
    T.pow128 = h_powmod31(16807u, 128);
    T.pow512 = h_powmod31(16807u, 512);
    T.inv16807 = h_powmod31(16807u, (uint64_t)GEV_M31 - 2);
    T.pow_lcg[0] = 0;
    for (int i = 1; i < 31; i++) T.pow_lcg[i] = h_powmod31(16807u, i - 1);
    // z_m = z_{m-31} + z_{m-3} (m >= 34), z_{31..33} = z_{0..2}; output k = z_{344+k}.
    // Coefficients by running the recurrence on unit vectors.
    for (int i = 0; i < 31; i++) {
        std::vector<uint32_t> z(344 + 64, 0);
        z[i] = 1;
        for (int m = 31; m < 34; m++) z[m] = z[m - 31];
        for (int m = 34; m < 344 + 64; m++) z[m] = z[m - 31] + z[m - 3];
        for (int k = 0; k < 64; k++) T.w_init[i * 64 + k] = z[344 + k];
        // generic continuation: given 31 consecutive words y_0..y_30 (y_j = z_{n+j}), next 64 words
        std::vector<uint32_t> y(31 + 64, 0);
        y[i] = 1;
        for (int m = 31; m < 31 + 64; m++) y[m] = y[m - 31] + y[m - 3];
        for (int k = 0; k < 64; k++) T.w_next[i * 64 + k] = y[31 + k];
    }
}

// generate_canonical<double,53,minstd_rand0> from digits (b = x1-1 low, a = x2-1 high);
// must be compiled without FMA contraction (hipcc host pass: -ffp-contract=off).
static inline double gev_canonical(uint32_t a, uint32_t b)
{
    const double R = 2147483646.0;
    // (double)(2147483646.0L * 2147483646.0L): the exact product 4611686009837453316 rounded to double
    const double R2 = 0x1.fffffff000000p+61;
    volatile double t = (double)a * R;
    volatile double sum = (double)b + t;
    double ret = sum / R2;
    if (ret >= 1.0) ret = std::nextafter(1.0, 0.0);
    return ret;
}

// exact two-level threshold of probability p.  Returns false if the window is wider than 2.
static inline bool gev_make_threshold(double p, GevThr& out)
{
    const uint32_t NMAX = 2147483646u;                 // digits are in [0, NMAX)
    auto hit = [&](uint32_t a, uint32_t b) { return gev_canonical(a, b) < p; };
    // number of b with hit(a,b): hit is non-increasing in b
    auto bthr = [&](uint32_t a) -> uint32_t {
        uint32_t lo = 0, hi = NMAX;                   // first b in [lo,hi] with !hit
        while (lo < hi) { uint32_t mid = lo + (hi - lo) / 2; if (hit(a, mid)) lo = mid + 1; else hi = mid; }
        return lo;
    };
    // a_lo = first a whose largest b misses (not all-hit); a_hi = first a whose b=0 misses (all-miss)
    uint32_t lo = 0, hi = NMAX;
    while (lo < hi) { uint32_t mid = lo + (hi - lo) / 2; if (hit(mid, NMAX - 1)) lo = mid + 1; else hi = mid; }
    const uint32_t a_lo = lo;
    lo = a_lo; hi = NMAX;
    while (lo < hi) { uint32_t mid = lo + (hi - lo) / 2; if (hit(mid, 0)) lo = mid + 1; else hi = mid; }
    const uint32_t a_hi = lo;
    if (a_hi - a_lo > 2) return false;
    out.a_lo = a_lo; out.a_hi = a_hi;
    out.b0 = (a_hi > a_lo) ? bthr(a_lo) : 0;
    out.b1 = (a_hi > a_lo + 1) ? bthr(a_lo + 1) : 0;
    return true;
}
