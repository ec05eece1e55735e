// gev_lists.h -- the sparse per-haplotype state (ancestry interval lists, mutation lists) as SHARED PIECES.
//
// Simulation::recombine (src/Simulation.cpp:2903-2958) copies the parent's parts unchanged between two crossovers, and
// ras_add_mutation (:2497-2552) appends to one part: per generation a haplotype's lists change around its ~1 crossover and its
// ~0.5 new mutations and nowhere else.  Rewriting the whole lists every generation (the CSR form of rounds 1-2) costs bytes
// proportional to the AGE of the run (lists lengthen by about one entry per generation).  Here the chromosome is cut into
// nseg <= LP_MAXSEG position ranges of 2^lgw bp, and the list of a haplotype row is the concatenation of one PIECE per range:
//
//   interval piece  (row, t) -> {offset, n}: n+1 LpPart entries in the arena: entry 0 = the part that covers the range's first
//                   position (carry-in, owned by an earlier range), entries 1..n = the parts whose st lies in the range.
//                   The parts of a haplotype tile [bp0, bp_end) (st ascending, en[i] = st[i+1]: recombine keeps that), so en is
//                   not stored: it is the next st of the row (bp_end for the last part).
//   mutation piece  (row, t) -> {offset, n}: the n positions of the row's (ascending) mutation list that lie in the range.
//
// A range that contains no crossover boundary of the gamete (and, for the mutation piece, no new mutation) NAMES the parent's
// piece -- one 8-byte table entry, nothing copied; only the ranges with an event get a new piece, appended to the arena with one
// atomic per workgroup.  Pieces are immutable.  Everything else that wants whole lists (downloads, output, migration, the
// plane-less genotype assembly) reads the CSR form, which is materialised from the pieces on demand (k_lp_count / k_lp_fill)
// and turned back into pieces by k_lp_import_*; when an arena fills up, that round trip is also its compaction.
// (included twice by gev_kernels.h: the types in front of ChrWork, the kernels -- GEV_LISTS_KERNELS -- behind the scan helpers)
#ifndef GEV_LISTS_TYPES
#define GEV_LISTS_TYPES

#define LP_MAXSEG 32
struct __attribute__((aligned(16))) LpPart { u64 st; u32 hap_index; u32 root_population; };
struct LpWork {
    const uint2* ptab_cur; uint2* ptab_alt;      // [rows][nseg] interval pieces of the parents / of the offspring
    const uint2* mtab_cur; uint2* mtab_alt;      // [rows][nseg] mutation pieces
    LpPart* parena; u64* marena;
    u32 pbase, mbase, pcap, mcap;                // arena cursors at the start of the attempt, capacities (entries)
    u32 nseg, lgw;
    u32 track;                                   // interval pieces are kept (gev_set_track_intervals)
    u32* items; u32* n_items;                    // work list of the generation: the (row, range) pairs with an event
};
#define LP_INF (~0ull)
__device__ __forceinline__ u32 lp_seg(u64 x, u64 bp0, u32 lgw, u32 nseg)
{
    if (x <= bp0) return 0u;
    const u64 t = (x - bp0) >> lgw;
    return t >= nseg ? nseg - 1u : (u32)t;
}
// first position of range t (0 for the first range: positions in front of bp0 count to it), first position behind it (none: last)
__device__ __forceinline__ u64 lp_lo(u32 t, u64 bp0, u32 lgw) { return t == 0 ? 0ull : bp0 + ((u64)t << lgw); }
__device__ __forceinline__ u64 lp_hi(u32 t, u64 bp0, u32 lgw, u32 nseg) { return t + 1 >= nseg ? LP_INF : bp0 + ((u64)(t + 1) << lgw); }

// is position x in the mutation list of haplotype row `row` (parents' table)?  (k_cv_newmut: the set semantics of a new mutation)
__device__ __forceinline__ bool lp_has_mutation(const LpWork& lp, size_t row, u64 x, u64 bp0)
{
    const uint2 me = lp.mtab_cur[row * lp.nseg + lp_seg(x, bp0, lp.lgw, lp.nseg)];
    const u64* a = lp.marena + me.x;
    u32 l = 0, r = me.y;
    while (l < r) { const u32 m = (l + r) >> 1; if (a[m] < x) l = m + 1; else r = m; }
    return l < me.y && a[l] == x;
}

#endif   // GEV_LISTS_TYPES
#ifdef GEV_LISTS_KERNELS
// One interval [Lc, Rc) of the crossover pattern on ONE piece of the source haplotype: Simulation::recombine's statements
// (:2918-2950) with en derived from the next entry.  e[0..n] = carry-in + owned entries, en_last = what the last entry's en is
// known to be (bp_end in the last range, otherwise "behind this range").  head: Lc lies in this range (the interval starts
// here); otherwise the interval began in an earlier range and only the whole-part run / the clip at Rc concern this piece.
template <bool FILL>
__device__ __forceinline__ u32 lp_interval(const LpPart* __restrict__ e, u32 n, u64 en_last, bool head, u64 Lc, u64 Rc, LpPart* __restrict__ out, u32 at)
{
#define LP_EN(i) ((i) < n ? e[(i) + 1].st : en_last)
    u32 i2 = head ? 0u : 1u, emitted = 0;
    if (head) {
        while (i2 <= n && LP_EN(i2) <= Lc) i2++;                                                     // :2918
        if (i2 <= n && e[i2].st < Lc && Lc < LP_EN(i2) && Rc < LP_EN(i2)) {                          // :2922
            if (FILL) { LpPart p = e[i2]; p.st = Lc; out[at + emitted] = p; } emitted++; i2++;
        }
        if (i2 <= n && e[i2].st < Lc && Lc < LP_EN(i2) && Rc >= LP_EN(i2)) {                         // :2931
            if (FILL) { LpPart p = e[i2]; p.st = Lc; out[at + emitted] = p; } emitted++; i2++;
        }
        if (i2 == 0) i2 = 1;                                                                         // the carry-in is owned (and was emitted) by an earlier range
    }
    while (i2 <= n && LP_EN(i2) <= Rc && Lc <= e[i2].st) { if (FILL) out[at + emitted] = e[i2]; emitted++; i2++; }   // :2939
    if (i2 <= n && i2 >= 1 && e[i2].st < Rc && Rc < LP_EN(i2)) { if (FILL) out[at + emitted] = e[i2]; emitted++; }    // :2947 (en = Rc is implicit)
#undef LP_EN
    return emitted;
}

// ------------------------------------------------------------------------------------------
// one generation: thread (offspring row, range t).  Ranges without an event copy two table entries; the others build their
// pieces in two passes over the (short) source pieces -- count, one atomic per workgroup and arena, fill.
// ------------------------------------------------------------------------------------------
// The same interval in closed form.  With c(x) = number of owned entries whose st is <= x (a bisection), recombine's statements
// come to: in an interval that starts here, the entry e[c(Lc)] covers Lc -- it is emitted clipped (st = Lc) if it starts in front
// of Lc, as it is otherwise (zero-length parts [Lc, Lc) in front of it are skipped by :2918); then every owned entry up to
// e[c(Rc)] follows, the last one only if it starts in front of Rc (a part that starts AT Rc belongs to the next interval; zero-
// length parts [Rc, Rc) in front of it do not: their en <= Rc, :2939).  An interval that starts at or behind the end of the map
// (Lc >= bp_end, last range) emits nothing.  lp_interval is the literal form (GEV_LP_LITERAL=1 builds with it: a cross-check).
__device__ __forceinline__ u32 lp_count_le(const LpPart* __restrict__ e, u32 n, u64 x)
{
    u32 lo = 0, hi = n;                                            // over e[1..n]
    while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (e[mid + 1].st <= x) lo = mid + 1; else hi = mid; }
    return lo;
}
template <bool FILL>
__device__ __forceinline__ u32 lp_interval_fast(const LpPart* __restrict__ e, u32 n, u64 en_last, bool head, u64 Lc, u64 Rc, LpPart* __restrict__ out, u32 at)
{
    u32 emitted = 0, first = 1;
    if (head) {
        if (Lc >= en_last) return 0u;                              // (only the last range knows a finite en_last)
        const u32 c0 = lp_count_le(e, n, Lc);
        if (e[c0].st < Lc) { if (FILL) { LpPart p = e[c0]; p.st = Lc; out[at] = p; } emitted = 1; first = c0 + 1; }
        else first = c0;
    }
    u32 last = Rc == LP_INF ? n : lp_count_le(e, n, Rc);
    if (last >= 1 && e[last].st == Rc) last--;
    if (first >= 1 && first <= last) {
        if (FILL) for (u32 j = first; j <= last; j++) out[at + emitted + (j - first)] = e[j];
        emitted += last - first + 1;
    }
    return emitted;
}
template <bool FILL, bool LITERAL>
__device__ __forceinline__ u32 lp_build_parts(const LpWork& lp, const u32 parent, const u32 start, const u64* __restrict__ bk, const u32 j0, const u32 j1,
                                              const u32 t, const u64 bp0, const u64 bp_end, LpPart* __restrict__ out)
{
    const bool last = t + 1 >= lp.nseg;
    const u64 en_last = last ? bp_end : LP_INF;
    u32 n = 0;
    for (u32 q = j0; q <= j1; q++) {                               // interval q: behind breakpoint q-1, in front of breakpoint q
        const u32 h = (start ^ q) & 1u;
        const uint2 pe = lp.ptab_cur[((size_t)2 * parent + h) * lp.nseg + t];
        const LpPart* e = lp.parena + pe.x;
        const bool head = q > j0 || t == 0;                        // the interval starts in this range (the very first one starts at bp0)
        const u64 Lc = q > j0 ? bk[q - 1] : bp0;
        const u64 Rc = q < j1 ? bk[q] : (last ? bp_end : LP_INF);
        if (FILL && q == j0) out[0] = e[0];                        // carry-in: the part that covers the range's first position
        n += LITERAL ? lp_interval<FILL>(e, pe.y, en_last, head, Lc, Rc, out, 1 + n) : lp_interval_fast<FILL>(e, pe.y, en_last, head, Lc, Rc, out, 1 + n);
    }
    return n;
}
// mutation piece of a range with events: the inherited entries are, per interval, a contiguous run of the source piece; the new
// mutations of this side that lie in the range are merged in (a new position goes behind every inherited entry that is not larger)
template <bool FILL>
__device__ __forceinline__ u32 lp_build_muts(const LpWork& lp, const u32 parent, const u32 start, const u64* __restrict__ bk, const u32 j0, const u32 j1,
                                             const u32 t, const u64* __restrict__ npos, const uint8_t* __restrict__ nside, u32 in, const u32 nn, const u32 s,
                                             const u64 bp0, const u64 bp_end, u64* __restrict__ out)
{
#define LP_NEXT_NEW() while (in < nn && !(nside[in] == s && npos[in] >= bp0 && npos[in] < bp_end && lp_seg(npos[in], bp0, lp.lgw, lp.nseg) == t)) in++
    u32 n = 0;
    LP_NEXT_NEW();
    for (u32 q = j0; q <= j1; q++) {
        const u32 h = (start ^ q) & 1u;
        const uint2 me = lp.mtab_cur[((size_t)2 * parent + h) * lp.nseg + t];
        const u64* a = lp.marena + me.x;
        u32 lo = 0, hi = me.y;
        if (q > j0) { const u64 v = bk[q - 1]; u32 l = 0, r = me.y; while (l < r) { const u32 m = (l + r) >> 1; if (a[m] < v) l = m + 1; else r = m; } lo = l; }
        if (q < j1) { const u64 v = bk[q]; u32 l = lo, r = me.y; while (l < r) { const u32 m = (l + r) >> 1; if (a[m] < v) l = m + 1; else r = m; } hi = l; }
        for (u32 x = lo; x < hi; x++) {
            const u64 v = a[x];
            while (in < nn && npos[in] < v) { if (FILL) out[n] = npos[in]; n++; in++; LP_NEXT_NEW(); }
            if (FILL) out[n] = v; n++;
        }
    }
    while (in < nn) { if (FILL) out[n] = npos[in]; n++; in++; LP_NEXT_NEW(); }
#undef LP_NEXT_NEW
    return n;
}
// events of (row, t): crossover boundaries in front of / up to the end of the range, a new mutation of the row's side inside it
struct LpEvents { u32 parent, start, s, j0, j1, in, nn; const u64* bk; bool fresh_p, fresh_m; };
__device__ __forceinline__ LpEvents lp_events(const ChrWork& w, size_t row, u32 t, int nchr, int has_mut, const SampleDev& sd)
{
    const LpWork& lp = w.lp;
    LpEvents ev;
    const u32 i = (u32)(row >> 1); ev.s = (u32)(row & 1);
    const size_t task = (size_t)i * nchr + w.chr, G_ = 2 * task + ev.s;
    ev.parent = ev.s ? sd.mother[i] : sd.father[i];
    ev.start = sd.start[G_];
    const u32 k = sd.k[G_];
    ev.bk = sd.bk + sd.bk_off[G_];
    const u64 lo_b = lp_lo(t, w.bp0, lp.lgw), hi_b = lp_hi(t, w.bp0, lp.lgw, lp.nseg);
    ev.j0 = 0; ev.j1 = 0;
    for (u32 j = 0; j < k; j++) { const u64 v = ev.bk[j]; ev.j0 += v < lo_b; ev.j1 += v < hi_b; }
    bool new_here = false;
    ev.in = 0; ev.nn = 0;
    if (has_mut) {
        ev.in = sd.nm_off[task]; ev.nn = ev.in + sd.nmut[task];
        for (u32 m = ev.in; m < ev.nn; m++) { const u64 x = sd.nm_pos[m]; new_here |= sd.nm_side[m] == ev.s && x >= w.bp0 && x < w.bp_end && lp_seg(x, w.bp0, lp.lgw, lp.nseg) == t; }
    }
    ev.fresh_p = lp.track && ev.j1 > ev.j0;
    ev.fresh_m = ev.j1 > ev.j0 || new_here;
    return ev;
}
// pass 1.  A workgroup takes 256 offspring rows.  First one thread per row turns the row's crossover boundaries and new mutations
// into three bit masks over the ranges (source haplotype at the range's start, ranges with a boundary, ranges with a new
// mutation) and puts the ranges with an event on the work list of pass 2 (items = row * LP_MAXSEG + t; one atomic per
// workgroup); then all threads stream the table entries of the 256 rows: a range without an event names the parent's pieces.
__global__ void __launch_bounds__(256) k_lp_inherit(const ChrWork* __restrict__ Wt, size_t n_rows_out, int nchr, int has_mut, SampleDev sd)
{
    __shared__ u32 lds[8];
    __shared__ u32 s_base;
    __shared__ u32 s_parent[256], s_hap[256], s_fp[256], s_fm[256];
    const ChrWork& w = Wt[blockIdx.y];
    const LpWork& lp = w.lp;
    const size_t row0 = (size_t)blockIdx.x * 256;
    const size_t row = row0 + threadIdx.x;
    u32 fresh = 0;
    if (row < n_rows_out) {
        const u32 i = (u32)(row >> 1), s = (u32)(row & 1);
        const size_t task = (size_t)i * nchr + w.chr, G_ = 2 * task + s;
        const u32 start = sd.start[G_], k = sd.k[G_];
        const u64* bk = sd.bk + sd.bk_off[G_];
        u32 cross = 0, flip = 0, mut = 0;                                  // flip bit t: an odd number of boundaries lies in front of range t
        for (u32 j = 0; j < k; j++) {
            const u32 g = lp_seg(bk[j], w.bp0, lp.lgw, lp.nseg);
            cross |= 1u << g;
            flip ^= g + 1 >= 32 ? 0u : ~0u << (g + 1);
        }
        if (has_mut) {
            const u32 in = sd.nm_off[task], nn = in + sd.nmut[task];
            for (u32 m = in; m < nn; m++) { const u64 x = sd.nm_pos[m]; if (sd.nm_side[m] == s && x >= w.bp0 && x < w.bp_end) mut |= 1u << lp_seg(x, w.bp0, lp.lgw, lp.nseg); }
        }
        s_parent[threadIdx.x] = s ? sd.mother[i] : sd.father[i];
        s_hap[threadIdx.x] = start ? ~flip : flip;
        s_fp[threadIdx.x] = lp.track ? cross : 0u;
        s_fm[threadIdx.x] = cross | mut;
        fresh = cross | mut;
    }
    u32 tot;
    const u32 ex = block_exclusive_scan_256((u32)__popc(fresh), lds, tot);
    if (threadIdx.x == 0) s_base = tot ? atomicAdd(lp.n_items, tot) : 0u;
    __syncthreads();
    for (u32 at = s_base + ex; fresh; fresh &= fresh - 1u) lp.items[at++] = (u32)(row * LP_MAXSEG) + (u32)__builtin_ctz(fresh);
    const u32 nseg = lp.nseg;
    const u32 n_here = (u32)(n_rows_out - row0 < 256 ? n_rows_out - row0 : 256);
    for (u32 e = threadIdx.x; e < n_here * nseg; e += 256) {
        const u32 r = e / nseg, t = e - r * nseg;
        const size_t src = ((size_t)2 * s_parent[r] + ((s_hap[r] >> t) & 1u)) * nseg + t, dst = (row0 + r) * nseg + t;
        if (lp.track && !((s_fp[r] >> t) & 1u)) lp.ptab_alt[dst] = lp.ptab_cur[src];
        if (!((s_fm[r] >> t) & 1u)) lp.mtab_alt[dst] = lp.mtab_cur[src];
    }
}
// pass 2, thread per item (persistent grid): the pieces of a range with an event, in two walks over the (short) source pieces --
// count, one atomic per workgroup and arena, fill
template <bool LITERAL>
__global__ void __launch_bounds__(256) k_lp_build(const ChrWork* __restrict__ Wt, int nchr, int has_mut, SampleDev sd)
{
    __shared__ u32 lds[8];
    __shared__ u32 s_base[2];
    const ChrWork& w = Wt[blockIdx.y];
    const LpWork& lp = w.lp;
    const u32 n_items = *lp.n_items;
    const u64 bp0 = w.bp0, bp_end = w.bp_end;
    for (u32 base = blockIdx.x * 256u; base < n_items; base += gridDim.x * 256u) {       // (uniform per workgroup)
        const u32 q = base + threadIdx.x;
        const bool live = q < n_items;
        size_t row = 0; u32 t = 0, np = 0, nm = 0;
        LpEvents ev{};
        if (live) {
            const u32 it = lp.items[q];
            row = it / LP_MAXSEG; t = it % LP_MAXSEG;
            ev = lp_events(w, row, t, nchr, has_mut, sd);
            if (ev.fresh_p) np = 1 + lp_build_parts<false, LITERAL>(lp, ev.parent, ev.start, ev.bk, ev.j0, ev.j1, t, bp0, bp_end, nullptr);
            if (ev.fresh_m) nm = lp_build_muts<false>(lp, ev.parent, ev.start, ev.bk, ev.j0, ev.j1, t, sd.nm_pos, sd.nm_side, ev.in, ev.nn, ev.s, bp0, bp_end, nullptr);
        }
        u32 tot_p, tot_m;
        const u32 ex_p = block_exclusive_scan_256(np, lds, tot_p);
        const u32 ex_m = block_exclusive_scan_256(nm, lds, tot_m);
        if (threadIdx.x == 0) {
            s_base[0] = tot_p ? atomicAdd(&sd.status[ST_TOTALS + ST_PER_CHR * w.chr + 1], tot_p) : 0u;
            s_base[1] = tot_m ? atomicAdd(&sd.status[ST_TOTALS + ST_PER_CHR * w.chr + 0], tot_m) : 0u;
        }
        __syncthreads();
        if (live) {
            const size_t dst = row * lp.nseg + t;
            if (ev.fresh_p) {
                const u64 off = (u64)lp.pbase + s_base[0] + ex_p;
                if (off + np > lp.pcap) { atomicOr(&sd.status[ST_FLAGS], (u32)FLAG_PARTS_CAP); lp.ptab_alt[dst] = make_uint2(0u, 0u); }
                else { lp_build_parts<true, LITERAL>(lp, ev.parent, ev.start, ev.bk, ev.j0, ev.j1, t, bp0, bp_end, lp.parena + off); lp.ptab_alt[dst] = make_uint2((u32)off, np - 1u); }
            }
            if (ev.fresh_m) {
                const u64 off = (u64)lp.mbase + s_base[1] + ex_m;
                if (off + nm > lp.mcap) { atomicOr(&sd.status[ST_FLAGS], (u32)FLAG_MUT_CAP); lp.mtab_alt[dst] = make_uint2(0u, 0u); }
                else { lp_build_muts<true>(lp, ev.parent, ev.start, ev.bk, ev.j0, ev.j1, t, sd.nm_pos, sd.nm_side, ev.in, ev.nn, ev.s, bp0, bp_end, lp.marena + off); lp.mtab_alt[dst] = make_uint2((u32)off, nm); }
            }
        }
        __syncthreads();                                  // s_base is written again in the next round
    }
}
// pass 2 with EIGHT lanes per item (the default): a piece is short (a few to a few dozen entries), and one lane that bisects it and
// copies it entry by entry spends its time in chains of dependent loads.  The eight lanes of a group load the piece's entries
// strided, all at once; c(Lc) and c(Rc) are the group's sums of two comparisons per entry (the entries are sorted: the number
// of entries <= x is all the closed form needs); the runs are copied eight entries (128 bytes) per step.  The mutation piece is
// merged the same way: an inherited entry moves up by the number of new positions in front of it, a new position goes behind the
// inherited entries that are not larger -- counted while they are copied.  More than LP8_NEW new mutations in one range of one
// row (hot maps): lane 0 merges sequentially.
#define LP8_NEW 4
__device__ __forceinline__ u32 lp_grp_sum8(u32 v) { v += (u32)__shfl_xor((int)v, 1, 8); v += (u32)__shfl_xor((int)v, 2, 8); v += (u32)__shfl_xor((int)v, 4, 8); return v; }
template <bool FILL>
__device__ __forceinline__ u32 lp_interval8(const LpPart* __restrict__ e, u32 n, u64 en_last, bool head, u64 Lc, u64 Rc, LpPart* __restrict__ out, u32 at, u32 sub)
{
    if (head && Lc >= en_last) return 0u;
    u32 x = 0, y = 0;
    for (u32 j = 1 + sub; j <= n; j += 8) { const u64 st = e[j].st; x += st <= Lc; y += st <= Rc; }
    const u32 c0 = lp_grp_sum8(x);
    u32 last = lp_grp_sum8(y);                                       // (Rc == LP_INF: every entry)
    u32 emitted = 0, first = 1;
    if (head) {
        const LpPart cov = e[c0];
        if (cov.st < Lc) { if (FILL && sub == 0) { LpPart p = cov; p.st = Lc; out[at] = p; } emitted = 1; first = c0 + 1; }
        else first = c0;
    }
    if (last >= 1 && Rc != LP_INF && e[last].st == Rc) last--;
    if (first >= 1 && first <= last) {
        if (FILL) for (u32 j = first + sub; j <= last; j += 8) out[at + emitted + (j - first)] = e[j];
        emitted += last - first + 1;
    }
    return emitted;
}
template <bool FILL>
__device__ __forceinline__ u32 lp_build_parts8(const LpWork& lp, const u32 parent, const u32 start, const u64* __restrict__ bk, const u32 j0, const u32 j1,
                                               const u32 t, const u64 bp0, const u64 bp_end, LpPart* __restrict__ out, u32 sub)
{
    const bool last = t + 1 >= lp.nseg;
    const u64 en_last = last ? bp_end : LP_INF;
    u32 n = 0;
    for (u32 q = j0; q <= j1; q++) {
        const u32 h = (start ^ q) & 1u;
        const uint2 pe = lp.ptab_cur[((size_t)2 * parent + h) * lp.nseg + t];
        const LpPart* e = lp.parena + pe.x;
        const bool head = q > j0 || t == 0;
        const u64 Lc = q > j0 ? bk[q - 1] : bp0;
        const u64 Rc = q < j1 ? bk[q] : (last ? bp_end : LP_INF);
        if (FILL && q == j0 && sub == 0) out[0] = e[0];
        n += lp_interval8<FILL>(e, pe.y, en_last, head, Lc, Rc, out, 1 + n, sub);
    }
    return n;
}
template <bool FILL>
__device__ __forceinline__ u32 lp_build_muts8(const LpWork& lp, const u32 parent, const u32 start, const u64* __restrict__ bk, const u32 j0, const u32 j1,
                                              const u32 t, const u64* __restrict__ nv, const u32 n_new, u64* __restrict__ out, u32 sub)
{
    u32 g = 0, cnt[LP8_NEW];
#pragma unroll
    for (int i = 0; i < LP8_NEW; i++) cnt[i] = 0;
    for (u32 q = j0; q <= j1; q++) {
        const u32 h = (start ^ q) & 1u;
        const uint2 me = lp.mtab_cur[((size_t)2 * parent + h) * lp.nseg + t];
        const u64* a = lp.marena + me.x;
        const u64 vlo = q > j0 ? bk[q - 1] : 0ull, vhi = q < j1 ? bk[q] : LP_INF;     // the run [lower_bound(vlo), lower_bound(vhi))
        u32 x = 0, y = 0;
        for (u32 j = sub; j < me.y; j += 8) { const u64 v = a[j]; x += v < vlo; y += (vhi == LP_INF) ? 1u : (u32)(v < vhi); }
        const u32 lo = lp_grp_sum8(x), hi = lp_grp_sum8(y);
        if (FILL)
            for (u32 j = lo + sub; j < hi; j += 8) {
                const u64 v = a[j];
                u32 shift = 0;
#pragma unroll
                for (int i = 0; i < LP8_NEW; i++) if ((u32)i < n_new) { shift += nv[i] < v; cnt[i] += v <= nv[i]; }
                out[g + (j - lo) + shift] = v;
            }
        g += hi > lo ? hi - lo : 0u;
    }
    if (FILL) {
#pragma unroll
        for (int i = 0; i < LP8_NEW; i++) if ((u32)i < n_new) { const u32 tot = lp_grp_sum8(cnt[i]); if (sub == 0) out[(u32)i + tot] = nv[i]; }
    }
    return g + n_new;
}
__global__ void __launch_bounds__(256) k_lp_build8(const ChrWork* __restrict__ Wt, int nchr, int has_mut, SampleDev sd)
{
    __shared__ u32 lds[8];
    __shared__ u32 s_base[2];
    const ChrWork& w = Wt[blockIdx.y];
    const LpWork& lp = w.lp;
    const u32 n_items = *lp.n_items;
    const u64 bp0 = w.bp0, bp_end = w.bp_end;
    const u32 sub = threadIdx.x & 7u, grp = threadIdx.x >> 3;
    for (u32 base = blockIdx.x * 32u; base < n_items; base += gridDim.x * 32u) {       // (uniform per workgroup)
        const u32 q = base + grp;
        const bool live = q < n_items;
        size_t row = 0; u32 t = 0, np = 0, nm = 0, n_new = 0;
        u64 nv[LP8_NEW];
#pragma unroll
        for (int i = 0; i < LP8_NEW; i++) nv[i] = 0;
        LpEvents ev{};
        if (live) {
            const u32 it = lp.items[q];
            row = it / LP_MAXSEG; t = it % LP_MAXSEG;
            ev = lp_events(w, row, t, nchr, has_mut, sd);
            for (u32 m = ev.in; m < ev.nn; m++) {                         // the new mutations of this row's side that lie in the range, ascending
                const u64 x = sd.nm_pos[m];
                if (sd.nm_side[m] == ev.s && x >= bp0 && x < bp_end && lp_seg(x, bp0, lp.lgw, lp.nseg) == t) { if (n_new < LP8_NEW) nv[n_new] = x; n_new++; }
            }
            if (ev.fresh_p) np = 1 + lp_build_parts8<false>(lp, ev.parent, ev.start, ev.bk, ev.j0, ev.j1, t, bp0, bp_end, nullptr, sub);
            if (ev.fresh_m) nm = n_new <= LP8_NEW ? lp_build_muts8<false>(lp, ev.parent, ev.start, ev.bk, ev.j0, ev.j1, t, nv, n_new, nullptr, sub)
                                                   : lp_build_muts<false>(lp, ev.parent, ev.start, ev.bk, ev.j0, ev.j1, t, sd.nm_pos, sd.nm_side, ev.in, ev.nn, ev.s, bp0, bp_end, nullptr);
        }
        u32 tot_p, tot_m;
        u32 ex_p = block_exclusive_scan_256(sub == 0 ? np : 0u, lds, tot_p);
        u32 ex_m = block_exclusive_scan_256(sub == 0 ? nm : 0u, lds, tot_m);
        ex_p = (u32)__shfl((int)ex_p, 0, 8); ex_m = (u32)__shfl((int)ex_m, 0, 8);
        if (threadIdx.x == 0) {
            s_base[0] = tot_p ? atomicAdd(&sd.status[ST_TOTALS + ST_PER_CHR * w.chr + 1], tot_p) : 0u;
            s_base[1] = tot_m ? atomicAdd(&sd.status[ST_TOTALS + ST_PER_CHR * w.chr + 0], tot_m) : 0u;
        }
        __syncthreads();
        if (live) {
            const size_t dst = row * lp.nseg + t;
            if (ev.fresh_p) {
                const u64 off = (u64)lp.pbase + s_base[0] + ex_p;
                if (off + np > lp.pcap) { if (sub == 0) { atomicOr(&sd.status[ST_FLAGS], (u32)FLAG_PARTS_CAP); lp.ptab_alt[dst] = make_uint2(0u, 0u); } }
                else { lp_build_parts8<true>(lp, ev.parent, ev.start, ev.bk, ev.j0, ev.j1, t, bp0, bp_end, lp.parena + off, sub); if (sub == 0) lp.ptab_alt[dst] = make_uint2((u32)off, np - 1u); }
            }
            if (ev.fresh_m) {
                const u64 off = (u64)lp.mbase + s_base[1] + ex_m;
                if (off + nm > lp.mcap) { if (sub == 0) { atomicOr(&sd.status[ST_FLAGS], (u32)FLAG_MUT_CAP); lp.mtab_alt[dst] = make_uint2(0u, 0u); } }
                else {
                    if (n_new <= LP8_NEW) lp_build_muts8<true>(lp, ev.parent, ev.start, ev.bk, ev.j0, ev.j1, t, nv, n_new, lp.marena + off, sub);
                    else if (sub == 0) lp_build_muts<true>(lp, ev.parent, ev.start, ev.bk, ev.j0, ev.j1, t, sd.nm_pos, sd.nm_side, ev.in, ev.nn, ev.s, bp0, bp_end, lp.marena + off);
                    if (sub == 0) lp.mtab_alt[dst] = make_uint2((u32)off, nm);
                }
            }
        }
        __syncthreads();                                  // s_base is written again in the next round
    }
}
// ------------------------------------------------------------------------------------------
// pieces -> CSR (downloads, output, migration, plane-less assembly, compaction).  Row r of the output is table row
// (map ? map[r] : r).  Half a wave per row (LP_MAXSEG = 32 lanes): counts by a butterfly sum, offsets by a 32-lane scan.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_lp_count(const uint2* __restrict__ ptab, const uint2* __restrict__ mtab, u32 nseg, const u32* __restrict__ map, size_t n_rows,
                                                  u32* __restrict__ pcnt, u32* __restrict__ mcnt)
{
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t r = tid / LP_MAXSEG; const u32 t = (u32)(tid % LP_MAXSEG);
    u32 np = 0, nm = 0;
    if (r < n_rows && t < nseg) {
        const size_t row = map ? (size_t)map[r] : r;
        if (ptab) np = ptab[row * nseg + t].y;
        nm = mtab[row * nseg + t].y;
    }
#pragma unroll
    for (int d = 1; d < LP_MAXSEG; d <<= 1) { np += __shfl_xor(np, d, LP_MAXSEG); nm += __shfl_xor(nm, d, LP_MAXSEG); }
    if (r < n_rows && t == 0) { if (pcnt) pcnt[r] = np; mcnt[r] = nm; }
}
__global__ void __launch_bounds__(256) k_lp_fill(const uint2* __restrict__ ptab, const uint2* __restrict__ mtab, const LpPart* __restrict__ parena, const u64* __restrict__ marena,
                                                 u32 nseg, const u32* __restrict__ map, size_t n_rows, u64 bp_end,
                                                 const u32* __restrict__ poff, gev_part* __restrict__ parts, const u32* __restrict__ moff, u64* __restrict__ mpos)
{
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t r = tid / LP_MAXSEG; const u32 t = (u32)(tid % LP_MAXSEG);
    const bool live = r < n_rows && t < nseg;
    const size_t row = live ? (map ? (size_t)map[r] : r) : 0;
    uint2 pe = make_uint2(0u, 0u), me = make_uint2(0u, 0u);
    if (live) { if (ptab) pe = ptab[row * nseg + t]; me = mtab[row * nseg + t]; }
    u32 ip = pe.y, im = me.y;                                          // inclusive scans over the row's ranges
#pragma unroll
    for (int d = 1; d < LP_MAXSEG; d <<= 1) { const u32 a = __shfl_up(ip, d, LP_MAXSEG), b = __shfl_up(im, d, LP_MAXSEG); if (t >= (u32)d) { ip += a; im += b; } }
    // en of a range's last part = st of the row's next part = first owned st of the next non-empty range (bp_end behind the last)
    u64 nxt = (live && pe.y) ? parena[pe.x + 1].st : LP_INF;           // this range's first owned st
    u64 after = LP_INF;                                                // min over the ranges behind this one
    {
        u64 run = nxt;                                                 // suffix minimum, exclusive: shift down by one first
        u64 v = __shfl_down(run, 1, LP_MAXSEG); if (t + 1 >= LP_MAXSEG) v = LP_INF;
        run = v;
#pragma unroll
        for (int d = 1; d < LP_MAXSEG; d <<= 1) { u64 o = __shfl_down(run, d, LP_MAXSEG); if (t + d >= LP_MAXSEG) o = LP_INF; run = run < o ? run : o; }
        after = run;
    }
    if (!live) return;
    if (ptab && pe.y) {
        gev_part* out = parts + poff[r] + (ip - pe.y);
        const LpPart* e = parena + pe.x + 1;
        for (u32 j = 0; j < pe.y; j++) {
            gev_part p; p.st = e[j].st; p.en = j + 1 < pe.y ? e[j + 1].st : (after == LP_INF ? bp_end : after);
            p.hap_index = e[j].hap_index; p.root_population = (int32_t)e[j].root_population; p.reserved = 0;
            out[j] = p;
        }
    }
    if (me.y) {
        u64* out = mpos + moff[r] + (im - me.y);
        const u64* a = marena + me.x;
        for (u32 j = 0; j < me.y; j++) out[j] = a[j];
    }
}
// CSR -> pieces for table rows [row0, row0 + n_rows) (generation 0, uploads, immigrants, compaction): CSR row r -> table row row0 + r.
// ctr[0], ctr[1]: entries appended to the interval / mutation arena by this launch (on top of pbase / mbase); flag: an arena is full
__global__ void __launch_bounds__(256) k_lp_import(const u32* __restrict__ poff, const gev_part* __restrict__ parts, const u32* __restrict__ moff, const u64* __restrict__ mpos,
                                                   size_t row0, size_t n_rows, uint2* __restrict__ ptab, uint2* __restrict__ mtab, LpPart* __restrict__ parena, u64* __restrict__ marena,
                                                   u32 pbase, u32 mbase, u32 pcap, u32 mcap, u32 nseg, u32 lgw, u64 bp0, u32* __restrict__ ctr, u32* __restrict__ flag)
{
    __shared__ u32 lds[8];
    __shared__ u32 s_base[2];
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t r = tid / LP_MAXSEG; const u32 t = (u32)(tid % LP_MAXSEG);
    const bool live = r < n_rows && t < nseg;
    u32 p_lo = 0, p_hi = 0, p_first = 0, m_lo = 0, m_hi = 0, np = 0, nm = 0;
    if (live) {
        const u64 lo_b = lp_lo(t, bp0, lgw), hi_b = lp_hi(t, bp0, lgw, nseg);
        if (ptab) {
            const u32 a = poff[r], b = poff[r + 1];
            p_first = a;
            u32 l = a, h = b; while (l < h) { const u32 m = (l + h) >> 1; if (parts[m].st < lo_b) l = m + 1; else h = m; } p_lo = l;
            if (hi_b == LP_INF) p_hi = b; else { l = p_lo; h = b; while (l < h) { const u32 m = (l + h) >> 1; if (parts[m].st < hi_b) l = m + 1; else h = m; } p_hi = l; }
            np = 1 + (p_hi - p_lo);
        }
        const u32 a = moff[r], b = moff[r + 1];
        u32 l = a, h = b; while (l < h) { const u32 m = (l + h) >> 1; if (mpos[m] < lo_b) l = m + 1; else h = m; } m_lo = l;
        if (hi_b == LP_INF) m_hi = b; else { l = m_lo; h = b; while (l < h) { const u32 m = (l + h) >> 1; if (mpos[m] < hi_b) l = m + 1; else h = m; } m_hi = l; }
        nm = m_hi - m_lo;
    }
    u32 tot_p, tot_m;
    const u32 ex_p = block_exclusive_scan_256(np, lds, tot_p);
    const u32 ex_m = block_exclusive_scan_256(nm, lds, tot_m);
    if (threadIdx.x == 0) { s_base[0] = tot_p ? atomicAdd(&ctr[0], tot_p) : 0u; s_base[1] = tot_m ? atomicAdd(&ctr[1], tot_m) : 0u; }
    __syncthreads();
    if (!live) return;
    const size_t dst = (row0 + r) * nseg + t;
    if (ptab) {
        const u64 off = (u64)pbase + s_base[0] + ex_p;
        if (off + np > pcap) { atomicOr(flag, 1u); ptab[dst] = make_uint2(0u, 0u); }
        else {
            LpPart* o = parena + off;
            const u32 have = poff[r + 1] - p_first;
            const u32 c = p_lo > p_first ? p_lo - 1 : p_lo;              // the part in front of the range's first owned one (first range: a copy of it)
            for (u32 j = 0; j < np; j++) {
                const u32 q = j == 0 ? c : p_lo + j - 1;
                LpPart e; e.st = 0; e.hap_index = 0; e.root_population = 0;
                if (have && q < poff[r + 1]) { e.st = parts[q].st; e.hap_index = (u32)parts[q].hap_index; e.root_population = (u32)parts[q].root_population; }
                o[j] = e;
            }
            ptab[dst] = make_uint2((u32)off, np - 1u);
        }
    }
    {
        const u64 off = (u64)mbase + s_base[1] + ex_m;
        if (off + nm > mcap) { atomicOr(flag, 2u); mtab[dst] = make_uint2(0u, 0u); }
        else { u64* o = marena + off; for (u32 j = 0; j < nm; j++) o[j] = mpos[m_lo + j]; mtab[dst] = make_uint2((u32)off, nm); }
    }
}
// table rows of selected individuals: dst row r <- src row map[r] (both tables; pieces are shared, nothing else moves)
__global__ void __launch_bounds__(256) k_lp_gather_rows(const uint2* __restrict__ sp, const uint2* __restrict__ sm, uint2* __restrict__ dp, uint2* __restrict__ dm,
                                                        const u32* __restrict__ map, size_t n_rows, u32 nseg)
{
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t r = tid / nseg; const u32 t = (u32)(tid % nseg);
    if (r >= n_rows) return;
    const size_t s = (size_t)map[r] * nseg + t, d = r * nseg + t;
    if (sp) dp[d] = sp[s];
    dm[d] = sm[s];
}
// ------------------------------------------------------------------------------------------
// Compaction of an arena that keeps the sharing.  Pieces no table entry names any more are dropped (a piece is named by the rows
// that inherited it unchanged: after k generations only ~2/k of a generation's pieces still are), the others move to the front in
// arena order: bit per entry = "a named piece starts here" (k_lpc_mark), rank of a piece = set bits in front of it (word popcounts
// + scan), length and old offset per rank (k_lpc_collect: every entry that names the piece writes the same values), scan of the
// lengths = new offsets, copy (k_lpc_copy), table rewritten (k_lpc_rewrite).  extra = 1 for interval pieces (carry-in entry).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 lpc_rank(const u32* __restrict__ bits, const u32* __restrict__ wpre, u32 off)
{
    return wpre[off >> 5] + (u32)__popc(bits[off >> 5] & ((1u << (off & 31u)) - 1u));
}
__global__ void __launch_bounds__(256) k_lpc_mark(const uint2* __restrict__ tab, size_t n_entries, u32 extra, u32* __restrict__ bits)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_entries) return;
    const uint2 v = tab[e];
    if (v.y + extra) atomicOr(&bits[v.x >> 5], 1u << (v.x & 31u));
}
__global__ void __launch_bounds__(256) k_lpc_popc(const u32* __restrict__ bits, size_t n_words, u32* __restrict__ cnt)
{
    const size_t w = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (w < n_words) cnt[w] = (u32)__popc(bits[w]);
}
__global__ void __launch_bounds__(256) k_lpc_collect(const uint2* __restrict__ tab, size_t n_entries, u32 extra, const u32* __restrict__ bits, const u32* __restrict__ wpre,
                                                     u32* __restrict__ plen, u32* __restrict__ pold)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_entries) return;
    const uint2 v = tab[e];
    if (!(v.y + extra)) return;
    const u32 r = lpc_rank(bits, wpre, v.x);
    plen[r] = v.y + extra; pold[r] = v.x;
}
// (ew = 8-byte words per entry: 2 for interval entries, 1 for mutation positions)
__global__ void __launch_bounds__(256) k_lpc_copy(const u64* __restrict__ src, u64* __restrict__ dst, const u32* __restrict__ plen, const u32* __restrict__ pold, const u32* __restrict__ pnew, size_t n_pieces, u32 ew)
{
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_pieces) return;
    const u64* a = src + (size_t)pold[r] * ew; u64* o = dst + (size_t)pnew[r] * ew;
    const u32 n = plen[r] * ew;
    for (u32 j = 0; j < n; j++) o[j] = a[j];
}
__global__ void __launch_bounds__(256) k_lpc_rewrite(uint2* __restrict__ tab, size_t n_entries, u32 extra, const u32* __restrict__ bits, const u32* __restrict__ wpre, const u32* __restrict__ pnew)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_entries) return;
    uint2 v = tab[e];
    v.x = (v.y + extra) ? pnew[lpc_rank(bits, wpre, v.x)] : 0u;
    tab[e] = v;
}
#endif   // GEV_LISTS_KERNELS
