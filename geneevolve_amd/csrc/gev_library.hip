// gev_library.hip -- C-ABI (include/geneevolve_amd.h) over the HIP kernels of gev_kernels.h.
// MI355X (gfx950) only.  Host side = plumbing: argument checks, device memory, launch order.
// No CPU fallback of any kernel exists: without a GPU every compute entry point fails loudly.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC gev_library.hip -o libgeneevolve_amd.so
//
// Tuning / diagnosis knobs (environment, read at gev_create; defaults are what the measurements of DESIGN.md chose):
//   GEV_OVERLAP=0|1|2|-1      stream overlap: never | everything (default) | sampling only | decide from two timed generations
//   GEV_SERIALIZE=1           same as GEV_OVERLAP=0
//   GEV_SAMPLE_BATCHED=0|1    sampling kernels: one task per wave | eight tasks per wave (default)
//   GEV_STITCH_MODE=0|1       dense stitch kernel: k_stitch_segments (default) | k_stitch_rows (same results)
//   GEV_SAMPLE_GRID=n         persistent workgroups of the sampling kernels (default 1792 up to 400k tasks per generation, 768 above; 1024 alone)
//   GEV_STITCH_WG_PER_CU=n|auto stitch workgroups per CU, by dynamic LDS padding (default: unlimited; auto: measured at run time)
//   GEV_STITCH_LDS_PAD=bytes  (experiments) that padding directly
//   GEV_ALIAS_ROWS=0|1        write every segment of every gamete row | segments without a crossover boundary share the parental unit (default)
//   GEV_LIST_SEGS=n           most position ranges per row the mutation / interval lists are cut into (default 32; 1 = a piece is a whole list)
//   GEV_LIST_ARENA=n          entries per row of the list-piece arenas (default: a tenth of the device memory, at most 4096 per row)
//   GEV_LIST_HEADROOM=0       size every list buffer exactly (tests: every generation overflows, grows and is enqueued again)
//   GEV_STREAM_PRIO=xxxxx     h|m|l for the main, head-start, mating, list and stitch streams (default hhhhh)
//   GEV_STITCH_PRIORITY=1|2   (older form) stitch stream high and the others low / all streams equal
//   GEV_STITCH_START=0|1|2    the stitch starts behind the unit table | the CV planes | the generation's whole small work (default 2)
//   GEV_STITCH_U=1|2|4        16-byte chunks per lane in flight in the segment stitch (default 1)
//   GEV_SIDE_STREAMS=0        mating and list kernels on the main stream
//   GEV_HEAD_START=1          the next generation's seeds + sampling are enqueued in front of this generation's work, and only the next
//                             generation's unit table waits for the sampling (default 0: behind this generation's work, waited for as a whole)
//   GEV_AD_WIDE=0             A/D sums with the 128-CV table pieces of k_ad_accumulate_tab instead of k_ad_accumulate_wide
//   GEV_CHAIN_WG=0            serial-chain mode (no mutation map): one wave per link instead of a workgroup
//   GEV_CHAIN_MAX_TASKS=n     most (offspring, chromosome) tasks accepted without a mutation map (default 4 000 000)
//   GEV_POOL_REBUILD=1        rebuild the free list of the segment pool every generation
//   GEV_STITCH_GRID=n         persistent workgroups of the segment stitch per chromosome (default 16384)
//   GEV_SEG_CHUNKS=2^k        16-byte chunks per row segment (default 128 = 2 KiB)
//   GEV_STITCH_WAVE_PRIO=0..3 s_setprio level of the stitch kernel's waves (default 0; measured: no effect next to the sampling kernels)
//   GEV_TABLE_RING_BYTES=n    minimum size of the pinned ring the per-generation work tables are staged in (default 256 KiB)
//   GEV_TRACE_HOST=1          stderr: host time per phase of gev_reproduce, per generation the host's and the main stream's time split by
//                             phase (timed events), allocations, deferred frees
#include <hip/hip_runtime.h>
#include <chrono>
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "gev_kernels.h"

static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
#define HIPC(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(GEV_EDEVICE, "HIP error %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, __LINE__, #expr); } while (0)
#define GEVC(expr) do { int rc_ = (expr); if (rc_ != GEV_OK) return rc_; } while (0)
#define KCHECK() HIPC(hipGetLastError())

static inline size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }
static inline size_t round_up(size_t a, size_t b) { return ceil_div(a, b) * b; }

// Device buffers that were outgrown while kernels may still be reading them: freeing (hipFree waits for the whole device)
// would stall the host behind the running stitch, so they are parked here and released at the next point where the
// device is idle anyway (gev_sync, serialised generations, gev_destroy) or when too much has piled up.
static bool g_trace_host = getenv("GEV_TRACE_HOST") != nullptr;      // stderr: where the host spends its time inside gev_reproduce
static double host_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double g_malloc_ms = 0; static size_t g_malloc_n = 0, g_malloc_bytes = 0;   // GEV_TRACE_HOST bookkeeping
struct Graveyard {
    struct Item { void* p; size_t bytes; int device; };
    std::vector<Item> items; size_t bytes = 0;
    void park(void* p, size_t n) { int d = 0; (void)hipGetDevice(&d); items.push_back({p, n, d}); bytes += n; }
    void drain(int device, bool device_is_idle)
    {
        bool any = false;
        for (auto& it : items) any |= it.device == device;
        if (!any) return;
        int cur = 0; (void)hipGetDevice(&cur);
        (void)hipSetDevice(device);
        const double t0 = host_ms();
        if (!device_is_idle) (void)hipDeviceSynchronize();
        const double t1 = host_ms(); const size_t n0 = items.size(), b0 = bytes;
        size_t w = 0;
        for (auto& it : items) { if (it.device == device) { (void)hipFree(it.p); bytes -= it.bytes; } else items[w++] = it; }
        items.resize(w);
        if (g_trace_host) fprintf(stderr, "[gev] graveyard drain: sync %.2f ms, %zu frees of %.1f MiB in %.2f ms\n", t1 - t0, n0 - w, (b0 - bytes) / 1048576.0, host_ms() - t1);
        (void)hipSetDevice(cur);
    }
};
static Graveyard g_graveyard;                 // one host thread calls the seam

static const size_t GRAVEYARD_LIMIT = (size_t)16 << 30;

struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept { p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
    int ensure(size_t need, hipStream_t st, bool keep = false, double slack = 1.0)
    {
        if (need <= bytes && p) return GEV_OK;
        size_t want = std::max<size_t>((size_t)(need * slack), 256);
        void* q = nullptr;
        const double tm0 = host_ms();
        hipError_t e = hipMalloc(&q, want);
        g_malloc_ms += host_ms() - tm0; g_malloc_n++; g_malloc_bytes += want;
        if (e != hipSuccess && g_graveyard.bytes) {                      // out of memory with parked buffers: release them and retry
            (void)hipGetLastError();
            int d = 0; (void)hipGetDevice(&d); g_graveyard.drain(d, false);
            e = hipMalloc(&q, want);
        }
        if (e != hipSuccess && want > need) { (void)hipGetLastError(); want = std::max<size_t>(need, 256); e = hipMalloc(&q, want); }
        if (e != hipSuccess) return fail(GEV_EDEVICE, "hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
        if (p) {
            if (keep && bytes) { HIPC(hipMemcpyAsync(q, p, bytes, hipMemcpyDeviceToDevice, st)); }
            g_graveyard.park(p, bytes);        // the old buffer may still be in use on either stream
            if (g_graveyard.bytes > GRAVEYARD_LIMIT) { int d = 0; (void)hipGetDevice(&d); g_graveyard.drain(d, false); }
        }
        p = q; bytes = want;
        return GEV_OK;
    }
    template <class T> T* as() const { return (T*)p; }
};

struct ChrStatic {                       // one population x one chromosome
    std::vector<u64> rbp; std::vector<double> rprob; u64 bp_dist = 0;
    std::vector<u64> mbp; std::vector<double> mrate; bool mut_set = false;
    std::vector<u64> pos;                // Legend.pos
    DevBuf d_rthr, d_rbp, d_mthr, d_mbp, d_pos, d_coarse;
    size_t L = 0, stride = 0;            // bytes of a whole genotype row in flat (host-side / staging) layouts, multiple of 128
    u32 nseg = 1, seg_shift = 9;         // the device keeps a row as nseg segments of 2^seg_shift 16-byte chunks (gev_kernels.h, PoolWork)
    size_t unit_bytes() const { return (size_t)16 << seg_shift; }
    size_t pitch() const { return (size_t)nseg * unit_bytes(); }      // founder rows: row r = units r*nseg .. r*nseg+nseg-1 = a flat row of this pitch
    u32 idx_lo = 0, idx_hi = 0;          // loci inside [bp0, bp_end)
    u32 r_amax = 0, m_amax = 0;
    size_t founder_rows = 0;
    DevBuf panel; size_t panel_rows = 0; // read-only founder panel of this ROOT population, flat rows of `stride` bytes (gev_*_founder_panel): immigrants' rows are rebuilt from it
};
struct CvStatic {                        // one population x phenotype x chromosome
    std::vector<u64> bp; std::vector<double> a, d; double vd = 0; bool set = false;
    std::vector<u32> col_of_icv;         // file index -> column in the sorted CV plane
    std::vector<u32> icv_of_col;
    DevBuf d_pos_sorted, d_pos_file, d_col_of_icv, d_icv_of_col, d_a, d_d, d_frq, d_counts, d_partial, d_aptr, d_dptr, d_tab;
    u32 C = 0, sub_w32 = 0, stride_w32 = 0;
    u32 idx_lo = 0, idx_hi = 0;
    size_t founder_rows = 0;
    bool frq_valid = false;
    bool cols_sorted = false;            // file order == position order (col_of_icv is the identity)
};
struct ChrState {
    // genotype rows: one pool of 4 * cap_people rows (two generations' worth); slot s of the current generation is pool row
    // phys[pcur][s] (gev_kernels.h, PoolWork).  phys has THREE buffers: the stitch of generation g (stream_big) still reads
    // phys of g-1 and g while the small work of g+1 writes the next one.
    DevBuf pool, phys[3], live, freel, pctr, items[2]; u32 pool_stamp = 0;
    // the free list of the pool is rebuilt (mark + collect) only when it runs low: valid = built for the current table lineage,
    // n_free / cursor = its length and how much of it has been handed out (from the last generation's status block)
    bool pool_list_valid = false, pool_force_rebuild = false; u32 pool_n_free = 0, pool_cursor = 0, pool_last_taken = 0;
    unsigned long long pool_rebuilds = 0;
    // Sparse state (mutation lists, ancestry interval lists) in TWO forms, each derived from the other on demand:
    //   pieces (gev_lists.h): what Simulation::reproduce reads and writes -- per-range pieces shared between parent and offspring
    //   CSR (moff/mpos, poff/parts of buffer P.cur): what everything else reads (downloads, output, migration, plane-less assembly)
    // csr_valid / lp.valid say which of them describe the current generation (at least one does).
    DevBuf moff[2], mpos[2], poff[2], parts[2];
    size_t mut_total[2] = {0, 0}, parts_total[2] = {0, 0};   // list sizes of the two CSR buffers
    bool csr_valid = true;
    struct LpState {
        DevBuf ptab[2], mtab[2], parena, marena, ctr, items;        // tables [P.cur] = current generation; ctr: {interval entries, mutation entries appended by an import, overflow flags}
        u32 p_used = 0, m_used = 0;                          // arena cursors (entries)
        u32 p_last = 0, m_last = 0;                          // what the last generation appended
        u32 nseg = 0, lgw = 0;
        bool valid = false, grow_p = false, grow_m = false;
        unsigned long long imports = 0, compactions = 0;     // rebuilds from whole lists (first use, after a CSR-side change); arena compactions
    } lp;
};
struct PopState {
    std::vector<ChrStatic> cs;                         // [chr]
    std::vector<std::vector<CvStatic>> cv;             // [phen][chr]
    std::vector<ChrState> st;                          // [chr]
    std::vector<std::vector<std::array<DevBuf, 2>>> cvp; // [phen][chr][2]
    DevBuf d_chrdev;
    DevBuf d_sex[2];                                   // Human::sex of the individuals (1 male, 2 female), physical row order, [cur]: what Simulation::random_mate reads (gev_random_mate)
    int cur = 0, pcur = 0;                             // current buffer of the double-buffered lists / of phys[3]
    size_t n_people = 0, cap_people = 0;
    // After a cross-GPU migration the individuals of the current generation are not moved physically:
    // `logical[i]` = physical individual index of logical position i (empty = identity).  The next
    // gev_reproduce maps parent positions through it and writes a dense generation again; anything
    // else that needs the dense order calls materialize_order() first.
    std::vector<u32> logical;
    size_t n_phys = 0;
    bool finalized = false, gen0 = false;
};

struct gev_ctx {
    int device = 0, n_pop = 0, nchr = 0, nphen = 0;
    u32 rp_bits = 0;
    bool ad_effects_shared = false;         // every root population has the same CV effects bit for bit (check_multipop): A/D uses the one-population term table
    hipStream_t stream = nullptr, stream_samp = nullptr, stream_aux = nullptr, stream_list = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    float last_ms[4] = {0, 0, 0, 0};
    bool track_intervals = true;
    std::vector<PopState> pop;
    DevBuf d_tables;
    // per-generation scratch: two sets, because the dense stitch of generation g (stream_big) still reads set g%2
    // while sampling / sparse state of generation g+1 (stream) fill the other one
    struct Scratch {
        DevBuf father, mother, mutseeds, globvals /* [2 + T] ras_glob_seed() values drawn on the device: mate seed, reproduce seed, mutation seeds */, seed_pat, seed_mat, k, bk_off, bk, bk_idx, start, nmut, nm_off, nm_pos, nm_side, sex, status, slow_mut, slow_rec, chrwork, cvwork;
        std::vector<uint8_t> chrwork_shadow, cvwork_shadow;     // what the device copies of the tables hold (upload_table_cached)
        unsigned n_chrwork = 0, n_cvwork = 0; float sampling_ms_saved = -1;
        size_t nseg_max = 1, cv_used_max = 0, lp_entries_per_row = 0; u32 cv_max = 0;     // launch shapes of the generation (enqueue_tables)
        bool pool_rebuild = false;                                                            // this attempt rebuilds the free list of the segment pool
        bool cv_count_fused = false;                                                          // k_stitch_small also counts the alleles per CV column (every grid <= 1024 columns)
        hipEvent_t ev_fork = nullptr, ev_aux = nullptr, ev_lists = nullptr, ev_forked = nullptr;   // joins of the attempt's side streams
        hipEvent_t ev_small_done = nullptr, ev_stitch_done = nullptr, t[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; hipEvent_t ev_status = nullptr, ev_sampled = nullptr, ev_seeded = nullptr;   // the generation's status block (and A/D results) have arrived on the host
        bool timing_pending = false, stitch_pending = false;
        hipEvent_t tc[3] = {nullptr, nullptr, nullptr};   // GEV_TRACE_HOST: start of the attempt's main stream, A/D done, lists joined
        double th_enq = 0;
        // gev_presample: the sampling kernels of the next gev_reproduce were already enqueued for exactly these inputs
        bool presampled = false; int ps_pop = -1; u32 ps_seed = 0; size_t ps_n_people = 0; bool ps_has_mut = false;
        bool ps_stale = false;      // the head start was dropped by a redo of the generation in flight (record capacities changed): gev_presample_sex samples again from the retained inputs
        // gev_random_mate: father / mother of this set hold the couples of the next gev_reproduce (couples == NULL) of mate_pop
        bool mated = false; int mate_pop = -1; size_t mate_n = 0;
        // gev_set_generation_chain: seeds drawn and sampling enqueued for the NEXT gev_generation_begin (same population, size) from the predicted engine state
        bool fused_ahead = false, fa_dropped = false, fa_has_mut = false; int fa_pop = -1; size_t fa_n = 0;
        hipEvent_t ev_chain = nullptr, ev_tab = nullptr;
    } sc[2];
    unsigned gen_counter = 0;
    std::vector<uint8_t> chr_active;        // 0: chromosome held by another context (gev_set_chr_active); sampling chain only
    bool any_inactive = false;
    bool migrant_rows = true;               // false (gev_set_migrant_rows): gev_export_rows packs no genotype rows, gev_import_rows rebuilds them from the founder panels
    DevBuf d_poff, d_prange;                      // rebuild of immigrants' rows: PanelRef per root population, offsets of the rows' parts in the payload
    bool dense = true;                      // false: no resident genotype planes (gev_set_dense_state); lists + CV planes only, output by gev_materialize
    hipStream_t stream_big = nullptr;
    bool planes_pending = false;            // a stitch may still be writing the current planes (stream_big)
    hipEvent_t ev_planes = nullptr;         // recorded after the most recent stitch
    double ms_sum[4] = {0, 0, 0, 0}; unsigned long long ms_count = 0;
    size_t bk_ovf_cap = 1 << 16, nm_ovf_cap = 1 << 16;       // overflow regions of the breakpoint / new-mutation records (GEV_OVF_CAP: initial size, tests force redos with a tiny one)
    int chain_draws = -1;                                    // gev_set_generation_chain: ras_glob_seed() draws the host makes between two generations (-1: unknown, no head start)
    bool chain_valid = false; u32 chain_state = 0;           // glob_generator state the queued head start assumed for the next gev_generation_begin
    unsigned long long chain_hits = 0, chain_misses = 0;
    unsigned long long redo_count = 0;                       // generations that were enqueued again with larger buffers (gev_redo_count)
    void* h_stage = nullptr; size_t h_stage_bytes = 0;       // pinned host staging
    void* h_seeds = nullptr; size_t h_seeds_bytes = 0;       // pinned copy of the mutation seeds handed to gev_presample
    // pinned A/D result cache, two buffers: [(gen_counter + 1) & 1] holds the PUBLISHED generation's values, the generation in flight
    // fills the other one -- gev_compute_ad may therefore be served between gev_generation_begin and _end (the host hands the next
    // generation over first and reads the last one's A/D while the device works)
    void* h_ad2[2] = {nullptr, nullptr}; size_t h_ad2_bytes[2] = {0, 0}; bool ad_dom_zero2[2] = {false, false};
    int ad_cached_pop = -1;                                  // population whose current-generation A/D sits in h_ad
    int ad_host_set_pop = -1;                                // population whose raw A/D totals on the device were supplied by gev_set_ad (locus-split: all-reduced)
    bool eager_ad = true;                                    // compute A/D inside gev_reproduce (same enqueue, same sync)
    int stitch_start = 2;          // when the dense stitch of a generation may start: 0 behind the unit table, 1 behind the CV planes, 2 behind the whole small work incl. A/D (default: with the stitch at the small kernels' priority it then runs alone for 0.16 ms, next to the host's turn-around; GEV_STITCH_START)
    unsigned cv_threads = 512; bool cv_count_fused_ok = true;   // k_stitch_small: threads per block (GEV_CV_THREADS=256|512|1024), column counts in the same pass (GEV_CV_COUNT_FUSED=0: separate k_cv_count)
    int stitch_u = 1;              // 16-byte chunks per lane in flight in the segment stitch: a 2 KiB segment is one step of a wave (GEV_STITCH_U=1|2|4; 8 KiB segments: 2: 722, 4: 778 generations/s)
    int head_start = 0;            // the next generation's seeds + sampling: 0 = enqueued behind this generation's work, the whole next generation waits for them; 1 = enqueued first, the next generation's mating waits for the seeds and its unit table for the sampling (GEV_HEAD_START; same rate at config 2)
    bool side_streams = true;      // mate + free list next to the sampling, lists next to CV planes + A/D (GEV_SIDE_STREAMS=0: one stream)
    int stitch_mode = 0;           // 0 = work-list form (production, k_stitch_segments), 1 = gamete-major (k_stitch_rows)
    bool sample_batched = true;               // K1-K3 as eight tasks per wave (gev_sample8.h); GEV_SAMPLE_BATCHED=0: one task per wave
    unsigned sample_grid = SAMPLE_GRID_MAX;   // persistent workgroups of the sampling kernels when they have the GPU to themselves (GEV_SAMPLE_GRID)
    bool sample_grid_env = false;             // GEV_SAMPLE_GRID fixed both
    unsigned sample_grid_shared = 768;        // ... and next to the other streams' kernels.  Next to the 5 ms whole-row stitch of the first half of round 2, 384 was
                                              // best (fewer of the stitch's slots taken for longer); with the 0.7 ms segment stitch and the pipelined host loop the
                                              // sampling is what gev_presample_sex waits for: 768 finishes in 0.45 instead of 0.95 ms (config 2 +4 %, the shard unchanged)
    bool serialize = false;        // wait for every stitch (no overlap between the two streams)
    int overlap_mode = 1;          // 1 everything (default), 0 never, 2 sampling only, -1 decide after two serialised generations
    bool sparse_after_stitch = false;   // mode 2: the memory-bound sparse/CV/A-D kernels wait for the running stitch, only the ALU-bound sampling shares the GPU with it
    int auto_gens = 0; double auto_small_ms = 0, auto_stitch_ms = 0;
    size_t stitch_grid = 16384;   // most workgroups (4 waves = 4 work-list entries each) of the segment stitch per chromosome (GEV_STITCH_GRID)
    int stitch_wave_prio = 0;      // s_setprio level of the stitch kernel's waves (GEV_STITCH_WAVE_PRIO)
    u32 seg_shift = 7;             // log2(16-byte chunks per row segment): 2 KiB.  Smaller: more table entries to manage per generation; larger: more bytes copied per
                                   // crossover.  Round 3 (free list kept across generations, one thread per table entry), config 2: 128 chunks 906, 256: 845, 512: 745
                                   // generations/s; round 2 (list rebuilt and every row's entries walked by one thread every generation): 256: 370, 512: 457, 1024: 450.
                                   // A row has at most 64 segments: longer rows get larger segments (gev_set_snps).  GEV_SEG_CHUNKS=<power of two>
    bool alias_rows = true;        // crossover-free gametes share their parent's pool row instead of copying it (GEV_ALIAS_ROWS=0: copy every row)
    unsigned long long chunks_written_sum = 0, chunks_total_sum = 0, segments_written_sum = 0, segments_total_sum = 0;   // over all generations and active chromosomes (gev_stitch_totals)
    // Stitch workgroups per CU (8 = every wave slot).  The hardware queue priority does not let the small kernels of the next
    // generation overtake a stitch grid that is still being dispatched: at 8 they start when the stitch is nearly over.  Slots
    // left free (6 of 8 used) let them run next to it, at the price of a slightly slower stitch.  That pays when the small-kernel
    // chain is the longer of the two: long rows (one 1M-SNP chromosome with shared rows: 6 -> +9 % generations/s), not when the
    // stitch dominates anyway (11 chromosomes of 227k SNPs: 8).  Default: by row length; GEV_STITCH_WG_PER_CU=<n> fixes it,
    // =auto measures it (a few generations per candidate, wall time between consecutive gev_reproduce returns).
    int stitch_occ = 0 /* 0 = by row length */, stitch_occ_env = 0; bool stitch_occ_auto = false;
    struct PendingRepro { bool active = false, has_mut = false, pre = false; int pop = 0, attempt = 0; size_t n_people = 0, n_status = 0; u32 seed = 0; u32* hstatus = nullptr; double th0 = 0, th1 = 0, th2 = 0;
                          bool fused = false /* gev_generation_begin: seeds and couples are made on the device */, has_svf = false, pool_rebuilt = false; u32 glob_state = 0; u32* hseeds2 = nullptr; uint8_t* hsex = nullptr; } pend;
    struct OccTune { int phase = 0 /* 0 idle, 1 measuring, 2 settled */, idx = 0, n = 0, best_occ = 8; double last = 0, cur_min = 0, best = 0; size_t people = 0; unsigned age = 0; } tune;
    DevBuf d_snpmajor, d_text;
    DevBuf d_mflag, d_mblk, d_posm, d_posf, d_pickblk, d_couples, d_svf, d_logical, d_globblk, d_mstat;   // gev_random_mate / gev_glob_seeds scratch
    DevBuf d_gef_flag, d_gef_first, d_gef_red, d_gef_io;
    DevBuf d_cvdone;
    DevBuf d_cnt, d_sums, d_map, d_cvm, d_addchr, d_domchr, d_add, d_dom, d_flag, d_stage, d_thr32, d_tmp;
    // per-generation work tables (gev_kernels.h: ChrWork / CvWork / AdWork) are written into a ring of pinned host memory and
    // copied to the device on the stream that uses them
    uint8_t* h_ring = nullptr; size_t h_ring_bytes = 0, h_ring_off = 0;
    DevBuf d_adwork2[2]; std::vector<uint8_t> adwork_shadow[2];
    DevBuf d_lpc_bits, d_lpc_cnt, d_lpc_wpre, d_lpc_len, d_lpc_old, d_lpc_new, d_lpc_tmp;     // scratch of an arena compaction (lp_compact)
    DevBuf d_lp_nitems;                                 // [nchr] length of each chromosome's work list of list pieces to build
    std::map<double, GevThr> thr_cache;
};

// ------------------------------------------------------------------------------------------
static int scan_u32_on(hipStream_t st, DevBuf& sums, const u32* in, size_t n, u32* out /*n+1*/)
{
    const size_t nb = ceil_div(n + 1, SCAN_ITEMS);
    GEVC(sums.ensure(nb * sizeof(u32), st));
    hipLaunchKernelGGL(k_scan_partial, dim3((unsigned)nb), dim3(256), 0, st, in, n, sums.as<u32>());
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, st, sums.as<u32>(), nb);
    hipLaunchKernelGGL(k_scan_final, dim3((unsigned)nb), dim3(256), 0, st, in, n, sums.as<u32>(), out);
    KCHECK();
    return GEV_OK;
}
static int scan_u32(gev_ctx* c, const u32* in, size_t n, u32* out /*n+1*/, u32* total_host)
{
    GEVC(scan_u32_on(c->stream, c->d_sums, in, n, out));
    if (total_host) {
        HIPC(hipMemcpyAsync(total_host, out + n, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
    }
    return GEV_OK;
}
static int h2d(gev_ctx* c, DevBuf& b, const void* src, size_t bytes)
{
    GEVC(b.ensure(std::max<size_t>(bytes, 16), c->stream));
    if (bytes) HIPC(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPC(hipStreamSynchronize(c->stream));          // src may be a caller-owned temporary
    return GEV_OK;
}
static int make_thresholds(gev_ctx* c, const std::vector<double>& p, std::vector<GevThr>& out)
{
    out.resize(p.size());
    for (size_t i = 0; i < p.size(); i++) {
        auto it = c->thr_cache.find(p[i]);
        if (it == c->thr_cache.end()) {
            GevThr t;
            if (p[i] != p[i]) return fail(GEV_EINVAL, "probability of map row %zu is NaN", i);
            if (!gev_make_threshold(p[i], t)) return fail(GEV_EUNSUPPORTED, "threshold window wider than 2 for probability %a", p[i]);
            it = c->thr_cache.emplace(p[i], t).first;
        }
        out[i] = it->second;
    }
    return GEV_OK;
}
static int check_idx(gev_ctx* c, int pop, int chr, int phen = 0)
{
    if (!c) return fail(GEV_EINVAL, "null context");
    if (c->pend.active) return fail(GEV_ESTATE, "a gev_reproduce_begin is pending: call gev_reproduce_end first");
    if (pop < 0 || pop >= c->n_pop) return fail(GEV_EINVAL, "population index %d out of range", pop);
    if (chr < 0 || chr >= c->nchr) return fail(GEV_EINVAL, "chromosome index %d out of range", chr);
    if (phen < 0 || phen >= c->nphen) return fail(GEV_EINVAL, "phenotype index %d out of range", phen);
    return GEV_OK;
}

// data of a chromosome this context does not hold / operations that move whole individuals
static int check_active(gev_ctx* c, int chr, const char* what)
{
    if (!c->chr_active[chr]) return fail(GEV_ESTATE, "%s: chromosome %d is not active on this context (gev_set_chr_active)", what, chr);
    return GEV_OK;
}
static int check_dense(gev_ctx* c, const char* what)
{
    if (!c->dense) return fail(GEV_ESTATE, "%s: this context keeps no resident genotype planes (gev_set_dense_state 0); use gev_materialize", what);
    return GEV_OK;
}
// bit-column permutation of small planes (CV grid): out column j <- in column src_col[j]
__global__ void k_permute_cols(const u32* __restrict__ in, size_t in_w32, u32* __restrict__ out, size_t out_w32, u32 used_w32,
                               const u32* __restrict__ src_col, u32 Cn, size_t n_rows)
{
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_rows * used_w32) return;
    const size_t r = q / used_w32; const u32 w = (u32)(q % used_w32);
    u32 v = 0;
    for (u32 b = 0; b < 32; b++) {
        const u32 j = w * 32 + b;
        if (j >= Cn) break;
        const u32 s = src_col[j];
        v |= ((in[r * in_w32 + (s >> 5)] >> (s & 31)) & 1u) << b;
    }
    out[r * out_w32 + w] = v;
}
__global__ void k_fill_rp(u32* __restrict__ plane, size_t stride_w32, u32 sub_w32, u32 rp_bits, u32 pop, size_t n_rows)
{
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 used = sub_w32 * rp_bits;
    if (q >= n_rows * used) return;
    const size_t r = q / used; const u32 w = (u32)(q % used);
    const u32 b = w / sub_w32;
    plane[r * stride_w32 + sub_w32 + w] = ((pop >> b) & 1u) ? 0xffffffffu : 0u;
}

// ------------------------------------------------------------------------------------------
extern "C" {

const char* gev_last_error(void) { return g_err.c_str(); }
const char* gev_version(void) { return "geneevolve_amd 0.1 (gfx950)"; }

int gev_create(gev_ctx** out, int device, int n_pop, int nchr, int nphen)
{
    if (!out) return fail(GEV_EINVAL, "gev_create: out is null");
    *out = nullptr;
    if (n_pop < 1 || nchr < 1 || nphen < 1) return fail(GEV_EINVAL, "gev_create: n_pop, nchr, nphen must be >= 1");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) return fail(GEV_EDEVICE, "no HIP device available (%s): this library has no CPU fallback", hipGetErrorString(e));
    if (device < 0) HIPC(hipGetDevice(&device));
    if (device >= ndev) return fail(GEV_EDEVICE, "device %d not present (%d devices)", device, ndev);
    HIPC(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPC(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(GEV_EDEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    std::unique_ptr<gev_ctx> c(new gev_ctx());
    c->device = device; c->n_pop = n_pop; c->nchr = nchr; c->nphen = nphen;
    while ((1u << c->rp_bits) < (u32)n_pop) c->rp_bits++;
    // small latency-critical kernels get the high-priority queue, the long HBM-bound stitch the low one:
    // stitch workgroups are short-lived, so freed CU slots go to the small kernels first
    int prio_least = 0, prio_greatest = 0;
    HIPC(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    if (const char* e = getenv("GEV_STITCH_PRIORITY")) {      // experiment knob: 1 = stitch stream high / small streams low, 2 = all equal (default: small high, stitch low)
        if (atoi(e) == 1) std::swap(prio_least, prio_greatest); else if (atoi(e) == 2) prio_least = prio_greatest = (prio_least + prio_greatest) / 2;
    }
    // GEV_STREAM_PRIO: five letters h / m / l for the main, head-start (sampling), mating, list and stitch streams (default hhhhh:
    // the stitch is short -- 0.15 ms at config 2 -- and at equal priority it is through before it is in anybody's way; at low
    // priority it lingered for 0.35 ms next to the latency-bound chain: 1060 against 1370 generations/s)
    const char* pr = getenv("GEV_STREAM_PRIO");
    auto level = [&](int i, int dflt) { if (!pr || strlen(pr) != 5) return dflt; return pr[i] == 'h' ? prio_greatest : (pr[i] == 'l' ? prio_least : (prio_least + prio_greatest) / 2); };
    HIPC(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, level(0, prio_greatest)));
    HIPC(hipStreamCreateWithPriority(&c->stream_samp, hipStreamNonBlocking, level(1, prio_greatest)));
    // (the runtime maps streams of one priority onto four hardware queues: the fifth stream created shares a queue with an earlier one.
    //  GEV_STREAM_ORDER=1 creates the list stream before the mating stream -- an experiment on which two share)
    static const bool swap_order = getenv("GEV_STREAM_ORDER") && atoi(getenv("GEV_STREAM_ORDER")) == 1;
    if (swap_order) HIPC(hipStreamCreateWithPriority(&c->stream_list, hipStreamNonBlocking, level(3, prio_greatest)));
    HIPC(hipStreamCreateWithPriority(&c->stream_aux, hipStreamNonBlocking, level(2, prio_greatest)));
    if (!swap_order) HIPC(hipStreamCreateWithPriority(&c->stream_list, hipStreamNonBlocking, level(3, prio_greatest)));
    for (auto& ev : c->ev) HIPC(hipEventCreate(&ev));
    HIPC(hipStreamCreateWithPriority(&c->stream_big, hipStreamNonBlocking, level(4, prio_greatest)));
    // Events that only order DEVICE work against device work carry no system-scope fence (by default recording an event makes the
    // preceding kernel write the dirty L2 lines back to memory: tens of microseconds behind a kernel that wrote 50 MB, paid by
    // whatever small kernel comes next).  Only the event the host waits on before it reads the results (ev_status) keeps it.
    static const bool sysfence = getenv("GEV_EVENT_SYSFENCE") != nullptr;            // (A/B knob: the default flags everywhere)
    const unsigned dev_only = hipEventDisableTiming | (sysfence ? 0u : hipEventDisableSystemFence), timed = sysfence ? hipEventDefault : hipEventDisableSystemFence;
    HIPC(hipEventCreateWithFlags(&c->ev_planes, dev_only));
    c->chr_active.assign(nchr, 1);
    for (auto& sc : c->sc) {
        HIPC(hipEventCreateWithFlags(&sc.ev_small_done, dev_only));
        HIPC(hipEventCreateWithFlags(&sc.ev_stitch_done, dev_only));
        HIPC(hipEventCreateWithFlags(&sc.ev_status, hipEventDisableTiming));
        HIPC(hipEventCreateWithFlags(&sc.ev_sampled, dev_only)); HIPC(hipEventCreateWithFlags(&sc.ev_seeded, dev_only));
        HIPC(hipEventCreateWithFlags(&sc.ev_fork, dev_only)); HIPC(hipEventCreateWithFlags(&sc.ev_aux, dev_only)); HIPC(hipEventCreateWithFlags(&sc.ev_lists, dev_only));
        HIPC(hipEventCreateWithFlags(&sc.ev_forked, dev_only)); HIPC(hipEventCreateWithFlags(&sc.ev_chain, dev_only)); HIPC(hipEventCreateWithFlags(&sc.ev_tab, dev_only));
        for (auto& e : sc.t) HIPC(hipEventCreateWithFlags(&e, timed));
        if (g_trace_host) for (auto& e : sc.tc) HIPC(hipEventCreateWithFlags(&e, timed));
    }
    c->pop.resize(n_pop);
    for (auto& P : c->pop) {
        P.cs.resize(nchr); P.st.resize(nchr);
        P.cv.resize(nphen); P.cvp.resize(nphen);
        for (int p = 0; p < nphen; p++) { P.cv[p].resize(nchr); P.cvp[p].resize(nchr); }
    }
    GevRngTables T; gev_build_rng_tables(T);
    GEVC(h2d(c.get(), c->d_tables, &T, sizeof T));
    if (const char* e = getenv("GEV_SERIALIZE")) c->overlap_mode = atoi(e) != 0 ? 0 : 1;
    if (const char* e = getenv("GEV_OVERLAP")) { const int v = atoi(e); c->overlap_mode = v < 0 ? -1 : std::min(v, 2); }
    c->serialize = c->overlap_mode <= 0;                             // auto starts serialised
    c->sparse_after_stitch = c->overlap_mode == 2;
    if (const char* e = getenv("GEV_SAMPLE_BATCHED")) c->sample_batched = atoi(e) != 0;
    if (const char* e = getenv("GEV_STITCH_GRID")) c->stitch_grid = (size_t)std::max(1, atoi(e));
    if (const char* e = getenv("GEV_SEG_CHUNKS")) { const int v = atoi(e); u32 sh = 0; while ((1 << (sh + 1)) <= v) sh++; if (v >= 1 && sh <= 20) c->seg_shift = sh; }
    if (const char* e = getenv("GEV_ALIAS_ROWS")) c->alias_rows = atoi(e) != 0;
    if (const char* e = getenv("GEV_STITCH_WAVE_PRIO")) c->stitch_wave_prio = std::max(0, std::min(atoi(e), 3));
    if (const char* e = getenv("GEV_STITCH_START")) c->stitch_start = std::max(0, std::min(atoi(e), 2));
    if (const char* e = getenv("GEV_SIDE_STREAMS")) c->side_streams = atoi(e) != 0;
    if (const char* e = getenv("GEV_HEAD_START")) c->head_start = atoi(e) == 1 ? 1 : 0;
    if (const char* e = getenv("GEV_STITCH_U")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) c->stitch_u = v; }
    if (const char* e = getenv("GEV_CV_THREADS")) { const int v = atoi(e); if (v == 256 || v == 512 || v == 1024) c->cv_threads = (unsigned)v; }
    if (const char* e = getenv("GEV_CV_COUNT_FUSED")) c->cv_count_fused_ok = atoi(e) != 0;
    if (const char* e = getenv("GEV_OVF_CAP")) { const long v = atol(e); if (v >= 1) c->bk_ovf_cap = c->nm_ovf_cap = (size_t)v; }
    if (const char* e = getenv("GEV_STITCH_MODE")) c->stitch_mode = std::max(0, std::min(atoi(e), 1));
    if (const char* e = getenv("GEV_SAMPLE_GRID")) { const int g = atoi(e); if (g >= 1) { c->sample_grid = c->sample_grid_shared = (unsigned)g; c->sample_grid_env = true; } }
    if (const char* e = getenv("GEV_STITCH_WG_PER_CU")) {       // fixed stitch workgroups per CU (default: measured, see OccTune)
        const int occ = atoi(e);
        if (!strcmp(e, "auto")) c->stitch_occ_auto = true;
        else if (occ >= 1 && occ <= 8) c->stitch_occ = c->stitch_occ_env = occ;
    }
    *out = c.release();
    return GEV_OK;
}
void gev_destroy(gev_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); }
    if (c->stream_big) { (void)hipStreamSynchronize(c->stream_big); (void)hipStreamDestroy(c->stream_big); }
    if (c->stream_samp) { (void)hipStreamSynchronize(c->stream_samp); (void)hipStreamDestroy(c->stream_samp); }
    if (c->stream_aux) { (void)hipStreamSynchronize(c->stream_aux); (void)hipStreamDestroy(c->stream_aux); }
    if (c->stream_list) { (void)hipStreamSynchronize(c->stream_list); (void)hipStreamDestroy(c->stream_list); }
    if (g_graveyard.bytes) g_graveyard.drain(c->device, false);
    for (auto& ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    if (c->ev_planes) (void)hipEventDestroy(c->ev_planes);
    for (auto& sc : c->sc) { if (sc.ev_small_done) (void)hipEventDestroy(sc.ev_small_done); if (sc.ev_stitch_done) (void)hipEventDestroy(sc.ev_stitch_done); if (sc.ev_status) (void)hipEventDestroy(sc.ev_status); if (sc.ev_sampled) (void)hipEventDestroy(sc.ev_sampled); if (sc.ev_seeded) (void)hipEventDestroy(sc.ev_seeded); if (sc.ev_fork) (void)hipEventDestroy(sc.ev_fork); if (sc.ev_aux) (void)hipEventDestroy(sc.ev_aux); if (sc.ev_lists) (void)hipEventDestroy(sc.ev_lists); if (sc.ev_forked) (void)hipEventDestroy(sc.ev_forked); if (sc.ev_chain) (void)hipEventDestroy(sc.ev_chain); if (sc.ev_tab) (void)hipEventDestroy(sc.ev_tab); for (auto& e : sc.t) if (e) (void)hipEventDestroy(e); for (auto& e : sc.tc) if (e) (void)hipEventDestroy(e); }
    hipStream_t s = c->stream;
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_seeds) (void)hipHostFree(c->h_seeds);
    for (void* h : c->h_ad2) if (h) (void)hipHostFree(h);
    if (c->h_ring) (void)hipHostFree(c->h_ring);
    delete c;
    if (s) (void)hipStreamDestroy(s);
}

// ---- static inputs -----------------------------------------------------------------------
int gev_set_rmap(gev_ctx* c, int pop, int chr, const u64* bp, const double* prob, size_t R, u64 bp_dist)
{
    GEVC(check_idx(c, pop, chr));
    if (!bp || !prob || R < 2) return fail(GEV_EINVAL, "set_rmap: need >= 2 map rows");
    if (bp_dist == 0) return fail(GEV_EINVAL, "set_rmap: bp_dist_in_rmap is 0 (rand() %% 0 in the reference, src/Simulation.cpp:2990)");
    if (R > 0x7fffffffu) return fail(GEV_EINVAL, "set_rmap: too many rows");
    // dense-equivalence precondition: breakpoints bp[j] + rand()%dist must come out ascending
    for (size_t j = 0; j + 1 < R; j++)
        if (bp[j + 1] < bp[j] + bp_dist) return fail(GEV_EUNSUPPORTED, "set_rmap: map rows %zu,%zu are closer than bp_dist_in_rmap=%llu: the reference would emit unsorted breakpoints (overlapping parts), which has no dense equivalent", j, j + 1, (unsigned long long)bp_dist);
    ChrStatic& S = c->pop[pop].cs[chr];
    S.rbp.assign(bp, bp + R); S.rprob.assign(prob, prob + R); S.bp_dist = bp_dist;
    std::vector<GevThr> thr;
    GEVC(make_thresholds(c, S.rprob, thr));
    S.r_amax = 0; for (const GevThr& t : thr) S.r_amax = std::max(S.r_amax, t.a_hi);
    HIPC(hipSetDevice(c->device));
    GEVC(h2d(c, S.d_rthr, thr.data(), R * sizeof(GevThr)));
    GEVC(h2d(c, S.d_rbp, bp, R * sizeof(u64)));
    c->pop[pop].finalized = false;
    return GEV_OK;
}
int gev_set_mutmap(gev_ctx* c, int pop, int chr, const u64* bp, const double* rate, size_t M)
{
    GEVC(check_idx(c, pop, chr));
    if (M && (!bp || !rate)) return fail(GEV_EINVAL, "set_mutmap: null arrays");
    if (M > 0x7fffffffu) return fail(GEV_EINVAL, "set_mutmap: too many rows");
    for (size_t i = 1; i < M; i++) {
        if (bp[i] < bp[i - 1]) return fail(GEV_EUNSUPPORTED, "set_mutmap: bp must be non-decreasing (row %zu)", i);
        if (bp[i] - bp[i - 1] >= 2147483645ull) return fail(GEV_EUNSUPPORTED, "set_mutmap: interval %zu spans >= 2^31 bp (uniform_int_distribution up-scaling branch)", i);
    }
    ChrStatic& S = c->pop[pop].cs[chr];
    S.mbp.assign(bp, bp + M); S.mrate.assign(rate, rate + M); S.mut_set = true;
    std::vector<GevThr> thr;
    GEVC(make_thresholds(c, S.mrate, thr));
    S.m_amax = 0; for (size_t i = 1; i < M; i++) S.m_amax = std::max(S.m_amax, thr[i].a_hi);
    HIPC(hipSetDevice(c->device));
    GEVC(h2d(c, S.d_mthr, thr.data(), M * sizeof(GevThr)));
    GEVC(h2d(c, S.d_mbp, bp, M * sizeof(u64)));
    c->pop[pop].finalized = false;
    return GEV_OK;
}
int gev_set_snps(gev_ctx* c, int pop, int chr, const u64* pos, size_t L)
{
    GEVC(check_idx(c, pop, chr));
    if (L && !pos) return fail(GEV_EINVAL, "set_snps: null positions");
    if (L >= 0xffffff00ull) return fail(GEV_EINVAL, "set_snps: too many loci");
    for (size_t i = 1; i < L; i++) if (pos[i] < pos[i - 1]) return fail(GEV_EUNSUPPORTED, "set_snps: positions must be non-decreasing (locus %zu)", i);
    ChrStatic& S = c->pop[pop].cs[chr];
    S.pos.assign(pos, pos + L); S.L = L;
    S.panel_rows = 0;                                              // a founder panel kept for another grid is void
    S.stride = std::max<size_t>(round_up(ceil_div(L, 8), 128), 128);
    S.seg_shift = c->seg_shift;                                    // 2 KiB segments (GEV_SEG_CHUNKS) unless the row would need more than 64 of them
    while (ceil_div(S.stride / 16, (size_t)1 << S.seg_shift) > POOL_SEG_MAX) S.seg_shift++;
    while (S.seg_shift > 0 && ((size_t)1 << (S.seg_shift - 1)) >= S.stride / 16) S.seg_shift--;     // a row shorter than a segment: the smallest power-of-two unit that holds it
    S.nseg = (u32)ceil_div(S.stride / 16, (size_t)1 << S.seg_shift);
    HIPC(hipSetDevice(c->device));
    GEVC(h2d(c, S.d_pos, pos, L * sizeof(u64)));
    c->pop[pop].finalized = false;
    return GEV_OK;
}
int gev_set_cvs(gev_ctx* c, int pop, int phen, int chr, const u64* bp, const double* a, const double* d, size_t C, double vd)
{
    GEVC(check_idx(c, pop, chr, phen));
    if (C && (!bp || !a || !d)) return fail(GEV_EINVAL, "set_cvs: null arrays");
    if (C > 0x7fffff00u) return fail(GEV_EINVAL, "set_cvs: too many CVs");
    CvStatic& V = c->pop[pop].cv[phen][chr];
    V.bp.assign(bp, bp + C); V.a.assign(a, a + C); V.d.assign(d, d + C); V.vd = vd; V.C = (u32)C; V.set = true;
    // sorted column order (stable: equal positions keep file order)
    V.icv_of_col.resize(C);
    for (u32 i = 0; i < C; i++) V.icv_of_col[i] = i;
    std::stable_sort(V.icv_of_col.begin(), V.icv_of_col.end(), [&](u32 x, u32 y) { return V.bp[x] < V.bp[y]; });
    V.col_of_icv.resize(C);
    for (u32 j = 0; j < C; j++) V.col_of_icv[V.icv_of_col[j]] = j;
    V.cols_sorted = true;
    for (u32 j = 0; j < C; j++) V.cols_sorted &= V.col_of_icv[j] == j;
    std::vector<u64> sorted(C);
    for (u32 j = 0; j < C; j++) sorted[j] = V.bp[V.icv_of_col[j]];
    V.sub_w32 = (u32)std::max<size_t>(ceil_div(C, 32), 1);
    V.stride_w32 = (u32)round_up((size_t)V.sub_w32 * (1 + c->rp_bits), 4);      // sub-rows: resolved alleles, root-population bits
    HIPC(hipSetDevice(c->device));
    GEVC(h2d(c, V.d_pos_sorted, sorted.data(), C * sizeof(u64)));
    GEVC(h2d(c, V.d_pos_file, V.bp.data(), C * sizeof(u64)));
    GEVC(h2d(c, V.d_col_of_icv, V.col_of_icv.data(), C * sizeof(u32)));
    GEVC(h2d(c, V.d_icv_of_col, V.icv_of_col.data(), C * sizeof(u32)));
    GEVC(h2d(c, V.d_a, a, C * sizeof(double)));
    GEVC(h2d(c, V.d_d, d, C * sizeof(double)));
    GEVC(V.d_frq.ensure(std::max<size_t>(C, 1) * sizeof(double), c->stream));
    GEVC(V.d_counts.ensure(std::max<size_t>(C, 1) * sizeof(u32), c->stream));
    HIPC(hipMemsetAsync(V.d_counts.p, 0, std::max<size_t>(C, 1) * sizeof(u32), c->stream));     // k_cv_count adds into them, k_cv_table / k_cv_freq consume and clear them
    V.frq_valid = false;
    c->ad_cached_pop = c->ad_host_set_pop = -1;
    c->pop[pop].finalized = false;
    return GEV_OK;
}

static int wait_planes(gev_ctx* c);
// pool descriptor of (population, chromosome); `alt` = the phys buffer a new generation / new slots are written to
static PoolWork pool_work(const gev_ctx* c, PopState& P, int chr, int alt)
{
    ChrState& cs = P.st[chr];
    PoolWork pw{};
    const ChrStatic& S = P.cs[chr];
    pw.pool = cs.pool.as<uint8_t>(); pw.phys_cur = cs.phys[P.pcur].as<u32>(); pw.phys_alt = cs.phys[alt].as<u32>();
    pw.live = cs.live.as<u32>(); pw.freel = cs.freel.as<u32>(); pw.pctr = cs.pctr.as<u32>();
    pw.nseg = S.nseg; pw.seg_shift = S.seg_shift;
    pw.pool_units = (u32)(4 * P.cap_people * S.nseg); pw.alias = c->alias_rows ? 1u : 0u;
    pw.items = cs.items[P.cur ^ 1].as<StitchItem>(); pw.items_cap = (u32)(2 * P.cap_people * S.nseg);
    pw.stamp = cs.pool_stamp;                             // (pool_new_stamp before a rebuild of the free list)
    return pw;
}
static void pool_new_stamp(ChrState& cs) { if (++cs.pool_stamp == 0) cs.pool_stamp = 1; }   // (a stale mark of 2^32 rebuilds ago could only keep a free unit out of one free list)
// read access to the current generation's rows / a flat buffer of whole rows, for the kernels that move or read rows
static RowMap row_map(const PopState& P, int chr)
{
    const ChrStatic& S = P.cs[chr]; const ChrState& cs = P.st[chr];
    return RowMap{cs.pool.as<uint8_t>(), cs.phys[P.pcur].as<u32>(), S.nseg, S.seg_shift};
}
static RowRef pool_rows(const PopState& P, int chr, const u32* table)
{
    const ChrStatic& S = P.cs[chr];
    return RowRef{P.st[chr].pool.as<uint8_t>(), table, 0, S.nseg, S.seg_shift};
}
static RowRef flat_rows(void* base, size_t stride_bytes) { return RowRef{(uint8_t*)base, nullptr, stride_bytes / 16, 1, 0}; }
// free rows of the pool given the slots that are in use (the kernels of gev_reproduce do the same from the work table)
static int pool_free_list(const PoolWork& pw, size_t n_slots, hipStream_t st)
{
    const unsigned blocks = (unsigned)std::min<size_t>(ceil_div(std::max<size_t>(pw.pool_units, 1), 256), 1024);
    hipLaunchKernelGGL(k_pool_mark, dim3(blocks), dim3(256), 0, st, pw, n_slots);
    hipLaunchKernelGGL(k_pool_collect, dim3(blocks), dim3(256), 0, st, pw);
    KCHECK();
    return GEV_OK;
}
// fresh rows for slots [slot0, slot0 + n) of pw.phys_alt (synchronous: migration / order-restoring paths only)
static int pool_take(const PoolWork& pw, size_t slot0, size_t n, hipStream_t st)
{
    if (!n) return GEV_OK;
    hipLaunchKernelGGL(k_pool_take, dim3((unsigned)ceil_div(n * pw.nseg, 256)), dim3(256), 0, st, pw, slot0, n, pw.pctr + 2);
    hipLaunchKernelGGL(k_pool_taken, dim3(1), dim3(64), 0, st, pw, (u32)(n * pw.nseg));
    KCHECK();
    u32 flag = 0;
    HIPC(hipMemcpyAsync(&flag, pw.pctr + 2, sizeof(u32), hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    if (flag) return fail(GEV_EDEVICE, "genotype row pool exhausted (internal error)");
    return GEV_OK;
}
static int ensure_capacity(gev_ctx* c, int pop, size_t people)
{
    PopState& P = c->pop[pop];
    if (people <= P.cap_people) return GEV_OK;
    const size_t rows = 2 * people;
    // (row, position range) pairs of the list pieces are 32-bit work items (gev_lists.h): 2^27 haplotype rows per population
    if (rows * LP_MAXSEG > 0xffffffffull) return fail(GEV_EINVAL, "population %d: %zu individuals are more than this library holds (%llu)", pop, people, 0xffffffffull / LP_MAXSEG / 2);
    // Growth copies the CURRENT buffers (keep = true) on `stream`; the stitch that produces the current planes may still be
    // running on stream_big (gev_reproduce does not wait for it), so order the copies behind it -- otherwise the new buffer
    // would receive a half-written generation.
    GEVC(wait_planes(c));
    for (int k = 0; k < c->nchr; k++) {
        if (!c->chr_active[k]) continue;
        if (!P.cs[k].stride) return fail(GEV_ESTATE, "set_snps must precede allocation (pop %d chr %d)", pop, k);
        if (c->dense) {
            ChrState& cs = P.st[k];
            const size_t units = rows * P.cs[k].nseg;                                          // per generation
            GEVC(cs.pool.ensure(2 * units * P.cs[k].unit_bytes(), c->stream, /*keep=*/true));   // unit numbers stay valid: the pool grows at its end
            for (int b = 0; b < 3; b++) GEVC(cs.phys[b].ensure(units * sizeof(u32), c->stream, b == P.pcur));
            for (int b = 0; b < 2; b++) GEVC(cs.items[b].ensure((units + 1) * sizeof(StitchItem), c->stream));
            GEVC(cs.live.ensure(2 * units * sizeof(u32), c->stream)); GEVC(cs.freel.ensure(2 * units * sizeof(u32), c->stream));
            HIPC(hipMemsetAsync(cs.live.p, 0, cs.live.bytes, c->stream)); cs.pool_stamp = 0;   // marks are generation stamps: start from a clean buffer
            GEVC(cs.pctr.ensure(8 * sizeof(u32), c->stream));
            cs.pool_list_valid = false;                                                         // the pool grew: its new units are in no free list yet
        }
        for (int b = 0; b < 2; b++) {
            GEVC(P.st[k].moff[b].ensure((rows + 1) * sizeof(u32), c->stream, b == P.cur));
            GEVC(P.st[k].poff[b].ensure((rows + 1) * sizeof(u32), c->stream, b == P.cur));
        }
        for (int p = 0; p < c->nphen; p++) {
            if (!P.cv[p][k].set) return fail(GEV_ESTATE, "set_cvs must precede allocation (pop %d phen %d chr %d)", pop, p, k);
            for (int b = 0; b < 2; b++)
                GEVC(P.cvp[p][k][b].ensure(rows * P.cv[p][k].stride_w32 * sizeof(u32), c->stream, b == P.cur));
        }
    }
    for (int b = 0; b < 2; b++) GEVC(P.d_sex[b].ensure(people, c->stream, b == P.cur));
    P.cap_people = people;
    return GEV_OK;
}
// Locus-split populations (BASELINE config 4: 125k individuals x 5M loci do not fit one GPU twice): a context can hold the
// genotype / CV / list state of a subset of the chromosomes only.  The rand() seed chain of Simulation::reproduce runs over
// ALL (offspring, chromosome) tasks in order (:2447-2501), so every context still evaluates the mutation-sampling chain of
// every task (maps of all chromosomes are required) but samples crossovers, tracks lists, stitches planes and accumulates
// A/D for its active chromosomes only; gev_compute_ad returns exact zeros for the others.  Needs a mutation map (without
// one the chain also depends on every crossover count: then all sampling runs everywhere).
int gev_set_chr_active(gev_ctx* c, int chr, int active)
{
    GEVC(check_idx(c, 0, chr));
    for (auto& P : c->pop) if (P.gen0) return fail(GEV_ESTATE, "set_chr_active: must precede gev_init_gen0");
    c->chr_active[chr] = active ? 1 : 0;
    c->any_inactive = false;
    for (uint8_t a : c->chr_active) c->any_inactive |= !a;
    for (auto& P : c->pop) P.finalized = false;
    return GEV_OK;
}
// Populations whose genotype matrix cannot be resident (BASELINE config 5: 1M individuals x 10M loci): keep the reference's
// own state only -- ancestry intervals + mutation sets (and the small CV planes A/D needs) -- and assemble genotype tiles
// on demand with gev_materialize.  Must precede gev_init_gen0; founder SNP panels are not uploaded in this mode.
int gev_set_dense_state(gev_ctx* c, int on)
{
    if (!c) return fail(GEV_EINVAL, "null context");
    if (c->pend.active) return fail(GEV_ESTATE, "a gev_reproduce_begin is pending: call gev_reproduce_end first");
    for (auto& P : c->pop) if (P.gen0) return fail(GEV_ESTATE, "set_dense_state: must precede gev_init_gen0");
    c->dense = on != 0;
    if (!c->dense) c->track_intervals = true;
    return GEV_OK;
}
int gev_reserve(gev_ctx* c, int pop, size_t max_people)
{
    GEVC(check_idx(c, pop, 0));
    HIPC(hipSetDevice(c->device));
    return ensure_capacity(c, pop, max_people);
}

int gev_upload_founders(gev_ctx* c, int pop, int chr, const u64* bits, size_t row_stride_words, size_t nhap, size_t L)
{
    if (c) GEVC(check_dense(c, "upload_founders"));
    GEVC(check_idx(c, pop, chr));
    PopState& P = c->pop[pop]; ChrStatic& S = P.cs[chr];
    if (!bits || nhap < 2 || (nhap & 1)) return fail(GEV_EINVAL, "upload_founders: need an even number (>=2) of haplotype rows");
    if (L != S.L) return fail(GEV_EINVAL, "upload_founders: L=%zu but set_snps gave %zu loci", L, S.L);
    if (row_stride_words * 64 < L) return fail(GEV_EINVAL, "upload_founders: row stride too small");
    HIPC(hipSetDevice(c->device));
    GEVC(gev_sync(c));
    GEVC(P.st[chr].pool.ensure(nhap * S.pitch(), c->stream));                       // founder haplotype r = units r*nseg ..: flat rows of pitch()
    HIPC(hipMemsetAsync(P.st[chr].pool.p, 0, nhap * S.pitch(), c->stream));
    HIPC(hipMemcpy2DAsync(P.st[chr].pool.p, S.pitch(), bits, row_stride_words * 8, ceil_div(L, 8), nhap, hipMemcpyHostToDevice, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    if (L % 8) {   // clear pad bits of the last byte (the contract says pad bits are zero; do not trust it)
        hipLaunchKernelGGL(k_mask_rows, dim3((unsigned)ceil_div(nhap * (S.pitch() / 4), 256)), dim3(256), 0, c->stream,
                           P.st[chr].pool.as<u32>(), S.pitch() / 4, nhap, 0u, (u32)L);
        KCHECK();
    }
    S.founder_rows = nhap; P.gen0 = false;
    return GEV_OK;
}
int gev_synth_founders(gev_ctx* c, int pop, int chr, size_t nhap, u64 seed)
{
    if (c) GEVC(check_dense(c, "synth_founders"));
    GEVC(check_idx(c, pop, chr));
    PopState& P = c->pop[pop]; ChrStatic& S = P.cs[chr];
    if (nhap < 2 || (nhap & 1)) return fail(GEV_EINVAL, "synth_founders: need an even number of haplotypes");
    if (!S.L) return fail(GEV_ESTATE, "synth_founders: set_snps first");
    HIPC(hipSetDevice(c->device));
    GEVC(gev_sync(c));
    GEVC(P.st[chr].pool.ensure(nhap * S.pitch(), c->stream));
    HIPC(hipMemsetAsync(P.st[chr].pool.p, 0, nhap * S.pitch(), c->stream));
    GEVC(c->d_thr32.ensure(S.L * sizeof(u32), c->stream));
    hipLaunchKernelGGL(k_synth_thresholds, dim3((unsigned)ceil_div(S.L, 256)), dim3(256), 0, c->stream, c->d_thr32.as<u32>(), S.L, seed);
    const size_t words = ceil_div(S.L, 64);
    hipLaunchKernelGGL(k_synth_rows, dim3((unsigned)ceil_div(nhap * words, 256)), dim3(256), 0, c->stream,
                       P.st[chr].pool.as<u64>(), S.pitch() / 8, nhap, S.L, c->d_thr32.as<u32>(), seed);
    KCHECK();
    HIPC(hipStreamSynchronize(c->stream));
    S.founder_rows = nhap; P.gen0 = false;
    return GEV_OK;
}
// Founder panels kept for good (read-only, flat rows of S.stride bytes), one per ROOT population x chromosome: what
// Simulation::ras_convert_interval_to_hap_matrix reads as pops[root_population].hap_snps (src/Simulation.cpp:1204).  With them a
// context can rebuild an immigrant's genotype row from its ancestry intervals, so the migration payload carries lists only
// (gev_set_migrant_rows).  `pop` is the root population; it need not be a population this context simulates.
static int panel_prepare(gev_ctx* c, int pop, int chr, size_t nhap, const char* who)
{
    if (c) GEVC(check_dense(c, who));
    GEVC(check_idx(c, pop, chr));
    ChrStatic& S = c->pop[pop].cs[chr];
    if (nhap < 2 || (nhap & 1)) return fail(GEV_EINVAL, "%s: need an even number (>=2) of haplotype rows", who);
    if (!S.L || !S.stride) return fail(GEV_ESTATE, "%s: set_snps first", who);
    HIPC(hipSetDevice(c->device));
    GEVC(gev_sync(c));
    GEVC(S.panel.ensure(nhap * S.stride, c->stream));
    HIPC(hipMemsetAsync(S.panel.p, 0, nhap * S.stride, c->stream));
    return GEV_OK;
}
int gev_upload_founder_panel(gev_ctx* c, int pop, int chr, const u64* bits, size_t row_stride_words, size_t nhap, size_t L)
{
    GEVC(panel_prepare(c, pop, chr, nhap, "upload_founder_panel"));
    ChrStatic& S = c->pop[pop].cs[chr];
    if (!bits) return fail(GEV_EINVAL, "upload_founder_panel: null bits");
    if (L != S.L) return fail(GEV_EINVAL, "upload_founder_panel: L=%zu but set_snps gave %zu loci", L, S.L);
    if (row_stride_words * 64 < L) return fail(GEV_EINVAL, "upload_founder_panel: row stride too small");
    HIPC(hipMemcpy2DAsync(S.panel.p, S.stride, bits, row_stride_words * 8, ceil_div(L, 8), nhap, hipMemcpyHostToDevice, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    if (L % 8) {   // pad bits of the last byte
        hipLaunchKernelGGL(k_mask_rows, dim3((unsigned)ceil_div(nhap * (S.stride / 4), 256)), dim3(256), 0, c->stream, S.panel.as<u32>(), S.stride / 4, nhap, 0u, (u32)L);
        KCHECK();
        HIPC(hipStreamSynchronize(c->stream));
    }
    S.panel_rows = nhap;
    return GEV_OK;
}
int gev_synth_founder_panel(gev_ctx* c, int pop, int chr, size_t nhap, u64 seed)
{
    GEVC(panel_prepare(c, pop, chr, nhap, "synth_founder_panel"));
    ChrStatic& S = c->pop[pop].cs[chr];
    GEVC(c->d_thr32.ensure(S.L * sizeof(u32), c->stream));
    hipLaunchKernelGGL(k_synth_thresholds, dim3((unsigned)ceil_div(S.L, 256)), dim3(256), 0, c->stream, c->d_thr32.as<u32>(), S.L, seed);
    hipLaunchKernelGGL(k_synth_rows, dim3((unsigned)ceil_div(nhap * ceil_div(S.L, 64), 256)), dim3(256), 0, c->stream,
                       S.panel.as<u64>(), S.stride / 8, nhap, S.L, c->d_thr32.as<u32>(), seed);      // the same generator as gev_synth_founders
    KCHECK();
    HIPC(hipStreamSynchronize(c->stream));
    S.panel_rows = nhap;
    return GEV_OK;
}
// 1 (default): gev_export_rows packs the migrants' genotype rows; 0: it packs none and gev_import_rows rebuilds them from the
// founder panels.  Both ends of an exchange must use the same setting (the payload's size gives a mismatch away).
int gev_set_migrant_rows(gev_ctx* c, int on)
{
    if (!c) return fail(GEV_EINVAL, "null context");
    if (c->pend.active) return fail(GEV_ESTATE, "a gev_reproduce_begin is pending: call gev_reproduce_end first");
    c->migrant_rows = on != 0;
    return GEV_OK;
}
// CV founders arrive in FILE column order; the plane keeps columns sorted by position
static int cv_founders_from_tmp(gev_ctx* c, int pop, int phen, int chr, size_t nhap, size_t tmp_w32)
{
    PopState& P = c->pop[pop]; CvStatic& V = P.cv[phen][chr];
    GEVC(P.cvp[phen][chr][P.cur].ensure(nhap * V.stride_w32 * sizeof(u32), c->stream));
    HIPC(hipMemsetAsync(P.cvp[phen][chr][P.cur].p, 0, nhap * V.stride_w32 * sizeof(u32), c->stream));
    if (V.C) {
        hipLaunchKernelGGL(k_permute_cols, dim3((unsigned)ceil_div(nhap * V.sub_w32, 256)), dim3(256), 0, c->stream,
                           c->d_tmp.as<u32>(), tmp_w32, P.cvp[phen][chr][P.cur].as<u32>(), (size_t)V.stride_w32, V.sub_w32,
                           V.d_icv_of_col.as<u32>(), V.C, nhap);
        KCHECK();
    }
    HIPC(hipStreamSynchronize(c->stream));
    c->ad_cached_pop = c->ad_host_set_pop = -1;
    V.founder_rows = nhap; P.gen0 = false;
    return GEV_OK;
}
int gev_upload_cv_founders(gev_ctx* c, int pop, int phen, int chr, const u64* bits, size_t row_stride_words, size_t nhap, size_t C)
{
    GEVC(check_idx(c, pop, chr, phen));
    CvStatic& V = c->pop[pop].cv[phen][chr];
    if (!V.set) return fail(GEV_ESTATE, "upload_cv_founders: set_cvs first");
    if (C != V.C) return fail(GEV_EINVAL, "upload_cv_founders: C=%zu but set_cvs gave %u", C, V.C);
    if (!bits || nhap < 2 || (nhap & 1) || row_stride_words * 64 < C) return fail(GEV_EINVAL, "upload_cv_founders: bad arguments");
    HIPC(hipSetDevice(c->device));
    GEVC(h2d(c, c->d_tmp, bits, nhap * row_stride_words * 8));
    return cv_founders_from_tmp(c, pop, phen, chr, nhap, row_stride_words * 2);
}
int gev_synth_cv_founders(gev_ctx* c, int pop, int phen, int chr, size_t nhap, u64 seed)
{
    GEVC(check_idx(c, pop, chr, phen));
    CvStatic& V = c->pop[pop].cv[phen][chr];
    if (!V.set) return fail(GEV_ESTATE, "synth_cv_founders: set_cvs first");
    if (nhap < 2 || (nhap & 1)) return fail(GEV_EINVAL, "synth_cv_founders: need an even number of haplotypes");
    HIPC(hipSetDevice(c->device));
    const size_t w64 = std::max<size_t>(ceil_div(V.C, 64), 1);
    GEVC(c->d_tmp.ensure(nhap * w64 * 8, c->stream));
    GEVC(c->d_thr32.ensure(std::max<size_t>(V.C, 1) * sizeof(u32), c->stream));
    HIPC(hipMemsetAsync(c->d_tmp.p, 0, nhap * w64 * 8, c->stream));
    if (V.C) {
        hipLaunchKernelGGL(k_synth_thresholds, dim3((unsigned)ceil_div(V.C, 256)), dim3(256), 0, c->stream, c->d_thr32.as<u32>(), (size_t)V.C, seed);
        hipLaunchKernelGGL(k_synth_rows, dim3((unsigned)ceil_div(nhap * w64, 256)), dim3(256), 0, c->stream,
                           c->d_tmp.as<u64>(), w64, nhap, (size_t)V.C, c->d_thr32.as<u32>(), seed);
        KCHECK();
    }
    return cv_founders_from_tmp(c, pop, phen, chr, nhap, w64 * 2);
}

// static tables that depend on several setters
static int finalize_static(gev_ctx* c, int pop)
{
    PopState& P = c->pop[pop];
    if (P.finalized) return GEV_OK;
    for (auto& sc : c->sc) { sc.presampled = false; if (sc.fused_ahead) { sc.fused_ahead = false; sc.fa_dropped = true; } }   // maps / grids changed: a head start sampled with the old ones is void
    std::vector<ChrDev> cd(c->nchr);
    for (int k = 0; k < c->nchr; k++) {
        ChrStatic& S = P.cs[k];
        if (S.rbp.empty()) return fail(GEV_ESTATE, "population %d chromosome %d: set_rmap missing", pop, k);
        const u64 bp0 = S.rbp.front(), bpe = S.rbp.back();
        S.idx_lo = (u32)(std::lower_bound(S.pos.begin(), S.pos.end(), bp0) - S.pos.begin());
        S.idx_hi = (u32)(std::lower_bound(S.pos.begin(), S.pos.end(), bpe) - S.pos.begin());
        // coarse index over the SNP positions: buckets of 2^shift base pairs holding about eight loci each (at most 2^21 buckets)
        u64 cbase = 0; u32 cshift = 0, cn = 0;
        if (S.L) {
            cbase = S.pos.front();
            const u64 span = S.pos.back() - cbase + 1;
            while (cshift < 63 && ((span >> cshift) > std::max<u64>(S.L / 8, 1) || (span >> cshift) > (1ull << 21))) cshift++;
            cn = (u32)((span >> cshift) + 1);
            std::vector<u32> coarse(cn + 1);
            size_t at = 0;
            for (u32 j = 0; j <= cn; j++) {                      // coarse[j] = lower_bound(pos, cbase + (j << cshift)): one sweep
                const u64 x = cbase + ((u64)j << cshift);
                while (at < S.L && S.pos[at] < x) at++;
                coarse[j] = (u32)at;
            }
            GEVC(h2d(c, S.d_coarse, coarse.data(), coarse.size() * sizeof(u32)));
        }
        cd[k] = ChrDev{S.d_rthr.as<GevThr>(), S.d_rbp.as<u64>(), S.d_mthr.as<GevThr>(), S.d_mbp.as<u64>(), S.bp_dist, bp0, bpe,
                       (u32)S.rbp.size(), (u32)S.mbp.size(), S.r_amax, S.m_amax, S.d_pos.as<u64>(), (u32)S.L, (u32)c->chr_active[k],
                       S.d_coarse.as<u32>(), cbase, cshift, cn};
        for (int p = 0; p < c->nphen; p++) {
            CvStatic& V = P.cv[p][k];
            if (!V.set) continue;
            std::vector<u64> sorted(V.C);
            for (u32 j = 0; j < V.C; j++) sorted[j] = V.bp[V.icv_of_col[j]];
            V.idx_lo = (u32)(std::lower_bound(sorted.begin(), sorted.end(), bp0) - sorted.begin());
            V.idx_hi = (u32)(std::lower_bound(sorted.begin(), sorted.end(), bpe) - sorted.begin());
        }
    }
    GEVC(h2d(c, P.d_chrdev, cd.data(), cd.size() * sizeof(ChrDev)));
    P.finalized = true;
    return GEV_OK;
}
// with several populations the dense state needs one shared coordinate system
static int check_multipop(gev_ctx* c)
{
    for (int pop = 1; pop < c->n_pop; pop++)
        for (int k = 0; k < c->nchr; k++) {
            if (!c->chr_active[k]) continue;
            const ChrStatic& A = c->pop[0].cs[k]; const ChrStatic& B = c->pop[pop].cs[k];
            if (A.pos != B.pos) return fail(GEV_EUNSUPPORTED, "populations 0 and %d have different SNP grids on chromosome %d", pop, k);
            if (A.rbp.front() != B.rbp.front() || A.rbp.back() != B.rbp.back()) return fail(GEV_EUNSUPPORTED, "populations 0 and %d have different map ranges on chromosome %d", pop, k);
            for (int p = 0; p < c->nphen; p++)
                if (c->pop[0].cv[p][k].bp != c->pop[pop].cv[p][k].bp) return fail(GEV_EUNSUPPORTED, "populations 0 and %d have different CV positions (phenotype %d chromosome %d)", pop, p, k);
        }
    // Root populations whose CV effects are the same arrays bit for bit (the usual multi-population run: one CV file for all): the mean
    // of two root populations' values (a0 + a1) / 2 (:2695-2696) is then (a + a) / 2 whatever the roots are, so the one-population
    // term table serves every haplotype and A/D need not look the root-population bits up
    c->ad_effects_shared = true;
    for (int pop = 1; pop < c->n_pop && c->ad_effects_shared; pop++)
        for (int p = 0; p < c->nphen && c->ad_effects_shared; p++)
            for (int k = 0; k < c->nchr; k++) {
                if (!c->chr_active[k]) continue;
                const CvStatic& A = c->pop[0].cv[p][k]; const CvStatic& B = c->pop[pop].cv[p][k];
                if (A.a.size() != B.a.size() || A.d.size() != B.d.size() || memcmp(&A.vd, &B.vd, sizeof(double)) ||
                    (A.a.size() && memcmp(A.a.data(), B.a.data(), A.a.size() * sizeof(double))) || (A.d.size() && memcmp(A.d.data(), B.d.data(), A.d.size() * sizeof(double)))) { c->ad_effects_shared = false; break; }
            }
    static const bool no_shared = getenv("GEV_AD_SHARED") && atoi(getenv("GEV_AD_SHARED")) == 0;      // (tests: force the per-haplotype lookup)
    if (no_shared) c->ad_effects_shared = false;
    // per (phen, chr): table of every population's a[] and d[] device arrays, stored with each population
    for (int pop = 0; pop < c->n_pop; pop++)
        for (int p = 0; p < c->nphen; p++)
            for (int k = 0; k < c->nchr; k++) {
                std::vector<const double*> ap(c->n_pop), dp(c->n_pop);
                for (int r = 0; r < c->n_pop; r++) { ap[r] = c->pop[r].cv[p][k].d_a.as<double>(); dp[r] = c->pop[r].cv[p][k].d_d.as<double>(); }
                GEVC(h2d(c, c->pop[pop].cv[p][k].d_aptr, ap.data(), ap.size() * sizeof(double*)));
                GEVC(h2d(c, c->pop[pop].cv[p][k].d_dptr, dp.data(), dp.size() * sizeof(double*)));
            }
    return GEV_OK;
}

// ---- Simulation::ras_initial_human_gen0 ---------------------------------------------------
int gev_init_gen0(gev_ctx* c, int pop, size_t n_people, uint32_t seed_gen0, uint8_t* sex_out)
{
    GEVC(check_idx(c, pop, 0));
    if (n_people < 1) return fail(GEV_EINVAL, "init_gen0: n_people must be >= 1");
    if (2 * n_people >= 0xffffffffull) return fail(GEV_EINVAL, "init_gen0: too many people");
    HIPC(hipSetDevice(c->device));
    GEVC(gev_sync(c));
    PopState& P = c->pop[pop];
    GEVC(finalize_static(c, pop));
    const size_t rows = 2 * n_people;
    for (int k = 0; k < c->nchr; k++) {
        if (!c->chr_active[k]) continue;
        if (c->dense && P.cs[k].founder_rows < rows) return fail(GEV_ESTATE, "init_gen0: population %d chromosome %d has %zu founder haplotypes, %zu needed", pop, k, P.cs[k].founder_rows, rows);
        for (int p = 0; p < c->nphen; p++)
            if (P.cv[p][k].founder_rows < rows) return fail(GEV_ESTATE, "init_gen0: population %d phenotype %d chromosome %d has %zu CV founder haplotypes, %zu needed", pop, p, k, P.cv[p][k].founder_rows, rows);
    }
    GEVC(ensure_capacity(c, pop, std::max(n_people, P.cap_people)));
    for (int k = 0; k < c->nchr; k++) {
        if (!c->chr_active[k]) continue;
        ChrStatic& S = P.cs[k]; ChrState& st = P.st[k];
        if (c->dense) {                                  // founder haplotype r is pool row r
            hipLaunchKernelGGL(k_mask_rows, dim3((unsigned)ceil_div(rows * (S.pitch() / 4), 256)), dim3(256), 0, c->stream,
                               st.pool.as<u32>(), S.pitch() / 4, rows, S.idx_lo, S.idx_hi);
            hipLaunchKernelGGL(k_iota_u32, dim3((unsigned)ceil_div(rows * S.nseg, 256)), dim3(256), 0, c->stream, st.phys[P.pcur].as<u32>(), rows * S.nseg);
        }
        HIPC(hipMemsetAsync(st.moff[P.cur].p, 0, (rows + 1) * sizeof(u32), c->stream));
        GEVC(st.parts[P.cur].ensure(rows * sizeof(gev_part), c->stream));
        hipLaunchKernelGGL(k_init_parts, dim3((unsigned)ceil_div(rows + 1, 256)), dim3(256), 0, c->stream,
                           st.parts[P.cur].as<gev_part>(), st.poff[P.cur].as<u32>(), rows, S.rbp.front(), S.rbp.back(), pop);
        for (int p = 0; p < c->nphen; p++) {
            CvStatic& V = P.cv[p][k];
            hipLaunchKernelGGL(k_mask_rows, dim3((unsigned)ceil_div(rows * V.stride_w32, 256)), dim3(256), 0, c->stream,
                               P.cvp[p][k][P.cur].as<u32>(), (size_t)V.stride_w32, rows, V.idx_lo, V.idx_hi);
            // k_mask_rows zeroed the (still empty) root-population sub-rows too; write them now
            if (c->rp_bits)
                hipLaunchKernelGGL(k_fill_rp, dim3((unsigned)ceil_div(rows * V.sub_w32 * c->rp_bits, 256)), dim3(256), 0, c->stream,
                                   P.cvp[p][k][P.cur].as<u32>(), (size_t)V.stride_w32, V.sub_w32, c->rp_bits, (u32)pop, rows);
            V.frq_valid = false;
        }
    }
    hipLaunchKernelGGL(k_sex_sequence, dim3(1), dim3(64), 0, c->stream, c->d_tables.as<GevRngTables>(), (u32)seed_gen0, n_people, P.d_sex[P.cur].as<uint8_t>());
    KCHECK();
    if (sex_out) HIPC(hipMemcpyAsync(sex_out, P.d_sex[P.cur].p, n_people, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    for (int k = 0; k < c->nchr; k++) { P.st[k].mut_total[P.cur] = 0; P.st[k].parts_total[P.cur] = c->chr_active[k] ? rows : 0; P.st[k].pool_list_valid = false; P.st[k].csr_valid = true; P.st[k].lp.valid = false; }
    c->ad_cached_pop = c->ad_host_set_pop = -1;
    P.n_people = n_people; P.n_phys = n_people; P.logical.clear(); P.gen0 = true;
    return GEV_OK;
}

// ---- Simulation::reproduce ----------------------------------------------------------------
// Two HIP streams.  `stream` carries the small kernels of generation g (sampling, sparse lists,
// CV planes, gamete grouping) and the status/sex read-back the host waits for; `stream_big`
// carries the HBM-bound dense stitch, which the host does NOT wait for: gev_reproduce returns as
// soon as the small work is done, so A/D of generation g, host mating and the small work of
// generation g+1 overlap the stitch of generation g.  Order on stream_big serialises consecutive
// stitches (each reads the plane the previous one wrote); the scratch set a stitch reads (g%2)
// is not reused before that stitch has finished (stream waits on its event).
// No host round trip inside a generation: variable-length outputs go to capacity-checked buffers
// sized from the previous totals; if one was too small the buffers are grown from the exact totals
// of the count passes and the small work is enqueued again (inputs are untouched until the flip).
static int enqueue_ad(gev_ctx* c, int pop, int buf, size_t n, bool counts_ready = false, int hbuf = -1);
static int enqueue_chain_head_start(gev_ctx* c);
static int prepare_eager_ad(gev_ctx* c, int pop);
static int check_not_pending(gev_ctx* c);
static int check_chain_size(size_t T);
static int materialize_order(gev_ctx* c, int pop);
static int ensure_csr(gev_ctx* c, int pop);
extern "C" int gev_presample(gev_ctx* c, int pop, uint32_t seed_reproduce, const uint32_t* mut_seeds, size_t n_mut_seeds, size_t n_people);
static int wait_planes(gev_ctx* c)
{
    if (c->planes_pending) HIPC(hipStreamWaitEvent(c->stream, c->ev_planes, 0));
    return GEV_OK;
}
// the sampling time of the set's previous generation, read before a head start records the sampling events again (never blocks:
// that sampling is long over; the stitch of that generation may still run, its time is collected later by harvest_timing)
static int harvest_sampling_time(gev_ctx::Scratch& sc)
{
    if (!sc.timing_pending || sc.sampling_ms_saved >= 0) return GEV_OK;
    float t;
    HIPC(hipEventElapsedTime(&t, sc.t[0], sc.t[1]));
    sc.sampling_ms_saved = t;
    return GEV_OK;
}
static int harvest_timing(gev_ctx* c, gev_ctx::Scratch& sc)
{
    if (!sc.timing_pending) return GEV_OK;
    HIPC(hipEventSynchronize(sc.t[3]));
    float t; float ms[4];
    if (sc.sampling_ms_saved >= 0) { ms[0] = sc.sampling_ms_saved; sc.sampling_ms_saved = -1; }
    else { HIPC(hipEventElapsedTime(&t, sc.t[0], sc.t[1])); ms[0] = t; }
    HIPC(hipEventElapsedTime(&t, sc.t[4], sc.t[3])); ms[1] = t;
    HIPC(hipEventElapsedTime(&t, sc.t[5], sc.t[2])); ms[2] = t;
    ms[3] = ms[0] + ms[1] + ms[2];
    for (int i = 0; i < 4; i++) { c->last_ms[i] = ms[i]; c->ms_sum[i] += ms[i]; }
    c->ms_count++;
    // overlap pays when the small kernels are a small fraction of the stitch (config 2: ~0.2); when they are
    // comparable (many chromosomes, short rows) the starved small stream only delays everything: stay serialised
    if (c->overlap_mode < 0 && c->serialize && c->auto_gens < 2) {
        c->auto_small_ms += ms[0] + ms[2]; c->auto_stitch_ms += ms[1];
        if (++c->auto_gens == 2) c->serialize = !(c->auto_small_ms < 4.0 * c->auto_stitch_ms);   // overlap unless the stitch is negligible
    }
    sc.timing_pending = false;
    return GEV_OK;
}
// host table -> device buffer `dst` on stream `st`, staged through the pinned ring (the host copy must stay valid until the
// asynchronous copy has run: a slot is reused only after a wrap, which waits for the stream)
// ... unless the device copy already holds exactly these bytes (the tables of a scratch set repeat from generation to generation
// as long as no buffer moved): `shadow` = host copy of what was uploaded last
static int upload_table_cached(gev_ctx* c, DevBuf& dst, std::vector<uint8_t>& shadow, const void* src, size_t bytes, hipStream_t st);
static int upload_table(gev_ctx* c, DevBuf& dst, const void* src, size_t bytes, hipStream_t st)
{
    GEVC(dst.ensure(std::max<size_t>(bytes, 16), st));
    if (!bytes) return GEV_OK;
    const size_t need = round_up(bytes, 64);
    if (c->h_ring_bytes < 4 * need) {
        HIPC(hipStreamSynchronize(st));
        if (c->h_ring) (void)hipHostFree(c->h_ring);
        c->h_ring = nullptr; c->h_ring_bytes = 0; c->h_ring_off = 0;
        static const size_t ring_min = getenv("GEV_TABLE_RING_BYTES") ? (size_t)atol(getenv("GEV_TABLE_RING_BYTES")) : (size_t)(256 << 10);   // (tests shrink it to force wrap-arounds)
        const size_t want = std::max<size_t>(8 * need, ring_min);
        HIPC(hipHostMalloc((void**)&c->h_ring, want, hipHostMallocDefault));
        c->h_ring_bytes = want;
    }
    if (c->h_ring_off + need > c->h_ring_bytes) { HIPC(hipStreamSynchronize(c->stream)); HIPC(hipStreamSynchronize(st)); c->h_ring_off = 0; }
    memcpy(c->h_ring + c->h_ring_off, src, bytes);
    HIPC(hipMemcpyAsync(dst.p, c->h_ring + c->h_ring_off, bytes, hipMemcpyHostToDevice, st));
    c->h_ring_off += need;
    return GEV_OK;
}
static int upload_table_cached(gev_ctx* c, DevBuf& dst, std::vector<uint8_t>& shadow, const void* src, size_t bytes, hipStream_t st)
{
    if (dst.p && shadow.size() == bytes && bytes && memcmp(shadow.data(), src, bytes) == 0) return GEV_OK;
    const void* before = dst.p;
    GEVC(upload_table(c, dst, src, bytes, st));
    (void)before;
    shadow.assign((const uint8_t*)src, (const uint8_t*)src + bytes);
    return GEV_OK;
}
static SampleDev make_sd(gev_ctx* c, gev_ctx::Scratch& sc, size_t T)
{
    SampleDev sd;
    sd.seed_pat = sc.seed_pat.as<u32>(); sd.seed_mat = sc.seed_mat.as<u32>(); sd.k = sc.k.as<u32>();
    sd.bk_off = sc.bk_off.as<u32>(); sd.bk = sc.bk.as<u64>(); sd.bk_idx = sc.bk_idx.as<u32>(); sd.start = sc.start.as<uint8_t>();
    sd.nmut = sc.nmut.as<u32>(); sd.nm_off = sc.nm_off.as<u32>(); sd.nm_pos = sc.nm_pos.as<u64>();
    sd.nm_side = sc.nm_side.as<uint8_t>(); sd.sex = sc.sex.as<uint8_t>();
    sd.father = sc.father.as<u32>(); sd.mother = sc.mother.as<u32>();
    sd.bk_ovf_base = (u32)(2 * T * GEV_BK_CAP); sd.bk_ovf_cap = (u32)c->bk_ovf_cap;
    sd.nm_ovf_base = (u32)(T * GEV_NM_CAP); sd.nm_ovf_cap = (u32)c->nm_ovf_cap;
    sd.status = sc.status.as<u32>();
    return sd;
}
// per-generation scratch every phase needs (sizes depend on n_people only)
static int ensure_scratch(gev_ctx* c, gev_ctx::Scratch& sc, size_t n_people, bool has_mut)
{
    hipStream_t st = c->stream;
    const size_t T = n_people * (size_t)c->nchr;
    const size_t n_status = ST_TOTALS + ST_PER_CHR * (size_t)c->nchr;
    GEVC(sc.father.ensure(n_people * sizeof(u32), st)); GEVC(sc.mother.ensure(n_people * sizeof(u32), st));
    GEVC(sc.seed_pat.ensure((T + 1) * sizeof(u32), st)); GEVC(sc.seed_mat.ensure(T * sizeof(u32), st));
    GEVC(sc.k.ensure(2 * T * sizeof(u32), st)); GEVC(sc.bk_off.ensure((2 * T + 1) * sizeof(u32), st));
    GEVC(sc.start.ensure(2 * T, st)); GEVC(sc.sex.ensure(n_people, st));
    GEVC(sc.nmut.ensure(T * sizeof(u32), st)); GEVC(sc.nm_off.ensure((T + 1) * sizeof(u32), st));
    GEVC(sc.nm_pos.ensure(16, st)); GEVC(sc.nm_side.ensure(16, st));
    GEVC(sc.status.ensure(n_status * sizeof(u32), st));
    if (has_mut) { GEVC(sc.mutseeds.ensure(T * sizeof(u32), st)); GEVC(sc.slow_mut.ensure(T * sizeof(u32), st)); GEVC(sc.slow_rec.ensure(T * sizeof(u32), st)); }
    return GEV_OK;
}
// K1-K3: crossover / mutation sampling and the rand() seed chain; depends on the seeds and n_people only, not on the couples
static int enqueue_sampling(gev_ctx* c, gev_ctx::Scratch& sc, int pop, size_t n_people, bool has_mut, u32 seed_reproduce, hipStream_t st,
                            const u32* seed_ptr = nullptr /* device copy of the reproduce seed */, const u32* mut_seeds_dev = nullptr /* default: sc.mutseeds */, bool clear_status = true)
{
    PopState& P = c->pop[pop];
    const int nchr = c->nchr;
    const size_t T = n_people * (size_t)nchr;
    const GevRngTables* Tb = c->d_tables.as<GevRngTables>();
    const ChrDev* chrs = P.d_chrdev.as<ChrDev>();
    const size_t bk_fixed = 2 * T * GEV_BK_CAP, nm_fixed = T * GEV_NM_CAP;
    if (bk_fixed + c->bk_ovf_cap >= 0xffffffffull || nm_fixed + c->nm_ovf_cap >= 0xffffffffull) return fail(GEV_EINVAL, "reproduce: breakpoint/mutation record space exceeds 32-bit offsets");
    GEVC(sc.bk.ensure((bk_fixed + c->bk_ovf_cap) * sizeof(u64), st));
    GEVC(sc.bk_idx.ensure((bk_fixed + c->bk_ovf_cap) * sizeof(u32), st));
    if (has_mut) { GEVC(sc.nm_pos.ensure((nm_fixed + c->nm_ovf_cap) * sizeof(u64), st)); GEVC(sc.nm_side.ensure(nm_fixed + c->nm_ovf_cap, st)); }
    const size_t n_status = ST_TOTALS + ST_PER_CHR * (size_t)nchr;
    if (clear_status) HIPC(hipMemsetAsync(sc.status.p, 0, n_status * sizeof(u32), st));
    SampleDev sd = make_sd(c, sc, T);
    const u32* mseeds = mut_seeds_dev ? mut_seeds_dev : sc.mutseeds.as<u32>();

    HIPC(hipEventRecord(sc.t[0], st));
    // ---- sampling: one map scan per gamete / per mutation task
    // persistent sampling workgroups.  Next to the other streams' kernels: 1792 (7 waves per SIMD: the kernels stall on dependent
    // integer chains and LDS, and more waves hide that -- +2 % at config 2) while the generation has few tasks; 768 for many tasks
    // (the 1.4 M tasks of the config-4 shard: its list and table kernels are the bound, and the sampling grid is in their way)
    const unsigned shared_grid = c->sample_grid_env ? c->sample_grid_shared : (T <= 400000 ? 1792u : c->sample_grid_shared);
    const unsigned task_blocks = (unsigned)std::min<size_t>(ceil_div(T, 4), (c->dense && !c->serialize) ? shared_grid : c->sample_grid);
    if (has_mut && c->sample_batched) {
        // eight tasks per wave; the rare tasks that need more than 8 rand() outputs go to the one-task-per-wave kernels
        const unsigned batch_blocks = (unsigned)std::min<size_t>(ceil_div(ceil_div(T, SB_TASKS), 4), task_blocks);
        const unsigned slow_blocks = (unsigned)std::min<size_t>(ceil_div(T, 4), 64);
        hipLaunchKernelGGL(k_mut_sample8, dim3(batch_blocks), dim3(256), 0, st, Tb, chrs, nchr, mseeds, seed_reproduce, seed_ptr, T, sd, sc.slow_mut.as<u32>());
        hipLaunchKernelGGL(k_mut_sample, dim3(slow_blocks), dim3(256), 0, st, Tb, chrs, nchr, mseeds, T, sd, sc.slow_mut.as<u32>(), sd.status + ST_SLOW_MUT);
        hipLaunchKernelGGL(k_rec_sample8, dim3(batch_blocks), dim3(256), 0, st, Tb, chrs, nchr, T, sd, sc.slow_rec.as<u32>());
        hipLaunchKernelGGL(k_rec_sample, dim3(slow_blocks), dim3(256), 0, st, Tb, chrs, nchr, seed_reproduce, seed_ptr, T, sd, sc.slow_rec.as<u32>(), sd.status + ST_SLOW_REC);
    } else if (has_mut) {
        hipLaunchKernelGGL(k_mut_sample, dim3(task_blocks), dim3(256), 0, st, Tb, chrs, nchr, mseeds, T, sd, (const u32*)nullptr, (const u32*)nullptr);
        hipLaunchKernelGGL(k_rec_sample, dim3(task_blocks), dim3(256), 0, st, Tb, chrs, nchr, seed_reproduce, seed_ptr, T, sd, (const u32*)nullptr, (const u32*)nullptr);
    } else {
        // no mutation map: the gametes of a generation form ONE serial chain (src/Simulation.cpp:2447-2455).  A workgroup per link
        // (k_rec_chain_wg); GEV_CHAIN_WG=0: the one-wave form
        static const bool wg = !(getenv("GEV_CHAIN_WG") && atoi(getenv("GEV_CHAIN_WG")) == 0);
        if (wg) {
            size_t rows_all = 0;
            for (int k = 0; k < nchr; k++) rows_all += c->pop[pop].cs[k].rbp.size();
            const u32 thr_lds = rows_all <= 2048 ? (u32)rows_all : 0u;            // thresholds of all chromosomes in LDS when they fit into 32 KiB
            hipLaunchKernelGGL(k_rec_chain_wg, dim3(1), dim3(CHAIN_THREADS), (size_t)thr_lds * sizeof(GevThr), st, Tb, chrs, nchr, seed_reproduce, seed_ptr, T, sd, thr_lds);
        } else hipLaunchKernelGGL(k_rec_chain, dim3(1), dim3(64), 0, st, Tb, chrs, nchr, seed_reproduce, seed_ptr, T, sd);
    }
    KCHECK();
    HIPC(hipEventRecord(sc.t[1], st));
    return GEV_OK;
}
// ---- sparse state as shared pieces (gev_lists.h) -------------------------------------------------------------------------
// ranges of 2^lgw bp, at most LP_MAXSEG of them (GEV_LIST_SEGS: fewer; 1 = one piece per row and list)
static void lp_geometry(const ChrStatic& S, u32& nseg, u32& lgw)
{
    const u32 max_seg = getenv("GEV_LIST_SEGS") ? (u32)std::min(std::max(atoi(getenv("GEV_LIST_SEGS")), 1), LP_MAXSEG) : (u32)LP_MAXSEG;
    const u64 span = S.rbp.back() - S.rbp.front();
    lgw = 0;
    while ((span >> lgw) + 1 > max_seg) lgw++;
    nseg = (u32)(span >> lgw) + 1;
}
static size_t lp_arena_entries(size_t rows, size_t live, int n_active_chr)
{
    // arena = room for the pieces of many generations (each appends a few entries per row): a tenth of the device memory, shared
    // out over the active chromosomes, at most 4096 entries per row (GEV_LIST_ARENA: entries per row); when it fills up, the
    // pieces are compacted (pieces -> whole lists -> pieces).  GEV_LIST_HEADROOM=0: just what is live (tests: every generation
    // then overflows, grows and is enqueued again)
    const size_t per_row_env = getenv("GEV_LIST_ARENA") ? (size_t)atol(getenv("GEV_LIST_ARENA")) : 0;      // (read at every call: tests set it per case)
    const bool tight = getenv("GEV_LIST_HEADROOM") && atol(getenv("GEV_LIST_HEADROOM")) == 0;
    if (tight) return live + 16;
    size_t per_row = per_row_env;
    if (!per_row) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); total_b = (size_t)64 << 30; }
        const size_t budget = total_b / 10 / (size_t)std::max(n_active_chr, 1);                 // bytes for this chromosome's two arenas (16 + 8 bytes per entry pair)
        per_row = std::min<size_t>(std::max<size_t>(budget / (std::max<size_t>(rows, 1) * 24), 128), 4096);
    }
    return std::min<size_t>(per_row_env ? live + rows * per_row : std::max<size_t>(2 * live + 4096, rows * per_row), 0xfffffff0u);
}
// CSR of buffer P.cur -> pieces (tables of buffer P.cur, arenas restarted): first use, after a CSR-side change, compaction
static int lp_import_all(gev_ctx* c, PopState& P, int k, hipStream_t st)
{
    ChrStatic& S = P.cs[k]; ChrState& cs = P.st[k]; ChrState::LpState& lp = cs.lp;
    const size_t rows = 2 * P.n_phys;
    lp_geometry(S, lp.nseg, lp.lgw);
    const bool track = c->track_intervals;
    const size_t live_p = track ? cs.parts_total[P.cur] + rows * lp.nseg : 0, live_m = cs.mut_total[P.cur];
    if (live_p >= 0xfffffff0u || live_m >= 0xfffffff0u) return fail(GEV_EDEVICE, "list pieces: more than 2^32 list entries on one chromosome");
    const size_t tab_rows = std::max<size_t>(rows, 2 * P.cap_people);
    for (int b = 0; b < 2; b++) {
        if (track) GEVC(lp.ptab[b].ensure(tab_rows * lp.nseg * sizeof(uint2), st));
        GEVC(lp.mtab[b].ensure(tab_rows * lp.nseg * sizeof(uint2), st));
    }
    int n_active = 0; for (int q = 0; q < c->nchr; q++) n_active += c->chr_active[q] ? 1 : 0;
    if (track) GEVC(lp.parena.ensure(lp_arena_entries(rows, live_p, n_active) * sizeof(LpPart), st));
    GEVC(lp.marena.ensure(std::max<size_t>(lp_arena_entries(rows, live_m, n_active), 16) * sizeof(u64), st));
    GEVC(lp.ctr.ensure(4 * sizeof(u32), st));
    HIPC(hipMemsetAsync(lp.ctr.p, 0, 4 * sizeof(u32), st));
    if (rows)
        hipLaunchKernelGGL(k_lp_import, dim3((unsigned)ceil_div(rows * LP_MAXSEG, 256)), dim3(256), 0, st,
                           track ? cs.poff[P.cur].as<u32>() : (const u32*)nullptr, track ? cs.parts[P.cur].as<gev_part>() : (const gev_part*)nullptr,
                           cs.moff[P.cur].as<u32>(), cs.mpos[P.cur].as<u64>(), (size_t)0, rows,
                           track ? lp.ptab[P.cur].as<uint2>() : (uint2*)nullptr, lp.mtab[P.cur].as<uint2>(), lp.parena.as<LpPart>(), lp.marena.as<u64>(),
                           0u, 0u, (u32)(lp.parena.bytes / sizeof(LpPart)), (u32)(lp.marena.bytes / sizeof(u64)), lp.nseg, lp.lgw, (u64)S.rbp.front(),
                           lp.ctr.as<u32>(), lp.ctr.as<u32>() + 2);
    KCHECK();
    u32 h[4] = {0, 0, 0, 0};
    HIPC(hipMemcpyAsync(h, lp.ctr.p, sizeof h, hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    if (h[2]) return fail(GEV_EDEVICE, "list pieces: import overflowed its arena (internal error: %u of %zu interval, %u of %zu mutation entries)", h[0], lp.parena.bytes / sizeof(LpPart), h[1], lp.marena.bytes / sizeof(u64));
    lp.p_used = h[0]; lp.m_used = h[1]; lp.p_last = 0; lp.m_last = 0; lp.valid = true; lp.grow_p = lp.grow_m = false; lp.imports++;
    return GEV_OK;
}
// one arena of the current generation's pieces compacted in place, sharing kept (gev_lists.h); *used = entries in use afterwards
static int lp_compact_arena(gev_ctx* c, uint2* tab, size_t n_entries, u32 extra, DevBuf& arena, size_t elem_bytes, u32* used, hipStream_t st)
{
    const size_t n_words = ceil_div(std::max<size_t>(*used, 1), 32);
    GEVC(c->d_lpc_bits.ensure(n_words * sizeof(u32), st)); GEVC(c->d_lpc_cnt.ensure((n_words + 1) * sizeof(u32), st)); GEVC(c->d_lpc_wpre.ensure((n_words + 1) * sizeof(u32), st));
    HIPC(hipMemsetAsync(c->d_lpc_bits.p, 0, n_words * sizeof(u32), st));
    const unsigned eb = (unsigned)ceil_div(std::max<size_t>(n_entries, 1), 256);
    hipLaunchKernelGGL(k_lpc_mark, dim3(eb), dim3(256), 0, st, (const uint2*)tab, n_entries, extra, c->d_lpc_bits.as<u32>());
    hipLaunchKernelGGL(k_lpc_popc, dim3((unsigned)ceil_div(n_words, 256)), dim3(256), 0, st, c->d_lpc_bits.as<u32>(), n_words, c->d_lpc_cnt.as<u32>());
    KCHECK();
    u32 n_pieces = 0, total = 0;
    GEVC(scan_u32(c, c->d_lpc_cnt.as<u32>(), n_words, c->d_lpc_wpre.as<u32>(), &n_pieces));
    GEVC(c->d_lpc_len.ensure(((size_t)n_pieces + 1) * sizeof(u32), st)); GEVC(c->d_lpc_old.ensure(((size_t)n_pieces + 1) * sizeof(u32), st)); GEVC(c->d_lpc_new.ensure(((size_t)n_pieces + 2) * sizeof(u32), st));
    hipLaunchKernelGGL(k_lpc_collect, dim3(eb), dim3(256), 0, st, (const uint2*)tab, n_entries, extra, c->d_lpc_bits.as<u32>(), c->d_lpc_wpre.as<u32>(), c->d_lpc_len.as<u32>(), c->d_lpc_old.as<u32>());
    KCHECK();
    GEVC(scan_u32(c, c->d_lpc_len.as<u32>(), n_pieces, c->d_lpc_new.as<u32>(), &total));
    if (n_pieces) {
        GEVC(c->d_lpc_tmp.ensure(std::max<size_t>(total, 1) * elem_bytes, st));
        hipLaunchKernelGGL(k_lpc_copy, dim3((unsigned)ceil_div(n_pieces, 256)), dim3(256), 0, st, (const u64*)arena.p, c->d_lpc_tmp.as<u64>(), c->d_lpc_len.as<u32>(), c->d_lpc_old.as<u32>(), c->d_lpc_new.as<u32>(), (size_t)n_pieces, (u32)(elem_bytes / 8));
        KCHECK();
        HIPC(hipMemcpyAsync(arena.p, c->d_lpc_tmp.p, (size_t)total * elem_bytes, hipMemcpyDeviceToDevice, st));
    }
    hipLaunchKernelGGL(k_lpc_rewrite, dim3(eb), dim3(256), 0, st, tab, n_entries, extra, c->d_lpc_bits.as<u32>(), c->d_lpc_wpre.as<u32>(), c->d_lpc_new.as<u32>());
    KCHECK();
    *used = total;
    return GEV_OK;
}
static int lp_compact(gev_ctx* c, PopState& P, int k, hipStream_t st)
{
    ChrState::LpState& lp = P.st[k].lp;
    const size_t n_entries = 2 * P.n_phys * (size_t)lp.nseg;
    if (st != c->stream) HIPC(hipStreamSynchronize(st));
    if (c->track_intervals) GEVC(lp_compact_arena(c, lp.ptab[P.cur].as<uint2>(), n_entries, 1u, lp.parena, sizeof(LpPart), &lp.p_used, c->stream));
    GEVC(lp_compact_arena(c, lp.mtab[P.cur].as<uint2>(), n_entries, 0u, lp.marena, sizeof(u64), &lp.m_used, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    lp.compactions++;
    return GEV_OK;
}
// the CSR form of the current generation (buffer P.cur), materialised from the pieces when it is not there
static int ensure_csr(gev_ctx* c, int pop)
{
    PopState& P = c->pop[pop];
    hipStream_t st = c->stream;
    const size_t rows = 2 * P.n_phys;
    for (int k = 0; k < c->nchr; k++) {
        ChrState& cs = P.st[k]; ChrState::LpState& lp = cs.lp;
        if (!c->chr_active[k] || cs.csr_valid) continue;
        if (!lp.valid) return fail(GEV_ESTATE, "internal error: neither form of the sparse state is valid (pop %d chr %d)", pop, k);
        const bool track = c->track_intervals;
        GEVC(cs.moff[P.cur].ensure((rows + 1) * sizeof(u32), st)); GEVC(cs.poff[P.cur].ensure((rows + 1) * sizeof(u32), st));
        GEVC(c->d_cnt.ensure(2 * (rows + 1) * sizeof(u32), st));
        u32* pcnt = c->d_cnt.as<u32>(); u32* mcnt = pcnt + rows + 1;
        const unsigned blocks = (unsigned)ceil_div(std::max<size_t>(rows, 1) * LP_MAXSEG, 256);
        hipLaunchKernelGGL(k_lp_count, dim3(blocks), dim3(256), 0, st, track ? lp.ptab[P.cur].as<uint2>() : (const uint2*)nullptr, lp.mtab[P.cur].as<uint2>(), lp.nseg,
                           (const u32*)nullptr, rows, track ? pcnt : (u32*)nullptr, mcnt);
        KCHECK();
        u32 tot_m = 0, tot_p = 0;
        GEVC(scan_u32(c, mcnt, rows, cs.moff[P.cur].as<u32>(), &tot_m));
        if (track) GEVC(scan_u32(c, pcnt, rows, cs.poff[P.cur].as<u32>(), &tot_p));
        GEVC(cs.mpos[P.cur].ensure(std::max<size_t>(tot_m, 2) * sizeof(u64), st, false, 1.25));
        if (track) GEVC(cs.parts[P.cur].ensure(std::max<size_t>(tot_p, 1) * sizeof(gev_part), st, false, 1.25));
        hipLaunchKernelGGL(k_lp_fill, dim3(blocks), dim3(256), 0, st, track ? lp.ptab[P.cur].as<uint2>() : (const uint2*)nullptr, lp.mtab[P.cur].as<uint2>(),
                           lp.parena.as<LpPart>(), lp.marena.as<u64>(), lp.nseg, (const u32*)nullptr, rows, (u64)P.cs[k].rbp.back(),
                           cs.poff[P.cur].as<u32>(), cs.parts[P.cur].as<gev_part>(), cs.moff[P.cur].as<u32>(), cs.mpos[P.cur].as<u64>());
        KCHECK();
        HIPC(hipStreamSynchronize(st));
        cs.mut_total[P.cur] = tot_m; cs.parts_total[P.cur] = track ? tot_p : 0; cs.csr_valid = true;
    }
    return GEV_OK;
}
// a CSR-side change of the current generation (generation 0, migration, import, order restoring): the pieces no longer describe it
static void lists_changed_in_csr(gev_ctx* c, PopState& P)
{
    for (int k = 0; k < c->nchr; k++) { P.st[k].csr_valid = true; P.st[k].lp.valid = false; }
}
static const size_t LIST_HEADROOM = getenv("GEV_LIST_HEADROOM") ? (size_t)atol(getenv("GEV_LIST_HEADROOM")) : 48;    // spare list entries per haplotype row when a list buffer is (re)allocated (tests force redos with 0)
// K4/K6 + grouping: everything of the small work that needs the couples (parents) on top of the sampling results.
// Every kernel covers ALL active chromosomes in one launch (blockIdx.y = entry of the generation's ChrWork / CvWork table).
// work tables of the generation (one ChrWork per active chromosome, one CvWork per (phenotype, active chromosome)): list buffers
// sized, tables uploaded on `st`; the launch shapes the later pieces need are kept in the scratch set
static int enqueue_tables(gev_ctx* c, gev_ctx::Scratch& sc, int pop, size_t n_people, hipStream_t st)
{
    PopState& P = c->pop[pop];
    const int nchr = c->nchr;
    const size_t rows = 2 * n_people;
    const int cur = P.cur, alt = P.cur ^ 1;
    std::vector<ChrWork> cw; std::vector<CvWork> vw;
    // The free list of the segment pool is kept across generations (units nobody named when it was built and that were not handed
    // out since are still unnamed) and rebuilt only when what is left of it may not cover the generation: four times what the last
    // generation took, at least a quarter of a generation's segments.  Should a generation need more than is left (k_pool_fresh
    // raises FLAG_POOL), it is enqueued again behind a rebuild.  GEV_POOL_REBUILD=1: every generation (the round-2 behaviour).
    static const bool always = getenv("GEV_POOL_REBUILD") && atoi(getenv("GEV_POOL_REBUILD")) != 0;
    sc.pool_rebuild = always;
    if (c->dense) for (int k = 0; k < nchr; k++) {
        if (!c->chr_active[k]) continue;
        ChrState& cs = P.st[k];
        const size_t left = cs.pool_n_free - std::min(cs.pool_cursor, cs.pool_n_free), gen_segs = 2 * n_people * P.cs[k].nseg;
        const size_t need = c->alias_rows ? std::max<size_t>(4 * (size_t)cs.pool_last_taken, gen_segs / 4) : gen_segs;
        sc.pool_rebuild |= !cs.pool_list_valid || cs.pool_force_rebuild || left < need;
    }
    if (sc.pool_rebuild) for (int k = 0; k < nchr; k++) if (c->chr_active[k]) pool_new_stamp(P.st[k]);     // the marks of the rebuild
    for (int k = 0; k < nchr; k++) {
        if (!c->chr_active[k]) continue;
        ChrStatic& S = P.cs[k]; ChrState& cs = P.st[k];
        // the parents' pieces: imported from the CSR form when that is what describes them; an arena that may not hold this
        // generation's pieces is compacted first (pieces -> CSR -> pieces: what no table names any more is dropped)
        ChrState::LpState& lp = cs.lp;
        const size_t rows_cur = 2 * P.n_phys;
        const size_t guess_p = std::max<size_t>(4 * (size_t)lp.p_last, 8 * rows), guess_m = std::max<size_t>(4 * (size_t)lp.m_last, 4 * rows);
        const bool tight = LIST_HEADROOM == 0;
        // room the arenas must have on top of what is used: an attempt overflowed (its status block holds what it wanted to append) ...
        size_t need_p = lp.grow_p ? (size_t)lp.p_last * 5 / 4 + 1024 : 0, need_m = lp.grow_m ? (size_t)lp.m_last * 5 / 4 + 1024 : 0;
        if (lp.valid && !tight) {
            const size_t want_p = c->track_intervals ? std::max(need_p, guess_p) : 0, want_m = std::max(need_m, guess_m);
            const bool full_p = c->track_intervals && lp.p_used + want_p > lp.parena.bytes / sizeof(LpPart);
            const bool full_m = lp.m_used + want_m > lp.marena.bytes / sizeof(u64);
            if (full_p || full_m) {                           // drop the pieces nobody names any more; should that not make room, the arena grows below
                GEVC(lp_compact(c, P, k, st));
                need_p = want_p; need_m = want_m;
            }
        }
        if (!lp.valid) {
            GEVC(lp_import_all(c, P, k, st));
            if (!tight) { need_p = c->track_intervals ? guess_p : 0; need_m = guess_m; }     // ... or a fresh import has to leave room for this generation
        }
        if (need_p && lp.p_used + need_p > lp.parena.bytes / sizeof(LpPart))
            GEVC(lp.parena.ensure(std::min<size_t>(lp.p_used + need_p, 0xfffffff0u) * sizeof(LpPart), st, /*keep=*/true, tight ? 1.0 : 1.5));
        if (need_m && lp.m_used + need_m > lp.marena.bytes / sizeof(u64))
            GEVC(lp.marena.ensure(std::min<size_t>(lp.m_used + need_m, 0xfffffff0u) * sizeof(u64), st, /*keep=*/true, tight ? 1.0 : 1.5));
        lp.grow_p = lp.grow_m = false;
        const size_t tab_bytes = std::max<size_t>(rows, rows_cur) * lp.nseg * sizeof(uint2);
        if (c->track_intervals) { GEVC(lp.ptab[alt].ensure(tab_bytes, st)); }
        GEVC(lp.mtab[alt].ensure(tab_bytes, st));
        ChrWork w{};
        w.lp.ptab_cur = lp.ptab[cur].as<uint2>(); w.lp.ptab_alt = lp.ptab[alt].as<uint2>(); w.lp.mtab_cur = lp.mtab[cur].as<uint2>(); w.lp.mtab_alt = lp.mtab[alt].as<uint2>();
        w.lp.parena = lp.parena.as<LpPart>(); w.lp.marena = lp.marena.as<u64>();
        w.lp.pbase = lp.p_used; w.lp.mbase = lp.m_used;
        w.lp.pcap = (u32)std::min<size_t>(lp.parena.bytes / sizeof(LpPart), 0xfffffff0u); w.lp.mcap = (u32)std::min<size_t>(lp.marena.bytes / sizeof(u64), 0xfffffff0u);
        w.lp.nseg = lp.nseg; w.lp.lgw = lp.lgw; w.lp.track = c->track_intervals ? 1u : 0u;
        GEVC(lp.items.ensure(rows * LP_MAXSEG * sizeof(u32), st));
        GEVC(c->d_lp_nitems.ensure((size_t)nchr * sizeof(u32), st));
        w.lp.items = lp.items.as<u32>(); w.lp.n_items = c->d_lp_nitems.as<u32>() + k;
        w.bp0 = S.rbp.front(); w.bp_end = S.rbp.back(); w.chr = k;
        if (c->dense) {
            w.pw = pool_work(c, P, k, (P.pcur + 1) % 3); w.snp_pos = S.d_pos.as<u64>();
            w.stride = S.stride; w.chunks = (u32)(S.stride / 16); w.L = (u32)S.L;
        }
        cw.push_back(w);
        for (int p = 0; p < c->nphen; p++) {
            CvStatic& V = P.cv[p][k];
            GEVC(V.d_partial.ensure(ceil_div(rows, SMALL_ROWS_PER_BLOCK) * 1024 * sizeof(uint16_t), st));
            vw.push_back(CvWork{P.cvp[p][k][alt].as<u32>(), P.cvp[p][k][cur].as<u32>(), V.d_pos_sorted.as<u64>(), V.d_counts.as<u32>(), V.d_partial.as<uint16_t>(), S.rbp.front(), S.rbp.back(), V.stride_w32, V.sub_w32, V.C, k, (u32)(cw.size() - 1)});
        }
    }
    sc.n_chrwork = (unsigned)cw.size(); sc.n_cvwork = (unsigned)vw.size();
    sc.lp_entries_per_row = 0;                            // interval entries the last generation appended per row (its new pieces' lengths)
    for (int k = 0; k < nchr; k++) if (c->chr_active[k]) sc.lp_entries_per_row = std::max<size_t>(sc.lp_entries_per_row, P.st[k].lp.p_last / std::max<size_t>(rows, 1));
    sc.nseg_max = 1; sc.cv_used_max = 0; sc.cv_max = 0;
    for (const ChrWork& w : cw) sc.nseg_max = std::max<size_t>(sc.nseg_max, w.pw.nseg);
    const u32 nsub = 1 + c->rp_bits;
    sc.cv_count_fused = c->cv_count_fused_ok;
    for (const CvWork& v : vw) { sc.cv_used_max = std::max<size_t>(sc.cv_used_max, (size_t)v.sub_w32 * nsub); sc.cv_count_fused &= v.sub_w32 <= 32; if (v.C <= SMALL_POS_LDS) sc.cv_max = std::max(sc.cv_max, v.C); }   // LDS copy of the CV grid: sized for the launch, not for the worst case (occupancy)
    if (sc.n_chrwork) {
        GEVC(upload_table_cached(c, sc.chrwork, sc.chrwork_shadow, cw.data(), cw.size() * sizeof(ChrWork), st));
        GEVC(upload_table_cached(c, sc.cvwork, sc.cvwork_shadow, vw.data(), vw.size() * sizeof(CvWork), st));
    }
    return GEV_OK;
}
// free units of the pool = units no (slot, segment) of the parents names: needs the parents' table only, not this generation's
// sampling or couples
static int enqueue_pool_free(gev_ctx* c, gev_ctx::Scratch& sc, int pop, size_t /*n_people*/, hipStream_t st)
{
    PopState& P = c->pop[pop];
    if (!c->dense || !sc.n_chrwork || !sc.pool_rebuild) return GEV_OK;      // (decided by enqueue_tables)
    const unsigned pool_blocks = (unsigned)std::min<size_t>(ceil_div(4 * P.cap_people * sc.nseg_max, 256), 1024);
    hipLaunchKernelGGL(k_pool_mark_tab, dim3(pool_blocks, sc.n_chrwork), dim3(256), 0, st, sc.chrwork.as<ChrWork>(), 2 * P.n_phys);
    hipLaunchKernelGGL(k_pool_collect_tab, dim3(pool_blocks, sc.n_chrwork), dim3(256), 0, st, sc.chrwork.as<ChrWork>());
    KCHECK();
    for (int k = 0; k < c->nchr; k++) if (c->chr_active[k]) { P.st[k].pool_list_valid = true; P.st[k].pool_force_rebuild = false; P.st[k].pool_rebuilds++; }
    return GEV_OK;
}
// units of the offspring rows (segments with a crossover boundary take a free unit and a work-list entry, the others name the
// parental unit) + the stitch's work-list length and the segment totals of the status block
// publish: k_pool_publish (counters into the status block, length of the stitch's work list) right behind; otherwise the caller
// launches it (enqueue_pool_publish) on a stream that is joined before the stitch starts and the status block leaves
static int enqueue_pool_publish(gev_ctx* c, gev_ctx::Scratch& sc, size_t n_people, hipStream_t st)
{
    if (!c->dense || !sc.n_chrwork) return GEV_OK;
    SampleDev sd = make_sd(c, sc, n_people * (size_t)c->nchr);
    hipLaunchKernelGGL(k_pool_publish, dim3(1), dim3(64), 0, st, sc.chrwork.as<ChrWork>(), sc.n_chrwork, sd.status);
    KCHECK();
    return GEV_OK;
}
static int enqueue_pool_assign(gev_ctx* c, gev_ctx::Scratch& sc, size_t n_people, hipStream_t st, bool publish)
{
    if (!c->dense || !sc.n_chrwork) return GEV_OK;
    const size_t rows = 2 * n_people, T = n_people * (size_t)c->nchr;
    SampleDev sd = make_sd(c, sc, T);
    u32 lg = 0; while ((1u << lg) * POOL_INH < sc.nseg_max) lg++;    // threads per row of k_pool_inherit: 2^lg, rows per block: 256 >> lg
    hipLaunchKernelGGL(k_pool_inherit, dim3((unsigned)ceil_div(rows, (size_t)(256u >> lg)), sc.n_chrwork), dim3(256), 0, st, sc.chrwork.as<ChrWork>(), rows, c->nchr, sd, lg);
    hipLaunchKernelGGL(k_pool_fresh, dim3((unsigned)ceil_div(rows, 256), sc.n_chrwork), dim3(256), 0, st, sc.chrwork.as<ChrWork>(), rows, c->nchr, sd);
    KCHECK();
    return publish ? enqueue_pool_publish(c, sc, n_people, st) : GEV_OK;
}
// sparse state: mutation lists + ancestry intervals (count -> segmented scan -> fill).  Needs the sampling results and the couples;
// nothing of the generation's dense / CV / A-D work reads its output.
static int enqueue_lists(gev_ctx* c, gev_ctx::Scratch& sc, size_t n_people, bool has_mut, hipStream_t st)
{
    const unsigned na = sc.n_chrwork;
    if (!na) return GEV_OK;
    const int nchr = c->nchr;
    const size_t T = n_people * (size_t)nchr, rows = 2 * n_people;
    SampleDev sd = make_sd(c, sc, T);
    // one launch: every (offspring row, position range) either names the parent's pieces or builds its own (gev_lists.h)
    HIPC(hipMemsetAsync(c->d_lp_nitems.p, 0, c->d_lp_nitems.bytes, st));
    hipLaunchKernelGGL(k_lp_inherit, dim3((unsigned)ceil_div(rows, 256), na), dim3(256), 0, st, sc.chrwork.as<ChrWork>(), rows, nchr, (int)has_mut, sd);
    const bool literal = getenv("GEV_LP_LITERAL") && atoi(getenv("GEV_LP_LITERAL")) != 0;      // recombine's statements one by one instead of their closed form (cross-check)
    const dim3 bgrid((unsigned)std::min<size_t>(ceil_div(rows * 4, 256), 2048), na);
    // one lane per piece while pieces are short, a group of eight once they are long (the one-lane kernel is faster up to about 20
    // interval entries per new piece -- generation ~450 at config 2 -- and its time grows with the piece length; the eight-lane
    // kernel's hardly does: 0.77 -> 0.89 ms per generation over 700 generations against 0.67 -> 0.97).  GEV_LP_LANES=1|8 fixes it.
    int lanes = getenv("GEV_LP_LANES") ? atoi(getenv("GEV_LP_LANES")) : 0;
    if (lanes != 1 && lanes != 8) lanes = sc.lp_entries_per_row >= 20 ? 8 : 1;
    if (literal) hipLaunchKernelGGL((k_lp_build<true>), bgrid, dim3(256), 0, st, sc.chrwork.as<ChrWork>(), nchr, (int)has_mut, sd);
    else if (lanes == 1) hipLaunchKernelGGL((k_lp_build<false>), bgrid, dim3(256), 0, st, sc.chrwork.as<ChrWork>(), nchr, (int)has_mut, sd);
    else hipLaunchKernelGGL(k_lp_build8, dim3((unsigned)std::min<size_t>(ceil_div(rows * 4, 32), 8192), na), dim3(256), 0, st, sc.chrwork.as<ChrWork>(), nchr, (int)has_mut, sd);
    KCHECK();
    return GEV_OK;
}
// CV planes of the offspring: every sub-row (resolved alleles, root-population bits) is inherited by the gamete's crossover
// pattern; the generation's new mutations that fall on a CV position flip the allele there (k_cv_newmut)
static int enqueue_cv_planes(gev_ctx* c, gev_ctx::Scratch& sc, size_t n_people, bool has_mut, bool count_cols, hipStream_t st)
{
    if (!sc.n_cvwork || !sc.cv_used_max) return GEV_OK;
    const int nchr = c->nchr;
    const size_t T = n_people * (size_t)nchr, rows = 2 * n_people;
    SampleDev sd = make_sd(c, sc, T);
    const CvWork* Vt = sc.cvwork.as<CvWork>();
    const dim3 grid((unsigned)ceil_div(rows, SMALL_ROWS_PER_BLOCK), sc.n_cvwork); const size_t lds = (size_t)sc.cv_max * sizeof(u64); const u32 nsub = 1u + c->rp_bits;
    if (c->cv_threads == 256) hipLaunchKernelGGL((k_stitch_small<256>), grid, dim3(256), lds, st, Vt, nsub, rows, nchr, sd, sc.cv_max, (int)count_cols);
    else if (c->cv_threads == 1024) hipLaunchKernelGGL((k_stitch_small<1024>), grid, dim3(1024), lds, st, Vt, nsub, rows, nchr, sd, sc.cv_max, (int)count_cols);
    else hipLaunchKernelGGL((k_stitch_small<512>), grid, dim3(512), lds, st, Vt, nsub, rows, nchr, sd, sc.cv_max, (int)count_cols);
    if (has_mut) hipLaunchKernelGGL(k_cv_newmut, dim3((unsigned)ceil_div(n_people, 256), sc.n_cvwork), dim3(256), 0, st, Vt, sc.chrwork.as<ChrWork>(), n_people, nchr, sd, (int)count_cols);
    KCHECK();
    return GEV_OK;
}
// Unused dynamic LDS per stitch workgroup that limits the workgroups per CU to `occ` (160 KiB of LDS per CU; the kernels' own
// static LDS is 3.4 KiB, allocation granularity taken as 512 B): the SMALLEST padding with which occ + 1 workgroups no longer
// fit, so that the rest of the LDS (about 20 KiB at occ = 6) stays available to the small kernels that run next to the stitch --
// the sampling kernels stage 17.5 KiB of tables.  The largest padding that still admits `occ` workgroups leaves 2 KiB: a small
// kernel then has to wait for a stitch workgroup to retire and displaces it (2.5 % fewer generations/s at config 2).
static unsigned stitch_lds_pad(int occ)
{
    if (const char* e = getenv("GEV_STITCH_LDS_PAD")) return (unsigned)atoi(e);       // experiments
    if (occ >= 8) return 0;
    const unsigned cu_lds = 160u * 1024u, own_min = 3072u, own_max = 3584u;            // the kernels' static LDS lies between these
    const unsigned per_wg = cu_lds / (unsigned)(occ + 1) + 1u;                          // occ + 1 of these exceed the CU
    const unsigned pad = (per_wg - own_min + 511u) / 512u * 512u;
    return std::min(pad, 64u * 1024u - own_max);
}
static const int OCC_CAND[] = {8, 7, 6};
// called at the end of every gev_reproduce: times the generations of each candidate and settles on the fastest
static void occ_tune_step(gev_ctx* c, size_t n_people)
{
    gev_ctx::OccTune& t = c->tune;
    if (!c->stitch_occ_auto || c->stitch_occ_env || !c->dense || c->serialize) return;
    const double now = host_ms(), delta = now - t.last;
    t.last = now;
    const bool resized = c->n_pop == 1 && t.people && (n_people > t.people + t.people / 4 || n_people + n_people / 4 < t.people);
    if (t.phase == 0 || resized || (t.phase == 2 && ++t.age >= 512)) {       // (re)start: first candidate, the first interval is a transition
        t.phase = 1; t.idx = 0; t.n = 0; t.best = 0; t.people = n_people; t.age = 0; c->stitch_occ = OCC_CAND[0];
        return;
    }
    if (t.phase != 1) return;
    const int GENS = 4;                                       // intervals per candidate; the first still contains the previous candidate's stitch
    if (t.n >= 1) t.cur_min = t.n == 1 ? delta : std::min(t.cur_min, delta);   // min: robust against a one-off host stall (a list buffer growing)
    if (++t.n < GENS) return;
    const bool better = t.idx == 0 || t.cur_min < t.best * 0.985;
    if (better) { t.best = t.cur_min; t.best_occ = OCC_CAND[t.idx]; }
    if (better && t.idx + 1 < (int)(sizeof OCC_CAND / sizeof OCC_CAND[0])) { t.idx++; t.n = 0; c->stitch_occ = OCC_CAND[t.idx]; return; }
    t.phase = 2; c->stitch_occ = t.best_occ;
    if (g_trace_host) fprintf(stderr, "[gev] stitch workgroups per CU: %d (best interval %.3f ms)\n", t.best_occ, t.best);
}
// the HBM-bound part, on stream_big, after the small work of the same generation: ONE launch over (parent, chromosome)
static int enqueue_stitch(gev_ctx* c, gev_ctx::Scratch& sc, int pop, size_t n_people)
{
    const int nchr = c->nchr;
    const size_t T = n_people * (size_t)nchr, rows = 2 * n_people;
    hipStream_t sb = c->stream_big;
    SampleDev sd = make_sd(c, sc, T);
    HIPC(hipStreamWaitEvent(sb, sc.ev_small_done, 0));        // recorded behind k_pool_assign: unit table, work list, sampling results and couples are complete
    HIPC(hipEventRecord(sc.t[4], sb));
    if (c->dense && sc.n_chrwork) {
        if (rows > 0x7fffffffull) return fail(GEV_EINVAL, "reproduce: stitch grid too large");
        if (c->stitch_mode == 0) {
            // persistent grid over the work list (its length is known to the device only): enough workgroups to fill every CU
            // one wave per entry when the list is as long as the last one; the kernel loops if it is longer
            size_t est = 0;
            for (int k = 0; k < nchr; k++) if (c->chr_active[k]) est = std::max<size_t>(est, c->pop[pop].st[k].pool_last_taken);
            if (!c->alias_rows || !est) est = rows * c->pop[pop].cs[0].nseg;
            const unsigned nblk = (unsigned)std::min<size_t>(std::max<size_t>(ceil_div(est + est / 8, 4), 256), c->stitch_grid);
            if (c->stitch_u == 4) hipLaunchKernelGGL((k_stitch_segments<true, 4>), dim3(nblk, sc.n_chrwork), dim3(256), stitch_lds_pad(c->stitch_occ ? c->stitch_occ : 8), sb, sc.chrwork.as<ChrWork>(), nchr, sd);
            else if (c->stitch_u == 1) hipLaunchKernelGGL((k_stitch_segments<true, 1>), dim3(nblk, sc.n_chrwork), dim3(256), stitch_lds_pad(c->stitch_occ ? c->stitch_occ : 8), sb, sc.chrwork.as<ChrWork>(), nchr, sd);
            else hipLaunchKernelGGL((k_stitch_segments<true, 2>), dim3(nblk, sc.n_chrwork), dim3(256), stitch_lds_pad(c->stitch_occ ? c->stitch_occ : 8), sb, sc.chrwork.as<ChrWork>(), nchr, sd);
        } else
            hipLaunchKernelGGL(k_stitch_rows, dim3((unsigned)rows, sc.n_chrwork), dim3(STITCH_THREADS), 0, sb, sc.chrwork.as<ChrWork>(), nchr, sd);
        KCHECK();
    }
    HIPC(hipEventRecord(sc.t[3], sb));
    HIPC(hipEventRecord(sc.ev_stitch_done, sb));
    HIPC(hipEventRecord(c->ev_planes, sb));
    sc.timing_pending = true; sc.stitch_pending = true; c->planes_pending = true;
    if (c->serialize) {                                  // no overlap: timings are final right away
        HIPC(hipStreamSynchronize(sb)); GEVC(harvest_timing(c, sc)); sc.stitch_pending = false;
        if (g_graveyard.bytes) g_graveyard.drain(c->device, false);
    }
    return GEV_OK;
}

// ---- Simulation::ras_glob_seed / Simulation::random_mate on the device (gev_mate.h) --------------------------------------------
// n ras_glob_seed() values into vals[0..n) and the engine state behind them into *state_out; the engine state in front comes from
// the host (state_val) or from a device word (state_ptr)
static int enqueue_glob(gev_ctx* c, hipStream_t st, u32 state_val, const u32* state_ptr, size_t n, u32* vals, u32* state_out, u32* flags)
{
    const u64 n_cand = (u64)n + n / 512 + 512;                  // expected rejections: n / 4444 (2147483646 - 2147000000 of 2147483646 outputs)
    const unsigned nb = (unsigned)ceil_div((size_t)n_cand, REJ_CHUNK);
    GEVC(c->d_globblk.ensure(nb * sizeof(u32), st));
    hipLaunchKernelGGL(k_glob_count, dim3(nb), dim3(256), 0, st, state_val, state_ptr, n_cand, c->d_globblk.as<u32>());
    hipLaunchKernelGGL(k_glob_emit, dim3(nb), dim3(256), 0, st, state_val, state_ptr, (u64)n, n_cand, c->d_globblk.as<u32>(), vals, state_out, flags);
    KCHECK();
    return GEV_OK;
}
// Simulation::random_mate of population P's current generation: pop_size couples into father / mother (physical parent rows, what
// the kernels of gev_reproduce read) and, when d_couples is given, as Couples_Info records; meta[0..1] = num_males_mate,
// num_females_mate; the seed (:2092) comes from the host (seed_val) or from a device word (seed_ptr)
static int enqueue_mate(gev_ctx* c, hipStream_t st, PopState& P, u32 seed_val, const u32* seed_ptr, const double* d_svf, size_t pop_size,
                        u32* father, u32* mother, gev_couple* d_couples, u32* meta, u32* flags)
{
    const size_t n_h = P.n_people;
    const u32* logical = nullptr;
    if (!P.logical.empty()) {                                   // after a cross-GPU migration positions are not rows
        GEVC(upload_table(c, c->d_logical, P.logical.data(), n_h * sizeof(u32), st));
        logical = c->d_logical.as<u32>();
    }
    const unsigned nb = (unsigned)ceil_div(n_h, MATE_CHUNK);
    GEVC(c->d_mflag.ensure(n_h, st)); GEVC(c->d_mblk.ensure(nb * sizeof(u32), st));
    GEVC(c->d_posm.ensure(n_h * sizeof(u32), st)); GEVC(c->d_posf.ensure(n_h * sizeof(u32), st));
    hipLaunchKernelGGL(k_mate_flags, dim3(nb), dim3(256), 0, st, P.d_sex[P.cur].as<uint8_t>(), logical, d_svf, n_h, seed_val, seed_ptr, c->d_mflag.as<uint8_t>(), c->d_mblk.as<u32>());
    hipLaunchKernelGGL(k_mate_compact, dim3(nb), dim3(256), 0, st, c->d_mflag.as<uint8_t>(), n_h, c->d_mblk.as<u32>(), nb, c->d_posm.as<u32>(), c->d_posf.as<u32>(), meta, flags);
    // candidates of the two pick streams: a draw is rejected with probability < n_h / 2^31
    const u64 n_cand = (u64)pop_size + (u64)(2.0 * (double)pop_size * (double)n_h / 2147483648.0) + 1024;
    const unsigned nbp = (unsigned)ceil_div((size_t)n_cand, REJ_CHUNK);
    GEVC(c->d_pickblk.ensure(2 * (size_t)nbp * sizeof(u32), st));
    hipLaunchKernelGGL(k_pick_count, dim3(nbp, 2), dim3(256), 0, st, seed_val, seed_ptr, (const u32*)meta, n_cand, c->d_pickblk.as<u32>());
    hipLaunchKernelGGL(k_pick_emit, dim3(nbp, 2), dim3(256), 0, st, seed_val, seed_ptr, (const u32*)meta, (u64)pop_size, n_cand, c->d_pickblk.as<u32>(),
                       c->d_posm.as<u32>(), c->d_posf.as<u32>(), logical, father, mother, d_couples, flags);
    KCHECK();
    return GEV_OK;
}

// One attempt of a generation: sampling (unless the head start covers it), lists + CV planes + unit table, the dense stitch on its
// own stream, A/D, and the status block on its way back -- everything enqueued, nothing waited for.  A generation begun with
// gev_generation_begin also draws its ras_glob_seed() values and forms its couples here, on the device.
static int enqueue_attempt(gev_ctx* c, int attempt)
{
    gev_ctx::PendingRepro& q = c->pend;
    gev_ctx::Scratch& sc = c->sc[c->gen_counter & 1];
    PopState& P = c->pop[q.pop];
    // Streams of one attempt.  S carries the chain the host waits for: seeds, sampling, unit table, CV planes, A/D, results.
    // X runs what needs neither the sampling nor (for the free list) the couples next to the sampling: Simulation::random_mate and
    // the free list of the segment pool.  L builds the mutation / interval lists, which nothing else of the generation reads, next
    // to the CV planes and A/D.  The dense stitch has its own stream as before.  All are joined into S before the status block
    // leaves.  Serialised mode (kernel timings without interference): everything on S.
    hipStream_t S = c->stream, X = (c->serialize || !c->side_streams) ? S : c->stream_aux, L = (c->serialize || !c->side_streams) ? S : c->stream_list;
    const size_t T = q.n_people * (size_t)c->nchr;
    q.th0 = host_ms();
    if (g_trace_host) HIPC(hipEventRecord(sc.tc[0], S));
    if (attempt == 0) GEVC(harvest_timing(c, sc));          // kernel times of the set's previous generation, before its events are recorded again
    u32* gv = sc.globvals.as<u32>(); u32* status = sc.status.as<u32>();
    const bool sampled = q.pre && attempt == 0;              // seeds drawn and sampling done by a head start
    if (q.fused && !sampled) {
        HIPC(hipMemsetAsync(sc.status.p, 0, q.n_status * sizeof(u32), S));
        // glob_generator's draws in the reference's order: random_mate (:2092), reproduce (:2398), then one per (offspring, chromosome) inside ras_add_mutation (:2500)
        GEVC(enqueue_glob(c, S, q.glob_state, nullptr, 2 + (q.has_mut ? T : 0), gv, status + ST_GLOB_STATE, status + ST_FLAGS));
    }
    // (the mating stream starts here: it needs the seeds, not the state behind them)
    if (X != S) { HIPC(hipEventRecord(sc.ev_fork, S)); HIPC(hipStreamWaitEvent(X, sc.ev_fork, 0)); }
    // overlap mode 2: only the ALU-bound sampling shares the GPU with the previous generation's stitch; everything latency-bound waits for it
    if (c->sparse_after_stitch && c->planes_pending) HIPC(hipStreamWaitEvent(X, c->ev_planes, 0));
    if (q.fused) {
        GEVC(enqueue_mate(c, X, P, 0u, gv, q.has_svf ? c->d_svf.as<double>() : nullptr, q.n_people, sc.father.as<u32>(), sc.mother.as<u32>(),
                          c->d_couples.as<gev_couple>(), status + ST_NM_MATE, status + ST_FLAGS));
    }
    if (q.fused && c->chain_draws >= 0) {                    // where glob_generator will stand when the host comes back for the next generation
        hipLaunchKernelGGL(k_glob_skip, dim3(1), dim3(64), 0, S, (const u32*)(status + ST_GLOB_STATE), (u32)c->chain_draws, status + ST_NEXT_STATE);
        KCHECK();
        HIPC(hipEventRecord(sc.ev_chain, S));
        // the NEXT generation's seeds and sampling go to their stream right away: they take most of a generation's time next to this
        // generation's chain, and the next generation's unit table cannot start before they are through
        if (attempt == 0 && c->head_start == 1) GEVC(enqueue_chain_head_start(c));
    }
    GEVC(enqueue_tables(c, sc, q.pop, q.n_people, S));        // (uploaded on S while X mates)
    if (X != S) { HIPC(hipEventRecord(sc.ev_tab, S)); HIPC(hipStreamWaitEvent(X, sc.ev_tab, 0)); }
    GEVC(enqueue_pool_free(c, sc, q.pop, q.n_people, X));
    if (X != S) HIPC(hipEventRecord(sc.ev_aux, X));
    if (q.fused && !sampled) GEVC(enqueue_sampling(c, sc, q.pop, q.n_people, q.has_mut, 0u, S, gv + 1, gv + 2, /*clear_status=*/false));
    else if (!q.fused && !sampled) GEVC(enqueue_sampling(c, sc, q.pop, q.n_people, q.has_mut, q.seed, S));
    q.th1 = host_ms();
    if (c->sparse_after_stitch && c->planes_pending && X != S) HIPC(hipStreamWaitEvent(S, c->ev_planes, 0));
    if (X != S) HIPC(hipStreamWaitEvent(S, sc.ev_aux, 0));
    if (sampled && q.fused && c->head_start == 1) HIPC(hipStreamWaitEvent(S, sc.ev_sampled, 0));   // crossovers and new mutations of this generation (head start)
    HIPC(hipEventRecord(sc.t[5], S));
    // (the counters of the unit table concern the stitch and the host only: with the stitch behind the small work they are
    // published from the list stream, one launch less on the chain the host waits for)
    const bool publish_aside = c->stitch_start >= 2 && L != S;
    GEVC(enqueue_pool_assign(c, sc, q.n_people, S, !publish_aside));
    // The dense stitch needs the sampling results, the couples and the unit table only.  It saturates HBM, and every latency-bound
    // kernel that runs next to it takes 2-5 times as long (and slows it down in turn): where it starts is a trade (stitch_start,
    // measured at config 2: behind the CV planes 711, behind the unit table 691, behind A/D 617 generations/s); whatever the
    // choice, the NEXT generation's ALU-bound sampling (head start) is what should share the GPU with it.
    HIPC(hipEventRecord(sc.ev_forked, S));
    if (c->stitch_start == 0) { HIPC(hipEventRecord(sc.ev_small_done, S)); GEVC(enqueue_stitch(c, sc, q.pop, q.n_people)); }
    // (the host enqueues slower than the device runs the first kernels of a generation: what the host waits for goes first, the
    // list kernels, which nothing of the generation reads, last)
    // the column counters are filled while the planes are written only if the A/D kernels that consume (and clear) them follow in this attempt
    const bool ad_now = c->eager_ad && c->pop[q.pop].cv[0][0].d_aptr.p;
    const bool count_cols = ad_now && sc.cv_count_fused && sc.n_cvwork && sc.cv_used_max;
    GEVC(enqueue_cv_planes(c, sc, q.n_people, q.has_mut, count_cols, S));
    HIPC(hipEventRecord(sc.t[2], S));
    if (c->stitch_start == 1) { HIPC(hipEventRecord(sc.ev_small_done, S)); GEVC(enqueue_stitch(c, sc, q.pop, q.n_people)); }
    q.th2 = host_ms();
    if (c->ad_cached_pop != q.pop) c->ad_cached_pop = -1;    // (the device-side arrays are about to be rewritten; the published values of q.pop stay readable in their pinned buffer)
    c->ad_host_set_pop = -1;
    if (ad_now) GEVC(enqueue_ad(c, q.pop, c->pop[q.pop].cur ^ 1, q.n_people, count_cols, (int)(c->gen_counter & 1)));   // Simulation::ras_compute_AD always follows (src/Simulation.cpp:1935)
    if (g_trace_host) HIPC(hipEventRecord(sc.tc[1], S));
    if (L != S) HIPC(hipStreamWaitEvent(L, sc.ev_forked, 0));
    if (publish_aside) GEVC(enqueue_pool_publish(c, sc, q.n_people, L));
    // Human::sex of the new generation (:2472) for the next gev_random_mate; a fused generation also sends them to the host with the status block
    HIPC(hipMemcpyAsync(P.d_sex[P.cur ^ 1].p, sc.sex.p, q.n_people, hipMemcpyDeviceToDevice, L));
    if (q.fused) { HIPC(hipMemcpyAsync(q.hsex, sc.sex.p, q.n_people, hipMemcpyDeviceToHost, L)); HIPC(hipMemcpyAsync(q.hseeds2, gv, 2 * sizeof(u32), hipMemcpyDeviceToHost, L)); }   // (copies nothing on the device waits for: off the main stream)
    GEVC(enqueue_lists(c, sc, q.n_people, q.has_mut, L));
    if (L != S) HIPC(hipEventRecord(sc.ev_lists, L));
    if (L != S) HIPC(hipStreamWaitEvent(S, sc.ev_lists, 0));
    if (g_trace_host) HIPC(hipEventRecord(sc.tc[2], S));
    if (c->stitch_start >= 2) { HIPC(hipEventRecord(sc.ev_small_done, S)); GEVC(enqueue_stitch(c, sc, q.pop, q.n_people)); }
    HIPC(hipMemcpyAsync(q.hstatus, sc.status.p, q.n_status * sizeof(u32), hipMemcpyDeviceToHost, S));
    HIPC(hipEventRecord(sc.ev_status, S));                  // gev_reproduce_end waits for THIS, not for whatever a head start queued behind it
    sc.th_enq = host_ms();
    return GEV_OK;
}
static int ensure_stage(gev_ctx* c, size_t bytes)
{
    if (c->h_stage_bytes >= bytes) return GEV_OK;
    HIPC(hipDeviceSynchronize());
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    c->h_stage = nullptr; c->h_stage_bytes = 0;
    HIPC(hipHostMalloc(&c->h_stage, bytes * 5 / 4 + 4096, hipHostMallocDefault));
    c->h_stage_bytes = bytes * 5 / 4 + 4096;
    return GEV_OK;
}
// gev_reproduce in two halves: _begin checks and stages the inputs and enqueues the generation's device work, _end waits for it,
// repeats it with larger buffers if a capacity was exceeded, and publishes the new generation.  Between the two the host is free
// (the bench forms the next generation's couples there); no other call on the context is allowed in between.
int gev_reproduce_begin(gev_ctx* c, int pop, const gev_couple* couples, size_t n_couples, uint32_t seed_reproduce,
                        const uint32_t* mut_seeds, size_t n_mut_seeds, size_t n_people)
{
    GEVC(check_idx(c, pop, 0));
    PopState& P = c->pop[pop];
    if (!P.gen0) return fail(GEV_ESTATE, "reproduce: population %d has no current generation (call gev_init_gen0)", pop);
    HIPC(hipSetDevice(c->device));
    const int nchr = c->nchr;
    const size_t T = n_people * (size_t)nchr;
    if (n_people == 0) return fail(GEV_EINVAL, "reproduce: no offspring");
    if (2 * T * GEV_BK_CAP >= 0xf0000000ull) return fail(GEV_EINVAL, "reproduce: too many gametes");
    const bool has_mut = mut_seeds != nullptr;
    if (has_mut && n_mut_seeds != T) return fail(GEV_EINVAL, "reproduce: n_mut_seeds=%zu, expected n_people*nchr=%zu", n_mut_seeds, T);
    if (!has_mut) GEVC(check_chain_size(T));
    gev_ctx::Scratch& sc = c->sc[c->gen_counter & 1];
    // couples == NULL: the couples gev_random_mate left on the device for this population
    const bool dev_couples = couples == nullptr;
    if (dev_couples && !(sc.mated && sc.mate_pop == pop && sc.mate_n == n_people))
        return fail(GEV_ESTATE, "reproduce: couples is NULL but no gev_random_mate of population %d for %zu offspring precedes", pop, n_people);
    // pinned staging: [father | mother | mut_seeds | status], written by the host, copied asynchronously
    const size_t n_status = ST_TOTALS + ST_PER_CHR * (size_t)nchr;
    const size_t stage_words = 2 * n_people + (has_mut ? T : 0) + n_status;
    GEVC(ensure_stage(c, stage_words * 4));
    u32* father = (u32*)c->h_stage; u32* mother = father + n_people; u32* hseeds = mother + n_people; u32* hstatus = hseeds + (has_mut ? T : 0);
    // offspring enumeration order of the couple loop (src/Simulation.cpp:2433-2443)
    size_t ip = 0;
    const u32* lg = P.logical.empty() ? nullptr : P.logical.data();
    for (size_t it = 0; it < n_couples && !dev_couples; it++) {
        if (couples[it].inbreed) continue;
        if (couples[it].num_offspring < 0) return fail(GEV_EINVAL, "reproduce: couple %zu has negative num_offspring", it);
        if (couples[it].num_offspring && (couples[it].pos_male >= P.n_people || couples[it].pos_female >= P.n_people))
            return fail(GEV_EINVAL, "reproduce: couple %zu references position beyond the population (%zu people)", it, P.n_people);
        for (int ns = 0; ns < couples[it].num_offspring; ns++) {
            if (ip >= n_people) return fail(GEV_EINVAL, "reproduce: n_people=%zu but the couples list yields more offspring", n_people);
            father[ip] = lg ? lg[couples[it].pos_male] : (u32)couples[it].pos_male;
            mother[ip] = lg ? lg[couples[it].pos_female] : (u32)couples[it].pos_female; ip++;
        }
    }
    if (!dev_couples && ip != n_people) return fail(GEV_EINVAL, "reproduce: n_people=%zu but the couples list yields %zu offspring", n_people, ip);
    GEVC(finalize_static(c, pop));
    GEVC(prepare_eager_ad(c, pop));
    GEVC(ensure_capacity(c, pop, n_people));
    hipStream_t st = c->stream;
    // sampling already enqueued by gev_presample for exactly these inputs?
    const bool pre = sc.presampled && sc.ps_pop == pop && sc.ps_seed == (u32)seed_reproduce && sc.ps_n_people == n_people && sc.ps_has_mut == has_mut &&
                     (!has_mut || (c->h_seeds && memcmp(c->h_seeds, mut_seeds, T * sizeof(u32)) == 0));
    sc.presampled = false; sc.ps_stale = false; sc.mated = false; sc.fused_ahead = false; sc.fa_dropped = false; c->chain_valid = false;
    if (pre) HIPC(hipStreamWaitEvent(st, sc.ev_sampled, 0));     // the head start ran on its own stream
    else HIPC(hipStreamSynchronize(c->stream_samp));            // a head start that does not match must not write into the set any more
    if (!pre) {
        // the stitch that last read this scratch set must be over before the set is refilled
        GEVC(harvest_timing(c, sc));
        if (sc.stitch_pending) { HIPC(hipStreamWaitEvent(st, sc.ev_stitch_done, 0)); sc.stitch_pending = false; }
        GEVC(ensure_scratch(c, sc, n_people, has_mut));
        if (has_mut) { memcpy(hseeds, mut_seeds, T * sizeof(u32)); HIPC(hipMemcpyAsync(sc.mutseeds.p, hseeds, T * sizeof(u32), hipMemcpyHostToDevice, st)); }
    }
    if (!dev_couples) {
        HIPC(hipMemcpyAsync(sc.father.p, father, n_people * sizeof(u32), hipMemcpyHostToDevice, st));
        HIPC(hipMemcpyAsync(sc.mother.p, mother, n_people * sizeof(u32), hipMemcpyHostToDevice, st));
    }

    gev_ctx::PendingRepro& q = c->pend;
    q.pop = pop; q.n_people = n_people; q.has_mut = has_mut; q.pre = pre; q.seed = (u32)seed_reproduce; q.attempt = 0; q.n_status = n_status; q.hstatus = hstatus;
    q.fused = false; q.has_svf = false; q.pool_rebuilt = false;
    GEVC(enqueue_attempt(c, 0));
    q.active = true;
    return GEV_OK;
}
static int enqueue_chain_head_start(gev_ctx* c)
{
    gev_ctx::PendingRepro& q = c->pend;
    gev_ctx::Scratch& sc = c->sc[c->gen_counter & 1]; gev_ctx::Scratch& nx = c->sc[(c->gen_counter + 1) & 1];
    hipStream_t SS = c->serialize ? c->stream : c->stream_samp;
    const size_t T = q.n_people * (size_t)c->nchr;
    // the stitch that last read the other scratch set (the previous generation's) may still be running: the sampling stream waits for
    // it, the host does not (its kernel time is collected when the set's next generation is enqueued)
    GEVC(harvest_sampling_time(nx));
    if (nx.stitch_pending) HIPC(hipStreamWaitEvent(SS, nx.ev_stitch_done, 0));
    HIPC(hipStreamWaitEvent(SS, sc.ev_chain, 0));
    GEVC(ensure_scratch(c, nx, q.n_people, q.has_mut));
    GEVC(nx.globvals.ensure((2 + T) * sizeof(u32), SS));
    u32* gv = nx.globvals.as<u32>(); u32* status = nx.status.as<u32>();
    HIPC(hipMemsetAsync(nx.status.p, 0, q.n_status * sizeof(u32), SS));
    GEVC(enqueue_glob(c, SS, 0u, sc.status.as<u32>() + ST_NEXT_STATE, 2 + (q.has_mut ? T : 0), gv, status + ST_GLOB_STATE, status + ST_FLAGS));
    HIPC(hipEventRecord(nx.ev_seeded, SS));               // status block cleared, seeds drawn: all that the next generation's mating needs
    GEVC(enqueue_sampling(c, nx, q.pop, q.n_people, q.has_mut, 0u, SS, gv + 1, gv + 2, /*clear_status=*/false));
    HIPC(hipEventRecord(nx.ev_sampled, SS));
    nx.fused_ahead = true; nx.fa_dropped = false; nx.fa_pop = q.pop; nx.fa_n = q.n_people; nx.fa_has_mut = q.has_mut;
    return GEV_OK;
}
// A/D is computed with the generation (same enqueue, same wait) once the per-population a/d tables exist; they are set up by the
// first gev_compute_ad, or here as soon as every population of the context has its static inputs
static int prepare_eager_ad(gev_ctx* c, int pop)
{
    if (c->pop[pop].cv[0][0].d_aptr.p) return GEV_OK;
    for (const PopState& Q : c->pop)
        for (int k = 0; k < c->nchr; k++) {
            if (Q.cs[k].rbp.empty() || (c->chr_active[k] && Q.cs[k].pos.empty() && Q.cs[k].L)) return GEV_OK;
            for (int p = 0; p < c->nphen; p++) if (c->chr_active[k] && !Q.cv[p][k].set) return GEV_OK;
        }
    return check_multipop(c);
}
// does every chromosome of the population have a mutation map (Simulation::reproduce's `_mutation_map.size() > 0`, :2459)?
// Without a mutation map every gamete's seed is a rand() of the srand() of the gamete before it (src/Simulation.cpp:2447-2455):
// the 2 T gametes form one serial chain, a few microseconds each on one workgroup (k_rec_chain_wg).  Beyond GEV_CHAIN_MAX_TASKS
// (default 4 000 000 tasks: tens of seconds per generation) the call refuses instead of running for minutes.
static int check_chain_size(size_t T)
{
    const size_t chain_max = getenv("GEV_CHAIN_MAX_TASKS") ? (size_t)atoll(getenv("GEV_CHAIN_MAX_TASKS")) : (size_t)4000000;
    if (T > chain_max)
        return fail(GEV_EUNSUPPORTED, "Error: %zu (offspring, chromosome) tasks without a mutation map form one serial rand() chain; more than %zu are refused (GEV_CHAIN_MAX_TASKS) -- give a mutation map (rates may be 0) to sample the tasks in parallel", T, chain_max);
    return GEV_OK;
}
static int population_has_mutmap(gev_ctx* c, int pop, bool& has_mut)
{
    PopState& P = c->pop[pop];
    has_mut = P.cs[0].mut_set;
    for (int k = 1; k < c->nchr; k++)
        if (P.cs[k].mut_set != has_mut) return fail(GEV_ESTATE, "population %d: a mutation map is set for some chromosomes only", pop);
    return GEV_OK;
}
// One generation of a randomly mating population in one piece: Simulation::random_mate -> Simulation::reproduce ->
// Simulation::ras_compute_AD (the body of sim_next_generation, src/Simulation.cpp:1907-1935) with every ras_glob_seed() value
// the three draw -- 1 (:2092) + 1 (:2398) + n_people * nchr (:2500) -- taken from glob_generator's state ON THE DEVICE.
int gev_generation_begin(gev_ctx* c, int pop, uint32_t glob_state, size_t pop_size, const double* selection_value_func)
{
    GEVC(check_idx(c, pop, 0));
    PopState& P = c->pop[pop];
    if (!P.gen0) return fail(GEV_ESTATE, "generation: population %d has no current generation (call gev_init_gen0)", pop);
    if (glob_state == 0 || glob_state >= GEV_M31) return fail(GEV_EINVAL, "generation: %u is not a state of std::minstd_rand0 (1 .. 2^31-2)", glob_state);
    HIPC(hipSetDevice(c->device));
    const int nchr = c->nchr;
    const size_t n_people = pop_size, T = n_people * (size_t)nchr;
    if (n_people == 0) return fail(GEV_EINVAL, "generation: no offspring");
    if (2 * T * GEV_BK_CAP >= 0xf0000000ull) return fail(GEV_EINVAL, "generation: too many gametes");
    bool has_mut = false;
    GEVC(population_has_mutmap(c, pop, has_mut));
    if (!has_mut) GEVC(check_chain_size(T));
    const size_t n_status = ST_TOTALS + ST_PER_CHR * (size_t)nchr;
    // pinned: [status | the two seeds | sexes]
    GEVC(ensure_stage(c, (n_status + 2) * 4 + n_people + 16));
    u32* hstatus = (u32*)c->h_stage;
    GEVC(finalize_static(c, pop));
    GEVC(prepare_eager_ad(c, pop));
    GEVC(ensure_capacity(c, pop, n_people));
    hipStream_t st = c->stream;
    gev_ctx::Scratch& sc = c->sc[c->gen_counter & 1];
    // seeds already drawn and sampling already enqueued for exactly this call (gev_set_generation_chain)?
    const bool pre = sc.fused_ahead && c->chain_valid && c->chain_state == (u32)glob_state && sc.fa_pop == pop && sc.fa_n == n_people && sc.fa_has_mut == has_mut;
    if (sc.presampled || sc.fa_dropped || (sc.fused_ahead && !pre)) HIPC(hipStreamSynchronize(c->stream_samp));     // a head start that does not match must not write into the set any more
    if (sc.fused_ahead) { if (pre) c->chain_hits++; else c->chain_misses++; }
    sc.presampled = false; sc.ps_stale = false; sc.mated = false; sc.fused_ahead = false; sc.fa_dropped = false; c->chain_valid = false;
    // the head start ran on its own stream: mating needs its seeds, the unit table its crossovers (waited for in enqueue_attempt)
    if (pre) HIPC(hipStreamWaitEvent(st, c->head_start == 1 ? sc.ev_seeded : sc.ev_sampled, 0));
    else {
        GEVC(harvest_timing(c, sc));
        if (sc.stitch_pending) { HIPC(hipStreamWaitEvent(st, sc.ev_stitch_done, 0)); sc.stitch_pending = false; }
        GEVC(ensure_scratch(c, sc, n_people, has_mut));
        GEVC(sc.globvals.ensure((2 + T) * sizeof(u32), st));
    }
    GEVC(c->d_couples.ensure(n_people * sizeof(gev_couple), st));
    if (selection_value_func) {
        GEVC(c->d_svf.ensure(P.n_people * sizeof(double), st));
        HIPC(hipMemcpyAsync(c->d_svf.p, selection_value_func, P.n_people * sizeof(double), hipMemcpyHostToDevice, st));
        HIPC(hipStreamSynchronize(st));                          // the caller's array is pageable memory of unknown lifetime
    }
    gev_ctx::PendingRepro& q = c->pend;
    q.pop = pop; q.n_people = n_people; q.has_mut = has_mut; q.pre = pre; q.seed = 0; q.attempt = 0; q.n_status = n_status; q.hstatus = hstatus;
    q.pool_rebuilt = false;
    q.fused = true; q.has_svf = selection_value_func != nullptr; q.glob_state = glob_state; q.hseeds2 = hstatus + n_status; q.hsex = (uint8_t*)(hstatus + n_status + 2);
    GEVC(enqueue_attempt(c, 0));
    q.active = true;
    if (c->chain_draws >= 0 && c->head_start == 0) GEVC(enqueue_chain_head_start(c));
    return GEV_OK;
}
// The host announced how many ras_glob_seed() values it draws itself between two generations (gev_set_generation_chain): the state
// glob_generator will have at the next gev_generation_begin is then known on the device as soon as this generation's seeds are
// drawn.  Draw the NEXT generation's seeds from it and run its sampling now, on the head-start stream, into the other scratch
// set: it shares the GPU with this generation's dense stitch (ALU-bound next to HBM-bound) instead of standing in front of the
// next generation's small work.  The next gev_generation_begin recognises it by the engine state it is handed.
int gev_set_generation_chain(gev_ctx* c, int draws_between)
{
    GEVC(check_not_pending(c));
    if (draws_between > 1000000) return fail(GEV_EINVAL, "set_generation_chain: %d draws between two generations", draws_between);
    c->chain_draws = draws_between < 0 ? -1 : draws_between;
    return GEV_OK;
}
static int generation_finish_inner(gev_ctx* c, uint8_t* sex_out, gev_generation_result* res, gev_couple* couples_out);
static int generation_finish(gev_ctx* c, uint8_t* sex_out, gev_generation_result* res, gev_couple* couples_out)
{
    const int rc = generation_finish_inner(c, sex_out, res, couples_out);
    if (rc != GEV_OK) c->ad_cached_pop = -1;                 // the device-side A/D arrays hold a generation that was not published
    return rc;
}
static int generation_finish_inner(gev_ctx* c, uint8_t* sex_out, gev_generation_result* res, gev_couple* couples_out)
{
    gev_ctx::PendingRepro& q = c->pend;
    q.active = false;                                       // whatever happens below, the generation is no longer pending
    HIPC(hipSetDevice(c->device));
    const int pop = q.pop, nchr = c->nchr; const size_t n_people = q.n_people;
    PopState& P = c->pop[pop];
    hipStream_t st = c->stream;
    gev_ctx::Scratch& sc = c->sc[c->gen_counter & 1];
    u32* hstatus = q.hstatus;
    const int alt = P.cur ^ 1;
    for (int attempt = q.attempt;; attempt++) {
        if (attempt > 0) GEVC(enqueue_attempt(c, attempt));
        const double th_w0 = host_ms();
        HIPC(hipEventSynchronize(sc.ev_status));
        const u32 flags = hstatus[ST_FLAGS];
        if (g_trace_host) {
            static double last_status = 0; const double now = host_ms();
            float a = 0, b = 0, d = 0, e = 0;
            (void)hipEventSynchronize(sc.tc[2]); (void)hipEventElapsedTime(&a, sc.tc[0], sc.t[5]); (void)hipEventElapsedTime(&b, sc.t[5], sc.t[2]); (void)hipEventElapsedTime(&d, sc.t[2], sc.tc[1]); (void)hipEventElapsedTime(&e, sc.tc[1], sc.tc[2]);
            fprintf(stderr, "[gev] gen %u host: since last status %.3f = to first launch %.3f + enqueue %.3f + host idle %.3f + wait %.3f | device chain: to unit table %.3f, unit table + CV planes %.3f, A/D %.3f, lists after A/D %.3f\n",
                    c->gen_counter, now - last_status, q.th0 - last_status, sc.th_enq - q.th0, th_w0 - sc.th_enq, now - th_w0, a, b, d, e);
            last_status = now;
        }
        if (g_trace_host) fprintf(stderr, "[gev] gen %u attempt %d pre %d: enqueue sampling %.2f ms, sparse %.2f ms, A/D + wait %.2f ms, flags %u, graveyard %.1f MiB\n",
                                  c->gen_counter, attempt, (int)q.pre, q.th1 - q.th0, q.th2 - q.th1, host_ms() - q.th2, flags, g_graveyard.bytes / 1048576.0);
        if (g_trace_host && g_malloc_n) { fprintf(stderr, "[gev]   %zu hipMalloc calls, %.1f MiB, %.2f ms\n", g_malloc_n, g_malloc_bytes / 1048576.0, g_malloc_ms); g_malloc_ms = 0; g_malloc_n = 0; g_malloc_bytes = 0; }
        for (int k = 0; k < nchr; k++) { P.st[k].lp.m_last = hstatus[ST_TOTALS + ST_PER_CHR * k]; P.st[k].lp.p_last = hstatus[ST_TOTALS + ST_PER_CHR * k + 1]; }   // entries appended to the arenas
        if (c->dense) for (int k = 0; k < nchr; k++) if (c->chr_active[k]) {
            ChrState& cs = P.st[k]; const u32* stw = hstatus + ST_TOTALS + ST_PER_CHR * k;
            cs.pool_n_free = stw[4]; cs.pool_cursor = stw[5]; cs.pool_last_taken = stw[2];
        }
        if ((flags & FLAG_POOL) && q.pool_rebuilt) return fail(GEV_EDEVICE, "reproduce: genotype row pool exhausted (internal error)");
        if (flags & FLAG_RNG_SHORT) return fail(GEV_EDEVICE, "generation: a rejection stream ran out of candidates (internal error)");
        if (flags & FLAG_NO_MATES)                          // Simulation::random_mate returns false (:2125-2129); nothing is published
            return fail(GEV_ENOMATE, "Error: No one can marry, num_males_mate=%u, num_females_mate=%u", hstatus[ST_NM_MATE], hstatus[ST_NF_MATE]);
        if (!(flags & (FLAG_REDO_MASK | FLAG_POOL))) break;
        if (flags & FLAG_POOL) { for (int k = 0; k < nchr; k++) P.st[k].pool_force_rebuild = true; q.pool_rebuilt = true; }   // the free list ran out: rebuild it, then the generation again
        if (attempt == 3) return fail(GEV_EDEVICE, "reproduce: buffers still too small after %d attempts (flags %u)", attempt + 1, flags);
        HIPC(hipStreamSynchronize(c->stream_big));        // the stitch of the failed attempt still reads the records that are sampled again below
        sc.timing_pending = false; sc.stitch_pending = false;   // (its kernel times are not counted)
        {   // a head start taken meanwhile used the old record capacities: it is sampled again (by gev_presample_sex from the retained
            // inputs, or by the next gev_reproduce_begin)
            gev_ctx::Scratch& nx = c->sc[(c->gen_counter + 1) & 1];
            if (nx.presampled) { nx.presampled = false; nx.ps_stale = true; }
            if (nx.fused_ahead) { nx.fused_ahead = false; nx.fa_dropped = true; }
        }
        if (flags & FLAG_BK_OVF) c->bk_ovf_cap = std::max<size_t>(2 * c->bk_ovf_cap, (size_t)hstatus[ST_BK_OVF_USED] * 5 / 4 + 1024);
        if (flags & FLAG_NM_OVF) c->nm_ovf_cap = std::max<size_t>(2 * c->nm_ovf_cap, (size_t)hstatus[ST_NM_OVF_USED] * 5 / 4 + 1024);
        if (flags & FLAG_PARTS_CAP) for (int k = 0; k < nchr; k++) P.st[k].lp.grow_p = c->chr_active[k] && c->track_intervals;   // an arena was full: doubled before the next attempt
        if (flags & FLAG_MUT_CAP) for (int k = 0; k < nchr; k++) P.st[k].lp.grow_m = c->chr_active[k];
        c->redo_count++;
    }
    for (int k = 0; k < nchr; k++) if (c->chr_active[k]) {      // the offspring's pieces are the state now; the CSR form is made when somebody asks for it
        ChrState::LpState& lp = P.st[k].lp;
        lp.p_used += lp.p_last; lp.m_used += lp.m_last; P.st[k].csr_valid = false;
    }
    if (c->dense) for (int k = 0; k < nchr; k++) if (c->chr_active[k]) {            // 16-byte chunks the stitch wrote / chunks of the generation's rows
        const ChrStatic& S = P.cs[k];
        const unsigned long long units = hstatus[ST_TOTALS + ST_PER_CHR * k + 2], last = hstatus[ST_TOTALS + ST_PER_CHR * k + 3];
        const unsigned long long chunks = S.stride / 16, seg = 1ull << S.seg_shift, last_chunks = chunks - (S.nseg - 1) * seg;
        c->chunks_written_sum += (units - last) * seg + last * last_chunks; c->chunks_total_sum += 2ull * n_people * chunks;
        c->segments_written_sum += units; c->segments_total_sum += 2ull * n_people * S.nseg;
    }
    const bool ad_done = c->eager_ad && c->pop[pop].cv[0][0].d_aptr.p;
    for (int p = 0; p < c->nphen; p++) for (int k = 0; k < nchr; k++) P.cv[p][k].frq_valid = ad_done;
    c->ad_cached_pop = ad_done ? pop : -1;
    if (q.fused) {
        c->chain_valid = c->chain_draws >= 0; c->chain_state = hstatus[ST_NEXT_STATE];
        if (sex_out) memcpy(sex_out, q.hsex, n_people);
        if (res) {
            res->glob_state = hstatus[ST_GLOB_STATE]; res->seed_mate = q.hseeds2[0]; res->seed_reproduce = q.hseeds2[1]; res->reserved = 0;
            res->num_males_mate = hstatus[ST_NM_MATE]; res->num_females_mate = hstatus[ST_NF_MATE];
        }
        if (couples_out) { HIPC(hipMemcpyAsync(couples_out, c->d_couples.p, n_people * sizeof(gev_couple), hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st)); }
    } else if (sex_out) { HIPC(hipMemcpyAsync(sex_out, sc.sex.p, n_people, hipMemcpyDeviceToHost, st)); HIPC(hipStreamSynchronize(st)); }
    P.cur = alt; P.pcur = (P.pcur + 1) % 3; P.n_people = n_people; P.n_phys = n_people; P.logical.clear();
    c->gen_counter++;
    occ_tune_step(c, n_people);
    return GEV_OK;
}
int gev_reproduce_end(gev_ctx* c, uint8_t* sex_out)
{
    if (!c) return fail(GEV_EINVAL, "null context");
    if (!c->pend.active) return fail(GEV_ESTATE, "reproduce_end: no gev_reproduce_begin is pending");
    if (c->pend.fused) return fail(GEV_ESTATE, "reproduce_end: the pending generation was begun with gev_generation_begin: call gev_generation_end");
    return generation_finish(c, sex_out, nullptr, nullptr);
}
int gev_generation_end(gev_ctx* c, gev_generation_result* res, gev_couple* couples_out, uint8_t* sex_out)
{
    if (!c) return fail(GEV_EINVAL, "null context");
    if (!c->pend.active) return fail(GEV_ESTATE, "generation_end: no gev_generation_begin is pending");
    if (!c->pend.fused) return fail(GEV_ESTATE, "generation_end: the pending generation was begun with gev_reproduce_begin: call gev_reproduce_end");
    return generation_finish(c, sex_out, res, couples_out);
}
int gev_reproduce(gev_ctx* c, int pop, const gev_couple* couples, size_t n_couples, uint32_t seed_reproduce,
                  const uint32_t* mut_seeds, size_t n_mut_seeds, size_t n_people, uint8_t* sex_out)
{
    GEVC(gev_reproduce_begin(c, pop, couples, n_couples, seed_reproduce, mut_seeds, n_mut_seeds, n_people));
    return gev_reproduce_end(c, sex_out);
}
// Simulation::random_mate (src/Simulation.cpp:2090-2157) of the population's current generation on the device.  The couples stay
// on the device for the next gev_reproduce of the population (couples == NULL there) and are copied out when couples_out is given.
int gev_random_mate(gev_ctx* c, int pop, uint32_t seed, const double* selection_value_func, size_t pop_size, gev_couple* couples_out,
                    size_t* num_males_mate, size_t* num_females_mate)
{
    GEVC(check_idx(c, pop, 0));
    PopState& P = c->pop[pop];
    if (!P.gen0) return fail(GEV_ESTATE, "random_mate: population %d has no current generation", pop);
    if (pop_size == 0) return fail(GEV_EINVAL, "random_mate: no couples asked for");
    if (pop_size >= 0x7fffffffull) return fail(GEV_EINVAL, "random_mate: too many couples");
    HIPC(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    gev_ctx::Scratch& sc = c->sc[c->gen_counter & 1];
    // father / mother of this scratch set were last read by the stitch two generations back
    GEVC(harvest_timing(c, sc));
    if (sc.stitch_pending) { HIPC(hipStreamWaitEvent(st, sc.ev_stitch_done, 0)); sc.stitch_pending = false; }
    GEVC(sc.father.ensure(pop_size * sizeof(u32), st)); GEVC(sc.mother.ensure(pop_size * sizeof(u32), st));
    GEVC(c->d_couples.ensure(pop_size * sizeof(gev_couple), st));
    GEVC(c->d_mstat.ensure(4 * sizeof(u32), st));
    HIPC(hipMemsetAsync(c->d_mstat.p, 0, 4 * sizeof(u32), st));
    const double* d_svf = nullptr;
    if (selection_value_func) {
        GEVC(c->d_svf.ensure(P.n_people * sizeof(double), st));
        HIPC(hipMemcpyAsync(c->d_svf.p, selection_value_func, P.n_people * sizeof(double), hipMemcpyHostToDevice, st));
        d_svf = c->d_svf.as<double>();
    }
    u32* ms = c->d_mstat.as<u32>();                             // {num_males_mate, num_females_mate, flags}
    GEVC(enqueue_mate(c, st, P, (u32)seed, nullptr, d_svf, pop_size, sc.father.as<u32>(), sc.mother.as<u32>(), couples_out ? c->d_couples.as<gev_couple>() : nullptr, ms, ms + 2));
    u32 h[4] = {0, 0, 0, 0};
    HIPC(hipMemcpyAsync(h, ms, sizeof h, hipMemcpyDeviceToHost, st));
    if (couples_out) HIPC(hipMemcpyAsync(couples_out, c->d_couples.p, pop_size * sizeof(gev_couple), hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    if (num_males_mate) *num_males_mate = h[0];
    if (num_females_mate) *num_females_mate = h[1];
    sc.mated = false;
    if (h[2] & FLAG_NO_MATES) return fail(GEV_ENOMATE, "Error: No one can marry, num_males_mate=%u, num_females_mate=%u", h[0], h[1]);
    if (h[2] & FLAG_RNG_SHORT) return fail(GEV_EDEVICE, "random_mate: a rejection stream ran out of candidates (internal error)");
    sc.mated = true; sc.mate_pop = pop; sc.mate_n = pop_size;
    return GEV_OK;
}
// Simulation::ras_glob_seed (src/Simulation.cpp:17-21) called n times, evaluated on the device: *engine_state = the state of
// glob_generator (std::minstd_rand0) before, and after on return; out (host, n values) may be NULL.
int gev_glob_seeds(gev_ctx* c, uint32_t* engine_state, size_t n, uint32_t* out)
{
    if (!c || !engine_state) return fail(GEV_EINVAL, "glob_seeds: null argument");
    if (c->pend.active) return fail(GEV_ESTATE, "a gev_reproduce_begin is pending: call gev_reproduce_end first");
    if (*engine_state == 0 || *engine_state >= GEV_M31) return fail(GEV_EINVAL, "glob_seeds: %u is not a state of std::minstd_rand0 (1 .. 2^31-2)", *engine_state);
    if (n == 0) return GEV_OK;
    if (n >= 0x7fffffffull) return fail(GEV_EINVAL, "glob_seeds: too many values");
    HIPC(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    GEVC(c->d_tmp.ensure((n + 4) * sizeof(u32), st));
    u32* vals = c->d_tmp.as<u32>() + 4; u32* stat = c->d_tmp.as<u32>();       // stat = {state, flags}
    HIPC(hipMemsetAsync(stat, 0, 16, st));
    GEVC(enqueue_glob(c, st, *engine_state, nullptr, n, vals, stat, stat + 1));
    u32 h[2] = {0, 0};
    HIPC(hipMemcpyAsync(h, stat, sizeof h, hipMemcpyDeviceToHost, st));
    if (out) HIPC(hipMemcpyAsync(out, vals, n * sizeof(u32), hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    if (h[1]) return fail(GEV_EDEVICE, "glob_seeds: the rejection stream ran out of candidates (internal error)");
    *engine_state = h[0];
    return GEV_OK;
}
// The sexes of the generation whose sampling gev_presample has enqueued (they come out of the rand() chain of the sampling
// kernels, src/Simulation.cpp:2472, and need nothing else): lets a host that mates at random form the NEXT couples while the
// device still works on this generation.  Waits for the sampling kernels only.
int gev_presample_sex(gev_ctx* c, int pop, uint8_t* sex_out, size_t n_people)
{
    GEVC(check_idx(c, pop, 0));
    if (!sex_out) return fail(GEV_EINVAL, "presample_sex: null output");
    gev_ctx::Scratch& sc = c->sc[c->gen_counter & 1];
    if (!sc.presampled && sc.ps_stale && sc.ps_pop == pop && sc.ps_n_people == n_people) {
        // the head start was dropped by a redo of the previous generation (larger record regions): sample again from the retained inputs
        sc.ps_stale = false;
        GEVC(gev_presample(c, pop, sc.ps_seed, sc.ps_has_mut ? (const uint32_t*)c->h_seeds : nullptr, sc.ps_has_mut ? n_people * (size_t)c->nchr : 0, n_people));
    }
    if (!sc.presampled || sc.ps_pop != pop || sc.ps_n_people != n_people) return fail(GEV_ESTATE, "presample_sex: no matching gev_presample is pending");
    HIPC(hipSetDevice(c->device));
    hipStream_t st = c->serialize ? c->stream : c->stream_samp;
    HIPC(hipMemcpyAsync(sex_out, sc.sex.p, n_people, hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    return GEV_OK;
}
// Enqueue the sampling kernels of the NEXT gev_reproduce of `pop` now (they need the seeds and the offspring count, not the
// couples) and return at once: the host can form the couples while the GPU samples.  gev_reproduce recognises the
// same (seed, mutation seeds, n_people) and skips its own sampling; anything else simply samples again.
int gev_presample(gev_ctx* c, int pop, uint32_t seed_reproduce, const uint32_t* mut_seeds, size_t n_mut_seeds, size_t n_people)
{
    if (!c) return fail(GEV_EINVAL, "null context");
    if (pop < 0 || pop >= c->n_pop) return fail(GEV_EINVAL, "population index %d out of range", pop);   // (allowed between gev_reproduce_begin and _end)
    PopState& P = c->pop[pop];
    if (!P.gen0) return fail(GEV_ESTATE, "presample: population %d has no current generation (call gev_init_gen0)", pop);
    HIPC(hipSetDevice(c->device));
    const size_t T = n_people * (size_t)c->nchr;
    if (n_people == 0) return fail(GEV_EINVAL, "presample: no offspring");
    if (2 * T * GEV_BK_CAP >= 0xf0000000ull) return fail(GEV_EINVAL, "presample: too many gametes");
    const bool has_mut = mut_seeds != nullptr;
    if (has_mut && n_mut_seeds != T) return fail(GEV_EINVAL, "presample: n_mut_seeds=%zu, expected n_people*nchr=%zu", n_mut_seeds, T);
    GEVC(finalize_static(c, pop));
    hipStream_t st = c->serialize ? c->stream : c->stream_samp;   // the head start has a stream of its own: it runs next to the lists / A-D of the generation in flight
    // the scratch set of the generation AFTER the one in flight when called between gev_reproduce_begin and _end
    gev_ctx::Scratch& sc = c->sc[(c->gen_counter + (c->pend.active ? 1u : 0u)) & 1];
    sc.presampled = false; sc.ps_stale = false;
    const bool seeds_in_place = has_mut && mut_seeds == (const uint32_t*)c->h_seeds;
    if (has_mut && !seeds_in_place) {
        if (c->h_seeds_bytes < T * sizeof(u32)) {
            HIPC(hipStreamSynchronize(st));              // an earlier copy out of the old buffer may be in flight
            if (c->h_seeds) (void)hipHostFree(c->h_seeds);
            c->h_seeds = nullptr; c->h_seeds_bytes = 0;
            HIPC(hipHostMalloc(&c->h_seeds, T * sizeof(u32) * 5 / 4 + 4096, hipHostMallocDefault));
            c->h_seeds_bytes = T * sizeof(u32) * 5 / 4 + 4096;
        }
        memcpy(c->h_seeds, mut_seeds, T * sizeof(u32));
    }
    // Called while a generation is in flight, the stitch that last used this scratch set may still be running: nothing here waits
    // for it on the host (its kernel time is collected when the set's next generation is enqueued); the sampling stream does.
    if (c->pend.active) GEVC(harvest_sampling_time(sc)); else GEVC(harvest_timing(c, sc));
    if (sc.stitch_pending) { HIPC(hipStreamWaitEvent(st, sc.ev_stitch_done, 0)); if (!c->pend.active) sc.stitch_pending = false; }
    GEVC(ensure_scratch(c, sc, n_people, has_mut));
    if (has_mut) {
        HIPC(hipMemcpyAsync(sc.mutseeds.p, c->h_seeds, T * sizeof(u32), hipMemcpyHostToDevice, st));
    }
    GEVC(enqueue_sampling(c, sc, pop, n_people, has_mut, seed_reproduce, st));
    HIPC(hipEventRecord(sc.ev_sampled, st));
    sc.presampled = true; sc.ps_pop = pop; sc.ps_seed = seed_reproduce; sc.ps_n_people = n_people; sc.ps_has_mut = has_mut;
    return GEV_OK;
}
// wait for all device work of the context (both streams) and collect pending kernel timings
int gev_sync(gev_ctx* c)
{
    if (!c) return fail(GEV_EINVAL, "null context");
    if (c->pend.active) return fail(GEV_ESTATE, "a gev_reproduce_begin is pending: call gev_reproduce_end first");
    HIPC(hipSetDevice(c->device));
    HIPC(hipStreamSynchronize(c->stream)); HIPC(hipStreamSynchronize(c->stream_big)); HIPC(hipStreamSynchronize(c->stream_samp));
    HIPC(hipStreamSynchronize(c->stream_aux)); HIPC(hipStreamSynchronize(c->stream_list));
    for (auto& sc : c->sc) { GEVC(harvest_timing(c, sc)); sc.stitch_pending = false; }
    c->planes_pending = false;
    if (g_graveyard.bytes) g_graveyard.drain(c->device, false);
    return GEV_OK;
}
// what the dense stitch wrote, summed over all gev_reproduce calls and active chromosomes: bytes written and bytes of the rows
// of the generations produced (the difference = row segments without a crossover boundary, which share the parental unit),
// and the same in segments
int gev_stitch_totals(gev_ctx* c, unsigned long long* bytes_written, unsigned long long* bytes_total, unsigned long long* segments_written, unsigned long long* segments_total)
{
    if (!c || !bytes_written || !bytes_total) return fail(GEV_EINVAL, "null");
    *bytes_written = 16ull * c->chunks_written_sum; *bytes_total = 16ull * c->chunks_total_sum;
    if (segments_written) *segments_written = c->segments_written_sum;
    if (segments_total) *segments_total = c->segments_total_sum;
    return GEV_OK;
}
// cumulative kernel time per phase over all harvested generations: sampling, dense stitch, sparse, sum
int gev_timing_totals(gev_ctx* c, double ms_sum[4], unsigned long long* n_generations)
{
    if (!c || !ms_sum || !n_generations) return fail(GEV_EINVAL, "null");
    GEVC(gev_sync(c));
    for (int i = 0; i < 4; i++) ms_sum[i] = c->ms_sum[i];
    *n_generations = c->ms_count;
    return GEV_OK;
}

// ---- Simulation::ras_compute_AD -----------------------------------------------------------
// kernels of ras_compute_AD on buffer set `buf` (current generation, or the one being produced)
// counts_ready: the allele counts of the CV columns were accumulated by k_stitch_small while it wrote the planes
// hbuf: the pinned result buffer (default: the published generation's)
static int enqueue_ad(gev_ctx* c, int pop, int buf, size_t n, bool counts_ready, int hbuf)
{
    if (hbuf < 0) hbuf = (int)((c->gen_counter + 1) & 1);
    PopState& P = c->pop[pop];
    hipStream_t st = c->stream;
    const size_t rows = 2 * n;
    const int nchr = c->nchr, nphen = c->nphen;
    GEVC(c->d_addchr.ensure(n * nchr * nphen * sizeof(double), st)); GEVC(c->d_domchr.ensure(n * nchr * nphen * sizeof(double), st));
    GEVC(c->d_add.ensure(n * nphen * sizeof(double), st)); GEVC(c->d_dom.ensure(n * nphen * sizeof(double), st));
    GEVC(c->d_flag.ensure(16, st));
    if (!c->d_cvdone.p) { GEVC(c->d_cvdone.ensure((size_t)nchr * nphen * sizeof(u32), st)); HIPC(hipMemsetAsync(c->d_cvdone.p, 0, (size_t)nchr * nphen * sizeof(u32), st)); }   // k_cv_finish's block counters (they re-zero themselves)
    c->ad_host_set_pop = -1;                                 // d_add / d_dom are rewritten below: totals handed in by gev_set_ad are gone
    if (c->any_inactive) {                                   // chromosomes held elsewhere contribute exact zeros here
        HIPC(hipMemsetAsync(c->d_addchr.p, 0, n * nchr * nphen * sizeof(double), st)); HIPC(hipMemsetAsync(c->d_domchr.p, 0, n * nchr * nphen * sizeof(double), st));
    }
    // one table entry per (phenotype, active chromosome); every kernel below covers all of them in one launch
    std::vector<AdWork> aw;
    u32 c_max = 0, sub_max = 0; bool all_have_cv = true;
    for (int p = 0; p < nphen; p++)
        for (int k = 0; k < nchr; k++) {
            if (!c->chr_active[k]) continue;
            CvStatic& V = P.cv[p][k];
            c_max = std::max(c_max, V.C); sub_max = std::max(sub_max, V.sub_w32); all_have_cv &= V.C > 0;
        }
    // fast path: one root population and the block's rows fit LDS; else the general per-individual kernel
    const u32 S1 = sub_max | 1u;
    int ipb = 0;
    const size_t ad_tab_lds = AD_CHUNK * 4 + AD_CHUNK * 6 * 8 + 8;          // column + table chunk behind the rows
    if ((c->rp_bits == 0 || c->ad_effects_shared) && all_have_cv) { for (int cand : {256, 128, 64}) if ((size_t)2 * cand * S1 * 4 + ad_tab_lds <= 64 * 1024) { ipb = cand; break; } }
    for (int p = 0; p < nphen; p++)
        for (int k = 0; k < nchr; k++) {
            if (!c->chr_active[k]) continue;
            CvStatic& V = P.cv[p][k]; ChrStatic& S = P.cs[k];
            if (ipb) GEVC(V.d_tab.ensure((size_t)V.C * 6 * sizeof(double), st));
            AdWork a{};
            a.cvp = P.cvp[p][k][buf].as<u32>();
            a.pos_sorted = V.d_pos_sorted.as<u64>(); a.pos_file = V.d_pos_file.as<u64>(); a.col_of_icv = V.d_col_of_icv.as<u32>();
            a.a = V.d_a.as<double>(); a.d = V.d_d.as<double>(); a.aptr = V.d_aptr.as<const double*>(); a.dptr = V.d_dptr.as<const double*>();
            a.counts = V.d_counts.as<u32>(); a.frq = V.d_frq.as<double>(); a.tab = ipb ? V.d_tab.as<double>() : nullptr;
            a.add_out = c->d_addchr.as<double>() + (size_t)k * nphen + p; a.dom_out = c->d_domchr.as<double>() + (size_t)k * nphen + p;
            const bool one_chr = nchr == 1;                                     // the sum over chromosomes is 0 + x: written with the per-chromosome value
            a.add_tot = one_chr ? c->d_add.as<double>() + p : nullptr; a.dom_tot = one_chr ? c->d_dom.as<double>() + p : nullptr;
            a.partial = counts_ready ? V.d_partial.as<uint16_t>() : nullptr;
            a.bp0 = S.rbp.front(); a.bp_end = S.rbp.back(); a.vd = V.vd;
            a.stride_w32 = V.stride_w32; a.sub_w32 = V.sub_w32; a.C = V.C; a.own_pop = pop; a.cols_sorted = V.cols_sorted ? 1u : 0u;
            aw.push_back(a);
        }
    const unsigned nw = (unsigned)aw.size();
    if (nw) {
        DevBuf& adw = c->d_adwork2[hbuf]; std::vector<uint8_t>& adw_shadow = c->adwork_shadow[hbuf];      // (one table per result buffer: they alternate with the generations)
        GEVC(upload_table_cached(c, adw, adw_shadow, aw.data(), aw.size() * sizeof(AdWork), st));
        const AdWork* At = adw.as<AdWork>();
        const size_t out_stride = (size_t)nchr * nphen, tot_stride = (size_t)nphen;
        u32* flag = c->d_flag.as<u32>();
        if (c_max && !counts_ready) {
            const unsigned gy = (unsigned)std::min<size_t>(std::max<size_t>(rows / 512, 1), 256);
            hipLaunchKernelGGL(k_cv_count, dim3((unsigned)ceil_div(c_max, 256), gy, nw), dim3(256), 0, st, At, rows);
        }
        // counts -> frequencies (+ the term table of the fast path), NaN flag reset: one launch
        const unsigned nblk = counts_ready ? (unsigned)ceil_div(rows, SMALL_ROWS_PER_BLOCK) : 0u;
        hipLaunchKernelGGL(k_cv_finish, dim3((unsigned)std::max<size_t>(ceil_div(c_max, 256), 1), nblk ? std::min((nblk + 7u) / 8u, 128u) : 1u, nw), dim3(256), 0, st, At, nblk, n, c->d_cvdone.as<u32>(), flag);
        if (ipb) {
            bool direct = true;                                  // every CV file in position order: rows are read from global memory, no row staging
            for (const AdWork& a : aw) direct &= a.cols_sorted != 0;
            bool chunked = direct, skip_d = true;                 // ... in whole 16-byte chunks: the term table in few large pieces (k_ad_accumulate_wide)
            for (const AdWork& a : aw) { chunked &= (a.sub_w32 & 3u) == 0 && (a.stride_w32 & 3u) == 0 && a.C > 0 && (size_t)a.sub_w32 * 32 >= (((size_t)a.C + 127) & ~(size_t)127); skip_d &= a.vd == 0; }
            static const bool no_dir = getenv("GEV_AD_WIDE") && atoi(getenv("GEV_AD_WIDE")) == 0;
            if (chunked && !no_dir) {
                const unsigned nb = (unsigned)ceil_div(n, 256);
                if (skip_d) hipLaunchKernelGGL((k_ad_accumulate_wide<true>), dim3(nb, nw), dim3(256), 0, st, At, n, out_stride, tot_stride, flag);
                else hipLaunchKernelGGL((k_ad_accumulate_wide<false>), dim3(nb, nw), dim3(256), 0, st, At, n, out_stride, tot_stride, flag);
            } else if (direct) {
                const unsigned nb = (unsigned)ceil_div(n, 256);
                hipLaunchKernelGGL((k_ad_accumulate_tab<256>), dim3(nb, nw), dim3(256), ad_tab_lds, st, At, 0u, n, out_stride, tot_stride, flag, 1);
            } else {
                const size_t lds = (size_t)2 * ipb * S1 * 4 + ad_tab_lds;
                const unsigned nb = (unsigned)ceil_div(n, ipb);
                if (ipb == 256) hipLaunchKernelGGL((k_ad_accumulate_tab<256>), dim3(nb, nw), dim3(256), lds, st, At, S1, n, out_stride, tot_stride, flag, 0);
                else if (ipb == 128) hipLaunchKernelGGL((k_ad_accumulate_tab<128>), dim3(nb, nw), dim3(128), lds, st, At, S1, n, out_stride, tot_stride, flag, 0);
                else hipLaunchKernelGGL((k_ad_accumulate_tab<64>), dim3(nb, nw), dim3(64), lds, st, At, S1, n, out_stride, tot_stride, flag, 0);
            }
        } else {
            // several root populations with their own CV effects: the piecewise kernel when every CV file is in position order
            bool sorted_all = c->rp_bits >= 1 && c->rp_bits <= 3;
            for (const AdWork& a : aw) sorted_all &= a.cols_sorted != 0;
            static const bool no_rp = getenv("GEV_AD_RP_FAST") && atoi(getenv("GEV_AD_RP_FAST")) == 0;
            const dim3 grid((unsigned)ceil_div(n, 256), nw);
            const size_t lds = (size_t)ADRP_PIECE * ((size_t)c->n_pop * 2 + 5) * sizeof(double);
            if (sorted_all && !no_rp && c->rp_bits == 1) hipLaunchKernelGGL((k_ad_accumulate_rp<1>), grid, dim3(256), lds, st, At, (u32)c->n_pop, n, out_stride, tot_stride, flag);
            else if (sorted_all && !no_rp && c->rp_bits == 2) hipLaunchKernelGGL((k_ad_accumulate_rp<2>), grid, dim3(256), lds, st, At, (u32)c->n_pop, n, out_stride, tot_stride, flag);
            else if (sorted_all && !no_rp && c->rp_bits == 3) hipLaunchKernelGGL((k_ad_accumulate_rp<3>), grid, dim3(256), lds, st, At, (u32)c->n_pop, n, out_stride, tot_stride, flag);
            else hipLaunchKernelGGL(k_ad_accumulate, grid, dim3(256), 0, st, At, c->rp_bits, n, out_stride, tot_stride, flag);
        }
        KCHECK();
    } else HIPC(hipMemsetAsync(c->d_flag.p, 0xff, 4, st));
    if (nchr > 1 || !nw) {                                   // one chromosome: the A/D kernels wrote the totals themselves
        hipLaunchKernelGGL(k_ad_sum_chr, dim3((unsigned)ceil_div(n * nphen, 256)), dim3(256), 0, st, c->d_addchr.as<double>(), c->d_domchr.as<double>(), c->d_add.as<double>(), c->d_dom.as<double>(), n, nchr, nphen);
        KCHECK();
    }
    // results to the pinned host cache: [flag | additive | dominance].  The per-chromosome arrays stay on the device and are
    // copied when gev_compute_ad is asked for them.  vd == 0 for every phenotype: the reference zeroes d (:2698-2699), every
    // D-term is (+-0) * ... and the running sums stay +0.0 exactly -- nothing to copy, gev_compute_ad fills zeros.
    const size_t nd = n * nphen;
    const size_t bytes = 16 + 2 * nd * sizeof(double);
    if (c->h_ad2_bytes[hbuf] < bytes) {
        HIPC(hipStreamSynchronize(st));                      // (copies into the old buffer may be in flight on this stream)
        void* old = c->h_ad2[hbuf];
        c->h_ad2[hbuf] = nullptr; c->h_ad2_bytes[hbuf] = 0;
        HIPC(hipHostMalloc(&c->h_ad2[hbuf], bytes * 5 / 4 + 4096, hipHostMallocDefault));
        c->h_ad2_bytes[hbuf] = bytes * 5 / 4 + 4096;
        if (old) (void)hipHostFree(old);
    }
    bool dom_zero = true;
    for (const AdWork& a : aw) dom_zero &= a.vd == 0;
    c->ad_dom_zero2[hbuf] = dom_zero;
    uint8_t* h = (uint8_t*)c->h_ad2[hbuf];
    HIPC(hipMemcpyAsync(h, c->d_flag.p, 4, hipMemcpyDeviceToHost, st));
    HIPC(hipMemcpyAsync(h + 16, c->d_add.p, nd * sizeof(double), hipMemcpyDeviceToHost, st));
    if (!dom_zero) HIPC(hipMemcpyAsync(h + 16 + nd * 8, c->d_dom.p, nd * sizeof(double), hipMemcpyDeviceToHost, st));
    return GEV_OK;
}
int gev_compute_ad(gev_ctx* c, int pop, double* additive, double* dominance, double* add_chr, double* dom_chr)
{
    if (!c) return fail(GEV_EINVAL, "null context");
    if (pop < 0 || pop >= c->n_pop) return fail(GEV_EINVAL, "population index %d out of range", pop);
    PopState& P = c->pop[pop];
    if (!P.gen0) return fail(GEV_ESTATE, "compute_ad: population %d has no current generation", pop);
    const int nchr = c->nchr, nphen = c->nphen;
    const int hb = (int)((c->gen_counter + 1) & 1);          // the published generation's result buffer
    if (c->pend.active) {
        // a generation is in flight: the CURRENT generation's values can still be handed out if they were computed with it (they
        // sit in the other pinned buffer); anything that needs device work has to wait for gev_reproduce_end / gev_generation_end
        if (c->ad_cached_pop != pop || add_chr || dom_chr)
            return fail(GEV_ESTATE, "a gev_reproduce_begin is pending: call gev_reproduce_end first");
    } else {
        HIPC(hipSetDevice(c->device));
        if (!c->pop[pop].cv[0][0].d_aptr.p) GEVC(check_multipop(c));
        GEVC(materialize_order(c, pop));
        if (c->ad_cached_pop != pop) {          // not computed eagerly by the last gev_reproduce of this population
            GEVC(enqueue_ad(c, pop, P.cur, P.n_people));
            HIPC(hipStreamSynchronize(c->stream));
            c->ad_cached_pop = pop;
            for (int p = 0; p < nphen; p++) for (int k = 0; k < nchr; k++) P.cv[p][k].frq_valid = true;
        }
    }
    const size_t n = P.n_people;
    const uint8_t* h = (const uint8_t*)c->h_ad2[hb];
    const size_t nd = n * nphen, ndc = n * (size_t)nchr * nphen;
    const u32 flag = *(const u32*)h;
    if (additive) memcpy(additive, h + 16, nd * sizeof(double));
    if (dominance) { if (c->ad_dom_zero2[hb]) memset(dominance, 0, nd * sizeof(double)); else memcpy(dominance, h + 16 + nd * 8, nd * sizeof(double)); }
    if (add_chr || dom_chr) {                                // (the arrays of the generation's A/D kernels are still in place: any later A/D run resets ad_cached_pop)
        if (add_chr) HIPC(hipMemcpyAsync(add_chr, c->d_addchr.p, ndc * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (dom_chr) HIPC(hipMemcpyAsync(dom_chr, c->d_domchr.p, ndc * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
    }
    if (flag != 0xffffffffu) return fail(GEV_ENAN, "Error: A or D is nan for human %u", flag);
    return GEV_OK;
}
int gev_compute_ad_device(gev_ctx* c, int pop, double** add_chr, double** dom_chr, size_t* n_doubles)
{
    GEVC(check_idx(c, pop, 0));
    GEVC(check_not_pending(c));
    if (!add_chr || !dom_chr || !n_doubles) return fail(GEV_EINVAL, "compute_ad_device: null argument");
    GEVC(gev_compute_ad(c, pop, nullptr, nullptr, nullptr, nullptr));          // (computes unless the last generation already did; checks the NaN flag)
    PopState& P = c->pop[pop];
    *add_chr = c->d_addchr.as<double>(); *dom_chr = c->d_domchr.as<double>(); *n_doubles = P.n_people * (size_t)c->nchr * c->nphen;
    return GEV_OK;
}
int gev_ad_finish_device(gev_ctx* c, int pop, double* additive, double* dominance, double* add_chr_out, double* dom_chr_out)
{
    GEVC(check_idx(c, pop, 0));
    GEVC(check_not_pending(c));
    PopState& P = c->pop[pop];
    if (!P.gen0 || c->ad_cached_pop != pop) return fail(GEV_ESTATE, "ad_finish_device: call gev_compute_ad_device for population %d first", pop);
    HIPC(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t n = P.n_people; const int nchr = c->nchr, nphen = c->nphen;
    const size_t nd = n * nphen, ndc = nd * nchr;
    GEVC(c->d_add.ensure(nd * sizeof(double), st)); GEVC(c->d_dom.ensure(nd * sizeof(double), st));
    hipLaunchKernelGGL(k_ad_sum_chr, dim3((unsigned)ceil_div(n * nphen, 256)), dim3(256), 0, st, c->d_addchr.as<double>(), c->d_domchr.as<double>(), c->d_add.as<double>(), c->d_dom.as<double>(), n, nchr, nphen);
    KCHECK();
    if (additive) HIPC(hipMemcpyAsync(additive, c->d_add.p, nd * sizeof(double), hipMemcpyDeviceToHost, st));
    if (dominance) HIPC(hipMemcpyAsync(dominance, c->d_dom.p, nd * sizeof(double), hipMemcpyDeviceToHost, st));
    if (add_chr_out) HIPC(hipMemcpyAsync(add_chr_out, c->d_addchr.p, ndc * sizeof(double), hipMemcpyDeviceToHost, st));
    if (dom_chr_out) HIPC(hipMemcpyAsync(dom_chr_out, c->d_domchr.p, ndc * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    c->ad_host_set_pop = pop;                                  // the totals in d_add / d_dom are the population's (as after gev_set_ad); the pinned cache holds this context's share only
    c->ad_cached_pop = -1;
    return GEV_OK;
}
// ---- Simulation::ras_scale_AD_compute_GEF ---------------------------------
static int normal_stream(gev_ctx* c, u32 engine_seed, size_t n, double sd, double* d_out)
{
    hipStream_t st = c->stream;
    size_t n_cand = (size_t)((n / 2 + 1) / 0.7) + 4096;           // acceptance = pi/4
    for (int attempt = 0; attempt < 4; attempt++, n_cand *= 2) {
        GEVC(c->d_gef_flag.ensure((n_cand + 1) * sizeof(u32) * 2, st)); GEVC(c->d_gef_first.ensure(n_cand * sizeof(double) * 2, st));
        u32* flag = c->d_gef_flag.as<u32>(); u32* off = flag + n_cand + 1;
        double* first = c->d_gef_first.as<double>(); double* second = first + n_cand;
        hipLaunchKernelGGL(k_polar_candidates, dim3((unsigned)ceil_div(n_cand, 256)), dim3(256), 0, st, engine_seed, n_cand, flag, first, second);
        KCHECK();
        u32 accepted = 0;
        GEVC(scan_u32(c, flag, n_cand, off, &accepted));
        if ((size_t)accepted * 2 < n) continue;                     // (practically never) not enough accepted pairs: more candidates
        hipLaunchKernelGGL(k_polar_emit, dim3((unsigned)ceil_div(n_cand, 256)), dim3(256), 0, st, flag, off, first, second, n_cand, n, sd, d_out);
        KCHECK();
        return GEV_OK;
    }
    return fail(GEV_EDEVICE, "normal_stream: acceptance far below pi/4");
}
static int device_sum(gev_ctx* c, const double* x, size_t n, double shift, int pw, double* host_out)
{
    hipStream_t st = c->stream;
    GEVC(c->d_gef_red.ensure(512 * sizeof(double), st));
    const int nb = (int)std::min<size_t>(ceil_div(n, 256), 256);
    hipLaunchKernelGGL(k_sum_partial, dim3(nb), dim3(256), 0, st, x, n, shift, pw, c->d_gef_red.as<double>());
    hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(256), 0, st, c->d_gef_red.as<double>(), nb, c->d_gef_red.as<double>() + 256);
    KCHECK();
    HIPC(hipMemcpyAsync(host_out, c->d_gef_red.as<double>() + 256, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    return GEV_OK;
}
int gev_scale_ad_compute_gef(gev_ctx* c, int pop, int phen, const gev_gef_params* par, uint32_t seed,
                             const double* common_sibling, const double* f_father, const double* f_mother,
                             double* additive, double* dominance, double* bv, double* e_noise, double* parental_effect, double* phen_out)
{
    GEVC(check_idx(c, pop, 0, phen));
    if (c->any_inactive && c->ad_host_set_pop != pop)
        return fail(GEV_ESTATE, "scale_ad_compute_gef: this context holds a subset of the chromosomes: give it the all-reduced raw A/D first (gev_set_ad)");
    if (!par) return fail(GEV_EINVAL, "scale_ad_compute_gef: null parameters");
    PopState& P = c->pop[pop];
    if (!P.gen0) return fail(GEV_ESTATE, "scale_ad_compute_gef: population %d has no current generation", pop);
    HIPC(hipSetDevice(c->device));
    if (c->ad_cached_pop != pop && c->ad_host_set_pop != pop) GEVC(gev_compute_ad(c, pop, nullptr, nullptr, nullptr, nullptr));      // raw A/D of this generation on the device
    hipStream_t st = c->stream;
    const size_t n = P.n_people;
    GEVC(c->d_gef_io.ensure(10 * n * sizeof(double), st));
    double* io = c->d_gef_io.as<double>();
    double *d_e = io, *d_par = io + n, *d_cs = io + 2 * n, *d_ff = io + 3 * n, *d_out = io + 4 * n;     // d_out: 6 vectors
    GEVC(normal_stream(c, seed, n, 1.0, d_e));                                                           // generator_e(seed), N(0,1), :3080-3102
    const bool need_par = par->vf > 0;
    if (par->gen_num == 0) { if (need_par) GEVC(normal_stream(c, seed + 1u, n, std::sqrt(par->vf), d_par)); }   // generator_f(seed+1), :3095-3114
    else if (need_par) {
        double* d_fm = d_out;                                                                             // staging before d_out is written
        if (f_father) HIPC(hipMemcpyAsync(d_ff, f_father, n * sizeof(double), hipMemcpyHostToDevice, st));
        if (f_mother) HIPC(hipMemcpyAsync(d_fm, f_mother, n * sizeof(double), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_par_eff, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, f_father ? d_ff : (const double*)nullptr, f_mother ? d_fm : (const double*)nullptr, n, par->beta, d_par);
        KCHECK();
        HIPC(hipStreamSynchronize(st));
    }
    if (common_sibling) HIPC(hipMemcpyAsync(d_cs, common_sibling, n * sizeof(double), hipMemcpyHostToDevice, st));
    double s_a = 1;
    if (par->va > 0) s_a = std::sqrt(par->s2_a_gen0 / par->va);
    double s_d = 0;
    if (par->vd > 0) s_d = std::sqrt(par->s2_d_gen0 / par->vd); else if (par->vd == -1) s_d = 1;
    double s_ev = 0;
    if (par->ve > 0) {                                                                                    // CommFunc::var(e): two passes, n-1
        double sum = 0, ss = 0, var_e = 0;
        if (n > 1) {
            GEVC(device_sum(c, d_e, n, 0.0, 1, &sum));
            const double mu = sum / (double)n;
            GEVC(device_sum(c, d_e, n, mu, 2, &ss));
            var_e = ss / (double)(n - 1);
        }
        s_ev = std::sqrt(var_e / par->ve);
    }
    hipLaunchKernelGGL(k_gef_apply, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, c->d_add.as<double>() + phen, c->d_dom.as<double>() + phen, (size_t)c->nphen,
                       d_e, d_par, common_sibling ? d_cs : (const double*)nullptr, n, s_a, s_d, s_ev, par->vf,
                       d_out, d_out + n, d_out + 2 * n, d_out + 3 * n, d_out + 4 * n, d_out + 5 * n);
    KCHECK();
    double* outs[6] = {additive, dominance, bv, e_noise, parental_effect, phen_out};
    for (int k = 0; k < 6; k++) if (outs[k]) HIPC(hipMemcpyAsync(outs[k], d_out + k * n, n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    return GEV_OK;
}

// Raw A/D totals of the current generation from the host: a locus-split population's contexts each hold partial sums; after the
// all-reduce (geneevolve_amd/distributed.py:compute_ad_locus_split) every context is given the totals, and
// gev_scale_ad_compute_gef then works as on an unsplit context.
int gev_set_ad(gev_ctx* c, int pop, const double* additive, const double* dominance)
{
    GEVC(check_idx(c, pop, 0));
    PopState& P = c->pop[pop];
    if (!P.gen0) return fail(GEV_ESTATE, "set_ad: population %d has no current generation", pop);
    if (!additive || !dominance) return fail(GEV_EINVAL, "set_ad: null array");
    HIPC(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t nd = P.n_people * (size_t)c->nphen;
    GEVC(c->d_add.ensure(nd * sizeof(double), st)); GEVC(c->d_dom.ensure(nd * sizeof(double), st));
    HIPC(hipMemcpyAsync(c->d_add.p, additive, nd * sizeof(double), hipMemcpyHostToDevice, st));
    HIPC(hipMemcpyAsync(c->d_dom.p, dominance, nd * sizeof(double), hipMemcpyHostToDevice, st));
    HIPC(hipStreamSynchronize(st));
    c->ad_host_set_pop = pop;
    return GEV_OK;
}
int gev_get_cv_freq(gev_ctx* c, int pop, int phen, int chr, double* frq, size_t C)
{
    GEVC(check_idx(c, pop, chr, phen));
    GEVC(check_active(c, chr, "get_cv_freq"));
    CvStatic& V = c->pop[pop].cv[phen][chr];
    if (!V.frq_valid) return fail(GEV_ESTATE, "get_cv_freq: call gev_compute_ad first");
    if (C != V.C || !frq) return fail(GEV_EINVAL, "get_cv_freq: C=%zu, expected %u", C, V.C);
    HIPC(hipSetDevice(c->device));
    HIPC(hipMemcpyAsync(frq, V.d_frq.p, C * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    return GEV_OK;
}

// ---- Simulation::ras_do_migration (row movement part) -------------------------------------
struct Seg { int src_pop; std::vector<u32> people; };     // individuals (positions in src_pop's current generation)
// rebuild population `dst` in its alternate buffers from segments of the CURRENT buffers; caller flips
static int gather_population(gev_ctx* c, int dst, const std::vector<Seg>& segs, size_t n_new)
{
    PopState& D = c->pop[dst];
    hipStream_t st = c->stream;
    const int alt = D.cur ^ 1;
    const size_t rows_new = 2 * n_new;
    for (const Seg& sg : segs) if (!sg.people.empty()) GEVC(ensure_csr(c, sg.src_pop));      // whole lists of the sources are read
    // capacity of the alternate buffers (the current ones keep their content)
    if (n_new > D.cap_people) GEVC(ensure_capacity(c, dst, n_new));
    GEVC(c->d_cnt.ensure((rows_new + 1) * sizeof(u32), st));
    for (int k = 0; k < c->nchr; k++) {
        if (!c->chr_active[k]) continue;                // held by another context of a locus-split population
        ChrStatic& S = D.cs[k]; ChrState& ds = D.st[k];
        for (int pass = 0; pass < 2; pass++) {          // 0: mutation lists, 1: interval lists
            if (pass == 1 && !c->track_intervals) continue;
            size_t row0 = 0;
            for (const Seg& sg : segs) {                // counts
                if (sg.people.empty()) continue;
                PopState& Sp = c->pop[sg.src_pop]; ChrState& ss = Sp.st[k];
                std::vector<u32> map(2 * sg.people.size());
                for (size_t j = 0; j < sg.people.size(); j++) { map[2 * j] = 2 * sg.people[j]; map[2 * j + 1] = 2 * sg.people[j] + 1; }
                GEVC(h2d(c, c->d_map, map.data(), map.size() * sizeof(u32)));
                const u32* soff = pass == 0 ? ss.moff[Sp.cur].as<u32>() : ss.poff[Sp.cur].as<u32>();
                hipLaunchKernelGGL(k_csr_gather_count, dim3((unsigned)ceil_div(map.size(), 256)), dim3(256), 0, st, soff, c->d_map.as<u32>(), map.size(), c->d_cnt.as<u32>() + row0);
                KCHECK();
                HIPC(hipStreamSynchronize(st));
                row0 += map.size();
            }
            u32 tot = 0;
            u32* doff = pass == 0 ? ds.moff[alt].as<u32>() : ds.poff[alt].as<u32>();
            GEVC(scan_u32(c, c->d_cnt.as<u32>(), rows_new, doff, &tot));
            if (pass == 0) { GEVC(ds.mpos[alt].ensure(std::max<size_t>(tot, 2) * sizeof(u64), st, false, 1.25)); ds.mut_total[alt] = tot; }
            else { GEVC(ds.parts[alt].ensure(std::max<size_t>(tot, 1) * sizeof(gev_part), st, false, 1.25)); ds.parts_total[alt] = tot; }
            row0 = 0;
            for (const Seg& sg : segs) {                // fill
                if (sg.people.empty()) continue;
                PopState& Sp = c->pop[sg.src_pop]; ChrState& ss = Sp.st[k];
                std::vector<u32> map(2 * sg.people.size());
                for (size_t j = 0; j < sg.people.size(); j++) { map[2 * j] = 2 * sg.people[j]; map[2 * j + 1] = 2 * sg.people[j] + 1; }
                GEVC(h2d(c, c->d_map, map.data(), map.size() * sizeof(u32)));
                if (pass == 0)
                    hipLaunchKernelGGL((k_csr_gather_fill<u64>), dim3((unsigned)ceil_div(map.size(), 256)), dim3(256), 0, st,
                                       ss.moff[Sp.cur].as<u32>(), ss.mpos[Sp.cur].as<u64>(), c->d_map.as<u32>(), map.size(), doff + row0, ds.mpos[alt].as<u64>());
                else
                    hipLaunchKernelGGL((k_csr_gather_fill<gev_part>), dim3((unsigned)ceil_div(map.size(), 256)), dim3(256), 0, st,
                                       ss.poff[Sp.cur].as<u32>(), ss.parts[Sp.cur].as<gev_part>(), c->d_map.as<u32>(), map.size(), doff + row0, ds.parts[alt].as<gev_part>());
                KCHECK();
                HIPC(hipStreamSynchronize(st));
                row0 += map.size();
            }
        }
        // genotype rows: fresh pool rows of the destination, filled from the sources' pools
        PoolWork dpw{};
        if (c->dense) {
            pool_new_stamp(ds);
            dpw = pool_work(c, D, k, (D.pcur + 1) % 3);
            GEVC(pool_free_list(dpw, 2 * D.n_phys, st));
            GEVC(pool_take(dpw, 0, rows_new, st));
            ds.pool_list_valid = false;                      // (the next generation builds its own list: the host's picture of this one is gone)
        }
        size_t row0 = 0;
        for (const Seg& sg : segs) {
            if (sg.people.empty()) continue;
            PopState& Sp = c->pop[sg.src_pop];
            std::vector<u32> map(2 * sg.people.size());
            for (size_t j = 0; j < sg.people.size(); j++) { map[2 * j] = 2 * sg.people[j]; map[2 * j + 1] = 2 * sg.people[j] + 1; }
            GEVC(h2d(c, c->d_map, map.data(), map.size() * sizeof(u32)));
            const u32 chunks = (u32)(S.stride / 16);
            if (c->dense)
                hipLaunchKernelGGL(k_copy_rows16, dim3((unsigned)ceil_div(map.size() * chunks, 256)), dim3(256), 0, st,
                                   pool_rows(D, k, dpw.phys_alt + row0 * S.nseg), pool_rows(Sp, k, Sp.st[k].phys[Sp.pcur].as<u32>()), c->d_map.as<u32>(), (size_t)0, map.size(), chunks);
            for (int p = 0; p < c->nphen; p++) {
                CvStatic& V = D.cv[p][k];
                const u32 cch = V.stride_w32 / 4;
                hipLaunchKernelGGL(k_gather_rows16, dim3((unsigned)ceil_div(map.size() * cch, 256)), dim3(256), 0, st,
                                   (uint4*)(D.cvp[p][k][alt].as<u32>() + row0 * V.stride_w32), (size_t)cch,
                                   (const uint4*)Sp.cvp[p][k][Sp.cur].p, (size_t)Sp.cv[p][k].stride_w32 / 4, c->d_map.as<u32>(), map.size(), cch);
                V.frq_valid = false;
            }
            KCHECK();
            HIPC(hipStreamSynchronize(st));
            row0 += map.size();
        }
    }
    // Human::sex follows the individuals (what gev_random_mate reads)
    size_t i0 = 0;
    for (const Seg& sg : segs) {
        if (sg.people.empty()) continue;
        PopState& Sp = c->pop[sg.src_pop];
        GEVC(h2d(c, c->d_map, sg.people.data(), sg.people.size() * sizeof(u32)));
        hipLaunchKernelGGL(k_gather_u8, dim3((unsigned)ceil_div(sg.people.size(), 256)), dim3(256), 0, st, D.d_sex[alt].as<uint8_t>() + i0, Sp.d_sex[Sp.cur].as<uint8_t>(), c->d_map.as<u32>(), 0u, sg.people.size());
        KCHECK();
        HIPC(hipStreamSynchronize(st));
        i0 += sg.people.size();
    }
    for (int k = 0; k < c->nchr; k++) { D.st[k].csr_valid = true; D.st[k].lp.valid = false; }     // (the caller flips to the buffers written here: CSR form, no pieces yet)
    return GEV_OK;
}
// bring the current generation back to dense logical order (no-op unless rows were removed/imported)
static int materialize_order(gev_ctx* c, int pop)
{
    PopState& P = c->pop[pop];
    if (P.logical.empty()) return GEV_OK;
    GEVC(gev_sync(c));
    Seg all; all.src_pop = pop; all.people = P.logical;
    std::vector<Seg> segs; segs.push_back(std::move(all));
    GEVC(gather_population(c, pop, segs, P.n_people));
    P.cur ^= 1; P.pcur = (P.pcur + 1) % 3; P.n_phys = P.n_people; P.logical.clear(); c->ad_cached_pop = c->ad_host_set_pop = -1;
    return GEV_OK;
}
static int check_not_pending(gev_ctx* c)
{
    if (!c) return fail(GEV_EINVAL, "null context");
    if (c->pend.active) return fail(GEV_ESTATE, "a gev_reproduce_begin is pending: call gev_reproduce_end first");
    return GEV_OK;
}
int gev_migrate(gev_ctx* c, const gev_move* moves, size_t n_moves)
{
    GEVC(check_not_pending(c));
    if (n_moves && !moves) return fail(GEV_EINVAL, "migrate: null moves");
    HIPC(hipSetDevice(c->device));
    GEVC(gev_sync(c));
    for (int p = 0; p < c->n_pop; p++) if (c->pop[p].gen0) GEVC(materialize_order(c, p));
    if (c->n_pop > 1 && !c->pop[0].cv[0][0].d_aptr.p) GEVC(check_multipop(c));
    std::vector<std::vector<uint8_t>> gone(c->n_pop);
    for (int p = 0; p < c->n_pop; p++) {
        if (!c->pop[p].gen0) return fail(GEV_ESTATE, "migrate: population %d has no current generation", p);
        gone[p].assign(c->pop[p].n_people, 0);
    }
    for (size_t i = 0; i < n_moves; i++) {
        const gev_move& m = moves[i];
        if (m.src_pop < 0 || m.src_pop >= c->n_pop || m.dst_pop < 0 || m.dst_pop >= c->n_pop || m.src_pop == m.dst_pop)
            return fail(GEV_EINVAL, "migrate: move %zu has bad population indices", i);
        if (m.src_pos >= c->pop[m.src_pop].n_people || gone[m.src_pop][m.src_pos]) return fail(GEV_EINVAL, "migrate: move %zu: position out of range or moved twice", i);
        gone[m.src_pop][m.src_pos] = 1;
    }
    std::vector<std::vector<Seg>> plan(c->n_pop);
    std::vector<size_t> n_new(c->n_pop);
    for (int p = 0; p < c->n_pop; p++) {
        Seg keep; keep.src_pop = p;
        for (size_t i = 0; i < c->pop[p].n_people; i++) if (!gone[p][i]) keep.people.push_back((u32)i);   // stayers keep their order (:960-966)
        plan[p].push_back(std::move(keep));
    }
    for (size_t i = 0; i < n_moves; i++) {                                                              // migrants appended in `moves` order (:971-981)
        const gev_move& m = moves[i];
        if (plan[m.dst_pop].back().src_pop != m.src_pop || plan[m.dst_pop].size() == 1) { Seg s; s.src_pop = m.src_pop; plan[m.dst_pop].push_back(std::move(s)); }
        plan[m.dst_pop].back().people.push_back((u32)m.src_pos);
    }
    for (int p = 0; p < c->n_pop; p++) {
        n_new[p] = 0;
        for (auto& s : plan[p]) n_new[p] += s.people.size();
        if (n_new[p] == 0) return fail(GEV_EINVAL, "migrate: population %d would become empty", p);
    }
    // grow every destination first (capacity growth copies the current buffers), then gather
    for (int p = 0; p < c->n_pop; p++) if (n_new[p] > c->pop[p].cap_people) GEVC(ensure_capacity(c, p, n_new[p]));
    for (int p = 0; p < c->n_pop; p++) GEVC(ensure_csr(c, p));          // whole lists of every population are read (before any flag of a destination changes)
    for (int p = 0; p < c->n_pop; p++) GEVC(gather_population(c, p, plan[p], n_new[p]));
    for (int p = 0; p < c->n_pop; p++) { c->pop[p].cur ^= 1; c->pop[p].pcur = (c->pop[p].pcur + 1) % 3; c->pop[p].n_people = n_new[p]; c->pop[p].n_phys = n_new[p]; }
    c->ad_cached_pop = c->ad_host_set_pop = -1;
    return GEV_OK;
}
// ---- cross-GPU form of the migration step ---------------------------------------------------
// Packed record buffer (all offsets 16-byte aligned), for n individuals:
//   [counts]  u32[n][nchr][2 haps][2] = (n_mut, n_parts) per haplotype row
//   [sex]     u8[n]  Human::sex
//   [planes]  per chr: 2n rows x stride bytes          [cv] per (phen, chr): 2n rows x stride_w32*4 bytes
//   [muts]    per chr: u64 lists concatenated in row order   [parts] per chr: gev_part lists likewise
struct PackLayout { size_t counts, sex, planes, cv, muts, parts, total; std::vector<size_t> mut_chr, parts_chr; };
static size_t al16(size_t x) { return (x + 15) & ~(size_t)15; }
static PackLayout pack_layout(gev_ctx* c, PopState& P, size_t n, const std::vector<u32>& counts)
{
    PackLayout L; const int nchr = c->nchr;
    size_t off = 0;
    // chromosomes this context does not hold (locus-split population) take no space: both ends of an exchange hold the same set
    L.counts = off; off = al16(off + n * nchr * 4 * sizeof(u32));
    L.sex = off; off = al16(off + n);
    L.planes = off; for (int k = 0; k < nchr; k++) if (c->chr_active[k] && c->dense && c->migrant_rows) off = al16(off + 2 * n * P.cs[k].stride);
    L.cv = off; for (int p = 0; p < c->nphen; p++) for (int k = 0; k < nchr; k++) if (c->chr_active[k]) off = al16(off + 2 * n * P.cv[p][k].stride_w32 * sizeof(u32));
    L.mut_chr.assign(nchr, 0); L.parts_chr.assign(nchr, 0);
    for (size_t i = 0; i < n; i++) for (int k = 0; k < nchr; k++) for (int h = 0; h < 2; h++) {
        L.mut_chr[k] += counts[((i * nchr + k) * 2 + h) * 2]; L.parts_chr[k] += counts[((i * nchr + k) * 2 + h) * 2 + 1];
    }
    L.muts = off; for (int k = 0; k < nchr; k++) off = al16(off + L.mut_chr[k] * sizeof(u64));
    L.parts = off; for (int k = 0; k < nchr; k++) off = al16(off + L.parts_chr[k] * sizeof(gev_part));
    L.total = off;
    return L;
}
// per-row list lengths of the selected individuals (device count kernels, one small D2H)
static int export_counts(gev_ctx* c, int pop, const uint64_t* positions, size_t n, std::vector<u32>& counts, std::vector<u32>& map)
{
    PopState& P = c->pop[pop];
    const int nchr = c->nchr;
    map.resize(2 * n);
    for (size_t i = 0; i < n; i++) {
        if (positions[i] >= P.n_people) return fail(GEV_EINVAL, "export: position %llu beyond the population (%zu people)", (unsigned long long)positions[i], P.n_people);
        const u32 ph = P.logical.empty() ? (u32)positions[i] : P.logical[positions[i]];
        map[2 * i] = 2 * ph; map[2 * i + 1] = 2 * ph + 1;
    }
    counts.assign(n * nchr * 4, 0);
    if (!n) return GEV_OK;
    GEVC(h2d(c, c->d_map, map.data(), map.size() * sizeof(u32)));
    GEVC(c->d_cnt.ensure(2 * n * sizeof(u32) * 2 + 16, c->stream));
    std::vector<u32> tmp(2 * n);
    for (int k = 0; k < nchr; k++) {
        if (!c->chr_active[k]) continue;
        ChrState::LpState& lp = P.st[k].lp;
        if (lp.valid) {                                   // the lists live as pieces: lengths of the selected rows straight from the tables
            u32* mcnt = c->d_cnt.as<u32>(); u32* pcnt = mcnt + 2 * n;
            hipLaunchKernelGGL(k_lp_count, dim3((unsigned)ceil_div(2 * n * LP_MAXSEG, 256)), dim3(256), 0, c->stream,
                               c->track_intervals ? lp.ptab[P.cur].as<uint2>() : (const uint2*)nullptr, lp.mtab[P.cur].as<uint2>(), lp.nseg, c->d_map.as<u32>(), 2 * n,
                               c->track_intervals ? pcnt : (u32*)nullptr, mcnt);
            KCHECK();
            for (int pass = 0; pass < 2; pass++) {
                if (pass == 1 && !c->track_intervals) continue;
                HIPC(hipMemcpyAsync(tmp.data(), pass == 0 ? mcnt : pcnt, 2 * n * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
                HIPC(hipStreamSynchronize(c->stream));
                for (size_t i = 0; i < n; i++) for (int h = 0; h < 2; h++) counts[((i * nchr + k) * 2 + h) * 2 + pass] = tmp[2 * i + h];
            }
            continue;
        }
        for (int pass = 0; pass < 2; pass++) {
            if (pass == 1 && !c->track_intervals) continue;
            const u32* soff = pass == 0 ? P.st[k].moff[P.cur].as<u32>() : P.st[k].poff[P.cur].as<u32>();
            hipLaunchKernelGGL(k_csr_gather_count, dim3((unsigned)ceil_div(2 * n, 256)), dim3(256), 0, c->stream, soff, c->d_map.as<u32>(), 2 * n, c->d_cnt.as<u32>());
            KCHECK();
            HIPC(hipMemcpyAsync(tmp.data(), c->d_cnt.p, 2 * n * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
            HIPC(hipStreamSynchronize(c->stream));
            for (size_t i = 0; i < n; i++) for (int h = 0; h < 2; h++) counts[((i * nchr + k) * 2 + h) * 2 + pass] = tmp[2 * i + h];
        }
    }
    return GEV_OK;
}
int gev_export_size(gev_ctx* c, int pop, const uint64_t* positions, size_t n, size_t* bytes)
{
    GEVC(check_idx(c, pop, 0));
    if (!bytes || (n && !positions)) return fail(GEV_EINVAL, "export_size: null argument");
    PopState& P = c->pop[pop];
    if (!P.gen0) return fail(GEV_ESTATE, "export_size: population %d has no current generation", pop);
    HIPC(hipSetDevice(c->device));
    std::vector<u32> counts, map;
    GEVC(export_counts(c, pop, positions, n, counts, map));
    *bytes = pack_layout(c, P, n, counts).total;
    return GEV_OK;
}
int gev_export_rows(gev_ctx* c, int pop, const uint64_t* positions, size_t n, void* device_buf, size_t bytes)
{
    GEVC(check_idx(c, pop, 0));
    if (n && (!positions || !device_buf)) return fail(GEV_EINVAL, "export_rows: null argument");
    PopState& P = c->pop[pop];
    if (!P.gen0) return fail(GEV_ESTATE, "export_rows: population %d has no current generation", pop);
    HIPC(hipSetDevice(c->device));
    GEVC(gev_sync(c));
    std::vector<u32> counts, map;
    GEVC(export_counts(c, pop, positions, n, counts, map));
    const PackLayout L = pack_layout(c, P, n, counts);
    if (bytes < L.total) return fail(GEV_EINVAL, "export_rows: buffer of %zu bytes, %zu needed", bytes, L.total);
    if (!n) return GEV_OK;
    hipStream_t st = c->stream;
    uint8_t* out = (uint8_t*)device_buf;
    const int nchr = c->nchr;
    HIPC(hipMemcpyAsync(out + L.counts, counts.data(), counts.size() * sizeof(u32), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_gather_u8, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, out + L.sex, P.d_sex[P.cur].as<uint8_t>(), c->d_map.as<u32>(), 1u, n);   // d_map holds haplotype rows 2*individual, 2*individual+1
    size_t po = L.planes, co = L.cv, mo = L.muts, pa = L.parts;
    GEVC(c->d_cnt.ensure((2 * n + 1) * sizeof(u32) * 4, st));
    u32* d_off = c->d_cnt.as<u32>() + 2 * n + 1;
    for (int k = 0; k < nchr; k++) {
        if (!c->chr_active[k]) continue;
        ChrStatic& S = P.cs[k]; ChrState& cs = P.st[k];
        const u32 chunks = (u32)(S.stride / 16);
        if (c->dense && c->migrant_rows) {
            hipLaunchKernelGGL(k_copy_rows16, dim3((unsigned)ceil_div(2 * n * chunks, 256)), dim3(256), 0, st, flat_rows(out + po, S.stride),
                               pool_rows(P, k, cs.phys[P.pcur].as<u32>()), c->d_map.as<u32>(), (size_t)0, 2 * n, chunks);
            po = al16(po + 2 * n * S.stride);
        }
        if (cs.lp.valid) {                                // pieces -> the records' lists, for the selected rows only
            ChrState::LpState& lp = cs.lp;
            const bool track = c->track_intervals;
            u32* mcnt = c->d_cnt.as<u32>(); u32* moff = mcnt + (2 * n + 1); u32* pcnt = moff + (2 * n + 1); u32* poff = pcnt + (2 * n + 1);
            const unsigned blocks = (unsigned)ceil_div(2 * n * LP_MAXSEG, 256);
            hipLaunchKernelGGL(k_lp_count, dim3(blocks), dim3(256), 0, st, track ? lp.ptab[P.cur].as<uint2>() : (const uint2*)nullptr, lp.mtab[P.cur].as<uint2>(), lp.nseg,
                               c->d_map.as<u32>(), 2 * n, track ? pcnt : (u32*)nullptr, mcnt);
            KCHECK();
            GEVC(scan_u32(c, mcnt, 2 * n, moff, nullptr));
            if (track) GEVC(scan_u32(c, pcnt, 2 * n, poff, nullptr));
            hipLaunchKernelGGL(k_lp_fill, dim3(blocks), dim3(256), 0, st, track ? lp.ptab[P.cur].as<uint2>() : (const uint2*)nullptr, lp.mtab[P.cur].as<uint2>(),
                               lp.parena.as<LpPart>(), lp.marena.as<u64>(), lp.nseg, c->d_map.as<u32>(), 2 * n, (u64)S.rbp.back(),
                               poff, (gev_part*)(out + pa), moff, (u64*)(out + mo));
            KCHECK();
        } else for (int pass = 0; pass < 2; pass++) {
            if (pass == 1 && !c->track_intervals) continue;
            const u32* soff = pass == 0 ? cs.moff[P.cur].as<u32>() : cs.poff[P.cur].as<u32>();
            hipLaunchKernelGGL(k_csr_gather_count, dim3((unsigned)ceil_div(2 * n, 256)), dim3(256), 0, st, soff, c->d_map.as<u32>(), 2 * n, c->d_cnt.as<u32>());
            KCHECK();
            GEVC(scan_u32(c, c->d_cnt.as<u32>(), 2 * n, d_off, nullptr));
            if (pass == 0) hipLaunchKernelGGL((k_csr_gather_fill<u64>), dim3((unsigned)ceil_div(2 * n, 256)), dim3(256), 0, st, soff, cs.mpos[P.cur].as<u64>(), c->d_map.as<u32>(), 2 * n, d_off, (u64*)(out + mo));
            else hipLaunchKernelGGL((k_csr_gather_fill<gev_part>), dim3((unsigned)ceil_div(2 * n, 256)), dim3(256), 0, st, soff, cs.parts[P.cur].as<gev_part>(), c->d_map.as<u32>(), 2 * n, d_off, (gev_part*)(out + pa));
            KCHECK();
        }
        mo = al16(mo + L.mut_chr[k] * sizeof(u64)); pa = al16(pa + L.parts_chr[k] * sizeof(gev_part));
    }
    for (int p = 0; p < c->nphen; p++) for (int k = 0; k < nchr; k++) {
        if (!c->chr_active[k]) continue;
        CvStatic& V = P.cv[p][k];
        const u32 cch = V.stride_w32 / 4;
        hipLaunchKernelGGL(k_gather_rows16, dim3((unsigned)ceil_div(2 * n * cch, 256)), dim3(256), 0, st, (uint4*)(out + co), (size_t)cch,
                           (const uint4*)P.cvp[p][k][P.cur].p, (size_t)cch, c->d_map.as<u32>(), 2 * n, cch);
        co = al16(co + 2 * n * V.stride_w32 * sizeof(u32));
    }
    KCHECK();
    HIPC(hipStreamSynchronize(st));
    return GEV_OK;
}
int gev_remove_rows(gev_ctx* c, int pop, const uint64_t* positions, size_t n)
{
    GEVC(check_idx(c, pop, 0));
    if (n && !positions) return fail(GEV_EINVAL, "remove_rows: null positions");
    PopState& P = c->pop[pop];
    if (!P.gen0) return fail(GEV_ESTATE, "remove_rows: population %d has no current generation", pop);
    std::vector<uint8_t> gone(P.n_people, 0);
    for (size_t i = 0; i < n; i++) {
        if (positions[i] >= P.n_people || gone[positions[i]]) return fail(GEV_EINVAL, "remove_rows: position out of range or listed twice");
        gone[positions[i]] = 1;
    }
    if (n >= P.n_people) return fail(GEV_EINVAL, "remove_rows: population %d would become empty", pop);
    // no row moves: only the logical order changes; stayers keep their order (src/Simulation.cpp:960-966)
    std::vector<u32> keep; keep.reserve(P.n_people - n);
    for (size_t i = 0; i < P.n_people; i++) if (!gone[i]) keep.push_back(P.logical.empty() ? (u32)i : P.logical[i]);
    P.logical.swap(keep); P.n_people = P.logical.size(); c->ad_cached_pop = c->ad_host_set_pop = -1;
    return GEV_OK;
}
int gev_import_rows(gev_ctx* c, int pop, const void* device_buf, size_t bytes, size_t n)
{
    GEVC(check_idx(c, pop, 0));
    PopState& P = c->pop[pop];
    if (!P.gen0) return fail(GEV_ESTATE, "import_rows: population %d has no current generation", pop);
    if (!n) return GEV_OK;
    if (!device_buf) return fail(GEV_EINVAL, "import_rows: null buffer");
    HIPC(hipSetDevice(c->device));
    GEVC(gev_sync(c));
    hipStream_t st = c->stream;
    const int nchr = c->nchr;
    const uint8_t* in = (const uint8_t*)device_buf;
    double tr_t = host_ms();
    auto mark = [&](const char* what) {                   // GEV_TRACE_HOST: where an import's time goes (serialises the steps)
        if (!g_trace_host) return;
        (void)hipStreamSynchronize(st);
        const double now = host_ms(); fprintf(stderr, "[gev] import_rows %s %.3f ms\n", what, now - tr_t); tr_t = now;
    };
    if (bytes < n * nchr * 4 * sizeof(u32)) return fail(GEV_EINVAL, "import_rows: buffer too small for its own header");
    std::vector<u32> counts(n * nchr * 4);
    HIPC(hipMemcpyAsync(counts.data(), in, counts.size() * sizeof(u32), hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    const PackLayout L = pack_layout(c, P, n, counts);
    if (bytes < L.total) return fail(GEV_EINVAL, "import_rows: buffer of %zu bytes, %zu expected from its header", bytes, L.total);
    const bool rebuild = c->dense && !c->migrant_rows;
    if (rebuild && !c->track_intervals) return fail(GEV_ESTATE, "import_rows: migrants without genotype rows need the interval state (gev_set_track_intervals)");
    if (rebuild) { GEVC(c->d_flag.ensure(16, st)); HIPC(hipMemsetAsync(c->d_flag.p, 0, 4, st)); }
    const size_t n_old = P.n_phys, n_new = n_old + n, r_old = 2 * n_old;     // physical append behind every existing row
    if (n_new * 2 >= 0xffffffffull) return fail(GEV_EINVAL, "import_rows: too many rows");
    mark("header");
    GEVC(ensure_capacity(c, pop, n_new));                              // keeps the current buffers' content
    mark("capacity");
    HIPC(hipMemcpyAsync(P.d_sex[P.cur].as<uint8_t>() + n_old, in + L.sex, n, hipMemcpyDeviceToDevice, st));
    size_t po = L.planes, co = L.cv, mo = L.muts, pa = L.parts;
    for (int k = 0; k < nchr; k++) {
        if (!c->chr_active[k]) continue;
        ChrStatic& S = P.cs[k]; ChrState& cs = P.st[k];
        if (c->dense) {
            // units for the immigrants' segments: from the free list the generations keep (its entries behind the cursor are named by
            // no table, and nothing is in flight after gev_sync) while it lasts, else from a new one
            const size_t need_units = 2 * n * S.nseg;
            static const bool keep_list = !(getenv("GEV_IMPORT_KEEP_LIST") && atoi(getenv("GEV_IMPORT_KEEP_LIST")) == 0);
            const bool reuse = keep_list && cs.pool_list_valid && !cs.pool_force_rebuild && (size_t)cs.pool_cursor + need_units <= cs.pool_n_free;
            if (!reuse) pool_new_stamp(cs);
            const PoolWork pw = pool_work(c, P, k, P.pcur);           // new slots of the CURRENT generation
            if (!reuse) GEVC(pool_free_list(pw, r_old, st));
            mark("free list");
            GEVC(pool_take(pw, r_old, 2 * n, st));
            if (reuse) cs.pool_cursor += (u32)need_units; else cs.pool_list_valid = false;
            mark("units");
            const u32 chunks = (u32)(S.stride / 16);
            if (c->migrant_rows) {
                hipLaunchKernelGGL(k_copy_rows16, dim3((unsigned)ceil_div(2 * n * chunks, 256)), dim3(256), 0, st, pool_rows(P, k, pw.phys_alt + r_old * S.nseg),
                                   flat_rows((void*)(in + po), S.stride), (const u32*)nullptr, (size_t)0, 2 * n, chunks);
                KCHECK();
                po = al16(po + 2 * n * S.stride);
            } else {
                // the migrants travelled as lists: their rows are the founder mosaic their ancestry intervals describe (:1198-1211),
                // assembled here from the founder panels of the parts' root populations
                std::vector<u32> offp(2 * n + 1, 0);
                for (size_t i = 0; i < n; i++) for (int h = 0; h < 2; h++) offp[2 * i + h + 1] = offp[2 * i + h] + counts[((i * nchr + k) * 2 + h) * 2 + 1];
                std::vector<PanelRef> pr(c->n_pop);
                for (int q = 0; q < c->n_pop; q++) {
                    const ChrStatic& F = c->pop[q].cs[k];
                    pr[q] = PanelRef{F.panel.as<u32>(), (u64)(F.stride / 4), (F.panel_rows && F.L == S.L) ? (u64)F.panel_rows : 0ull};
                }
                // one upload: the rows' part offsets, then the panel table (8-byte aligned behind them)
                const size_t off_words = (offp.size() + 1) & ~(size_t)1;
                std::vector<u32> up(off_words + pr.size() * sizeof(PanelRef) / sizeof(u32), 0);
                memcpy(up.data(), offp.data(), offp.size() * sizeof(u32)); memcpy(up.data() + off_words, pr.data(), pr.size() * sizeof(PanelRef));
                GEVC(h2d(c, c->d_poff, up.data(), up.size() * sizeof(u32)));
                const PanelRef* d_panels = (const PanelRef*)(c->d_poff.as<u32>() + off_words);
                GEVC(c->d_prange.ensure(std::max<size_t>(offp[2 * n], 1) * sizeof(uint2), st));
                if (offp[2 * n]) hipLaunchKernelGGL(k_parts_locus_range, dim3((unsigned)ceil_div((size_t)offp[2 * n], 256)), dim3(256), 0, st, (const gev_part*)(in + pa), (size_t)offp[2 * n],
                                                    P.d_chrdev.as<ChrDev>(), k, c->d_prange.as<uint2>());
                hipLaunchKernelGGL(k_rebuild_rows, dim3((unsigned)(2 * n), (unsigned)ceil_div(chunks, 4 * REBUILD_CPW)), dim3(256), 0, st, c->d_poff.as<u32>(), (const gev_part*)(in + pa),
                                   c->d_prange.as<uint2>(), (u32)S.L, d_panels, c->n_pop, pool_rows(P, k, pw.phys_alt + r_old * S.nseg), chunks, c->d_flag.as<u32>());
                KCHECK();
            }
            mark("rows");
        }
        if (cs.lp.valid) {
            // the lists live as pieces: the immigrants' lists are cut into pieces of their own, appended to the arenas, and their
            // table rows written behind the existing ones; nothing of the residents is touched
            ChrState::LpState& lp = cs.lp;
            const bool track = c->track_intervals;
            std::vector<u32> offm(2 * n + 1, 0), offp(2 * n + 1, 0);
            for (size_t i = 0; i < n; i++) for (int h = 0; h < 2; h++) {
                offm[2 * i + h + 1] = offm[2 * i + h] + counts[((i * nchr + k) * 2 + h) * 2];
                offp[2 * i + h + 1] = offp[2 * i + h] + counts[((i * nchr + k) * 2 + h) * 2 + 1];
            }
            GEVC(c->d_cnt.ensure((2 * n + 1) * sizeof(u32) * 2, st));
            u32* d_offm = c->d_cnt.as<u32>(); u32* d_offp = d_offm + (2 * n + 1);
            HIPC(hipMemcpyAsync(d_offm, offm.data(), (2 * n + 1) * sizeof(u32), hipMemcpyHostToDevice, st));
            HIPC(hipMemcpyAsync(d_offp, offp.data(), (2 * n + 1) * sizeof(u32), hipMemcpyHostToDevice, st));
            const size_t tab_bytes = (r_old + 2 * n) * lp.nseg * sizeof(uint2);
            if (track) GEVC(lp.ptab[P.cur].ensure(tab_bytes, st, /*keep=*/true, 1.25));
            GEVC(lp.mtab[P.cur].ensure(tab_bytes, st, /*keep=*/true, 1.25));
            const size_t need_p = track ? (size_t)lp.p_used + offp[2 * n] + 2 * n * lp.nseg : 0, need_m = (size_t)lp.m_used + offm[2 * n];
            if (need_p >= 0xfffffff0u || need_m >= 0xfffffff0u) return fail(GEV_EDEVICE, "import_rows: list arenas beyond 2^32 entries");
            if (track && need_p > lp.parena.bytes / sizeof(LpPart)) GEVC(lp.parena.ensure(need_p * sizeof(LpPart), st, true, 1.5));
            if (need_m > lp.marena.bytes / sizeof(u64)) GEVC(lp.marena.ensure(std::max<size_t>(need_m, 16) * sizeof(u64), st, true, 1.5));
            HIPC(hipMemsetAsync(lp.ctr.p, 0, 4 * sizeof(u32), st));
            hipLaunchKernelGGL(k_lp_import, dim3((unsigned)ceil_div(2 * n * LP_MAXSEG, 256)), dim3(256), 0, st,
                               track ? d_offp : (const u32*)nullptr, track ? (const gev_part*)(in + pa) : (const gev_part*)nullptr, d_offm, (const u64*)(in + mo),
                               r_old, 2 * n, track ? lp.ptab[P.cur].as<uint2>() : (uint2*)nullptr, lp.mtab[P.cur].as<uint2>(), lp.parena.as<LpPart>(), lp.marena.as<u64>(),
                               lp.p_used, lp.m_used, (u32)(lp.parena.bytes / sizeof(LpPart)), (u32)(lp.marena.bytes / sizeof(u64)), lp.nseg, lp.lgw, (u64)S.rbp.front(),
                               lp.ctr.as<u32>(), lp.ctr.as<u32>() + 2);
            KCHECK();
            u32 h[4] = {0, 0, 0, 0};
            HIPC(hipMemcpyAsync(h, lp.ctr.p, sizeof h, hipMemcpyDeviceToHost, st));
            HIPC(hipStreamSynchronize(st));
            if (h[2]) return fail(GEV_EDEVICE, "import_rows: list arena overflow (internal error)");
            lp.p_used += h[0]; lp.m_used += h[1];
            cs.csr_valid = false;
            mark("list pieces");
        } else for (int pass = 0; pass < 2; pass++) {
            if (pass == 1 && !c->track_intervals) continue;
            size_t& total = pass == 0 ? cs.mut_total[P.cur] : cs.parts_total[P.cur];
            std::vector<u32> off(2 * n + 1);
            off[0] = (u32)total;
            for (size_t i = 0; i < n; i++) for (int h = 0; h < 2; h++) off[2 * i + h + 1] = off[2 * i + h] + counts[((i * nchr + k) * 2 + h) * 2 + pass];
            const size_t add = off[2 * n] - off[0];
            DevBuf& doff = pass == 0 ? cs.moff[P.cur] : cs.poff[P.cur];
            // offsets of the appended rows: entries r_old .. r_old + 2n (entry r_old already equals the old total)
            HIPC(hipMemcpyAsync(doff.as<u32>() + r_old, off.data(), (2 * n + 1) * sizeof(u32), hipMemcpyHostToDevice, st));
            HIPC(hipStreamSynchronize(st));
            if (pass == 0) {
                GEVC(cs.mpos[P.cur].ensure(std::max<size_t>(total + add, 2) * sizeof(u64), st, /*keep=*/true, 1.25));
                if (add) HIPC(hipMemcpyAsync(cs.mpos[P.cur].as<u64>() + total, in + mo, add * sizeof(u64), hipMemcpyDeviceToDevice, st));
            } else {
                GEVC(cs.parts[P.cur].ensure(std::max<size_t>(total + add, 1) * sizeof(gev_part), st, true, 1.25));
                if (add) HIPC(hipMemcpyAsync(cs.parts[P.cur].as<gev_part>() + total, in + pa, add * sizeof(gev_part), hipMemcpyDeviceToDevice, st));
            }
            total += add;
        }
        mo = al16(mo + L.mut_chr[k] * sizeof(u64)); pa = al16(pa + L.parts_chr[k] * sizeof(gev_part));
    }
    for (int p = 0; p < c->nphen; p++) for (int k = 0; k < nchr; k++) {
        if (!c->chr_active[k]) continue;
        CvStatic& V = P.cv[p][k];
        HIPC(hipMemcpyAsync(P.cvp[p][k][P.cur].as<u32>() + r_old * V.stride_w32, in + co, 2 * n * V.stride_w32 * sizeof(u32), hipMemcpyDeviceToDevice, st));
        co = al16(co + 2 * n * V.stride_w32 * sizeof(u32));
        V.frq_valid = false;
    }
    u32 rb_flag = 0;
    if (rebuild) HIPC(hipMemcpyAsync(&rb_flag, c->d_flag.p, sizeof(u32), hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    if (rb_flag & 2u) return fail(GEV_ESTATE, "import_rows: an immigrant descends from a root population whose founder panel is not held here (gev_upload_founder_panel / gev_synth_founder_panel)");
    if (rb_flag & 1u) return fail(GEV_EINVAL, "import_rows: Error: p.hap_index is not in range");
    if (P.logical.empty()) { P.logical.resize(P.n_people); for (size_t i = 0; i < P.n_people; i++) P.logical[i] = (u32)i; }
    for (size_t i = 0; i < n; i++) P.logical.push_back((u32)(n_old + i));
    P.n_phys = n_new; P.n_people = P.logical.size(); c->ad_cached_pop = c->ad_host_set_pop = -1;
    return GEV_OK;
}

// ---- output materialisation ---------------------------------------------------------------
// haplotype slots [slot0, slot0 + n) of the current generation, contiguous, into c->d_stage (already large enough)
static int stage_rows(gev_ctx* c, PopState& P, int chr, size_t slot0, size_t n)
{
    ChrStatic& S = P.cs[chr]; ChrState& cs = P.st[chr];
    const u32 chunks = (u32)(S.stride / 16);
    if (!n) return GEV_OK;
    hipLaunchKernelGGL(k_copy_rows16, dim3((unsigned)ceil_div(n * chunks, 256)), dim3(256), 0, c->stream, flat_rows(c->d_stage.p, S.stride),
                       pool_rows(P, chr, cs.phys[P.pcur].as<u32>()), (const u32*)nullptr, slot0, n, chunks);
    KCHECK();
    return GEV_OK;
}
int gev_download_haps(gev_ctx* c, int pop, int chr, size_t row_begin, size_t n_rows, u64* bits, size_t row_stride_words)
{
    if (c) GEVC(check_dense(c, "download_haps"));
    GEVC(check_idx(c, pop, chr));
    GEVC(check_active(c, chr, "download_haps"));
    PopState& P = c->pop[pop]; ChrStatic& S = P.cs[chr]; ChrState& cs = P.st[chr];
    if (!P.gen0) return fail(GEV_ESTATE, "download_haps: population %d has no current generation", pop);
    GEVC(materialize_order(c, pop));
    GEVC(ensure_csr(c, pop));                     // (the mutation overlay reads whole lists)
    if (row_begin + n_rows > 2 * P.n_people) return fail(GEV_EINVAL, "download_haps: rows [%zu,%zu) beyond 2*n_people=%zu", row_begin, row_begin + n_rows, 2 * P.n_people);
    if (n_rows && (!bits || row_stride_words * 64 < S.L)) return fail(GEV_EINVAL, "download_haps: bad output buffer");
    HIPC(hipSetDevice(c->device));
    GEVC(wait_planes(c));
    hipStream_t st = c->stream;
    const size_t max_rows = std::max<size_t>((64u << 20) / S.stride, 1);
    GEVC(c->d_stage.ensure(std::min(max_rows, std::max<size_t>(n_rows, 1)) * S.stride, st));
    const size_t copy_bytes = std::min(row_stride_words * 8, S.stride);
    for (size_t r0 = 0; r0 < n_rows; r0 += max_rows) {
        const size_t nr = std::min(max_rows, n_rows - r0);
        GEVC(stage_rows(c, P, chr, row_begin + r0, nr));
        hipLaunchKernelGGL(k_snp_apply_mut, dim3((unsigned)ceil_div(nr, 256)), dim3(256), 0, st,
                           row_map(P, chr), c->d_stage.as<u32>(), S.stride / 4, row_begin + r0, nr,
                           cs.moff[P.cur].as<u32>(), cs.mpos[P.cur].as<u64>(), S.d_pos.as<u64>(), (u32)S.L);
        KCHECK();
        if (row_stride_words * 8 > copy_bytes)
            for (size_t r = 0; r < nr; r++) memset((uint8_t*)(bits + (r0 + r) * row_stride_words) + copy_bytes, 0, row_stride_words * 8 - copy_bytes);
        HIPC(hipMemcpy2DAsync(bits + r0 * row_stride_words, row_stride_words * 8, c->d_stage.p, S.stride, copy_bytes, nr, hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
    }
    return GEV_OK;
}
// ---- K9: SNP-major outputs ---------------------------------------------------
// SNP-major bit matrix of SNPs [s0, s0+ns) of the current generation (mutations applied) into c->d_snpmajor
static int snp_major_device(gev_ctx* c, int pop, int chr, size_t s0, size_t ns, size_t& stride_w64)
{
    PopState& P = c->pop[pop]; ChrStatic& S = P.cs[chr]; ChrState& cs = P.st[chr];
    hipStream_t st = c->stream;
    GEVC(ensure_csr(c, pop));
    const size_t rows = 2 * P.n_people;
    stride_w64 = ceil_div(rows, 64);
    GEVC(c->d_snpmajor.ensure(std::max<size_t>(ns * stride_w64 * 8, 16), st));
    HIPC(hipMemsetAsync(c->d_snpmajor.p, 0, ns * stride_w64 * 8, st));
    const u32 n_words = (u32)(((s0 + ns - 1) >> 6) - (s0 >> 6) + 1);
    const u32 wpw = 16;                                              // 16 words = one 128-byte line of every row per wave
    const unsigned gy = (unsigned)ceil_div(ceil_div(n_words, wpw), 4);
    hipLaunchKernelGGL(k_transpose_tiles, dim3((unsigned)stride_w64, gy), dim3(256), 0, st, (const u64*)nullptr, row_map(P, chr), S.stride / 8, rows, (u32)S.L,
                       (u32)s0, (u32)ns, c->d_snpmajor.as<u64>(), stride_w64, wpw);
    hipLaunchKernelGGL(k_snpmajor_apply_mut, dim3((unsigned)ceil_div(rows, 256)), dim3(256), 0, st, row_map(P, chr), rows,
                       cs.moff[P.cur].as<u32>(), cs.mpos[P.cur].as<u64>(), S.d_pos.as<u64>(), (u32)S.L, (u32)s0, (u32)ns,
                       (unsigned long long*)c->d_snpmajor.p, stride_w64);
    KCHECK();
    return GEV_OK;
}
static int snp_range_check(gev_ctx* c, int pop, int chr, size_t s0, size_t ns, const char* who)
{
    GEVC(check_idx(c, pop, chr));
    PopState& P = c->pop[pop];
    if (!P.gen0) return fail(GEV_ESTATE, "%s: population %d has no current generation", who, pop);
    if (s0 + ns > P.cs[chr].L) return fail(GEV_EINVAL, "%s: SNPs [%zu,%zu) beyond the %zu loci of chromosome %d", who, s0, s0 + ns, P.cs[chr].L, chr);
    HIPC(hipSetDevice(c->device));
    GEVC(materialize_order(c, pop));
    GEVC(wait_planes(c));
    return GEV_OK;
}
// SNPs per device chunk so that the staging buffers stay <= ~256 MB
static size_t snp_chunk(size_t bytes_per_snp) { return std::max<size_t>((256u << 20) / std::max<size_t>(bytes_per_snp, 1), 64) & ~(size_t)63; }
int gev_download_snp_major(gev_ctx* c, int pop, int chr, size_t snp_begin, size_t n_snps, u64* bits, size_t row_stride_words)
{
    if (c) GEVC(check_dense(c, "download_snp_major"));
    GEVC(snp_range_check(c, pop, chr, snp_begin, n_snps, "download_snp_major"));
    GEVC(check_active(c, chr, "download_snp_major"));
    PopState& P = c->pop[pop];
    const size_t rows = 2 * P.n_people, w = ceil_div(rows, 64);
    if (n_snps && (!bits || row_stride_words < w)) return fail(GEV_EINVAL, "download_snp_major: bad output buffer");
    const size_t chunk = snp_chunk(w * 8);
    for (size_t s = 0; s < n_snps; s += chunk) {
        const size_t ns = std::min(chunk, n_snps - s); size_t stride;
        GEVC(snp_major_device(c, pop, chr, snp_begin + s, ns, stride));
        if (row_stride_words > w) for (size_t j = 0; j < ns; j++) memset(bits + (s + j) * row_stride_words + w, 0, (row_stride_words - w) * 8);
        HIPC(hipMemcpy2DAsync(bits + s * row_stride_words, row_stride_words * 8, c->d_snpmajor.p, stride * 8, w * 8, ns, hipMemcpyDeviceToHost, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
    }
    return GEV_OK;
}
// bytes of the reference's .hap file for SNP lines [snp_begin, snp_begin+n_snps): n_snps * (4*n_people + 1)
int gev_format_hap_text(gev_ctx* c, int pop, int chr, size_t snp_begin, size_t n_snps, char* out, size_t out_bytes)
{
    if (c) GEVC(check_dense(c, "format_hap_text"));
    GEVC(snp_range_check(c, pop, chr, snp_begin, n_snps, "format_hap_text"));
    GEVC(check_active(c, chr, "format_hap_text"));
    PopState& P = c->pop[pop];
    const size_t rows = 2 * P.n_people, line = 2 * rows + 1;
    if (out_bytes < n_snps * line || (n_snps && !out)) return fail(GEV_EINVAL, "format_hap_text: buffer of %zu bytes, %zu needed", out_bytes, n_snps * line);
    const size_t chunk = snp_chunk(line);
    for (size_t s = 0; s < n_snps; s += chunk) {
        const size_t ns = std::min(chunk, n_snps - s); size_t stride;
        GEVC(snp_major_device(c, pop, chr, snp_begin + s, ns, stride));
        GEVC(c->d_text.ensure(ns * line, c->stream));
        hipLaunchKernelGGL(k_format_hap_text, dim3((unsigned)ceil_div(ceil_div(ns * line, 16), 256)), dim3(256), 0, c->stream, c->d_snpmajor.as<u64>(), stride, rows, (u32)ns, c->d_text.as<char>());
        KCHECK();
        HIPC(hipMemcpyAsync(out + s * line, c->d_text.p, ns * line, hipMemcpyDeviceToHost, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
    }
    return GEV_OK;
}
// hap-major rows (mutations applied) of individuals [ind0, ind0+n) into c->d_stage
static int stage_individuals(gev_ctx* c, int pop, int chr, size_t ind0, size_t n)
{
    PopState& P = c->pop[pop]; ChrStatic& S = P.cs[chr]; ChrState& cs = P.st[chr];
    hipStream_t st = c->stream;
    GEVC(c->d_stage.ensure(std::max<size_t>(2 * n * S.stride, 16), st));
    if (!n) return GEV_OK;
    GEVC(ensure_csr(c, pop));
    GEVC(stage_rows(c, P, chr, 2 * ind0, 2 * n));
    hipLaunchKernelGGL(k_snp_apply_mut, dim3((unsigned)ceil_div(2 * n, 256)), dim3(256), 0, st,
                       row_map(P, chr), c->d_stage.as<u32>(), S.stride / 4, 2 * ind0, 2 * n,
                       cs.moff[P.cur].as<u32>(), cs.mpos[P.cur].as<u64>(), S.d_pos.as<u64>(), (u32)S.L);
    KCHECK();
    return GEV_OK;
}
static int ind_range_check(gev_ctx* c, int pop, int chr, size_t ind0, size_t n, const char* what)
{
    GEVC(check_idx(c, pop, chr));
    PopState& P = c->pop[pop];
    if (!P.gen0) return fail(GEV_ESTATE, "%s: population %d has no current generation", what, pop);
    GEVC(materialize_order(c, pop));
    if (ind0 + n > P.n_people) return fail(GEV_EINVAL, "%s: individuals [%zu,%zu) beyond n_people=%zu", what, ind0, ind0 + n, P.n_people);
    HIPC(hipSetDevice(c->device));
    GEVC(wait_planes(c));
    return GEV_OK;
}
// genotype columns of the PLINK .ped text, one line per individual (the caller writes the six id columns in front)
int gev_format_ped_text(gev_ctx* c, int pop, int chr, size_t ind_begin, size_t n_ind, const char* al0, const char* al1, char* out, size_t out_bytes)
{
    if (c) GEVC(check_dense(c, "format_ped_text"));
    GEVC(ind_range_check(c, pop, chr, ind_begin, n_ind, "format_ped_text"));
    GEVC(check_active(c, chr, "format_ped_text"));
    ChrStatic& S = c->pop[pop].cs[chr];
    const size_t line = 4 * S.L + 1;
    if ((al0 == nullptr) != (al1 == nullptr)) return fail(GEV_EINVAL, "format_ped_text: al0 and al1 must both be given or both be NULL");
    if (out_bytes < n_ind * line || (n_ind && !out)) return fail(GEV_EINVAL, "format_ped_text: buffer of %zu bytes, %zu needed", out_bytes, n_ind * line);
    hipStream_t st = c->stream;
    const char *d_al0 = nullptr, *d_al1 = nullptr;
    if (al0) {
        GEVC(c->d_tmp.ensure(std::max<size_t>(2 * S.L, 16), st));
        HIPC(hipMemcpyAsync(c->d_tmp.p, al0, S.L, hipMemcpyHostToDevice, st));
        HIPC(hipMemcpyAsync(c->d_tmp.as<char>() + S.L, al1, S.L, hipMemcpyHostToDevice, st));
        d_al0 = c->d_tmp.as<char>(); d_al1 = d_al0 + S.L;
    }
    const size_t chunk = std::max<size_t>((256u << 20) / line, 1);
    for (size_t i = 0; i < n_ind; i += chunk) {
        const size_t n = std::min(chunk, n_ind - i);
        GEVC(stage_individuals(c, pop, chr, ind_begin + i, n));
        GEVC(c->d_text.ensure(n * line, st));
        hipLaunchKernelGGL(k_format_ped_text, dim3((unsigned)ceil_div(ceil_div(n * line, 16), 256)), dim3(256), 0, st, c->d_stage.as<u32>(), S.stride / 4, n, (u32)S.L, d_al0, d_al1, c->d_text.as<char>());
        KCHECK();
        HIPC(hipMemcpyAsync(out + i * line, c->d_text.p, n * line, hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
    }
    return GEV_OK;
}
// matrix_plink_ped rows (bit 2*snp + hap), ceil(2L/64) words per individual
int gev_download_plink_matrix(gev_ctx* c, int pop, int chr, size_t ind_begin, size_t n_ind, u64* bits, size_t row_stride_words)
{
    if (c) GEVC(check_dense(c, "download_plink_matrix"));
    GEVC(ind_range_check(c, pop, chr, ind_begin, n_ind, "download_plink_matrix"));
    GEVC(check_active(c, chr, "download_plink_matrix"));
    ChrStatic& S = c->pop[pop].cs[chr];
    const size_t w32 = ceil_div(S.L, 32);                  // one source word -> one 64-bit output word
    if (n_ind && (!bits || row_stride_words < w32)) return fail(GEV_EINVAL, "download_plink_matrix: bad output buffer (need %zu words per row)", w32);
    hipStream_t st = c->stream;
    const size_t chunk = std::max<size_t>((256u << 20) / std::max<size_t>(w32 * 8, 1), 1);
    for (size_t i = 0; i < n_ind; i += chunk) {
        const size_t n = std::min(chunk, n_ind - i);
        GEVC(stage_individuals(c, pop, chr, ind_begin + i, n));
        GEVC(c->d_text.ensure(std::max<size_t>(n * w32 * 8, 16), st));
        if (w32) hipLaunchKernelGGL(k_interleave_haps, dim3((unsigned)ceil_div(n * w32, 256)), dim3(256), 0, st, c->d_stage.as<u32>(), S.stride / 4, n, (u32)w32, c->d_text.as<u64>(), w32);
        KCHECK();
        if (row_stride_words > w32)
            for (size_t r = 0; r < n; r++) memset(bits + (i + r) * row_stride_words + w32, 0, (row_stride_words - w32) * 8);
        if (w32) HIPC(hipMemcpy2DAsync(bits + i * row_stride_words, row_stride_words * 8, c->d_text.p, w32 * 8, w32 * 8, n, hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
    }
    return GEV_OK;
}
// CommFunc::ras_rank on the device (the O(n^2) host loop of assort_mate, src/Simulation.cpp:2278-2279): stable radix sort of
// order-preserving keys with the reference's tie / NaN rule (gev_sort.hip); the all-pairs form (k_rank_f64) stays selectable with
// GEV_RANK_ALLPAIRS=1 as an independent cross-check
extern "C" size_t gev_rank_scratch_bytes(size_t n);
extern "C" int gev_rank_device(const double* d_x, size_t n, unsigned long long* d_rank, void* d_tmp, hipStream_t st);
int gev_rank_f64(gev_ctx* c, const double* x, size_t n, unsigned long long* rank_out)
{
    if (!c) return fail(GEV_EINVAL, "null context");
    if (n && (!x || !rank_out)) return fail(GEV_EINVAL, "rank_f64: null buffer");
    if (!n) return GEV_OK;
    if (n >= 0xffffffffull) return fail(GEV_EINVAL, "rank_f64: too many values");
    HIPC(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    GEVC(c->d_tmp.ensure(n * 16, st));
    double* dx = c->d_tmp.as<double>(); unsigned long long* dr = (unsigned long long*)(dx + n);
    HIPC(hipMemcpyAsync(dx, x, n * sizeof(double), hipMemcpyHostToDevice, st));
    static const bool allpairs = getenv("GEV_RANK_ALLPAIRS") != nullptr;
    if (allpairs) {
        hipLaunchKernelGGL(k_rank_f64, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, dx, n, dr);
        KCHECK();
    } else {
        GEVC(c->d_text.ensure(gev_rank_scratch_bytes(n), st));
        const int e = gev_rank_device(dx, n, dr, c->d_text.p, st);
        if (e) return fail(GEV_EDEVICE, "rank_f64: HIP error %s in the sort", hipGetErrorString((hipError_t)e));
    }
    HIPC(hipMemcpyAsync(rank_out, dr, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    return GEV_OK;
}
// K8: rows [row_begin, +n_rows) x loci [snp_begin, +n_snps) of the genotype matrix from the interval state and founder tiles
// (one per root population: after migration a part may descend from another population's founders, :1204)
// checks + founder tiles to the device (c->d_snpmajor, stacked population after population; row offsets in c->d_map)
static int materialize_prepare(gev_ctx* c, int pop, int chr, size_t row_begin, size_t n_rows, size_t snp_begin, size_t n_snps,
                               const u64* const* founder_bits, const size_t* founder_stride_words, const size_t* n_founder_rows)
{
    GEVC(check_idx(c, pop, chr));
    GEVC(check_active(c, chr, "materialize"));
    PopState& P = c->pop[pop]; ChrStatic& S = P.cs[chr];
    if (!P.gen0) return fail(GEV_ESTATE, "materialize: population %d has no current generation", pop);
    if (!c->track_intervals) return fail(GEV_ESTATE, "materialize: interval tracking is disabled");
    GEVC(materialize_order(c, pop));
    if (row_begin + n_rows > 2 * P.n_people) return fail(GEV_EINVAL, "materialize: rows [%zu,%zu) beyond 2*n_people=%zu", row_begin, row_begin + n_rows, 2 * P.n_people);
    if (snp_begin + n_snps > S.L) return fail(GEV_EINVAL, "materialize: SNPs [%zu,%zu) beyond L=%zu", snp_begin, snp_begin + n_snps, S.L);
    const size_t w64 = ceil_div(n_snps, 64);
    if (!n_rows || !n_snps) return GEV_OK;
    if (!founder_bits || !founder_stride_words || !n_founder_rows) return fail(GEV_EINVAL, "materialize: bad founder tile");
    std::vector<u64> row0(c->n_pop + 1, 0);
    for (int p = 0; p < c->n_pop; p++) {
        if (n_founder_rows[p] && (!founder_bits[p] || founder_stride_words[p] < w64)) return fail(GEV_EINVAL, "materialize: bad founder tile of population %d", p);
        row0[p + 1] = row0[p] + n_founder_rows[p];
    }
    if (!row0[c->n_pop]) return fail(GEV_EINVAL, "materialize: bad founder tile");
    HIPC(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    GEVC(c->d_snpmajor.ensure(row0[c->n_pop] * w64 * 8, st));
    for (int p = 0; p < c->n_pop; p++)
        if (n_founder_rows[p])
            HIPC(hipMemcpy2DAsync(c->d_snpmajor.as<u64>() + row0[p] * w64, w64 * 8, founder_bits[p], founder_stride_words[p] * 8, w64 * 8, n_founder_rows[p], hipMemcpyHostToDevice, st));
    GEVC(c->d_map.ensure((c->n_pop + 1) * sizeof(u64) + 16, st));
    HIPC(hipMemcpyAsync(c->d_map.p, row0.data(), (c->n_pop + 1) * sizeof(u64), hipMemcpyHostToDevice, st));
    HIPC(hipStreamSynchronize(st));                      // row0 is a local
    GEVC(c->d_flag.ensure(16, st));
    HIPC(hipMemsetAsync(c->d_flag.p, 0, 4, st));
    return GEV_OK;
}
// rows [row0, row0+nr) of the tile into `out` (nr x w64 words, mutations applied); `plain` = scratch of the same size
static int materialize_rows(gev_ctx* c, int pop, int chr, size_t row0, size_t nr, size_t snp_begin, size_t n_snps, u64* plain, u64* out)
{
    PopState& P = c->pop[pop]; ChrStatic& S = P.cs[chr]; ChrState& cs = P.st[chr];
    hipStream_t st = c->stream;
    const size_t w64 = ceil_div(n_snps, 64), w32 = 2 * w64;
    GEVC(ensure_csr(c, pop));
    HIPC(hipMemsetAsync(plain, 0, nr * w64 * 8, st));                                     // pad bits of the last word stay 0
    hipLaunchKernelGGL(k_materialize_tile, dim3((unsigned)ceil_div(nr * ceil_div(n_snps, 32), 256)), dim3(256), 0, st,
                       cs.poff[P.cur].as<u32>(), cs.parts[P.cur].as<gev_part>(), row0, nr, S.d_pos.as<u64>(), (u32)snp_begin, (u32)n_snps,
                       c->d_snpmajor.as<u32>(), w32, c->d_map.as<u64>(), c->n_pop, (u32*)plain, w32, c->d_flag.as<u32>());
    HIPC(hipMemcpyAsync(out, plain, nr * w64 * 8, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_tile_apply_mut, dim3((unsigned)ceil_div(nr, 256)), dim3(256), 0, st, (const u32*)plain, (u32*)out, w32, row0, nr,
                       cs.moff[P.cur].as<u32>(), cs.mpos[P.cur].as<u64>(), S.d_pos.as<u64>(), (u32)snp_begin, (u32)n_snps);
    KCHECK();
    return GEV_OK;
}
static int materialize_status(gev_ctx* c)
{
    u32 flag = 0;
    HIPC(hipMemcpy(&flag, c->d_flag.p, 4, hipMemcpyDeviceToHost));
    if (flag) return fail(GEV_EINVAL, "materialize: Error: p.hap_index is not in range");
    return GEV_OK;
}
int gev_materialize_pops(gev_ctx* c, int pop, int chr, size_t row_begin, size_t n_rows, size_t snp_begin, size_t n_snps,
                         const u64* const* founder_bits, const size_t* founder_stride_words, const size_t* n_founder_rows, u64* bits, size_t row_stride_words)
{
    GEVC(materialize_prepare(c, pop, chr, row_begin, n_rows, snp_begin, n_snps, founder_bits, founder_stride_words, n_founder_rows));
    const size_t w64 = ceil_div(n_snps, 64);
    if (!n_rows || !n_snps) return GEV_OK;
    if (!bits || row_stride_words < w64) return fail(GEV_EINVAL, "materialize: bad output buffer");
    hipStream_t st = c->stream;
    const size_t max_rows = std::max<size_t>((128u << 20) / (w64 * 8), 1);
    for (size_t r0 = 0; r0 < n_rows; r0 += max_rows) {
        const size_t nr = std::min(max_rows, n_rows - r0);
        GEVC(c->d_stage.ensure(nr * w64 * 8, st)); GEVC(c->d_text.ensure(nr * w64 * 8, st));
        GEVC(materialize_rows(c, pop, chr, row_begin + r0, nr, snp_begin, n_snps, c->d_stage.as<u64>(), c->d_text.as<u64>()));
        if (row_stride_words > w64)
            for (size_t r = 0; r < nr; r++) memset(bits + (r0 + r) * row_stride_words + w64, 0, (row_stride_words - w64) * 8);
        HIPC(hipMemcpy2DAsync(bits + r0 * row_stride_words, row_stride_words * 8, c->d_text.p, w64 * 8, w64 * 8, nr, hipMemcpyDeviceToHost, st));
        HIPC(hipStreamSynchronize(st));
    }
    return materialize_status(c);
}
// PLINK .bed body of SNPs [snp_begin, +n_snps) straight from the interval state (BASELINE config 5: "PLINK bit-packed output" of a
// population whose genotype matrix is never resident): tile of ALL haplotype rows -> 64x64 bit-tile transpose -> 2-bit packing
int gev_materialize_bed(gev_ctx* c, int pop, int chr, size_t snp_begin, size_t n_snps,
                        const u64* const* founder_bits, const size_t* founder_stride_words, const size_t* n_founder_rows, uint8_t* out, size_t out_bytes)
{
    if (!c) return fail(GEV_EINVAL, "null context");
    GEVC(check_idx(c, pop, chr));
    PopState& P = c->pop[pop];
    const size_t rows = 2 * P.n_people, bpl = ceil_div(P.n_people, 4), w64 = ceil_div(n_snps, 64);
    GEVC(materialize_prepare(c, pop, chr, 0, rows, snp_begin, n_snps, founder_bits, founder_stride_words, n_founder_rows));
    if (!n_snps || !rows) return GEV_OK;
    if (!out || out_bytes < n_snps * bpl) return fail(GEV_EINVAL, "materialize_bed: buffer of %zu bytes, %zu needed", out_bytes, n_snps * bpl);
    hipStream_t st = c->stream;
    GEVC(c->d_stage.ensure(rows * w64 * 8, st)); GEVC(c->d_tmp.ensure(rows * w64 * 8, st));
    GEVC(materialize_rows(c, pop, chr, 0, rows, snp_begin, n_snps, c->d_stage.as<u64>(), c->d_tmp.as<u64>()));
    const size_t stride_sm = ceil_div(rows, 64);
    GEVC(c->d_text.ensure(std::max<size_t>(n_snps * stride_sm * 8 + n_snps * bpl, 16), st));
    u64* snpmajor = c->d_text.as<u64>(); uint8_t* bed = c->d_text.as<uint8_t>() + n_snps * stride_sm * 8;
    HIPC(hipMemsetAsync(snpmajor, 0, n_snps * stride_sm * 8, st));
    const u32 wpw = 16;
    hipLaunchKernelGGL(k_transpose_tiles, dim3((unsigned)stride_sm, (unsigned)ceil_div(ceil_div(w64, wpw), 4)), dim3(256), 0, st, c->d_tmp.as<u64>(), RowMap{}, w64, rows, (u32)n_snps,
                       0u, (u32)n_snps, snpmajor, stride_sm, wpw);
    hipLaunchKernelGGL(k_format_bed, dim3((unsigned)ceil_div(n_snps * bpl, 256)), dim3(256), 0, st, snpmajor, stride_sm, P.n_people, (u32)n_snps, bed);
    KCHECK();
    HIPC(hipMemcpyAsync(out, bed, n_snps * bpl, hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    return materialize_status(c);
}
// one population per context: a single founder tile
int gev_materialize(gev_ctx* c, int pop, int chr, size_t row_begin, size_t n_rows, size_t snp_begin, size_t n_snps,
                    const u64* founder_bits, size_t founder_stride_words, size_t n_founder_rows, u64* bits, size_t row_stride_words)
{
    if (!c) return fail(GEV_EINVAL, "null context");
    if (c->n_pop > 1) return fail(GEV_EINVAL, "materialize: this context has %d populations: use gev_materialize_pops (one founder tile per root population)", c->n_pop);
    return gev_materialize_pops(c, pop, chr, row_begin, n_rows, snp_begin, n_snps, &founder_bits, &founder_stride_words, &n_founder_rows, bits, row_stride_words);
}
// GT columns of the VCF data lines: n_snps * (4*n_people + 1) bytes; the caller writes the nine fixed columns in front
int gev_format_vcf_gt(gev_ctx* c, int pop, int chr, size_t snp_begin, size_t n_snps, char* out, size_t out_bytes)
{
    if (c) GEVC(check_dense(c, "format_vcf_gt"));
    GEVC(snp_range_check(c, pop, chr, snp_begin, n_snps, "format_vcf_gt"));
    GEVC(check_active(c, chr, "format_vcf_gt"));
    PopState& P = c->pop[pop];
    const size_t line = 4 * P.n_people + 1;
    if (out_bytes < n_snps * line || (n_snps && !out)) return fail(GEV_EINVAL, "format_vcf_gt: buffer of %zu bytes, %zu needed", out_bytes, n_snps * line);
    const size_t chunk = snp_chunk(line);
    for (size_t s = 0; s < n_snps; s += chunk) {
        const size_t ns = std::min(chunk, n_snps - s); size_t stride;
        GEVC(snp_major_device(c, pop, chr, snp_begin + s, ns, stride));
        GEVC(c->d_text.ensure(ns * line, c->stream));
        hipLaunchKernelGGL(k_format_vcf_gt, dim3((unsigned)ceil_div(ceil_div(ns * line, 16), 256)), dim3(256), 0, c->stream, c->d_snpmajor.as<u64>(), stride, P.n_people, (u32)ns, c->d_text.as<char>());
        KCHECK();
        HIPC(hipMemcpyAsync(out + s * line, c->d_text.p, ns * line, hipMemcpyDeviceToHost, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
    }
    return GEV_OK;
}
// PLINK .bed body (SNP-major, without the 3 magic bytes 0x6c 0x1b 0x01): n_snps * ceil(n_people/4) bytes
int gev_format_bed(gev_ctx* c, int pop, int chr, size_t snp_begin, size_t n_snps, uint8_t* out, size_t out_bytes)
{
    if (c) GEVC(check_dense(c, "format_bed"));
    GEVC(snp_range_check(c, pop, chr, snp_begin, n_snps, "format_bed"));
    GEVC(check_active(c, chr, "format_bed"));
    PopState& P = c->pop[pop];
    const size_t bpl = ceil_div(P.n_people, 4);
    if (out_bytes < n_snps * bpl || (n_snps && !out)) return fail(GEV_EINVAL, "format_bed: buffer of %zu bytes, %zu needed", out_bytes, n_snps * bpl);
    const size_t chunk = snp_chunk(ceil_div(2 * P.n_people, 64) * 8);
    for (size_t s = 0; s < n_snps; s += chunk) {
        const size_t ns = std::min(chunk, n_snps - s); size_t stride;
        GEVC(snp_major_device(c, pop, chr, snp_begin + s, ns, stride));
        GEVC(c->d_text.ensure(ns * bpl, c->stream));
        hipLaunchKernelGGL(k_format_bed, dim3((unsigned)ceil_div(ns * bpl, 256)), dim3(256), 0, c->stream, c->d_snpmajor.as<u64>(), stride, P.n_people, (u32)ns, c->d_text.as<uint8_t>());
        KCHECK();
        HIPC(hipMemcpyAsync(out + s * bpl, c->d_text.p, ns * bpl, hipMemcpyDeviceToHost, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
    }
    return GEV_OK;
}

int gev_download_cv(gev_ctx* c, int pop, int phen, int chr, u64* bits, size_t row_stride_words)
{
    GEVC(check_idx(c, pop, chr, phen));
    GEVC(check_active(c, chr, "download_cv"));
    PopState& P = c->pop[pop]; CvStatic& V = P.cv[phen][chr];
    if (!P.gen0) return fail(GEV_ESTATE, "download_cv: population %d has no current generation", pop);
    GEVC(materialize_order(c, pop));
    if (!bits || row_stride_words * 64 < V.C) return fail(GEV_EINVAL, "download_cv: bad output buffer");
    HIPC(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const size_t rows = 2 * P.n_people;
    GEVC(c->d_cvm.ensure(std::max<size_t>(rows * V.sub_w32 * sizeof(u32), 16), st));
    GEVC(c->d_tmp.ensure(rows * row_stride_words * 8, st));
    hipLaunchKernelGGL(k_cv_resolve, dim3((unsigned)ceil_div(rows * V.sub_w32, 256)), dim3(256), 0, st,
                       P.cvp[phen][chr][P.cur].as<u32>(), V.stride_w32, V.sub_w32, c->d_cvm.as<u32>(), rows);
    HIPC(hipMemsetAsync(c->d_tmp.p, 0, rows * row_stride_words * 8, st));
    if (V.C)
        hipLaunchKernelGGL(k_permute_cols, dim3((unsigned)ceil_div(rows * V.sub_w32, 256)), dim3(256), 0, st,
                           c->d_cvm.as<u32>(), (size_t)V.sub_w32, c->d_tmp.as<u32>(), row_stride_words * 2, V.sub_w32, V.d_col_of_icv.as<u32>(), V.C, rows);
    KCHECK();
    HIPC(hipMemcpyAsync(bits, c->d_tmp.p, rows * row_stride_words * 8, hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    return GEV_OK;
}
int gev_download_intervals(gev_ctx* c, int pop, int chr, gev_part* out, u64* hap_offsets, size_t* n_parts)
{
    GEVC(check_idx(c, pop, chr));
    GEVC(check_active(c, chr, "download_intervals"));
    PopState& P = c->pop[pop]; ChrState& cs = P.st[chr];
    if (!P.gen0) return fail(GEV_ESTATE, "download_intervals: population %d has no current generation", pop);
    GEVC(materialize_order(c, pop));
    GEVC(ensure_csr(c, pop));
    if (!c->track_intervals) return fail(GEV_ESTATE, "download_intervals: interval tracking is disabled");
    if (!n_parts) return fail(GEV_EINVAL, "download_intervals: n_parts is null");
    HIPC(hipSetDevice(c->device));
    const size_t rows = 2 * P.n_people;
    std::vector<u32> off(rows + 1);
    HIPC(hipMemcpyAsync(off.data(), cs.poff[P.cur].p, (rows + 1) * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    *n_parts = off[rows];
    if (hap_offsets) for (size_t r = 0; r <= rows; r++) hap_offsets[r] = off[r];
    if (out && off[rows]) {
        HIPC(hipMemcpyAsync(out, cs.parts[P.cur].p, (size_t)off[rows] * sizeof(gev_part), hipMemcpyDeviceToHost, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
    }
    return GEV_OK;
}
int gev_download_mutations(gev_ctx* c, int pop, int chr, u64* out, u64* hap_offsets, size_t* n_mut)
{
    GEVC(check_idx(c, pop, chr));
    GEVC(check_active(c, chr, "download_mutations"));
    PopState& P = c->pop[pop]; ChrState& cs = P.st[chr];
    if (!P.gen0) return fail(GEV_ESTATE, "download_mutations: population %d has no current generation", pop);
    GEVC(materialize_order(c, pop));
    GEVC(ensure_csr(c, pop));
    if (!n_mut) return fail(GEV_EINVAL, "download_mutations: n_mut is null");
    HIPC(hipSetDevice(c->device));
    const size_t rows = 2 * P.n_people;
    std::vector<u32> off(rows + 1);
    HIPC(hipMemcpyAsync(off.data(), cs.moff[P.cur].p, (rows + 1) * sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    *n_mut = off[rows];
    if (hap_offsets) for (size_t r = 0; r <= rows; r++) hap_offsets[r] = off[r];
    if (out && off[rows]) {
        HIPC(hipMemcpyAsync(out, cs.mpos[P.cur].p, (size_t)off[rows] * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
        HIPC(hipStreamSynchronize(c->stream));
    }
    return GEV_OK;
}

// ---- introspection ------------------------------------------------------------------------
int gev_pop_size(gev_ctx* c, int pop, size_t* n)          // (the published generation's size; allowed while a generation is in flight)
{
    if (!c || !n) return fail(GEV_EINVAL, "null");
    if (pop < 0 || pop >= c->n_pop) return fail(GEV_EINVAL, "population index %d out of range", pop);
    *n = c->pop[pop].n_people;
    return GEV_OK;
}
int gev_plane_ptr(gev_ctx* c, int pop, int chr, void** dptr, size_t* unit_bytes, size_t* n_slots, const uint32_t** unit_of, uint32_t* segments_per_row)
{
    if (c) GEVC(check_dense(c, "plane_ptr"));
    GEVC(check_idx(c, pop, chr));
    GEVC(check_active(c, chr, "plane_ptr"));
    GEVC(gev_sync(c));
    if (c->pop[pop].gen0) GEVC(materialize_order(c, pop));
    PopState& P = c->pop[pop];
    if (dptr) *dptr = P.st[chr].pool.p;
    if (unit_of) *unit_of = P.st[chr].phys[P.pcur].as<u32>();
    if (unit_bytes) *unit_bytes = P.cs[chr].unit_bytes();
    if (segments_per_row) *segments_per_row = P.cs[chr].nseg;
    if (n_slots) *n_slots = 2 * P.n_people;
    return GEV_OK;
}
int gev_stream(gev_ctx* c, void** s) { if (!c || !s) return fail(GEV_EINVAL, "null"); *s = (void*)c->stream; return GEV_OK; }
int gev_last_reproduce_ms(gev_ctx* c, float ms[4]) { if (!c || !ms) return fail(GEV_EINVAL, "null"); GEVC(gev_sync(c)); for (int i = 0; i < 4; i++) ms[i] = c->last_ms[i]; return GEV_OK; }
int gev_set_track_intervals(gev_ctx* c, int on)
{
    if (!c) return fail(GEV_EINVAL, "null");
    if (c->pend.active) return fail(GEV_ESTATE, "a gev_reproduce_begin is pending: call gev_reproduce_end first");
    if (c->track_intervals == (on != 0)) return GEV_OK;
    for (int p = 0; p < c->n_pop; p++) if (c->pop[p].gen0) { GEVC(ensure_csr(c, p)); lists_changed_in_csr(c, c->pop[p]); }   // the pieces are rebuilt with / without the interval tables
    c->track_intervals = on != 0;
    return GEV_OK;
}
int gev_list_stats(gev_ctx* c, int pop, int chr, unsigned long long out[10])
{
    GEVC(check_idx(c, pop, chr));
    if (!out) return fail(GEV_EINVAL, "list_stats: null argument");
    const ChrState::LpState& lp = c->pop[pop].st[chr].lp;
    out[0] = lp.imports; out[1] = lp.p_used; out[2] = lp.m_used; out[3] = lp.parena.bytes / sizeof(LpPart); out[4] = lp.marena.bytes / sizeof(u64);
    out[5] = lp.nseg; out[6] = lp.p_last; out[7] = lp.m_last; out[8] = lp.compactions; out[9] = 0;
    return GEV_OK;
}
int gev_redo_count(gev_ctx* c, unsigned long long* n) { if (!c || !n) return fail(GEV_EINVAL, "null"); *n = c->redo_count; return GEV_OK; }
int gev_set_overlap(gev_ctx* c, int on)
{
    if (!c) return fail(GEV_EINVAL, "null");
    GEVC(gev_sync(c));
    c->overlap_mode = on < 0 ? -1 : std::min(on, 2);
    c->sparse_after_stitch = c->overlap_mode == 2;
    if (c->overlap_mode >= 0) c->serialize = c->overlap_mode == 0; else { c->serialize = true; c->auto_gens = 0; c->auto_small_ms = c->auto_stitch_ms = 0; }
    return GEV_OK;
}
int gev_set_stitch_mode(gev_ctx* c, int mode) { if (c && c->pend.active) return fail(GEV_ESTATE, "a gev_reproduce_begin is pending: call gev_reproduce_end first"); if (!c || mode < 0 || mode > 1) return fail(GEV_EINVAL, "stitch mode must be 0 (work list of the segments to write) or 1 (gamete-major)"); c->stitch_mode = mode; return GEV_OK; }

// ---- diagnostics (tests only; no simulation state involved) --------------------------------
__global__ void __launch_bounds__(64) k_dbg_rand(const GevRngTables* __restrict__ T, u32 seed, u32 n, int* __restrict__ out)
{
    GlibcWave g; g.seed(T, seed);
    for (u32 i = 0; i < n; i++) { const u32 v = g.out(T, i); if (threadIdx.x == 0) out[i] = (int)v; }
}
// one gamete of ras_sim_loc_rec: breakpoints + the next two rand() outputs
__global__ void __launch_bounds__(64) k_dbg_sim_loc_rec(const GevRngTables* __restrict__ T, const ChrDev* __restrict__ chrs, int chr, u32 seed,
                                                        u64* __restrict__ locs, u32 cap, u32* __restrict__ n_out, int* __restrict__ next2)
{
    const ChrDev& C = chrs[chr];
    GlibcWave g; g.seed(T, seed);
    u32 h = 0;
    wave_scan_hits(T, seed + 1u, C.rthr, C.r_amax, 0, C.R, [&](u32 row) {
        const u64 v = C.rbp[row] + (u64)g.out(T, h) % C.bp_dist;
        if (threadIdx.x == 0 && h < cap) locs[h] = v;
        h++;
    });
    const u32 a = g.out(T, h), b = g.out(T, h + 1);
    if (threadIdx.x == 0) { *n_out = h; next2[0] = (int)a; next2[1] = (int)b; }
}
// every word of the resident plane (dense stitch) against the interval state (k_parts) + the synthetic founder panel
int gev_dbg_verify_planes(gev_ctx* c, int pop, int chr, const uint64_t* founder_seeds, unsigned long long* n_bad_words, unsigned long long* n_bad_parts)
{
    if (c) GEVC(check_dense(c, "dbg_verify_planes"));
    GEVC(check_idx(c, pop, chr));
    GEVC(check_active(c, chr, "dbg_verify_planes"));
    PopState& P = c->pop[pop]; ChrStatic& S = P.cs[chr]; ChrState& cs = P.st[chr];
    if (!P.gen0) return fail(GEV_ESTATE, "dbg_verify_planes: population %d has no current generation", pop);
    if (!c->track_intervals) return fail(GEV_ESTATE, "dbg_verify_planes: interval tracking is disabled");
    if (!founder_seeds || !n_bad_words || !n_bad_parts) return fail(GEV_EINVAL, "dbg_verify_planes: null argument");
    HIPC(hipSetDevice(c->device));
    GEVC(materialize_order(c, pop));
    GEVC(ensure_csr(c, pop));
    GEVC(gev_sync(c));
    hipStream_t st = c->stream;
    const size_t rows = 2 * P.n_people, words = ceil_div(S.L, 32);
    if (!rows || !words) { *n_bad_words = 0; *n_bad_parts = 0; return GEV_OK; }
    if (ceil_div(rows * words, 256) > 0x7fffffffull) return fail(GEV_EINVAL, "dbg_verify_planes: grid too large");
    // per ROOT population: allele-frequency thresholds of its synthetic panel, its seed, its number of founder haplotypes
    // (a population this context never generated founders for is taken to have as many as this one: same panel shape on every GPU)
    const int np = c->n_pop;
    GEVC(c->d_thr32.ensure((size_t)np * S.L * sizeof(u32), st));
    std::vector<u64> meta(2 * (size_t)np);
    for (int r = 0; r < np; r++) {
        meta[r] = founder_seeds[r];
        meta[np + r] = c->pop[r].cs[chr].founder_rows ? c->pop[r].cs[chr].founder_rows : S.founder_rows;
        hipLaunchKernelGGL(k_synth_thresholds, dim3((unsigned)ceil_div(S.L, 256)), dim3(256), 0, st, c->d_thr32.as<u32>() + (size_t)r * S.L, S.L, founder_seeds[r]);
    }
    GEVC(h2d(c, c->d_map, meta.data(), meta.size() * sizeof(u64)));
    GEVC(c->d_flag.ensure(16, st));
    HIPC(hipMemsetAsync(c->d_flag.p, 0, 16, st));
    hipLaunchKernelGGL(k_verify_plane, dim3((unsigned)ceil_div(rows * words, 256)), dim3(256), 0, st, cs.poff[P.cur].as<u32>(), cs.parts[P.cur].as<gev_part>(), rows,
                       S.d_pos.as<u64>(), (u32)S.L, row_map(P, chr), c->d_thr32.as<u32>(), c->d_map.as<u64>(), np, c->d_map.as<u64>() + np,
                       (unsigned long long*)c->d_flag.p);
    KCHECK();
    unsigned long long h[2] = {0, 0};
    HIPC(hipMemcpyAsync(h, c->d_flag.p, 16, hipMemcpyDeviceToHost, st));
    HIPC(hipStreamSynchronize(st));
    *n_bad_words = h[0]; *n_bad_parts = h[1];
    return GEV_OK;
}
// Exhaustive check of the FP64 prefilter of the batched sampling kernels (gev_sample8.h:scan_candidates) over engine states
// [x_begin, x_end) and all SB_J multipliers: with the exact high digit x2 = x * c_j mod M and F = floor(x2 * 2^20 / M), the 20 low
// mantissa bits L of fma(x, fl(c_j / M), 1.5 * 2^32 + 4 u) must satisfy L - 4 - F in {-1, 0, 1, 2} (mod 2^20): then every draw
// whose x2 is <= amax passes L < floor(amax * 2^20 / M) + 7, for every amax.  out[0] = violations, out[1] = largest (L - 4 - F),
// out[2] = largest (F + 4 - L).
__global__ void __launch_bounds__(256) k_dbg_prefilter_sweep(const GevRngTables* __restrict__ Tg, u32 x_begin, u32 x_end, unsigned long long* __restrict__ out)
{
    __shared__ double s_kd[SB_J]; __shared__ u32 s_ci[SB_J];
    if (threadIdx.x < SB_J) { const u32 c = powmod31(Tg->pow128, threadIdx.x); s_ci[threadIdx.x] = c; s_kd[threadIdx.x] = (double)c / 2147483647.0; }
    __syncthreads();
    const double C = 6442450944.0 + 4.0 / 1048576.0;
    unsigned long long bad = 0; int dpos = 0, dneg = 0;
    for (u64 x = (u64)x_begin + (u64)blockIdx.x * 256 + threadIdx.x; x < x_end; x += (u64)gridDim.x * 256) {
        const double xd = (double)(u32)x;
        for (u32 j = 0; j < SB_J; j++) {
            const u32 x2 = mulmod31((u32)x, s_ci[j]);
            const u32 F = (u32)(((u64)x2 << 20) / 2147483647ull);
            const u32 L = (u32)__double2loint(__fma_rn(xd, s_kd[j], C)) & 0xfffffu;
            int d = (int)((L - 4u - F) & 0xfffffu);
            if (d >= (1 << 19)) d -= (1 << 20);
            if (d < -1 || d > 2) bad++;
            dpos = max(dpos, d); dneg = max(dneg, -d);
        }
    }
    if (bad) atomicAdd(&out[0], bad);
    atomicMax(&out[1], (unsigned long long)dpos); atomicMax(&out[2], (unsigned long long)dneg);
}
int gev_dbg_prefilter_sweep(gev_ctx* c, uint32_t x_begin, uint32_t x_end, unsigned long long out[3])
{
    if (!c || !out) return fail(GEV_EINVAL, "null");
    if (x_begin < 1 || x_end > 2147483647u || x_begin > x_end) return fail(GEV_EINVAL, "dbg_prefilter_sweep: engine states are 1 .. 2^31-2");
    HIPC(hipSetDevice(c->device));
    GEVC(c->d_flag.ensure(32, c->stream));
    HIPC(hipMemsetAsync(c->d_flag.p, 0, 24, c->stream));
    hipLaunchKernelGGL(k_dbg_prefilter_sweep, dim3(8192), dim3(256), 0, c->stream, c->d_tables.as<GevRngTables>(), x_begin, x_end, (unsigned long long*)c->d_flag.p);
    KCHECK();
    HIPC(hipMemcpyAsync(out, c->d_flag.p, 24, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    return GEV_OK;
}
int gev_dbg_tables(void* out, size_t bytes)
{
    if (bytes != sizeof(GevRngTables)) return fail(GEV_EINVAL, "gev_dbg_tables: expected %zu bytes", sizeof(GevRngTables));
    GevRngTables T; gev_build_rng_tables(T); memcpy(out, &T, sizeof T); return GEV_OK;
}
int gev_dbg_threshold(double p, uint32_t out[4])
{
    GevThr t;
    if (!gev_make_threshold(p, t)) return fail(GEV_EUNSUPPORTED, "window wider than 2");
    out[0] = t.a_lo; out[1] = t.a_hi; out[2] = t.b0; out[3] = t.b1; return GEV_OK;
}
double gev_dbg_canonical(uint32_t a, uint32_t b) { return gev_canonical(a, b); }
int gev_dbg_rand(gev_ctx* c, uint32_t seed, uint32_t n, int* out)
{
    if (!c || !out) return fail(GEV_EINVAL, "null");
    HIPC(hipSetDevice(c->device));
    GEVC(c->d_tmp.ensure(std::max<size_t>(n, 4) * sizeof(int), c->stream));
    hipLaunchKernelGGL(k_dbg_rand, dim3(1), dim3(64), 0, c->stream, c->d_tables.as<GevRngTables>(), seed, n, c->d_tmp.as<int>());
    KCHECK();
    HIPC(hipMemcpyAsync(out, c->d_tmp.p, n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    return GEV_OK;
}
int gev_dbg_sim_loc_rec(gev_ctx* c, int pop, int chr, uint32_t seed, uint64_t* locs, uint32_t cap, uint32_t* n, int next2[2])
{
    GEVC(check_idx(c, pop, chr));
    HIPC(hipSetDevice(c->device));
    GEVC(finalize_static(c, pop));
    GEVC(c->d_tmp.ensure((size_t)cap * 8 + 64, c->stream));
    u64* dl = c->d_tmp.as<u64>(); u32* dn = (u32*)(dl + cap); int* dx = (int*)(dn + 2);
    hipLaunchKernelGGL(k_dbg_sim_loc_rec, dim3(1), dim3(64), 0, c->stream, c->d_tables.as<GevRngTables>(), c->pop[pop].d_chrdev.as<ChrDev>(), chr, seed, dl, cap, dn, dx);
    KCHECK();
    HIPC(hipMemcpyAsync(n, dn, 4, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipMemcpyAsync(next2, dx, 8, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    HIPC(hipMemcpyAsync(locs, dl, (size_t)std::min(*n, cap) * 8, hipMemcpyDeviceToHost, c->stream));
    HIPC(hipStreamSynchronize(c->stream));
    return GEV_OK;
}

} // extern "C"
