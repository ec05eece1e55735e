/* geneevolve_amd.h -- C-ABI of the MI355X-native GeneEvolve reproduction hot path.
 *
 * The reference (MMesbahU/GeneEvolve) has no plugin/FFI interface; the seam is the set of
 * private members of class Simulation (reference src/Simulation.h:64-144) listed below.  A
 * maintainer replaces their bodies with calls to this library (INTEGRATION.md shows the
 * patch).  Every entry point names the reference code it replaces.
 *
 *   reference seam (src/Simulation.h)                      entry point here
 *   ------------------------------------------------------ ---------------------------------
 *   ras_init_parameters (readers, src/Simulation.cpp:164)  gev_set_rmap / gev_set_mutmap /
 *                                                          gev_set_snps / gev_set_cvs /
 *                                                          gev_upload_founders / gev_upload_cv_founders
 *   bool ras_initial_human_gen0(int ipop)          :86     gev_init_gen0
 *   bool random_mate(int ipop,int gen_ind)         :78     gev_random_mate
 *   unsigned ras_glob_seed(void)                   :67     gev_glob_seeds; inside gev_generation_begin
 *   sim_next_generation: random_mate -> reproduce -> ras_compute_AD as one unit     gev_generation_begin / _end
 *   std::vector<Human> reproduce(int,int)          :81     gev_reproduce
 *   bool ras_compute_AD(int ipop,int gen_num)      :80     gev_compute_ad
 *   bool ras_do_migration(int gen_ind)             :71     gev_migrate (+ gev_export_rows /
 *                                                          gev_import_rows across GPUs)
 *   ras_convert_interval_to_hap_matrix             :105    gev_download_haps
 *   ras_write_hap_to_interval_format               :116    gev_download_intervals
 *
 * Conventions (mirroring the reference's): every call returns 0 on success and a negative
 * GEV_E* code on failure (the reference returns bool false and prints one line; the line is
 * available from gev_last_error()).  No exceptions cross the boundary.  The caller owns every
 * host buffer it passes; the library owns all device memory.  Exactly one host thread drives
 * one context (the reference is single threaded).  One context lives on one GPU; one process
 * per GPU.
 *
 * Bit packing of every genotype buffer crossing the boundary: haplotype-major rows
 * (row = 2*individual + chromatid, as Hap_SNP.hap in reference src/format_hap.h:18-31),
 * locus ii of a row is bit (ii & 63) of 64-bit little-endian word ii >> 6; rows are
 * `row_stride_words` words apart; pad bits are zero.
 *
 * RNG coupling: the host keeps drawing Simulation::ras_glob_seed()
 * (src/Simulation.cpp:17-21) exactly as the reference does and hands the values in:
 * 1 value per gev_init_gen0, 1 per gev_reproduce plus 1 per (offspring, chromosome) when a
 * mutation map is loaded (the draw inside ras_add_mutation, src/Simulation.cpp:2500).
 */
#ifndef GENEEVOLVE_AMD_H
#define GENEEVOLVE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GEV_OK          0
#define GEV_EINVAL     -1   /* bad argument / inconsistent sizes                                   */
#define GEV_ESTATE     -2   /* call order violated (e.g. reproduce before init_gen0)               */
#define GEV_EDEVICE    -3   /* HIP runtime error, no device, out of device memory                  */
#define GEV_ENAN       -4   /* "A or D is nan" (reference src/Simulation.cpp:2716-2720)            */
#define GEV_EUNSUPPORTED -5 /* input outside the dense-equivalence preconditions   */
#define GEV_ENOMATE    -6   /* "Error: No one can marry" (reference src/Simulation.cpp:2125-2129)  */

typedef struct gev_ctx gev_ctx;

/* class Couples_Info, reference src/Population.h:165-180.  pos_* index the CURRENT
 * generation of the population by position (not by ID). */
typedef struct gev_couple {
    uint64_t pos_male;
    uint64_t pos_female;
    int32_t  inbreed;        /* != 0: couple is skipped (src/Simulation.cpp:2440) */
    int32_t  num_offspring;
} gev_couple;

/* class part, reference src/Population.h:20-51, without gen0_indv (a pure function of
 * hap_index and root_population: founder id + ".1"/".2", src/Simulation.cpp:3031-3033)
 * and without mutation_pos (see gev_download_mutations). */
typedef struct gev_part {
    uint64_t st;
    uint64_t en;
    uint64_t hap_index;
    int32_t  root_population;
    int32_t  reserved;
} gev_part;

/* one migrant: individual at position src_pos of population src_pop (positions BEFORE the
 * migration step) moves to dst_pop.  Order = the order in which the reference appends
 * migrants (src/Simulation.cpp:971-981). */
typedef struct gev_move {
    int32_t  src_pop;
    int32_t  dst_pop;
    uint64_t src_pos;
} gev_move;

const char* gev_last_error(void);
const char* gev_version(void);

/* device < 0: use the current HIP device.
 * Environment read here (tuning / cross-check knobs; none changes a result):
 *   GEV_OVERLAP=0|1|2|-1        stream overlap mode (gev_set_overlap)
 *   GEV_ALIAS_ROWS=0            write every segment of every gamete row (default 1: a segment without a crossover boundary shares the parental unit)
 *   GEV_STITCH_WG_PER_CU=n|auto dense-stitch workgroups per CU (default: unlimited; auto = measured at run time)
 *   GEV_SEG_CHUNKS=2^k          16-byte chunks per row segment (default 128 = 2 KiB; a row has at most 64 segments, longer rows get larger ones)
 *   GEV_STITCH_MODE=0|1         stitch kernel (gev_set_stitch_mode)
 *   GEV_STITCH_LDS_PAD=bytes    (experiments) dynamic LDS padding of the stitch workgroups, overriding the one derived from WG_PER_CU
 *   GEV_SAMPLE_BATCHED=0        one sampling task per wave (the round-1 kernels) instead of eight
 *   GEV_LIST_SEGS=n             most position ranges per row of the shared list pieces (default 32; 1 = one piece per list), GEV_LIST_ARENA=n
 *                               entries per row of their arenas (default: a tenth of the device memory) -- csrc/gev_lists.h
 *   GEV_STREAM_PRIO=xxxxx       h|m|l: priorities of the main, head-start, mating, list and stitch streams (default hhhhh);
 *                               GEV_STITCH_START=0|1|2: where in a generation the stitch is enqueued (default 2: behind the small work)
 *                               GEV_HEAD_START=1: the next generation's seeds and sampling are enqueued in front of the generation's own
 *                               work and waited for separately (seeds: mating, sampling: unit table); default 0 (same rate at config 2)
 *   GEV_CHAIN_WG=0              serial-chain mode with one wave per link; GEV_CHAIN_MAX_TASKS=n: most tasks accepted without a mutation map
 *   GEV_TABLE_RING_BYTES=n      minimum size of the pinned ring the per-generation work tables are staged through
 *   GEV_AD_SHARED=0             several root populations with bit-identical CV effects: per-haplotype a/d lookup anyway (default: the one-population term table)
 *   GEV_AD_RP_FAST=0            several root populations with their own CV effects: per-haplotype a/d lookup in global memory (k_ad_accumulate) even
 *                               when the CV files are in position order (default then: k_ad_accumulate_rp, the piece's values in LDS)
 *   GEV_IMPORT_KEEP_LIST=0      gev_import_rows builds a new free list of row units instead of taking from the one the generations keep
 *   GEV_OVF_CAP=n, GEV_LIST_HEADROOM=n  (tests) initial size of the breakpoint / new-mutation overflow regions, spare list entries per row:
 *                               tiny values make generations overflow their buffers, so that the grow-and-enqueue-again path runs */
int  gev_create(gev_ctx** out, int device, int n_pop, int nchr, int nphen);
void gev_destroy(gev_ctx* ctx);

/* ---- static inputs (what ras_init_parameters loads) -------------------------------------- */
/* rMap + _recom_prob of one chromosome (src/Population.h:183-189, src/Population.cpp:471-480).
 * bp_dist = rMap::bp_dist_in_rmap (src/Population.cpp:396-397). */
int gev_set_rmap(gev_ctx*, int pop, int chr, const uint64_t* bp, const double* recom_prob,
                 size_t R, uint64_t bp_dist);
/* MutationMap of one chromosome (src/Population.h:192-197); rates already range-checked by the
 * reader (src/Population.cpp:458).  M may be 0. */
int gev_set_mutmap(gev_ctx*, int pop, int chr, const uint64_t* bp, const double* rate, size_t M);
/* Legend.pos of the founder panel (src/format_hap.h); must be non-decreasing. */
int gev_set_snps(gev_ctx*, int pop, int chr, const uint64_t* pos, size_t L);
/* CV_INFO of one phenotype x chromosome (src/Population.h:201-208): bp, genetic_value_a,
 * genetic_value_d in FILE order (the A/D sum runs in this order); vd = Phenotype_scheme::_vd. */
int gev_set_cvs(gev_ctx*, int pop, int phen, int chr, const uint64_t* bp, const double* a,
                const double* d, size_t C, double vd);
/* Hap_SNP.hap founder panel, haplotype-major bit rows (nhap rows x L loci). */
int gev_upload_founders(gev_ctx*, int pop, int chr, const uint64_t* bits, size_t row_stride_words,
                        size_t nhap, size_t L);
/* CV.val of one phenotype x chromosome (src/Population.h:212-216), columns in FILE order. */
int gev_upload_cv_founders(gev_ctx*, int pop, int phen, int chr, const uint64_t* bits,
                           size_t row_stride_words, size_t nhap, size_t C);
/* Synthetic founder panel generated on the device (bench / large parity cases): allele of
 * (hap, locus) ~ Bernoulli(f_locus), f_locus ~ U(0.05,0.5), counter-based hash of `seed`.
 * Not a reference function: SURVEY.md section 8(d) prescribes device-side generation. */
int gev_synth_founders(gev_ctx*, int pop, int chr, size_t nhap, uint64_t seed);
int gev_synth_cv_founders(gev_ctx*, int pop, int phen, int chr, size_t nhap, uint64_t seed);

/* ---- Simulation::ras_initial_human_gen0 (src/Simulation.cpp:3000-3072) --------------------
 * individual i = founder haplotypes 2i, 2i+1 as two whole-chromosome parts
 * [rmap.bp[0], rmap.bp[last]); sex[i] = rand()%2+1 after srand(seed_gen0).
 * sex_out: n_people bytes (1 = male, 2 = female) or NULL. */
int gev_init_gen0(gev_ctx*, int pop, size_t n_people, uint32_t seed_gen0, uint8_t* sex_out);

/* ---- Simulation::reproduce (src/Simulation.cpp:2394-2493) ---------------------------------
 * Replaces the population's current generation by its offspring.
 *  seed_reproduce : the ras_glob_seed() value drawn at :2398
 *  mut_seeds      : NULL when no mutation map is loaded (_mutation_map.size()==0, :2459);
 *                   else the n_people*nchr ras_glob_seed() values drawn inside
 *                   ras_add_mutation (:2500), in call order (offspring-major, chromosome-minor)
 *  n_people       : sum of num_offspring over couples with inbreed==0 (checked)
 *  sex_out        : n_people bytes, Human::sex of each offspring (:2472), or NULL
 * Pedigree ids and common_sibling (:2473-2484) are pure host bookkeeping and stay with the host.
 *  couples == NULL : the couples the preceding gev_random_mate of `pop` left on the device (n_couples is ignored, n_people must be
 *                   that call's pop_size).
 * Cost note: with a mutation map every (offspring, chromosome) task restarts its rand() chain at srand(mut_seed), so all tasks are
 * sampled in parallel.  WITHOUT one (mut_seeds == NULL) the reference's chain runs through every gamete in order (the seed of a
 * gamete is the rand() output that follows the previous gamete's breakpoints): one wave walks it, about 10 us per gamete, i.e.
 * seconds per generation beyond ~10^5 gametes -- a parity mode for Example1-sized inputs, not a production path. */
int gev_reproduce(gev_ctx*, int pop, const gev_couple* couples, size_t n_couples,
                  uint32_t seed_reproduce, const uint32_t* mut_seeds, size_t n_mut_seeds,
                  size_t n_people, uint8_t* sex_out);

/* Optional head start for the next gev_reproduce of `pop`: crossover / mutation sampling and the rand() seed chain depend
 * on the ras_glob_seed() values and the offspring count only, not on the couples, so they can be enqueued as soon as those
 * are known (random mating with one child per couple: n_people = pop_size) while the host is still forming couples.
 * Returns without waiting.  gev_reproduce called next with the same seed_reproduce, mut_seeds and n_people uses the
 * results; with anything else it samples again.  Same argument checks / errors as gev_reproduce. */
int gev_presample(gev_ctx*, int pop, uint32_t seed_reproduce, const uint32_t* mut_seeds, size_t n_mut_seeds, size_t n_people);
/* gev_reproduce in two halves.  _begin checks and stages the inputs and enqueues the generation's device work; _end waits for it
 * (repeating it with larger buffers if a list capacity was exceeded), publishes the new generation and returns the sexes.
 * gev_reproduce = _begin + _end.  Between the two the host is free -- bench.py forms the next generation's couples there -- but no
 * other call on the context is allowed (GEV_ESTATE). */
int gev_reproduce_begin(gev_ctx*, int pop, const gev_couple* couples, size_t n_couples, uint32_t seed_reproduce,
                        const uint32_t* mut_seeds, size_t n_mut_seeds, size_t n_people);
int gev_reproduce_end(gev_ctx*, uint8_t* sex_out);
/* The sexes of the generation whose sampling gev_presample has enqueued: they come out of the rand() chain of the sampling
 * (src/Simulation.cpp:2472) and depend on nothing else, so a host that mates at random (Simulation::random_mate reads nothing but
 * the sexes) can form the couples of the generation AFTER the one it is about to hand over.  Waits for the sampling kernels only;
 * call it after gev_presample and before the gev_reproduce(_begin) of that generation. */
int gev_presample_sex(gev_ctx*, int pop, uint8_t* sex_out, size_t n_people);

/* ---- Simulation::random_mate (src/Simulation.cpp:2090-2157)  [SURVEY 8(f) row 2] ----------------
 * Forms pop_size couples of the population's current generation on the device, bit for bit as the reference does:
 *  seed                 : the ras_glob_seed() value drawn at :2092
 *  selection_value_func : Human::selection_value_func of every individual [n_people] (:2113), or NULL when every value is 1
 *                         (selection function "none" / generation 0): r < 1 always holds, so the draws of generator(seed)
 *                         cannot change the outcome and are skipped
 *  couples_out          : pop_size Couples_Info records (pos_male, pos_female, inbreed = 0, num_offspring = 1), or NULL
 *  num_*_mate           : the two counts the reference prints (:2122-2123), or NULL
 * Marriageable males / females are compacted in position order (:2109-2118), father and mother of couple i are
 * pos_male[i_f], pos_female[i_m] with i_f / i_m the i-th values of uniform_int_distribution<unsigned long>(0, count-1) on
 * default_random_engine(seed+1) / (seed+2) (:2132-2147).  The sexes are the ones the library itself produced (gev_init_gen0,
 * gev_reproduce; they follow the individuals through gev_migrate / gev_export_rows / gev_import_rows).
 * The couples also stay on the device: the next gev_reproduce of `pop` takes them when its `couples` is NULL.
 * Returns GEV_ENOMATE with the reference's message when no male or no female may marry (:2125-2129). */
int gev_random_mate(gev_ctx*, int pop, uint32_t seed, const double* selection_value_func, size_t pop_size,
                    gev_couple* couples_out, size_t* num_males_mate, size_t* num_females_mate);
/* ---- Simulation::ras_glob_seed (src/Simulation.cpp:17-21) called n times, evaluated on the device ----
 * *engine_state: the state of glob_generator (std::minstd_rand0: `os << glob_generator` prints it, `is >> glob_generator` sets
 * it) before the n calls; on return the state after them.  out: n values (host), or NULL. */
int gev_glob_seeds(gev_ctx*, uint32_t* engine_state, size_t n, uint32_t* out);
/* ---- one whole generation of a randomly mating population: the body of Simulation::sim_next_generation for --RM
 * (src/Simulation.cpp:1907-1935): random_mate -> reproduce -> ras_compute_AD, enqueued as ONE unit.  Every ras_glob_seed() value
 * the three draw -- 1 (:2092) + 1 (:2398) + pop_size * nchr (:2500, only with a mutation map) -- is taken from glob_state on the
 * device, the couples never leave it, and the host waits once (in _end).
 *  glob_state            : state of glob_generator in front of random_mate's draw
 *  selection_value_func  : as gev_random_mate
 * gev_generation_end waits (enqueues the generation again with larger buffers if a list outgrew its capacity), publishes the new
 * generation and returns: res (may be NULL), the couples (pop_size records, or NULL), the offspring sexes (pop_size bytes, or
 * NULL).  gev_compute_ad then returns the generation's A/D without further device work.  Between _begin and _end no other call
 * on the context is allowed (GEV_ESTATE).  GEV_ENOMATE as gev_random_mate (nothing is published). */
typedef struct gev_generation_result {
    uint32_t glob_state;                 /* glob_generator behind the generation's last draw: the host stores it back */
    uint32_t seed_mate, seed_reproduce;  /* the values drawn at :2092 and :2398 (common_sibling needs seed_reproduce + 1, :2417) */
    uint32_t reserved;
    uint64_t num_males_mate, num_females_mate;   /* :2119-2123 */
} gev_generation_result;
int gev_generation_begin(gev_ctx*, int pop, uint32_t glob_state, size_t pop_size, const double* selection_value_func);
int gev_generation_end(gev_ctx*, gev_generation_result* res, gev_couple* couples_out, uint8_t* sex_out);
/* Head start across generations.  draws_between >= 0: the host promises that glob_generator makes exactly that many ras_glob_seed()
 * draws of its own (ras_scale_AD_compute_GEF :3078, ras_do_migration :921, ...) between the state gev_generation_end returns and the
 * state it passes to the next gev_generation_begin.  The state of the next call is then known on the device as soon as this
 * generation's seeds are drawn: the library draws the next generation's seeds and runs its sampling right away, next to this
 * generation's dense stitch, and the next gev_generation_begin (same population, same pop_size) only adds what depends on the
 * couples.  A call that arrives with another state, population or size samples again: results never depend on the promise.
 * draws_between < 0 (default): no head start. */
int gev_set_generation_chain(gev_ctx*, int draws_between);
/* ---- Simulation::ras_compute_AD + ras_find_cv (src/Simulation.cpp:2624-2815) --------------
 * additive/dominance : [n_people * nphen], index ih*nphen + iphen  (Human::additive/dominance, raw)
 * add_chr/dom_chr    : [n_people * nchr * nphen], index (ih*nchr + ichr)*nphen + iphen, or NULL
 * bv = additive + dominance is left to the host (:2715, :2744). */
int gev_compute_ad(gev_ctx*, int pop, double* additive, double* dominance,
                   double* add_chr, double* dom_chr);
/* (gev_reproduce / gev_generation_end compute the new generation's A/D with the generation, so the call after them only copies.
 * The totals of the PUBLISHED generation can also be read while the next one is in flight -- between gev_generation_begin and _end:
 * a host without selection hands the next generation over first and reads this one's values while the device works; the
 * per-chromosome arrays need the device and are refused there.) */
/* ---- Simulation::ras_scale_AD_compute_GEF (src/Simulation.cpp:3075-3206)  [SURVEY 8(f) row 1] ----
 * Scales the raw A/D of the last gev_compute_ad to the generation-0 variances, draws the noise term
 * e ~ N(0,1) from std::normal_distribution on minstd_rand0(seed) (seed = the ras_glob_seed() value
 * drawn at :3078), standardises it with CommFunc::var, and assembles phen = A + D + C + E + F.
 * The host supplies what the reference reads from host-side records: common_sibling[i]
 * (Human::common_sibling), and for gen_num > 0 the parents' values f_father[i], f_mother[i]
 * (_Pop_info_prev_gen phen or parental_effect looked up by ID_Father / ID_Mother, :3118-3131); any may be
 * NULL (= 0).  Outputs: n doubles each, any may be NULL.  Floats agree with the reference within 1e-12
 * relative (device log() and parallel sums differ from glibc / sequential sums in the last bits). */
typedef struct gev_gef_params {
    double va, vd, ve, vf, beta;        /* Phenotype_scheme::_va,_vd,_ve,_vf,_beta                        */
    double s2_a_gen0, s2_d_gen0;        /* Population::_var_a_gen0 / _var_d_gen0 of this phenotype        */
    int32_t gen_num, reserved;
} gev_gef_params;
int gev_scale_ad_compute_gef(gev_ctx*, int pop, int phen, const gev_gef_params* par, uint32_t seed,
                             const double* common_sibling, const double* f_father, const double* f_mother,
                             double* additive, double* dominance, double* bv, double* e_noise,
                             double* parental_effect, double* phen_out);

/* Locus-split populations (gev_set_chr_active): every context of the population computes the A/D of its own chromosomes; after
 * the all-reduce + chromosome-ordered sum (geneevolve_amd/distributed.py:compute_ad_locus_split) each context is handed the raw
 * totals [n_people * nphen] of the current generation, which is what gev_scale_ad_compute_gef scales (:3104-3105). */
int gev_set_ad(gev_ctx*, int pop, const double* additive, const double* dominance);

/* population allele frequency frq[icv] of the last gev_compute_ad (:2647-2655), FILE order. */
int gev_get_cv_freq(gev_ctx*, int pop, int phen, int chr, double* frq, size_t C);

/* ---- Simulation::ras_do_migration (src/Simulation.cpp:877-989) ----------------------------
 * The host decides WHO moves exactly as the reference does; the library moves the rows:
 * migrants are removed from their origin populations (stable order of the remainder, :960-966)
 * and appended to their destinations in `moves` order (:971-981). */
int gev_migrate(gev_ctx*, const gev_move* moves, size_t n_moves);
/* Cross-GPU form of the same step (one context per GPU, exchange done by the caller over
 * RCCL): pack the complete state of the individuals at `positions` into one contiguous device
 * buffer / remove them / append individuals packed by a peer.  Record layout is opaque but
 * identical across contexts with identical static inputs; gev_export_size gives the byte count. */
int gev_export_size(gev_ctx*, int pop, const uint64_t* positions, size_t n, size_t* bytes);
int gev_export_rows(gev_ctx*, int pop, const uint64_t* positions, size_t n, void* device_buf, size_t bytes);
int gev_remove_rows(gev_ctx*, int pop, const uint64_t* positions, size_t n);
int gev_import_rows(gev_ctx*, int pop, const void* device_buf, size_t bytes, size_t n);
/* Migrants that travel as lists.  A migrant's genotype row is redundant with what travels beside it: it is the founder mosaic its
 * ancestry intervals describe -- out[ii] = pops[part.root_population].hap_snps[part.hap_index][ii] for the part that contains
 * pos[ii], Simulation::ras_convert_interval_to_hap_matrix (src/Simulation.cpp:1198-1211) -- with the mutations kept beside it as
 * a sparse list.  A context that holds the founder panels of the root populations (read-only copies: what the reference keeps
 * as Population::hap_snps for the whole run, src/Population.h) can therefore rebuild the rows where they arrive:
 *  gev_upload_founder_panel / gev_synth_founder_panel : Hap_SNP.hap of ROOT population `pop`, chromosome chr, kept for good
 *                            (same arguments as gev_upload_founders / gev_synth_founders; `pop` need not be simulated here;
 *                            gev_set_snps of (pop, chr) first).  Needs resident planes (not gev_set_dense_state 0).
 *  gev_set_migrant_rows(0) : gev_export_size / gev_export_rows leave the genotype rows out of the records (sexes, CV rows, mutation
 *                            and interval lists travel as before), gev_import_rows assembles the immigrants' rows from the
 *                            panels.  Needs the interval state (gev_set_track_intervals 1).  Both ends of an exchange use the
 *                            same setting; gev_import_rows returns GEV_ESTATE when an immigrant descends from a root population
 *                            without a panel here, GEV_EINVAL with the reference's "p.hap_index is not in range" (:1205-1209).
 * At BASELINE config 2's shape a migrant is 250 KB of rows against 0.2-20 KB of lists. */
int gev_upload_founder_panel(gev_ctx*, int pop, int chr, const uint64_t* bits, size_t row_stride_words, size_t nhap, size_t L);
int gev_synth_founder_panel(gev_ctx*, int pop, int chr, size_t nhap, uint64_t seed);
int gev_set_migrant_rows(gev_ctx*, int on);

/* ---- output materialisation --------------------------------------------------------------
 * == Simulation::ras_convert_interval_to_hap_matrix (src/Simulation.cpp:1186-1230):
 * rows [row_begin, row_begin+n_rows) of the 2*n_people x L matrix, mutations applied. */
int gev_download_haps(gev_ctx*, int pop, int chr, size_t row_begin, size_t n_rows,
                      uint64_t* bits, size_t row_stride_words);
int gev_materialize_pops(gev_ctx*, int pop, int chr, size_t row_begin, size_t n_rows, size_t snp_begin, size_t n_snps,
                         const uint64_t* const* founder_bits, const size_t* founder_stride_words, const size_t* n_founder_rows,
                         uint64_t* bits, size_t row_stride_words);
/* PLINK .bed body (as gev_format_bed) of SNPs [snp_begin, +n_snps) assembled from the interval state + founder tiles: the output
 * path of plane-less contexts (BASELINE config 5: "PLINK bit-packed output"); n_snps * ceil(n_people/4) bytes. */
int gev_materialize_bed(gev_ctx*, int pop, int chr, size_t snp_begin, size_t n_snps,
                        const uint64_t* const* founder_bits, const size_t* founder_stride_words, const size_t* n_founder_rows,
                        uint8_t* out, size_t out_bytes);
/* ---- K9 output packing [SURVEY 8(f) row 3]: SNP-major forms of the same matrix, built on the device
 * (64x64 bit-tile transposes + mutation overlay).
 *  gev_download_snp_major : row = SNP (snp_begin + j), bit h = haplotype row h; ceil(2*n_people/64) words per row
 *  gev_format_hap_text    : the exact bytes format_hap::write_hap (src/format_hap.cpp:6-30) writes for these SNP
 *                           lines: per line "b b ... b \n", i.e. n_snps * (4*n_people + 1) bytes
 *  gev_format_bed         : PLINK .bed body (SNP-major, 2 bits / individual, A1 = allele 1: 00 = 1/1, 10 = het,
 *                           11 = 0/0, pad 00; without the 3 magic bytes): n_snps * ceil(n_people/4) bytes.  The
 *                           reference has no .bed writer (BASELINE config 5 asks for one): checked by definition.
 *  gev_format_vcf_gt      : the sample columns format_vcf::write_vcf_file (src/format_vcf.cpp:55-59) appends to each data
 *                           line: per individual "\ta|b", then '\n' = n_snps * (4*n_people + 1) bytes; the nine fixed
 *                           columns (CHROM..FORMAT, host strings of the input VCF) are the caller's.  Pinned on the
 *                           reference's own .vcf output (fixture `dense`; the reference's VCF-panel path needs the one
 *                           missing `return` of format_vcf.cpp:367-389 supplied, see oracle/ref_vcf_return.cpp). */
int gev_download_snp_major(gev_ctx*, int pop, int chr, size_t snp_begin, size_t n_snps, uint64_t* bits, size_t row_stride_words);
int gev_format_hap_text(gev_ctx*, int pop, int chr, size_t snp_begin, size_t n_snps, char* out, size_t out_bytes);
int gev_format_bed(gev_ctx*, int pop, int chr, size_t snp_begin, size_t n_snps, uint8_t* out, size_t out_bytes);
int gev_format_vcf_gt(gev_ctx*, int pop, int chr, size_t snp_begin, size_t n_snps, char* out, size_t out_bytes);
/* ---- mating support [SURVEY 8(f) row 2] ------------------------------------------------------
 * == CommFunc::ras_rank (src/CommFunc.cpp:152-161), the O(n^2) zero-based rank assort_mate applies to its bivariate-normal
 * template (src/Simulation.cpp:2278-2279): rank[k] = #{x[j] < x[k]} + #{j < k : x[j] == x[k]}.  Host buffers. */
int gev_rank_f64(gev_ctx*, const double* x, size_t n, unsigned long long* rank_out);
/* ---- PLINK siblings of the dense assembly (src/Simulation.cpp:1308-1416, src/format_plink.cpp:5-141), individual-major:
 *  gev_download_plink_matrix : matrix_plink_ped of ras_convert_interval_to_format_plink (:1335-1362): row = individual
 *                              ind_begin + i, bit 2*snp + hap; ceil(L/32) words per row
 *  gev_format_ped_text       : the genotype columns format_plink::write_ped_map (:42-49) / write_ped01_map (:114-121) append
 *                              to each line: L x " a b" then '\n' = 4*L + 1 bytes per individual.  al0/al1 = one allele
 *                              letter per SNP (Legend.al0/al1, plink_map.al0/al1), or both NULL for "0"/"1" (--out_plink01).
 *                              The six id columns (FID IID PID MID sex phen, :1391-1402) are host pedigree strings: the caller
 *                              writes them in front of each line. */
int gev_download_plink_matrix(gev_ctx*, int pop, int chr, size_t ind_begin, size_t n_ind, uint64_t* bits, size_t row_stride_words);
int gev_format_ped_text(gev_ctx*, int pop, int chr, size_t ind_begin, size_t n_ind, const char* al0, const char* al1, char* out, size_t out_bytes);
/* ---- K8: genotype tiles from the interval state ------------------------------------------------
 * gev_set_dense_state(ctx, 0): keep no resident genotype planes (populations whose individuals x loci matrix exceeds HBM,
 * BASELINE config 5); the per-generation work is then sampling + interval/mutation lists + CV planes + A/D.  Call before
 * gev_init_gen0; founder SNP panels are not uploaded (gev_upload_founders is refused), CV founders are.
 * gev_materialize == ras_convert_interval_to_hap_matrix (src/Simulation.cpp:1186-1230) for haplotype rows
 * [row_begin, +n_rows) and SNPs [snp_begin, +n_snps): founder_bits = the same SNP range of every founder haplotype
 * (row = founder haplotype = part::hap_index, bit j = SNP snp_begin + j); output rows as gev_download_haps, bit j = SNP
 * snp_begin + j.  Works on dense contexts too (same result as the resident planes).  gev_materialize_pops takes one
 * founder tile per ROOT population (arrays of n_pop entries; after migration a part may descend from another population's
 * founders, :1204); gev_materialize is the single-population form. */
int gev_set_dense_state(gev_ctx*, int on);
int gev_materialize(gev_ctx*, int pop, int chr, size_t row_begin, size_t n_rows, size_t snp_begin, size_t n_snps,
                    const uint64_t* founder_bits, size_t founder_stride_words, size_t n_founder_rows,
                    uint64_t* bits, size_t row_stride_words);
/* CV genotype matrix of ras_find_cv (the --debug .cvval dump, :2665-2683), FILE column order. */
int gev_download_cv(gev_ctx*, int pop, int phen, int chr, uint64_t* bits, size_t row_stride_words);
/* ancestry interval lists (the .int output, :1596-1633): hap_offsets has 2*n_people+1 entries;
 * parts of haplotype row r are out[hap_offsets[r] .. hap_offsets[r+1]).  Call with out==NULL to
 * get the total count in *n_parts. */
int gev_download_intervals(gev_ctx*, int pop, int chr, gev_part* out, uint64_t* hap_offsets,
                           size_t* n_parts);
/* mutation positions carried by each haplotype row (union of part::mutation_pos over its
 * parts), ascending per row, duplicates kept.  Same calling convention as above. */
int gev_download_mutations(gev_ctx*, int pop, int chr, uint64_t* out, uint64_t* hap_offsets,
                           size_t* n_mut);

/* ---- introspection ------------------------------------------------------------------------ */
int gev_pop_size(gev_ctx*, int pop, size_t* n_people);
/* The resident genotype rows of the current generation (founder alleles only: mutations are kept as a sparse overlay, see
 * DESIGN.md).  A row is kept as segments_per_row segments of unit_bytes bytes; segment g of haplotype slot s = 2*individual +
 * chromatid is the unit_bytes bytes at dptr + unit_of[s * segments_per_row + g] * unit_bytes (device pointers; the last segment of
 * a row is used up to the row's length).  A segment without a crossover boundary shares the parental unit, so several entries can
 * name the same unit.  Valid until the next call that changes the population. */
int gev_plane_ptr(gev_ctx*, int pop, int chr, void** dptr, size_t* unit_bytes, size_t* n_slots, const uint32_t** unit_of, uint32_t* segments_per_row);
/* Bytes the dense stitch wrote and bytes of the rows of the generations produced, summed over all gev_reproduce calls and active
 * chromosomes of the context, and the same in row segments (the last two may be NULL).  The difference: segments that contain no
 * crossover boundary -- Simulation::recombine copies the parental parts unchanged there (src/Simulation.cpp:2939-2946, :2910) --
 * name the parent's unit instead of being copied. */
int gev_stitch_totals(gev_ctx*, unsigned long long* bytes_written, unsigned long long* bytes_total, unsigned long long* segments_written, unsigned long long* segments_total);
/* Locus-split population: mark the chromosomes whose genotype / CV / list state THIS context holds (default: all).
 * Inactive chromosomes still need gev_set_rmap / gev_set_mutmap (the rand() seed chain of Simulation::reproduce,
 * src/Simulation.cpp:2447-2501, runs through every (offspring, chromosome) task), nothing else; their A/D entries come back
 * as exact zeros, so the contexts sharing one population add their per-chromosome arrays (all-reduce) and sum over
 * chromosomes in order (geneevolve_amd/distributed.py:compute_ad_locus_split).  Call before gev_init_gen0.
 * Row movement works on such a context (gev_export_rows / gev_import_rows / gev_migrate carry the active chromosomes only: both
 * ends of an exchange must hold the same chromosomes); gev_scale_ad_compute_gef needs the all-reduced totals first (gev_set_ad). */
int gev_set_chr_active(gev_ctx*, int chr, int active);
/* reserve device capacity for populations of up to max_people (avoids reallocation) */
int gev_reserve(gev_ctx*, int pop, size_t max_people);
/* HIP stream the context launches on (hipStream_t as void*), for event timing by the caller */
int gev_stream(gev_ctx*, void** stream);
/* gev_reproduce returns when the small per-generation work is done; the dense stitch of the
 * genotype planes keeps running on a second HIP stream and every later call is ordered after it
 * where it needs the planes.  gev_sync waits for all device work of the context. */
int gev_sync(gev_ctx*);
/* on == 1 (default): the stitch overlaps later work as described above; on == 0: gev_reproduce waits for it (kernel
 * timings without interference between the two streams); on == 2: only the ALU-bound sampling of the next generation
 * shares the GPU with the stitch, its memory-bound sparse/A-D kernels wait for it; on < 0: two serialised generations
 * are timed first and overlap is switched on unless the stitch is negligible (< 1/4 of the small kernels).
 * Environment: GEV_OVERLAP=<on> sets the initial mode. */
int gev_set_overlap(gev_ctx*, int on);
/* kernel timing measured with HIP events on the library's own streams: ms[0] = sampling
 * (crossover + mutation + seed chain), ms[1] = dense stitch (genotype planes), ms[2] = sparse state
 * (mutation lists, intervals, CV planes, grouping), ms[3] = sum.  _last = most recent generation
 * (implies gev_sync); _totals = cumulative sums over all generations so far (implies gev_sync). */
int gev_last_reproduce_ms(gev_ctx*, float ms[4]);
int gev_timing_totals(gev_ctx*, double ms_sum[4], unsigned long long* n_generations);
/* generations that outgrew a buffer and were enqueued again (inside gev_reproduce_end / gev_generation_end), over the context's life */
int gev_redo_count(gev_ctx*, unsigned long long* n);
/* Simulation::ras_compute_AD (src/Simulation.cpp:2624-2749) for a population split along chromosomes over several contexts
 * (gev_set_chr_active) WITHOUT a host round trip: the per-chromosome arrays of the current generation stay on the device.
 * *add_chr / *dom_chr = device pointers to [n_people][nchr][nphen] doubles (exact zeros for the chromosomes this context does not hold),
 * *n_doubles = their length; valid until the next call that changes the population.  The caller all-reduces (SUM) both arrays in place
 * across the population's group (RCCL; every entry is x + 0 + ... + 0: bit for bit) and then calls gev_ad_finish_device. */
int gev_compute_ad_device(gev_ctx*, int pop, double** add_chr, double** dom_chr, size_t* n_doubles);
/* ... behind the in-place all-reduce: Human::additive / dominance = the sum over chromosomes in order (:2729-2746), computed on the device
 * from the (reduced) per-chromosome arrays.  The totals stay in the context for gev_scale_ad_compute_gef (as after gev_set_ad) and are
 * copied out; add_chr_out / dom_chr_out receive the reduced per-chromosome arrays.  Any of the four pointers may be NULL. */
int gev_ad_finish_device(gev_ctx*, int pop, double* additive, double* dominance, double* add_chr_out, double* dom_chr_out);
/* state of the shared list pieces of one (population, chromosome) (csrc/gev_lists.h): out[0] = times the pieces were (re)built from
 * whole lists (first use, after a migration / import / upload, arena compactions), out[1] / out[2] = interval / mutation arena
 * entries in use, out[3] / out[4] = their capacities, out[5] = position ranges per row, out[6] / out[7] = entries the last
 * generation appended, out[8] = arena compactions (unnamed pieces dropped, sharing kept), out[9] = 0.  Diagnostic: nothing observable
 * depends on it (the oracle build reports zeros). */
int gev_list_stats(gev_ctx*, int pop, int chr, unsigned long long out[10]);
/* enable/disable keeping the ancestry interval state on the device (default on) */
int gev_set_track_intervals(gev_ctx*, int on);
/* dense-stitch kernel: 0 = k_stitch_segments (default: one workgroup per entry of the list of segments to write), 1 = gamete-major
 * k_stitch_rows (every output row finds its own segments and sources).  Same results; kept for parity cross-checks. */
int gev_set_stitch_mode(gev_ctx*, int mode);

/* ---- diagnostics: RNG building blocks exposed for the parity tests (no simulation state) ----
 * gev_dbg_tables / gev_dbg_threshold / gev_dbg_canonical run on the host (table and threshold
 * construction); gev_dbg_rand / gev_dbg_sim_loc_rec run the device generators:
 * glibc srand(seed);rand()xN and one Simulation::ras_sim_loc_rec call (src/Simulation.cpp:2973). */
/* gev_dbg_verify_planes: full-coverage self-check of the dense state of a population whose founder panels came from
 * gev_synth_founders: every 32-bit word of the resident genotype plane (written by the dense stitch from breakpoints) is
 * compared on the device with ras_convert_interval_to_hap_matrix (src/Simulation.cpp:1186-1230) applied to the ancestry
 * intervals (written by the interval kernel) -- two independent paths.  founder_seeds[r] = the seed the panel of ROOT
 * population r was generated with (n_pop entries: after migration a part may descend from another population's founders, on
 * another GPU).  *n_bad_words = mismatching words, *n_bad_parts = parts whose founder is unknown / out of range (both must be 0). */
int    gev_dbg_verify_planes(gev_ctx*, int pop, int chr, const uint64_t* founder_seeds, unsigned long long* n_bad_words, unsigned long long* n_bad_parts);
/* gev_dbg_prefilter_sweep: exhaustive check of the FP64 candidate prefilter of the batched sampling kernels over the engine states
 * [x_begin, x_end) (the whole state space of minstd_rand0 is 1 .. 2^31-2) and all 32 per-lane multipliers: out[0] = states x multipliers
 * for which the prefilter's 20-bit fraction deviates from the exact one by more than its proven bound (must be 0: the prefilter
 * then flags a superset of the exact candidates for every threshold), out[1] / out[2] = largest deviations seen. */
int    gev_dbg_prefilter_sweep(gev_ctx*, uint32_t x_begin, uint32_t x_end, unsigned long long out[3]);
int    gev_dbg_tables(void* out, size_t bytes);
int    gev_dbg_threshold(double p, uint32_t out[4] /* a_lo, a_hi, b0, b1 */);
double gev_dbg_canonical(uint32_t a, uint32_t b);
int    gev_dbg_rand(gev_ctx*, uint32_t seed, uint32_t n, int* out);
int    gev_dbg_sim_loc_rec(gev_ctx*, int pop, int chr, uint32_t seed, uint64_t* locs, uint32_t cap,
                           uint32_t* n, int next2[2]);

#ifdef __cplusplus
}
#endif
#endif /* GENEEVOLVE_AMD_H */
