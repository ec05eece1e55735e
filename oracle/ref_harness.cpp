// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Function-level driver for the UNMODIFIED reference (MMesbahU/GeneEvolve), built by
// oracle/Makefile.ref against the reference objects where they
// lie under /root/reference.  It runs the reference's own code and dumps, in lossless
// text (%a doubles), everything the C-ABI of include/geneevolve_amd.h must reproduce.
// tests/golden/make_golden.py turns these dumps into the committed fixtures.
//
// Two modes:
//   ref_harness KAT <out.txt>
//       known-answer vectors of the third-party RNG arithmetic the hot path uses
//       (glibc srand/rand, libstdc++ minstd_rand0 / uniform_real / uniform_int /
//       normal), plus direct calls of Simulation::ras_sim_loc_rec and
//       Simulation::recombine (reference src/Simulation.cpp:2973, :2903).
//   GEV_DUMP=<prefix> [GEV_DUMP_GENS=1,2,..] [GEV_DUMP_DENSE=1 [GEV_DENSE_GENS=0,1,..]] ref_harness <GeneEvolve args...>
//       drives ras_init_parameters -> ras_init_generation0 -> one generation at a time
//       using the reference's own member functions in the order of
//       Simulation::sim_next_generation (src/Simulation.cpp:1890-2082), and dumps the
//       inputs (couples, ras_glob_seed() values) and outputs (sex, interval lists,
//       mutation lists, raw A/D, dense haplotype matrix) of the hot path.
//       <prefix>.timing.txt receives the wall time of every reproduce / ras_compute_AD call (the
//       CPU baseline of bench.py when this binary is present).
//       The driver order is validated by comparing the .info files it writes with the
//       ones the stock CLI build (oracle/_ref/GeneEvolve_ref) writes for the same seed.
// Standard headers first (they must not see the macro below), then open the reference's
// private members for THIS translation unit only.  Access specifiers do not change class
// layout or symbol names, so this TU links against the reference objects compiled
// without any such define.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <random>
#include <set>
#include <sstream>
#include <string>
#include <vector>
#define private public
#include "Simulation.h"
#include "CommFunc.h"
#undef private

static FILE* g_out = nullptr;
static FILE* g_timing = nullptr;   // <GEV_DUMP>.timing.txt: gen pop n_people reproduce_seconds compute_AD_seconds

static void dump_humans(Simulation& sim, int ipop, const char* tag)
{
    Population& P = sim.population[ipop];
    unsigned long n = P.h.size();
    int nchr = n ? (int)P.h[0].chr.size() : 0;
    fprintf(g_out, "%s pop %d n %lu nchr %d\n", tag, ipop, n, nchr);
    for (unsigned long ih = 0; ih < n; ih++) {
        Human& h = P.h[ih];
        fprintf(g_out, "H %lu %d %lu %lu %lu\n", ih, h.sex, h.ID, h.ID_Father, h.ID_Mother);
        for (int c = 0; c < nchr; c++)
            for (int hp = 0; hp < 2; hp++) {
                std::vector<part>& v = h.chr[c].Hap[hp];
                for (size_t j = 0; j < v.size(); j++) {
                    fprintf(g_out, "P %lu %d %d %lu %lu %lu %d %zu", ih, c, hp, v[j].st, v[j].en,
                            v[j].hap_index, v[j].root_population, v[j].mutation_pos.size());
                    for (size_t m = 0; m < v[j].mutation_pos.size(); m++)
                        fprintf(g_out, " %lu", v[j].mutation_pos[m]);
                    fprintf(g_out, "\n");
                }
            }
    }
}

static void dump_ad(Simulation& sim, int ipop)
{
    Population& P = sim.population[ipop];
    unsigned long n = P.h.size();
    int nchr = (int)P.h[0].chr.size();
    int nphen = (int)P._pheno_scheme.size();
    fprintf(g_out, "AD pop %d n %lu nchr %d nphen %d\n", ipop, n, nchr, nphen);
    for (unsigned long ih = 0; ih < n; ih++) {
        for (int p = 0; p < nphen; p++) {
            fprintf(g_out, "A %lu %d %a %a %a", ih, p, P.h[ih].additive[p], P.h[ih].dominance[p], P.h[ih].bv[p]);
            for (int c = 0; c < nchr; c++)
                fprintf(g_out, " %a %a", P.h[ih].chr[c].additive_chr[p], P.h[ih].chr[c].dominance_chr[p]);
            fprintf(g_out, "\n");
        }
    }
}

// dense genotype assembly through the reference's own materialiser
// (Simulation::ras_convert_interval_to_hap_matrix, src/Simulation.cpp:1186-1230)
static void dump_dense(Simulation& sim)
{
    int nchr = (int)sim.population[0].h[0].chr.size();
    for (int c = 0; c < nchr; c++) {
        std::vector<Legend> pops_legend(sim._n_pop);
        std::vector<Hap_SNP> pops_hap(sim._n_pop);
        sim.ras_read_hap_legend_sample_chr(pops_legend, pops_hap, c);
        for (int ipop = 0; ipop < sim._n_pop; ipop++) {
            Hap_SNP out;
            sim.ras_convert_interval_to_hap_matrix(ipop, pops_hap, pops_legend, c, out);
            size_t nrow = out.hap.size(), ncol = nrow ? out.hap[0].size() : 0;
            fprintf(g_out, "DENSE pop %d chr %d nrow %zu ncol %zu\n", ipop, c, nrow, ncol);
            for (size_t r = 0; r < nrow; r++) {
                fputc('D', g_out); fputc(' ', g_out);
                for (size_t k = 0; k < ncol; k++) fputc(out.hap[r][k] ? '1' : '0', g_out);
                fputc('\n', g_out);
            }
        }
    }
}

static bool want_gen(const std::set<int>& gens, int g) { return gens.empty() || gens.count(g); }

static int run_sim(int argc, char** argv)
{
    const char* prefix = getenv("GEV_DUMP");
    std::set<int> gens;
    if (const char* s = getenv("GEV_DUMP_GENS")) {
        std::string t(s); size_t p = 0;
        while (p < t.size()) { gens.insert(atoi(t.c_str() + p)); p = t.find(',', p); if (p == std::string::npos) break; p++; }
    }
    bool dense = getenv("GEV_DUMP_DENSE") != nullptr;
    g_timing = fopen((std::string(prefix) + ".timing.txt").c_str(), "w");
    std::set<int> dense_gens;
    if (const char* s = getenv("GEV_DENSE_GENS")) {
        std::string t(s); size_t p = 0;
        while (p < t.size()) { dense_gens.insert(atoi(t.c_str() + p)); p = t.find(',', p); if (p == std::string::npos) break; p++; }
    }

    Parameters par;
    std::vector<std::string> vec_arg(argc + 2);
    for (int i = 0; i < argc; i++) vec_arg[i] = argv[i];
    vec_arg[argc] = "nothing"; vec_arg[argc + 1] = "nothing";
    if (!par.read(vec_arg) || !par.check()) return 2;

    Simulation sim;
    sim.par = par;
    sim.glob_generator.seed(par._seed);                  // Simulation::run, src/Simulation.cpp:75-76
    if (!sim.ras_init_parameters()) return 3;            // :83
    std::default_random_engine snap0 = sim.glob_generator;   // the ras_glob_seed() stream gen 0 will consume
    if (!sim.ras_init_generation0()) return 4;           // :91

    int nphen = (int)sim.population[0]._pheno_scheme.size();
    {
        std::string f = std::string(prefix) + ".gen0.txt";
        g_out = fopen(f.c_str(), "w");
        fprintf(g_out, "GEN 0 npop %d\n", sim._n_pop);
        {   // first 64 ras_glob_seed() values from the state before ras_init_generation0: the seed of
            // ras_initial_human_gen0 (:3003) of each population is one of them (identified by the packer)
            std::uniform_int_distribution<unsigned> d(1, 1000000);
            fprintf(g_out, "GLOBSEQ");
            for (int i = 0; i < 64; i++) fprintf(g_out, " %u", d(snap0));
            fprintf(g_out, "\n");
        }
        for (int ipop = 0; ipop < sim._n_pop; ipop++) {
            dump_humans(sim, ipop, "POST");
            // additive/dominance were rescaled by ras_scale_AD_compute_GEF; per-chr values are raw
            dump_ad(sim, ipop);
        }
        if (dense && want_gen(dense_gens, 0)) dump_dense(sim);
        fclose(g_out);
    }

    for (int gen_num = 1; gen_num <= sim._tot_gen; gen_num++) {
        bool w = want_gen(gens, gen_num);
        std::string f = std::string(prefix) + ".gen" + std::to_string(gen_num) + ".txt";
        g_out = fopen(w ? f.c_str() : "/dev/null", "w");
        fprintf(g_out, "GEN %d npop %d\n", gen_num, sim._n_pop);
        // ---- Simulation::sim_next_generation, src/Simulation.cpp:1895-1966
        for (int ipop = 0; ipop < sim._n_pop; ipop++) {
            Population& P = sim.population[ipop];
            // inputs of the mating step: sexes and selection_value_func of the parent generation,
            // and, for random_mate (exactly one ras_glob_seed() draw, :2092), its seed
            {
                std::default_random_engine snapm = sim.glob_generator;
                std::uniform_int_distribution<unsigned> dm(1, 1000000);
                fprintf(g_out, "MATE pop %d rm %d seed %u popsize %lu n %zu\n", ipop, (int)P._RM, dm(snapm), (unsigned long)P._pop_size[gen_num - 1], P.h.size());
                for (size_t i = 0; i < P.h.size(); i++) fprintf(g_out, "MS %d %a\n", P.h[i].sex, P.h[i].selection_value_func);
                if (!P._RM) {   // assort_mate (:2167-2360): its parameters, the four ras_glob_seed() values it may draw, mating values, pedigree
                    unsigned s1 = dm(snapm), s2 = dm(snapm), s3 = dm(snapm);          // (the MATE line consumed the first one)
                    fprintf(g_out, "MATEP pop %d matcor %a mm %a avoid %d dist %s seeds %u %u %u\n", ipop, P._mat_cor[gen_num - 1], P._MM_percent,
                            (int)P._avoid_inbreeding, P._offspring_dist[gen_num - 1].c_str(), s1, s2, s3);
                    for (size_t i = 0; i < P.h.size(); i++)
                        fprintf(g_out, "MV %a %ld %ld %ld %ld %ld\n", P.h[i].mating_value, (long)P.h[i].ID_Father, (long)P.h[i].ID_Fathers_Father,
                                (long)P.h[i].ID_Fathers_Mother, (long)P.h[i].ID_Mothers_Father, (long)P.h[i].ID_Mothers_Mother);
                }
            }
            bool ok = P._RM ? sim.random_mate(ipop, gen_num - 1) : sim.assort_mate(ipop, gen_num - 1);   // :1907-1918
            if (!ok) return 5;
            // inputs of the hot path
            unsigned long n_couples = P._couples_info.size(), n_people = 0;
            fprintf(g_out, "COUPLES pop %d n %lu\n", ipop, n_couples);
            for (unsigned long i = 0; i < n_couples; i++) {
                Couples_Info& c = P._couples_info[i];
                fprintf(g_out, "C %lu %lu %d %d\n", c.pos_male, c.pos_female, (int)c.inbreed, c.num_offspring);
                if (!c.inbreed) n_people += c.num_offspring;
            }
            int nchr = (int)P.h[0].chr.size();
            bool has_mut = P._mutation_map.size() > 0;
            std::default_random_engine snap = sim.glob_generator;       // state before reproduce
            const auto t_r0 = std::chrono::steady_clock::now();
            P.h = sim.reproduce(ipop, gen_num);                          // :1929
            const auto t_r1 = std::chrono::steady_clock::now();
            // replay the ras_glob_seed() draws reproduce made (1 + n_people*nchr with a mutation map)
            {
                std::uniform_int_distribution<unsigned> d(1, 1000000);   // ras_glob_seed, :17-21
                unsigned s0 = d(snap);
                fprintf(g_out, "SEED_REPRODUCE pop %d %u\n", ipop, s0);
                unsigned long nm = has_mut ? n_people * (unsigned long)nchr : 0;
                fprintf(g_out, "MUTSEEDS pop %d n %lu\n", ipop, nm);
                for (unsigned long i = 0; i < nm; i++) fprintf(g_out, "S %u\n", d(snap));
                if (!(snap == sim.glob_generator)) { fprintf(stderr, "harness: glob_generator replay mismatch\n"); return 6; }
            }
            const auto t_a0 = std::chrono::steady_clock::now();
            if (!sim.ras_compute_AD(ipop, gen_num)) return 7;            // :1935
            const auto t_a1 = std::chrono::steady_clock::now();
            if (g_timing)                                                // wall time of the two hot-path calls of the REAL reference
                fprintf(g_timing, "%d %d %lu %.6f %.6f\n", gen_num, ipop, (unsigned long)P.h.size(),
                        std::chrono::duration<double>(t_r1 - t_r0).count(), std::chrono::duration<double>(t_a1 - t_a0).count());
            dump_humans(sim, ipop, "OFFSPRING");
            dump_ad(sim, ipop);                                          // raw A/D (before scaling)
            for (int iphen = 0; iphen < nphen; iphen++) {                // :1943-1946
                // inputs of ras_scale_AD_compute_GEF (:3075): its ras_glob_seed() value, gen-0 variances, the parents' values it looks up
                std::default_random_engine snap2 = sim.glob_generator;
                std::uniform_int_distribution<unsigned> d2(1, 1000000);
                fprintf(g_out, "GEF pop %d phen %d seed %u s2a %a s2d %a va %a vd %a ve %a vf %a beta %a vt %d\n", ipop, iphen, d2(snap2),
                        P._var_a_gen0[iphen], P._var_d_gen0[iphen], P._pheno_scheme[iphen]._va, P._pheno_scheme[iphen]._vd,
                        P._pheno_scheme[iphen]._ve, P._pheno_scheme[iphen]._vf, P._pheno_scheme[iphen]._beta, sim._vt_type);
                for (unsigned long ih = 0; ih < P.h.size(); ih++) {
                    const unsigned long f = P.h[ih].ID_Father, m = P.h[ih].ID_Mother;
                    const double pf = sim._vt_type == 1 ? sim._Pop_info_prev_gen[ipop].phen[iphen][f] : (sim._vt_type == 2 ? sim._Pop_info_prev_gen[ipop].parental_effect[iphen][f] : 0.0);
                    const double pm = sim._vt_type == 1 ? sim._Pop_info_prev_gen[ipop].phen[iphen][m] : (sim._vt_type == 2 ? sim._Pop_info_prev_gen[ipop].parental_effect[iphen][m] : 0.0);
                    fprintf(g_out, "GI %lu %a %a %a\n", ih, P.h[ih].common_sibling[iphen], pf, pm);
                }
                if (!sim.ras_scale_AD_compute_GEF(gen_num, ipop, iphen, P._var_a_gen0[iphen], P._var_d_gen0[iphen])) return 8;
                if (!(snap2 == sim.glob_generator)) { fprintf(stderr, "harness: GEF glob replay mismatch\n"); return 6; }
                for (unsigned long ih = 0; ih < P.h.size(); ih++)
                    fprintf(g_out, "GO %lu %a %a %a %a %a %a\n", ih, P.h[ih].additive[iphen], P.h[ih].dominance[iphen], P.h[ih].bv[iphen],
                            P.h[ih].e_noise[iphen], P.h[ih].parental_effect[iphen], P.h[ih].phen[iphen]);
            }
        }
        for (int iphen = 0; iphen < nphen; iphen++)                      // :1973-1977
            sim.sim_environmental_effects_specific_to_each_population(iphen);
        for (int ipop = 0; ipop < sim._n_pop; ipop++)                    // :1983-1989
            sim.ras_compute_mating_value_selection_value(gen_num, ipop);
        if (sim._n_pop > 1) {                                            // :1994-1999
            for (int ipop = 0; ipop < sim._n_pop; ipop++) {
                fprintf(g_out, "PREMIG pop %d n %zu\n", ipop, sim.population[ipop].h.size());
                // fingerprint = phenotype value (contains the N(0,1) noise term): identifies an individual across the move
                for (size_t i = 0; i < sim.population[ipop].h.size(); i++) fprintf(g_out, "I %lu %a\n", sim.population[ipop].h[i].ID, sim.population[ipop].h[i].phen[0]);
            }
            if (!sim.ras_do_migration(gen_num - 1)) return 9;
            for (int ipop = 0; ipop < sim._n_pop; ipop++) {
                dump_humans(sim, ipop, "POSTMIG");
                fprintf(g_out, "POSTFP pop %d n %zu\n", ipop, sim.population[ipop].h.size());
                for (size_t i = 0; i < sim.population[ipop].h.size(); i++) fprintf(g_out, "J %lu %a\n", sim.population[ipop].h[i].ID, sim.population[ipop].h[i].phen[0]);
            }
        }
        for (int ipop = 0; ipop < sim._n_pop; ipop++)                    // :2003-2006
            sim.ras_save_human_info_to_Pop_info_prev_gen(ipop);
        for (int ipop = 0; ipop < sim._n_pop; ipop++)                    // :2018 (the .info file, used to validate this driver)
            sim.population[ipop].ras_save_human_info(gen_num);
        if (w && dense && want_gen(dense_gens, gen_num)) dump_dense(sim);
        fclose(g_out);
        if (g_timing) fflush(g_timing);
    }
    if (g_timing) fclose(g_timing);
    return 0;
}

// ---------------------------------------------------------------------------------------
static int run_kat(const char* path)
{
    g_out = fopen(path, "w");
    // glibc srand/rand (TYPE_3) -- call sites src/Simulation.cpp:2400,2447,2449,2453,2455,2472,2501,2522,2977,2990
    const unsigned seeds[] = {0u, 1u, 2u, 12345u, 1000000u, 2147483646u, 2147483647u, 2147483648u, 4294967295u, 123456789u, 987654321u, 3000000000u};
    for (unsigned s : seeds) {
        srand(s);
        fprintf(g_out, "RAND %u", s);
        for (int i = 0; i < 40; i++) fprintf(g_out, " %d", rand());
        fprintf(g_out, "\n");
    }
    // libstdc++ minstd_rand0 raw, uniform_real(0,1) (generate_canonical<double,53>)
    for (unsigned s : seeds) {
        std::default_random_engine g(s);
        fprintf(g_out, "MINSTD %u", s);
        for (int i = 0; i < 8; i++) fprintf(g_out, " %lu", (unsigned long)g());
        fprintf(g_out, "\n");
        std::default_random_engine g2(s);
        std::uniform_real_distribution<double> d(0.0, 1.0);
        fprintf(g_out, "U01 %u", s);
        for (int i = 0; i < 8; i++) fprintf(g_out, " %a", d(g2));
        fprintf(g_out, "\n");
    }
    // uniform_int_distribution<unsigned long>(lo,hi) as in ras_add_mutation (:2519-2520), fresh object per draw
    {
        const unsigned long rng[][2] = {{738555, 788555}, {0, 50000}, {1000, 1100}, {5, 5}, {7, 8}, {100, 2147483744ul}, {0, 2147483644ul}, {10, 2147483654ul}};
        for (auto& r : rng) for (unsigned s : {1u, 12345u, 999999u}) {
            std::default_random_engine g(s + 1);
            fprintf(g_out, "UINT %u %lu %lu", s, r[0], r[1]);
            for (int i = 0; i < 12; i++) { std::uniform_int_distribution<unsigned long> d(r[0], r[1]); fprintf(g_out, " %lu", d(g)); }
            fprintf(g_out, "\n");
        }
        // ras_glob_seed: uniform_int_distribution<unsigned>(1,1000000) (:17-21)
        for (unsigned s : {1u, 12345u, 999999u}) {
            std::default_random_engine g(s);
            std::uniform_int_distribution<unsigned> d(1, 1000000);
            fprintf(g_out, "GLOB %u", s);
            for (int i = 0; i < 12; i++) fprintf(g_out, " %u", d(g));
            fprintf(g_out, "\n");
        }
    }
    // normal_distribution (reproduce :2417-2426; host side)
    for (unsigned s : {2u, 12346u}) for (double sd : {1.0, 0.5}) {
        std::default_random_engine g(s);
        std::normal_distribution<double> d(0.0, sd);
        fprintf(g_out, "NORMAL %u %a", s, sd);
        for (int i = 0; i < 9; i++) fprintf(g_out, " %a", d(g));
        fprintf(g_out, "\n");
    }
    // CommFunc::ras_rank (src/CommFunc.cpp:152-161) on vectors with ties, signed zeros and repeated blocks
    for (unsigned s : {5u, 6u, 7u}) {
        std::mt19937 g(s);
        const int n = s == 5u ? 37 : (s == 6u ? 200 : 1);
        std::vector<double> x(n);
        for (int i = 0; i < n; i++) x[i] = (double)((int)(g() % 23) - 11) * 0.25;      // many exact ties
        if (n > 10) { x[3] = 0.0; x[7] = -0.0; x[9] = 1e300; x[10] = -1e300; }
        std::vector<unsigned long int> r = CommFunc::ras_rank(x);
        fprintf(g_out, "RANK %d", n);
        for (int i = 0; i < n; i++) fprintf(g_out, " %a", x[i]);
        for (int i = 0; i < n; i++) fprintf(g_out, " %lu", r[i]);
        fprintf(g_out, "\n");
    }
    // Simulation::ras_sim_loc_rec (:2973-2995) on a synthetic uniform map, then the next two rand() outputs
    {
        Simulation sim;
        rMap rm; std::vector<double> prob;
        const unsigned long R = 401;
        for (unsigned long j = 0; j < R; j++) { rm.bp.push_back(1000 + 5000 * j); rm.cM.push_back(0.0); }
        rm.bp_dist_in_rmap = rm.bp[1] - rm.bp[0];
        prob.resize(R);
        for (unsigned long j = 0; j < R; j++) prob[j] = (j == 0) ? 0.0 : (j % 7 == 0 ? 0.02 : 0.004);
        prob[R - 1] = 0.5;   // exercises a hit on the last row (location >= bp[R-1])
        fprintf(g_out, "LOCMAP %lu %lu %lu", R, rm.bp[0], rm.bp_dist_in_rmap);
        for (unsigned long j = 0; j < R; j++) fprintf(g_out, " %a", prob[j]);
        fprintf(g_out, "\n");
        std::default_random_engine sg(777);
        for (int t = 0; t < 400; t++) {
            unsigned seed = (t < 6) ? seeds[t + 0] : (unsigned)(sg() % 2147483648u);
            std::vector<unsigned long> locs = sim.ras_sim_loc_rec(prob, rm, seed);
            int r1 = rand(), r2 = rand();
            fprintf(g_out, "LOC %u %d %d %zu", seed, r1, r2, locs.size());
            for (unsigned long x : locs) fprintf(g_out, " %lu", x);
            fprintf(g_out, "\n");
        }
        // Simulation::recombine (:2903-2958) + modify_part_for_mutation_pos (:2961-2970)
        std::default_random_engine rg(4242);
        for (int t = 0; t < 300; t++) {
            chromosome ch; ch.Hap.resize(2);
            unsigned long lo = rm.bp[0], hi = rm.bp[R - 1];
            for (int hp = 0; hp < 2; hp++) {
                int np = 1 + (int)(rg() % 6);
                std::set<unsigned long> cuts;
                while ((int)cuts.size() < np - 1) cuts.insert(lo + 1 + rg() % (hi - lo - 1));
                std::vector<unsigned long> b(cuts.begin(), cuts.end());
                b.insert(b.begin(), lo); b.push_back(hi);
                for (int j = 0; j + 1 < (int)b.size(); j++) {
                    part p(b[j], b[j + 1], rg() % 50, "x", (int)(rg() % 2));
                    int nm = (int)(rg() % 4);
                    for (int m = 0; m < nm; m++) p.mutation_pos.push_back(b[j] + rg() % (b[j + 1] - b[j]));
                    if (nm && (t % 5 == 0)) p.mutation_pos.push_back(p.mutation_pos[0]);   // duplicate
                    ch.Hap[hp].push_back(p);
                }
            }
            // breakpoints: sorted, sometimes exactly on a part boundary, sometimes >= hi
            int k = (int)(rg() % 5);
            std::set<unsigned long> bs;
            while ((int)bs.size() < k) {
                unsigned long c = lo + 1 + rg() % (hi - lo + 3000);
                if (t % 7 == 0 && bs.empty()) c = ch.Hap[rg() % 2][0].en;   // exactly on a part boundary (or == hi)
                bs.insert(c);
            }
            std::vector<unsigned long> locs; locs.push_back(lo);
            for (unsigned long c : bs) locs.push_back(c);
            locs.push_back(hi);
            int start = (int)(rg() % 2);
            fprintf(g_out, "RECIN %d %d %zu", t, start, locs.size());
            for (unsigned long x : locs) fprintf(g_out, " %lu", x);
            fprintf(g_out, "\n");
            for (int hp = 0; hp < 2; hp++) for (auto& p : ch.Hap[hp]) {
                fprintf(g_out, "RECP %d %d %lu %lu %lu %d %zu", t, hp, p.st, p.en, p.hap_index, p.root_population, p.mutation_pos.size());
                for (unsigned long m : p.mutation_pos) fprintf(g_out, " %lu", m);
                fprintf(g_out, "\n");
            }
            std::vector<part> out = sim.recombine(ch, start, locs);
            for (auto& p : out) {
                fprintf(g_out, "RECO %d %lu %lu %lu %d %zu", t, p.st, p.en, p.hap_index, p.root_population, p.mutation_pos.size());
                for (unsigned long m : p.mutation_pos) fprintf(g_out, " %lu", m);
                fprintf(g_out, "\n");
            }
        }
    }
    fclose(g_out);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc >= 3 && strcmp(argv[1], "KAT") == 0) return run_kat(argv[2]);
    if (!getenv("GEV_DUMP")) { fprintf(stderr, "set GEV_DUMP=<prefix>\n"); return 1; }
    return run_sim(argc, argv);
}
