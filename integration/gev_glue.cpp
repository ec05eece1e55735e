// integration/gev_glue.cpp -- the reference-side binding of INTEGRATION.md, for real: GeneEvolve's own host (Main.cpp,
// parameters, every reader/writer, mating, phenotype scaling, migration decisions, summaries) with the reproduction hot
// path running in libgeneevolve_amd.so.  integration/build_gpu_cli.py makes an edited build copy of the reference's
// Simulation.{h,cpp} (its call sites rerouted to the functions below, see the list in build_gpu_cli.py; nothing of the reference is stored in this repo)
// and links it with this file, the reference's other objects and the library into oracle/_ref/GeneEvolve_gpu.
// tests/test_gpu_parity.py runs that program on the reference's own kind of input files and compares its output files
// with the unmodified reference's (hashes in tests/golden).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <random>
#include <string>
#include <chrono>
#include <thread>
#include <vector>
#include "Simulation.h"          // the EDITED copy: class Simulation has `friend struct GevGlue;`
#include "CommFunc.h"
#include "geneevolve_amd.h"

struct GevGlue {
    static gev_ctx*& ctx() { static gev_ctx* g = nullptr; return g; }
    static bool fail(const char* where)
    {
        std::cout << "Error: " << where << ": " << gev_last_error() << std::endl;    // the reference's convention: message + false
        return false;
    }
    static std::vector<uint64_t> pack(const std::vector<std::vector<bool> >& rows, size_t ncol, size_t& words)
    {
        words = (ncol + 63) / 64;
        std::vector<uint64_t> out(rows.size() * std::max<size_t>(words, 1), 0);
        for (size_t r = 0; r < rows.size(); r++)
            for (size_t j = 0; j < ncol; j++)
                if (rows[r][j]) out[r * words + (j >> 6)] |= 1ull << (j & 63);
        return out;
    }

    // end of Simulation::ras_init_parameters (src/Simulation.cpp:164-525): hand the static inputs over
    static bool init_static(Simulation& S)
    {
        const int n_pop = S._n_pop, nchr = S.population[0]._nchr, nphen = (int)S.population[0]._pheno_scheme.size();
        if (gev_create(&ctx(), -1, n_pop, nchr, nphen)) return fail("gev_create");
        for (int ipop = 0; ipop < n_pop; ipop++) {
            Population& P = S.population[ipop];
            for (int c = 0; c < nchr; c++) {
                if (gev_set_rmap(ctx(), ipop, c, P._rmap[c].bp.data(), P._recom_prob[c].data(), P._rmap[c].bp.size(), P._rmap[c].bp_dist_in_rmap)) return fail("gev_set_rmap");
                if (P._mutation_map.size() > 0 &&
                    gev_set_mutmap(ctx(), ipop, c, P._mutation_map[c].bp.data(), P._mutation_map[c].mutation_rate.data(), P._mutation_map[c].bp.size())) return fail("gev_set_mutmap");
                // the founder panel the reference itself only reads at output time (ras_read_hap_legend_sample_chr, :1105;
                // VCF panel: ras_read_vcf_pops_chr, :1764)
                size_t w; std::vector<uint64_t> bits;
                if (S._ref_is_vcf) {
                    vcf_structure vs;
                    if (!format_vcf::read_vcf_file(P._ref_vcf_address[c], vs)) { std::cout << "Error in reading vcf file." << std::endl; return false; }
                    std::vector<unsigned long int> pos(vs.POS.begin(), vs.POS.end());
                    if (gev_set_snps(ctx(), ipop, c, pos.data(), pos.size())) return fail("gev_set_snps");
                    bits = pack(vs.data, pos.size(), w);
                    if (gev_upload_founders(ctx(), ipop, c, bits.data(), w, vs.data.size(), pos.size())) return fail("gev_upload_founders");
                } else {
                    Legend legend; Hap_SNP hs;
                    const long nind = CommFunc::ras_FileLineNumber(P._hap_legend_sample_name[c][2]);
                    const long nsnp = format_hap::read_legend(legend, P._hap_legend_sample_name[c][1]);
                    if (!format_hap::read_hap(hs, P._hap_legend_sample_name[c][0], nind, nsnp, false)) return false;
                    if (gev_set_snps(ctx(), ipop, c, legend.pos.data(), legend.pos.size())) return fail("gev_set_snps");
                    bits = pack(hs.hap, legend.pos.size(), w);
                    if (gev_upload_founders(ctx(), ipop, c, bits.data(), w, hs.hap.size(), legend.pos.size())) return fail("gev_upload_founders");
                }
                for (int p = 0; p < nphen; p++) {
                    CV_INFO& I = P._pheno_scheme[p]._cv_info[c];
                    if (gev_set_cvs(ctx(), ipop, p, c, I.bp.data(), I.genetic_value_a.data(), I.genetic_value_d.data(), I.bp.size(), P._pheno_scheme[p]._vd)) return fail("gev_set_cvs");
                    bits = pack(P._pheno_scheme[p]._cvs[c].val, I.bp.size(), w);
                    if (gev_upload_cv_founders(ctx(), ipop, p, c, bits.data(), w, P._pheno_scheme[p]._cvs[c].val.size(), I.bp.size())) return fail("gev_upload_cv_founders");
                }
            }
        }
        return true;
    }

    // end of Simulation::ras_initial_human_gen0 (:3000-3072): the reference has just built its gen-0 humans from `seed`
    static bool after_gen0(Simulation& S, int ipop, unsigned seed)
    {
        std::vector<Human>& h = S.population[ipop].h;
        std::vector<uint8_t> sex(h.size());
        if (gev_init_gen0(ctx(), ipop, h.size(), seed, sex.data())) return fail("gev_init_gen0");
        for (size_t i = 0; i < h.size(); i++)
            if (h[i].sex != (int)sex[i]) { std::cout << "Error: gev_init_gen0: sex of founder " << i << " differs from the host's." << std::endl; return false; }
        return true;
    }

    // Simulation::ras_allocate_memory_for_humans (:2366-2392): the same resizes (12 + 4 nchr heap allocations per individual), the
    // individuals shared out over a few host threads -- with the genotype work on the GPU this was the longest part of
    // Simulation::reproduce at 100k individuals.  Same stdout text.
    static void allocate_humans(std::vector<Human>& h, unsigned long n_people, int nchr, int nphen)
    {
        std::cout << "        Allocating memory: " << std::flush;
        h.resize(n_people);
        auto fill = [&h, nchr, nphen](size_t b, size_t e) {
            for (size_t it = b; it < e; it++) {
                h[it].chr.resize(nchr);
                for (int ichr = 0; ichr < nchr; ichr++) {
                    h[it].chr[ichr].Hap.resize(2);
                    h[it].chr[ichr].bv_chr.resize(nphen);
                    h[it].chr[ichr].additive_chr.resize(nphen);
                    h[it].chr[ichr].dominance_chr.resize(nphen);
                }
                h[it].additive.resize(nphen); h[it].dominance.resize(nphen); h[it].common_sibling.resize(nphen);
                h[it].bv.resize(nphen); h[it].e_noise.resize(nphen); h[it].parental_effect.resize(nphen); h[it].phen.resize(nphen);
            }
        };
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t nt = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(hw ? hw : 1, 8), n_people / 4096 + 1));
        std::vector<std::thread> th;
        for (size_t t = 1; t < nt; t++) th.emplace_back(fill, n_people * t / nt, n_people * (t + 1) / nt);
        fill(0, n_people / nt);
        for (auto& x : th) x.join();
        std::cout << "done." << std::endl;
    }

    // Simulation::reproduce (:2394-2493): same ras_glob_seed() draws, same host bookkeeping, genotype work in the library
    static std::vector<Human> reproduce(Simulation& S, int ipop, int gen_num)
    {
        Population& P = S.population[ipop];
        const unsigned seed = S.ras_glob_seed();                                   // :2398
        std::srand(seed);                                                          // :2400 (process rand() state as the reference leaves it)
        const unsigned long n_couples = P._couples_info.size();
        unsigned long n_people = 0;
        for (unsigned long i = 0; i < n_couples; i++) if (!P._couples_info[i].inbreed) n_people += P._couples_info[i].num_offspring;
        const int nchr = (int)P.h[0].chr.size(), nphen = (int)P._pheno_scheme.size();
        std::vector<Human> h_ret;
        const bool trace = getenv("GEV_GLUE_TRACE") != NULL;
        const auto tr0 = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) { if (trace) fprintf(stderr, "[glue] reproduce: %s at %.2f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count()); };
        allocate_humans(h_ret, n_people, nchr, nphen);
        lap("records allocated");
        std::default_random_engine generator(seed + 1);                            // common effect, :2417-2429
        std::vector<std::vector<double> > val_common(nphen, std::vector<double>(n_couples, 0));
        for (int iphen = 0; iphen < nphen; iphen++)
            if (P._pheno_scheme[iphen]._vc > 0) {
                std::normal_distribution<double> distribution(0.0, sqrt(P._pheno_scheme[iphen]._vc));
                for (unsigned long it = 0; it < n_couples; it++) val_common[iphen][it] = distribution(generator);
            }
        std::vector<gev_couple> cpl(n_couples);
        for (unsigned long it = 0; it < n_couples; it++)
            cpl[it] = gev_couple{P._couples_info[it].pos_male, P._couples_info[it].pos_female, P._couples_info[it].inbreed ? 1 : 0, P._couples_info[it].num_offspring};
        std::vector<uint32_t> mut_seeds;
        if (P._mutation_map.size() > 0) {                                          // the draws ras_add_mutation would make (:2500), same order
            mut_seeds.resize(n_people * (size_t)nchr);
            for (size_t t = 0; t < mut_seeds.size(); t++) mut_seeds[t] = S.ras_glob_seed();
        }
        std::vector<uint8_t> sex(n_people);
        lap("couples and seeds ready");
        if (gev_reproduce(ctx(), ipop, cpl.data(), n_couples, seed, mut_seeds.empty() ? NULL : mut_seeds.data(), mut_seeds.size(), n_people, sex.data())) {
            fail("gev_reproduce"); return std::vector<Human>();
        }
        lap("gev_reproduce returned");
        unsigned long i_people = 0;
        for (unsigned long it = 0; it < n_couples; it++) {
            if (P._couples_info[it].inbreed) continue;
            const Human& h_pat = P.h[P._couples_info[it].pos_male]; const Human& h_mat = P.h[P._couples_info[it].pos_female];
            for (int ns = 0; ns < P._couples_info[it].num_offspring; ns++) {
                Human& o = h_ret[i_people];
                o.gen_num = gen_num; o.sex = sex[i_people]; o.ID = i_people;                       // :2471-2479
                o.ID_Father = h_pat.ID; o.ID_Fathers_Mother = h_pat.ID_Mother; o.ID_Fathers_Father = h_pat.ID_Father;
                o.ID_Mother = h_mat.ID; o.ID_Mothers_Mother = h_mat.ID_Mother; o.ID_Mothers_Father = h_mat.ID_Father;
                for (int iphen = 0; iphen < nphen; iphen++) o.common_sibling[iphen] = val_common[iphen][it];   // :2481-2484
                i_people++;
            }
        }
        lap("pedigree fields set");
        release_humans(P.h);              // the caller replaces P.h with the return value (:1929): nothing reads the parents any more
        lap("parents released");
        return h_ret;
    }
    // The parents' records are destroyed by the assignment `population[ipop].h = reproduce(...)` (src/Simulation.cpp:1929), one
    // individual after the other on the calling thread (16 frees each).  Their inner vectors are released here first, by a few
    // threads; the assignment then destroys empty shells.
    static void release_humans(std::vector<Human>& h)
    {
        const size_t n = h.size();
        auto drop = [&h](size_t b, size_t e) {
            for (size_t it = b; it < e; it++) {
                Human& x = h[it];
                std::vector<chromosome>().swap(x.chr);
                std::vector<double>().swap(x.additive); std::vector<double>().swap(x.dominance); std::vector<double>().swap(x.common_sibling);
                std::vector<double>().swap(x.bv); std::vector<double>().swap(x.e_noise); std::vector<double>().swap(x.parental_effect); std::vector<double>().swap(x.phen);
            }
        };
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t nt = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(hw ? hw : 1, 8), n / 4096 + 1));
        std::vector<std::thread> th;
        for (size_t t = 1; t < nt; t++) th.emplace_back(drop, n * t / nt, n * (t + 1) / nt);
        drop(0, n / nt);
        for (auto& x : th) x.join();
    }

    // Simulation::sim_next_generation, just before the mating of population ipop (:1907): under random mating (one child per
    // couple, :2146-2151) the offspring count is _pop_size[gen_num-1] and the seeds reproduce() is going to draw are the next
    // values of glob_generator whatever couples random_mate forms.  Draw them from a COPY of the engine (the real stream is
    // not advanced: random_mate and reproduce will draw the very same values) and let the GPU sample while the host mates.
    static bool presample(Simulation& S, int ipop, int gen_num)
    {
        Population& P = S.population[ipop];
        if (!P._RM || P._mutation_map.size() == 0) return true;        // offspring count unknown before mating / serial seed chain
        std::default_random_engine peek = S.glob_generator;
        auto draw = [&peek]() { std::uniform_int_distribution<unsigned> d(1, 1000000); return d(peek); };   // == ras_glob_seed (:17-21)
        draw();                                                        // the draw random_mate makes (:2092)
        const unsigned seed_rep = draw();                              // :2398
        const size_t n_people = P._pop_size[gen_num - 1], nchr = P.h[0].chr.size();
        std::vector<uint32_t> mut_seeds(n_people * nchr);
        for (size_t t = 0; t < mut_seeds.size(); t++) mut_seeds[t] = draw();      // :2500
        if (gev_presample(ctx(), ipop, seed_rep, mut_seeds.data(), mut_seeds.size(), n_people)) return fail("gev_presample");
        return true;
    }

    // Simulation::random_mate (:2090-2157) in the library: the same ras_glob_seed() draw, the same stdout lines, the couples
    // filled into _couples_info from what gev_random_mate returns.  GEV_MATE_HOST=1 keeps the reference's own function.
    static bool use_device_mate() { static const bool off = getenv("GEV_MATE_HOST") && atoi(getenv("GEV_MATE_HOST")) != 0; return !off; }
    static bool random_mate(Simulation& S, int ipop, int gen_ind)
    {
        Population& P = S.population[ipop];
        const unsigned seed = S.ras_glob_seed();                                        // :2092
        const unsigned long n_h = P.h.size(), pop_size = P._pop_size[gen_ind];
        if (S._debug) {
            std::cout << "Debug: Simulation::random_mate; seed=" << seed << std::endl;
            std::cout << "Debug: Simulation::random_mate; n_h=" << n_h << std::endl;
        }
        std::vector<double> svf(n_h);
        bool all_one = true;
        for (unsigned long i = 0; i < n_h; i++) { svf[i] = P.h[i].selection_value_func; all_one &= svf[i] == 1.0; }
        std::vector<gev_couple> cpl(pop_size);
        size_t nm = 0, nf = 0;
        const int rc = gev_random_mate(ctx(), ipop, seed, all_one ? NULL : svf.data(), pop_size, cpl.data(), &nm, &nf);
        if (rc == GEV_OK || rc == GEV_ENOMATE) {
            std::cout << "        num_males_mate    = " << nm << std::endl;              // :2122-2123
            std::cout << "        num_females_mate  = " << nf << std::endl;
        }
        if (rc == GEV_ENOMATE) { std::cout << gev_last_error() << std::endl; return false; }   // the reference's own line (:2127)
        if (rc) return fail("gev_random_mate");
        std::vector<Couples_Info> couples_info(pop_size);
        for (unsigned long i = 0; i < pop_size; i++) {
            couples_info[i].pos_male = cpl[i].pos_male; couples_info[i].pos_female = cpl[i].pos_female;
            couples_info[i].inbreed = false; couples_info[i].num_offspring = 1;          // :2148-2149
        }
        P._couples_info = couples_info;
        return true;
    }

    // CommFunc::ras_rank (src/CommFunc.cpp:152-161) for assort_mate (:2278-2279): O(n^2) on the host, a stable sort on the device
    static std::vector<unsigned long int> rank(std::vector<double>& x)
    {
        std::vector<unsigned long long> r(x.size());
        if (gev_rank_f64(ctx(), x.data(), x.size(), r.data())) { fail("gev_rank_f64"); return CommFunc::ras_rank(x); }
        return std::vector<unsigned long int>(r.begin(), r.end());
    }

    // Simulation::ras_scale_AD_compute_GEF (:3075-3206) on the device.  Its floats agree with the host's to ~1e-12 relative (device
    // log(), parallel variance), not bit for bit, so the bound program keeps the host version unless GEV_GEF_DEVICE=1.
    static bool use_device_gef() { static const bool on = getenv("GEV_GEF_DEVICE") && atoi(getenv("GEV_GEF_DEVICE")) != 0; return on; }
    static bool scale_gef(Simulation& S, int gen_num, int ipop, int iphen, double s2_a_gen0, double s2_d_gen0)
    {
        Population& P = S.population[ipop];
        const unsigned seed = S.ras_glob_seed();                                        // :3078
        const size_t n = P.h.size();
        const Phenotype_scheme& ps = P._pheno_scheme[iphen];
        gev_gef_params par = {ps._va, ps._vd, ps._ve, ps._vf, ps._beta, s2_a_gen0, s2_d_gen0, gen_num, 0};
        std::vector<double> cs(n), ff(n, 0.0), fm(n, 0.0), add(n), dom(n), bv(n), e(n), pe(n), phen(n);
        for (size_t i = 0; i < n; i++) {
            cs[i] = P.h[i].common_sibling[iphen];
            if (gen_num > 0) {                                                          // :3118-3131
                const unsigned long ind_f = P.h[i].ID_Father, ind_m = P.h[i].ID_Mother;
                if (S._vt_type == 1) { ff[i] = S._Pop_info_prev_gen[ipop].phen[iphen][ind_f]; fm[i] = S._Pop_info_prev_gen[ipop].phen[iphen][ind_m]; }
                else if (S._vt_type == 2) { ff[i] = S._Pop_info_prev_gen[ipop].parental_effect[iphen][ind_f]; fm[i] = S._Pop_info_prev_gen[ipop].parental_effect[iphen][ind_m]; }
            }
        }
        if (gev_scale_ad_compute_gef(ctx(), ipop, iphen, &par, seed, cs.data(), ff.data(), fm.data(), add.data(), dom.data(), bv.data(), e.data(), pe.data(), phen.data()))
            return fail("gev_scale_ad_compute_gef");
        for (size_t i = 0; i < n; i++) {
            Human& h = P.h[i];
            h.additive[iphen] = add[i]; h.dominance[iphen] = dom[i]; h.bv[iphen] = bv[i]; h.e_noise[iphen] = e[i]; h.parental_effect[iphen] = pe[i]; h.phen[iphen] = phen[i];
        }
        return true;
    }

    // Population::ras_save_human_info (src/Population.cpp:510-568), the per-generation .info text dump -- after the GPU took over
    // the genotype work this is the longest host phase of a generation (tools/cli_timing.py: 0.18 of 0.25 s at 100k individuals),
    // because the reference ends every line with std::endl (one flush = one write() per individual).  Same bytes (default
    // ostream formatting of a double = "%g", 6 significant digits; ids printed 1-based), built in memory by up to 8 host threads
    // and written once.
    static void format_humans(const Population& P, size_t i0, size_t i1, int npheno, std::string& out)
    {
        char buf[64];
        auto num = [&](double v, char end) { const int n = snprintf(buf, sizeof buf, "%g", v); out.append(buf, (size_t)n); out.push_back(end); };
        auto id = [&](unsigned long v) { const int n = snprintf(buf, sizeof buf, "%lu ", v); out.append(buf, (size_t)n); };
        out.reserve((i1 - i0) * (64 + 90 * (size_t)npheno));
        for (size_t i = i0; i < i1; i++) {
            const Human& h = P.h[i];
            id(h.ID + 1); id(h.ID_Father + 1); id(h.ID_Mother + 1); id(h.ID_Fathers_Father + 1); id(h.ID_Fathers_Mother + 1);
            id(h.ID_Mothers_Father + 1); id(h.ID_Mothers_Mother + 1);
            { const int n = snprintf(buf, sizeof buf, "%d ", (int)h.sex); out.append(buf, (size_t)n); }
            for (int j = 0; j < npheno; j++) {
                num(h.additive[j], ' '); num(h.dominance[j], ' '); num(h.bv[j], ' '); num(h.common_sibling[j], ' ');
                num(h.e_noise[j], ' '); num(h.parental_effect[j], ' '); num(h.phen[j], ' ');
            }
            num(h.mating_value, ' '); num(h.selection_value, ' '); num(h.selection_value_func, '\n');
        }
    }
    static bool save_human_info(Simulation& S, int ipop, int gen_num)
    {
        Population& P = S.population[ipop];
        const int npheno = (int)P._pheno_scheme.size();
        std::string head = "ID ID_Father ID_Mother ID_Fathers_Father ID_Fathers_Mother ID_Mothers_Father ID_Mothers_Mother sex ";
        for (int j = 0; j < npheno; j++) {
            const std::string ph = "ph" + std::to_string(j + 1);
            for (const char* c : {"_A ", "_D ", "_G ", "_C ", "_E ", "_F ", "_P "}) head += ph + c;
        }
        head += "MV SV SV_f\n";
        // formatting is independent per individual: a few host threads each format a contiguous slice, the slices are written in order
        const size_t n = P.h.size();
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t nt = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(hw ? hw : 1, 8), n / 4096 + 1));
        std::vector<std::string> part(nt);
        std::vector<std::thread> th;
        for (size_t t = 1; t < nt; t++) th.emplace_back([&, t]() { format_humans(P, n * t / nt, n * (t + 1) / nt, npheno, part[t]); });
        format_humans(P, 0, n / nt, npheno, part[0]);
        for (auto& x : th) x.join();
        std::ofstream f((P._out_prefix + ".info.pop" + std::to_string(P._pop_num + 1) + ".gen" + std::to_string(gen_num) + ".txt").c_str(), std::ios::binary);
        f.write(head.data(), (std::streamsize)head.size());
        for (const std::string& p : part) f.write(p.data(), (std::streamsize)p.size());
        return true;
    }

    // Simulation::ras_compute_AD (:2624-2749)
    static bool compute_AD(Simulation& S, int ipop, int /*gen_num*/)
    {
        std::vector<Human>& h = S.population[ipop].h;
        const size_t n = h.size(), nchr = h[0].chr.size(), nphen = S.population[ipop]._pheno_scheme.size();
        std::vector<double> A(n * nphen), D(n * nphen), Ac(n * nchr * nphen), Dc(n * nchr * nphen);
        if (gev_compute_ad(ctx(), ipop, A.data(), D.data(), Ac.data(), Dc.data())) return fail("gev_compute_ad");
        for (size_t ih = 0; ih < n; ih++)
            for (size_t p = 0; p < nphen; p++) {
                double bv = 0;
                for (size_t k = 0; k < nchr; k++) {
                    const double a = Ac[(ih * nchr + k) * nphen + p], d = Dc[(ih * nchr + k) * nphen + p];
                    h[ih].chr[k].additive_chr[p] = a; h[ih].chr[k].dominance_chr[p] = d; h[ih].chr[k].bv_chr[p] = a + d;
                    bv += a + d;                                                                   // :2738
                }
                h[ih].additive[p] = A[ih * nphen + p]; h[ih].dominance[p] = D[ih * nphen + p]; h[ih].bv[p] = bv;
            }
        return true;
    }

    // row movement of Simulation::ras_do_migration (:959-981); WHO moves was decided by the reference's own code just above
    static bool migrate(Simulation& S, const std::vector<std::vector<unsigned long int> >& move_sample_pop, const std::vector<std::vector<unsigned long int> >& num_move)
    {
        std::vector<gev_move> moves;                   // append order of :971-981: for i, for j != i, for k (with the reference's running k, :925-936)
        for (int i = 0; i < S._n_pop; i++) {
            unsigned long k = 0;
            for (int j = 0; j < S._n_pop; j++) {
                if (i == j) continue;
                while (k < num_move[i][j]) { moves.push_back(gev_move{i, j, move_sample_pop[i][k]}); k++; }
            }
            if (k != move_sample_pop[i].size()) {      // with more than two populations the reference's running k (:930) leaves sampled people in no camp: they vanish
                std::cout << "Error: the GPU build does not reproduce the loss of unassigned emigrants (more than two populations)." << std::endl;
                return false;
            }
        }
        if (gev_migrate(ctx(), moves.data(), moves.size())) return fail("gev_migrate");
        return true;
    }

    // Simulation::ras_convert_interval_to_hap_matrix (:1186-1230)
    static bool hap_matrix(Simulation& S, int ipop, std::vector<Legend>& pops_legend, int ichr, Hap_SNP& hap_snp)
    {
        const size_t n_human = S.population[ipop].h.size(), nsnp = pops_legend[ipop].id.size(), w = (nsnp + 63) / 64;
        std::vector<uint64_t> bits(2 * n_human * std::max<size_t>(w, 1));
        if (gev_download_haps(ctx(), ipop, ichr, 0, 2 * n_human, bits.data(), w)) return fail("gev_download_haps");
        hap_snp.hap.assign(2 * n_human, std::vector<bool>(nsnp, false));
        for (size_t r = 0; r < 2 * n_human; r++)
            for (size_t ii = 0; ii < nsnp; ii++) hap_snp.hap[r][ii] = (bits[r * w + (ii >> 6)] >> (ii & 63)) & 1;
        return true;
    }

    // rows x nsnp genotype bits of a population from the library into the reference's vector<vector<bool>> (2 rows per individual)
    static bool fetch_haps(int ipop, int ichr, size_t n_human, size_t nsnp, std::vector<std::vector<bool> >& rows)
    {
        const size_t w = (nsnp + 63) / 64;
        std::vector<uint64_t> bits(2 * n_human * std::max<size_t>(w, 1));
        if (gev_download_haps(ctx(), ipop, ichr, 0, 2 * n_human, bits.data(), w)) return fail("gev_download_haps");
        rows.assign(2 * n_human, std::vector<bool>(nsnp, false));
        for (size_t r = 0; r < 2 * n_human; r++)
            for (size_t ii = 0; ii < nsnp; ii++) rows[r][ii] = (bits[r * w + (ii >> 6)] >> (ii & 63)) & 1;
        return true;
    }
    // the VCF-panel siblings: the panel was handed to the library at start-up (init_static), the genotype rows are resident
    // Simulation::ras_convert_interval_from_vcf_to_hap_matrix (:1838-1881)
    static bool vcf_hap_matrix(Simulation& S, std::vector<vcf_structure>& vcf_all, int ipop, int ichr, Hap_SNP& hap_snp)
    {
        const size_t n_human = S.population[ipop].h.size(), nsnp = vcf_all[ipop].ID.size();
        std::cout << "      n_human=" << n_human << std::endl;
        return fetch_haps(ipop, ichr, n_human, nsnp, hap_snp.hap);
    }
    // Simulation::ras_convert_interval_from_vcf_to_vcf_structure (:1690-1758): columns, meta lines and sample names as :1701-1733
    static bool vcf_structure_out(Simulation& S, vcf_structure& vcf_out, int gen_num, int ipop, int ichr, std::vector<vcf_structure>& vcf_all)
    {
        Population& P = S.population[ipop];
        const size_t n_human = P.h.size(), nsnp = vcf_all[ipop].ID.size();
        std::cout << "      n_human=" << n_human << std::endl;
        vcf_out.SAMPLES.resize(n_human);
        vcf_out.CHROM = vcf_all[ipop].CHROM; vcf_out.POS = vcf_all[ipop].POS; vcf_out.ID = vcf_all[ipop].ID; vcf_out.REF = vcf_all[ipop].REF;
        vcf_out.ALT = vcf_all[ipop].ALT; vcf_out.QUAL = vcf_all[ipop].QUAL; vcf_out.FILTER = vcf_all[ipop].FILTER;
        vcf_out.INFO = std::vector<std::string>(nsnp, "."); vcf_out.FORMAT = std::vector<std::string>(nsnp, "GT");
        std::time_t t = std::time(NULL);
        char day[100];
        std::strftime(day, sizeof(day), "%Y%m%d", std::localtime(&t));
        vcf_out.meta_lines = {"##fileformat=VCFv4.1", "##Phasing=phased", "##CreatedBy=GeneEvolve", "##fileDate=" + std::string(day),
                              "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">"};
        for (size_t ih = 0; ih < n_human; ih++) vcf_out.SAMPLES[ih] = "g" + std::to_string(gen_num) + "_" + std::to_string(P.h[ih].ID + 1);
        return fetch_haps(ipop, ichr, n_human, nsnp, vcf_out.data);
    }
    // Simulation::ras_convert_interval_from_vcf_to_plink (:1477-1571): matrix from the device, ids and map exactly as :1546-1568
    static bool vcf_plink_matrix(Simulation& S, std::vector<std::vector<bool> >& matrix_plink_ped, plink_PED_ids& plink_ped_ids, plink_MAP& plink_map,
                                 int ipop, int ichr, std::vector<vcf_structure>& vcf_all)
    {
        Population& P = S.population[ipop];
        const size_t n_human = P.h.size(), nsnp = vcf_all[ipop].ID.size(), w = (2 * nsnp + 63) / 64;
        std::cout << "      n_human=" << n_human << std::endl;
        std::vector<uint64_t> bits(n_human * std::max<size_t>(w, 1));
        if (gev_download_plink_matrix(ctx(), ipop, ichr, 0, n_human, bits.data(), w)) return fail("gev_download_plink_matrix");
        matrix_plink_ped.assign(n_human, std::vector<bool>(nsnp * 2, false));
        plink_ped_ids.alloc(n_human); plink_map.alloc(nsnp);
        for (size_t ih = 0; ih < n_human; ih++) {
            for (size_t b = 0; b < 2 * nsnp; b++) matrix_plink_ped[ih][b] = (bits[ih * w + (b >> 6)] >> (b & 63)) & 1;
            plink_ped_ids.FID[ih] = std::to_string(P.h[ih].ID_Father + 1); plink_ped_ids.IID[ih] = std::to_string(P.h[ih].ID + 1);
            plink_ped_ids.PID[ih] = std::to_string(P.h[ih].ID_Father + 1); plink_ped_ids.MID[ih] = std::to_string(P.h[ih].ID_Mother + 1);
            plink_ped_ids.sex[ih] = P.h[ih].sex; plink_ped_ids.phen[ih] = -9;
        }
        for (size_t i = 0; i < nsnp; i++) {
            plink_map.chr[i] = std::to_string(S._all_active_chrs[ichr]); plink_map.rs[i] = vcf_all[ipop].ID[i]; plink_map.cM[i] = 0;
            plink_map.pos[i] = vcf_all[ipop].POS[i]; plink_map.al0[i] = vcf_all[ipop].REF[i]; plink_map.al1[i] = vcf_all[ipop].ALT[i];
        }
        return true;
    }

    // Simulation::ras_convert_interval_to_format_plink (:1308-1416): matrix from the device, ids and map exactly as :1391-1413
    static bool plink_matrix(Simulation& S, int ipop, std::vector<Legend>& pops_legend, int ichr, std::vector<std::vector<bool> >& matrix_plink_ped,
                             plink_PED_ids& plink_ped_ids, plink_MAP& plink_map)
    {
        Population& P = S.population[ipop];
        const size_t n_human = P.h.size(), nsnp = pops_legend[ipop].id.size(), w = (2 * nsnp + 63) / 64;
        std::vector<uint64_t> bits(n_human * std::max<size_t>(w, 1));
        if (gev_download_plink_matrix(ctx(), ipop, ichr, 0, n_human, bits.data(), w)) return fail("gev_download_plink_matrix");
        matrix_plink_ped.assign(n_human, std::vector<bool>(nsnp * 2, false));
        plink_ped_ids.alloc(n_human); plink_map.alloc(nsnp);
        for (size_t ih = 0; ih < n_human; ih++) {
            for (size_t b = 0; b < 2 * nsnp; b++) matrix_plink_ped[ih][b] = (bits[ih * w + (b >> 6)] >> (b & 63)) & 1;
            plink_ped_ids.FID[ih] = std::to_string(P.h[ih].ID_Father + 1); plink_ped_ids.IID[ih] = std::to_string(P.h[ih].ID + 1);
            plink_ped_ids.PID[ih] = std::to_string(P.h[ih].ID_Father + 1); plink_ped_ids.MID[ih] = std::to_string(P.h[ih].ID_Mother + 1);
            plink_ped_ids.sex[ih] = P.h[ih].sex; plink_ped_ids.phen[ih] = -9;
        }
        for (size_t i = 0; i < nsnp; i++) {
            plink_map.chr[i] = std::to_string(S._all_active_chrs[ichr]); plink_map.rs[i] = pops_legend[ipop].id[i]; plink_map.cM[i] = 0;
            plink_map.pos[i] = pops_legend[ipop].pos[i]; plink_map.al0[i] = pops_legend[ipop].al0[i]; plink_map.al1[i] = pops_legend[ipop].al1[i];
        }
        return true;
    }

    // Simulation::ras_write_hap_to_interval_format (:1582-1633): the interval state lives in the library
    static bool write_interval(Simulation& S, int gen_num)
    {
        const std::string sep = " ";
        const int n_chr = (int)S.population[0].h[0].chr.size();
        for (int ipop = 0; ipop < S._n_pop; ipop++) {
            Population& P = S.population[ipop];
            const size_t nind = P.h.size();
            for (int ichr = 0; ichr < n_chr; ichr++) {
                size_t n_parts = 0;
                std::vector<uint64_t> off(2 * nind + 1);
                if (gev_download_intervals(ctx(), ipop, ichr, NULL, off.data(), &n_parts)) return fail("gev_download_intervals");
                std::vector<gev_part> parts(std::max<size_t>(n_parts, 1));
                if (gev_download_intervals(ctx(), ipop, ichr, parts.data(), off.data(), &n_parts)) return fail("gev_download_intervals");
                std::ofstream file_out((S._out_prefix + ".pop" + std::to_string(ipop + 1) + ".gen" + std::to_string(gen_num) + ".chr" + std::to_string(S._all_active_chrs[ichr]) + ".int").c_str());
                file_out << "h_ID" << sep << "chr" << sep << "hap" << sep << "st" << sep << "en" << sep << "hap_index" << sep << "gen0_indv" << sep << "root_pop" << std::endl;
                for (size_t ih = 0; ih < nind; ih++)
                    for (int ihap = 0; ihap < 2; ihap++)
                        for (uint64_t ip = off[2 * ih + ihap]; ip < off[2 * ih + ihap + 1]; ip++) {
                            const gev_part& p = parts[ip];
                            const std::string gen0_indv = S.population[p.root_population]._indv_id[p.hap_index / 2] + ((p.hap_index & 1) ? ".2" : ".1");   // :3031-3033
                            file_out << P.h[ih].ID + 1 << sep << S._all_active_chrs[ichr] << sep << ihap << sep << p.st << sep << p.en << sep << p.hap_index + 1 << sep
                                     << gen0_indv << sep << p.root_population + 1 << std::endl;
                        }
            }
        }
        return true;
    }
};

// free functions the edited Simulation.cpp calls
bool gevglue_plink_matrix(Simulation& S, int ipop, std::vector<Legend>& pops_legend, int ichr, std::vector<std::vector<bool> >& m, plink_PED_ids& ids, plink_MAP& map) { return GevGlue::plink_matrix(S, ipop, pops_legend, ichr, m, ids, map); }
bool gevglue_write_interval(Simulation& S, int gen_num) { return GevGlue::write_interval(S, gen_num); }
bool gevglue_vcf_hap_matrix(Simulation& S, std::vector<vcf_structure>& v, int ipop, int ichr, Hap_SNP& hap_snp) { return GevGlue::vcf_hap_matrix(S, v, ipop, ichr, hap_snp); }
bool gevglue_vcf_structure(Simulation& S, vcf_structure& out, int gen_num, int ipop, int ichr, std::vector<vcf_structure>& v) { return GevGlue::vcf_structure_out(S, out, gen_num, ipop, ichr, v); }
bool gevglue_vcf_plink_matrix(Simulation& S, std::vector<std::vector<bool> >& m, plink_PED_ids& ids, plink_MAP& map, int ipop, int ichr, std::vector<vcf_structure>& v) { return GevGlue::vcf_plink_matrix(S, m, ids, map, ipop, ichr, v); }
bool gevglue_init_static(Simulation& S) { return GevGlue::init_static(S); }
bool gevglue_after_gen0(Simulation& S, int ipop, unsigned seed) { return GevGlue::after_gen0(S, ipop, seed); }
std::vector<Human> gevglue_reproduce(Simulation& S, int ipop, int gen_num) { return GevGlue::reproduce(S, ipop, gen_num); }
bool gevglue_compute_AD(Simulation& S, int ipop, int gen_num) { return GevGlue::compute_AD(S, ipop, gen_num); }
bool gevglue_migrate(Simulation& S, const std::vector<std::vector<unsigned long int> >& a, const std::vector<std::vector<unsigned long int> >& b) { return GevGlue::migrate(S, a, b); }
bool gevglue_hap_matrix(Simulation& S, int ipop, std::vector<Legend>& pops_legend, int ichr, Hap_SNP& hap_snp) { return GevGlue::hap_matrix(S, ipop, pops_legend, ichr, hap_snp); }
bool gevglue_presample(Simulation& S, int ipop, int gen_num) { return GevGlue::presample(S, ipop, gen_num); }
bool gevglue_save_human_info(Simulation& S, int ipop, int gen_num) { return GevGlue::save_human_info(S, ipop, gen_num); }
std::vector<unsigned long int> gevglue_rank(std::vector<double>& x) { return GevGlue::rank(x); }
bool gevglue_use_device_gef() { return GevGlue::use_device_gef(); }
bool gevglue_use_device_mate() { return GevGlue::use_device_mate(); }
bool gevglue_random_mate(Simulation& S, int ipop, int gen_ind) { return GevGlue::random_mate(S, ipop, gen_ind); }
bool gevglue_scale_gef(Simulation& S, int gen_num, int ipop, int iphen, double s2_a_gen0, double s2_d_gen0) { return GevGlue::scale_gef(S, gen_num, ipop, iphen, s2_a_gen0, s2_d_gen0); }
