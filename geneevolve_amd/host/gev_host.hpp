// gev_host.hpp -- C++ host-side mirror of the reference's seam (header only, no HIP, no torch).
//
// The reference host is C++ (`class Simulation`, reference src/Simulation.h:64-144).  This class keeps
// what the reference keeps on the host -- the ras_glob_seed() stream (src/Simulation.cpp:17-21), the
// couples list, sexes -- and forwards the seam functions, under their reference names, to the C-ABI
// of include/geneevolve_amd.h.  INTEGRATION.md shows the same calls as a patch to the reference.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/geneevolve_amd.h"

namespace gev {

// std::default_random_engine (minstd_rand0) + std::uniform_int_distribution<unsigned>(1,1000000):
// libstdc++ down-scaling branch, scaling = 2147483645/1000000, reject engine()-1 >= 1000000*scaling
class GlobSeed {
    uint64_t x_;
public:
    explicit GlobSeed(unsigned seed) { x_ = seed % 2147483647ull; if (!x_) x_ = 1; }   // glob_generator.seed(par._seed), :75-76
    unsigned operator()()                                                            // Simulation::ras_glob_seed, :17-21
    {
        const uint64_t scaling = 2147483645ull / 1000000ull, past = 1000000ull * scaling;
        uint64_t r;
        do { x_ = x_ * 16807ull % 2147483647ull; r = x_ - 1; } while (r >= past);
        return (unsigned)(r / scaling + 1);
    }
};

// std::default_random_engine + the two libstdc++ distributions Simulation::random_mate uses
class Minstd {
    uint64_t x_;
public:
    explicit Minstd(unsigned seed) { x_ = seed % 2147483647ull; if (!x_) x_ = 1; }
    uint64_t operator()() { x_ = x_ * 16807ull % 2147483647ull; return x_; }
    double u01()                                                      // generate_canonical<double,53>: first call = low digit
    {
        const double lo = (double)((*this)() - 1), hi = (double)((*this)() - 1);
        const double R2 = (double)(2147483646.0L * 2147483646.0L);
        double r = (lo + hi * 2147483646.0) / R2;
        return r >= 1.0 ? std::nextafter(1.0, 0.0) : r;
    }
    uint64_t uniform_int(uint64_t lo, uint64_t hi)                     // down-scaling branch (hi-lo < 2147483645)
    {
        const uint64_t uerange = hi - lo + 1, scaling = 2147483645ull / uerange, past = uerange * scaling;
        uint64_t r;
        do r = (*this)() - 1; while (r >= past);
        return r / scaling + lo;
    }
};
// Simulation::random_mate (src/Simulation.cpp:2090-2157); seed = the ras_glob_seed() value drawn at :2092
inline bool random_mate(const std::vector<uint8_t>& sex, const std::vector<double>& selection_value_func, size_t pop_size, unsigned seed,
                        std::vector<gev_couple>& couples)
{
    Minstd generator(seed);
    std::vector<uint64_t> pos_male, pos_female;
    for (size_t i = 0; i < sex.size(); i++) {
        const double r = generator.u01();
        if (r < selection_value_func[i]) { if (sex[i] == 1) pos_male.push_back(i); else if (sex[i] == 2) pos_female.push_back(i); }
    }
    if (pos_male.empty() || pos_female.empty()) { printf("Error: No one can marry, num_males_mate=%zu, num_females_mate=%zu\n", pos_male.size(), pos_female.size()); return false; }
    Minstd g_f(seed + 1), g_m(seed + 2);
    couples.assign(pop_size, gev_couple{0, 0, 0, 1});
    for (size_t i = 0; i < pop_size; i++) {
        couples[i].pos_male = pos_male[g_f.uniform_int(0, pos_male.size() - 1)];
        couples[i].pos_female = pos_female[g_m.uniform_int(0, pos_female.size() - 1)];
    }
    return true;
}

inline void check(int rc) { if (rc != GEV_OK) throw std::runtime_error(std::string(gev_last_error())); }

class Simulation {
public:
    gev_ctx* ctx = nullptr;
    int n_pop, nchr, nphen;
    bool has_mutation_map;
    GlobSeed glob;
    std::vector<std::vector<gev_couple>> couples;     // population[ipop]._couples_info
    std::vector<std::vector<uint8_t>> sex;            // Human::sex of the current generation

    Simulation(int device, int n_pop_, int nchr_, int nphen_, unsigned seed, bool has_mut)
        : n_pop(n_pop_), nchr(nchr_), nphen(nphen_), has_mutation_map(has_mut), glob(seed), couples(n_pop_), sex(n_pop_)
    { check(gev_create(&ctx, device, n_pop, nchr, nphen)); }
    ~Simulation() { gev_destroy(ctx); }
    Simulation(const Simulation&) = delete;

    unsigned ras_glob_seed() { return glob(); }

    bool ras_initial_human_gen0(int ipop, size_t n_people)                      // src/Simulation.cpp:3000
    {
        sex[ipop].resize(n_people);
        check(gev_init_gen0(ctx, ipop, n_people, ras_glob_seed(), sex[ipop].data()));
        return true;
    }
    // src/Simulation.cpp:2394: draws 1 + n_people*nchr glob seeds exactly as the reference would (:2398, :2500)
    const std::vector<uint8_t>& reproduce(int ipop)
    {
        size_t n_people = 0;
        for (const gev_couple& c : couples[ipop]) if (!c.inbreed) n_people += c.num_offspring;
        const unsigned seed = ras_glob_seed();
        std::vector<uint32_t> mut_seeds;
        if (has_mutation_map) { mut_seeds.resize(n_people * nchr); for (auto& s : mut_seeds) s = ras_glob_seed(); }
        std::vector<uint8_t> s(n_people);
        check(gev_reproduce(ctx, ipop, couples[ipop].data(), couples[ipop].size(), seed,
                            has_mutation_map ? mut_seeds.data() : nullptr, mut_seeds.size(), n_people, s.data()));
        sex[ipop].swap(s);
        return sex[ipop];
    }
    // random mating with a head start for the GPU: the ras_glob_seed() values of random_mate (:2092) and of
    // reproduce (:2398, :2500) are drawn in the reference's order BEFORE the couples are formed (one child per couple, so
    // the offspring count is pop_size, :2146-2151); gev_presample starts the sampling kernels, the host mates meanwhile.
    bool random_mate_and_reproduce(int ipop, const std::vector<double>& selection_value_func, size_t pop_size)
    {
        const unsigned seed_mate = ras_glob_seed();
        const unsigned seed = ras_glob_seed();
        std::vector<uint32_t> mut_seeds;
        if (has_mutation_map) { mut_seeds.resize(pop_size * nchr); for (auto& s : mut_seeds) s = ras_glob_seed(); }
        check(gev_presample(ctx, ipop, seed, has_mutation_map ? mut_seeds.data() : nullptr, mut_seeds.size(), pop_size));
        if (!random_mate(sex[ipop], selection_value_func, pop_size, seed_mate, couples[ipop])) return false;
        std::vector<uint8_t> s(pop_size);
        check(gev_reproduce(ctx, ipop, couples[ipop].data(), couples[ipop].size(), seed,
                            has_mutation_map ? mut_seeds.data() : nullptr, mut_seeds.size(), pop_size, s.data()));
        sex[ipop].swap(s);
        return true;
    }
    // CommFunc::ras_rank (src/CommFunc.cpp:152-161) on the device: the O(n^2) step of assort_mate (:2278-2279)
    std::vector<unsigned long long> ras_rank(const std::vector<double>& x)
    {
        std::vector<unsigned long long> r(x.size());
        check(gev_rank_f64(ctx, x.data(), x.size(), r.data()));
        return r;
    }
    bool ras_compute_AD(int ipop, std::vector<double>& additive, std::vector<double>& dominance)   // :2624
    {
        size_t n = 0; check(gev_pop_size(ctx, ipop, &n));
        additive.resize(n * nphen); dominance.resize(n * nphen);
        const int rc = gev_compute_ad(ctx, ipop, additive.data(), dominance.data(), nullptr, nullptr);
        if (rc != GEV_OK) { printf("%s\n", gev_last_error()); return false; }                         // error convention of the reference: message + false
        return true;
    }
    bool ras_do_migration(const std::vector<gev_move>& moves) { check(gev_migrate(ctx, moves.data(), moves.size())); return true; }   // :877
};

} // namespace gev
