// tools/host_demo.cpp -- a C++ host driving the library through the C-ABI only (no Python, no torch):
// synthetic config-1-shaped inputs, G generations of reproduce + ras_compute_AD, FNV checksums of the
// final genotype matrix and A values.  tests/test_gpu_parity.py runs it and compares the checksums with
// the same scenario driven from Python through ctypes.
//   g++ -O2 -std=c++14 tools/host_demo.cpp -Lgeneevolve_amd/csrc -lgeneevolve_amd -Wl,-rpath,'$ORIGIN/../geneevolve_amd/csrc' -o tools/host_demo
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../geneevolve_amd/host/gev_host.hpp"

static uint64_t fnv(const void* p, size_t n, uint64_t h = 1469598103934665603ull)
{
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}
int main(int argc, char** argv)
{
    const size_t N = argc > 1 ? atol(argv[1]) : 500, L = argc > 2 ? atol(argv[2]) : 5000;
    const int G = argc > 3 ? atoi(argv[3]) : 4;
    const bool presample = argc > 4 && atoi(argv[4]) != 0;     // draw the seeds first and let the GPU sample while the host mates
    const size_t R = 201, C = 64;
    try {
        gev::Simulation sim(-1, 1, 1, 1, 12345, true);
        std::vector<uint64_t> bp(R), pos(L), cvbp(C); std::vector<double> prob(R, 5e-3), rate(R, 5e-3), a(C), d(C, 0.0);
        for (size_t j = 0; j < R; j++) bp[j] = 1000 + 10000 * j;
        prob[0] = 0; rate[0] = 0;
        for (size_t i = 0; i < L; i++) pos[i] = 1000 + (2000000 / L) * i;
        for (size_t i = 0; i < C; i++) { cvbp[i] = 1500 + 31000 * i; a[i] = (double)((i * 37) % 11) - 5.0; }
        gev::check(gev_set_rmap(sim.ctx, 0, 0, bp.data(), prob.data(), R, 10000));
        gev::check(gev_set_mutmap(sim.ctx, 0, 0, bp.data(), rate.data(), R));
        gev::check(gev_set_snps(sim.ctx, 0, 0, pos.data(), L));
        gev::check(gev_set_cvs(sim.ctx, 0, 0, 0, cvbp.data(), a.data(), d.data(), C, 0.0));
        gev::check(gev_synth_founders(sim.ctx, 0, 0, 2 * N, 77));
        gev::check(gev_synth_cv_founders(sim.ctx, 0, 0, 0, 2 * N, 78));
        sim.ras_initial_human_gen0(0, N);
        std::vector<double> A, D;
        for (int g = 1; g <= G; g++) {
            // Simulation::sim_next_generation order (src/Simulation.cpp:1907-1935): random_mate, reproduce, ras_compute_AD
            const std::vector<double> svf(sim.sex[0].size(), 1.0);                    // selection_value_func: everybody may marry
            if (presample) { if (!sim.random_mate_and_reproduce(0, svf, N)) return 1; }
            else {
                if (!gev::random_mate(sim.sex[0], svf, N, sim.ras_glob_seed(), sim.couples[0])) return 1;
                sim.reproduce(0);
            }
            if (!sim.ras_compute_AD(0, A, D)) return 1;
        }
        const size_t w = (L + 63) / 64;
        std::vector<uint64_t> bits(2 * N * w);
        gev::check(gev_download_haps(sim.ctx, 0, 0, 0, 2 * N, bits.data(), w));
        {   // the device rank against the definition
            std::vector<double> x = {0.5, -1.0, 0.5, 2.0, -1.0, 0.0};
            const std::vector<unsigned long long> r = sim.ras_rank(x), want = {3, 0, 4, 5, 1, 2};
            if (r != want) { printf("Error: ras_rank\n"); return 1; }
        }
        printf("HAPS %016llx\nADD %016llx\nSEX %016llx\n", (unsigned long long)fnv(bits.data(), bits.size() * 8),
               (unsigned long long)fnv(A.data(), A.size() * 8), (unsigned long long)fnv(sim.sex[0].data(), sim.sex[0].size()));
    } catch (const std::exception& e) { printf("Error: %s\n", e.what()); return 1; }
    return 0;
}
