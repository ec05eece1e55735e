// Test infrastructure (tests/build_glue_on_oracle.py): the few gev_* entry points integration/gev_glue.cpp uses, forwarded to the CPU oracle (gevo_*), so that
// the glue and the edit script of build_gpu_cli.py can be checked in a container without a GPU
// (oracle/_ref/GeneEvolve_glue_on_oracle).  Never part of the product: the product program links libgeneevolve_amd.so.
#include <cstddef>
#include <cstdint>
#include "geneevolve_amd.h"
extern "C" {
struct Ctx;
const char* gevo_last_error(void);
int gevo_create(Ctx** out, int n_pop, int nchr, int nphen);
int gevo_set_rmap(Ctx*, int, int, const uint64_t*, const double*, size_t, uint64_t);
int gevo_set_mutmap(Ctx*, int, int, const uint64_t*, const double*, size_t);
int gevo_set_snps(Ctx*, int, int, const uint64_t*, size_t);
int gevo_set_cvs(Ctx*, int, int, int, const uint64_t*, const double*, const double*, size_t, double);
int gevo_upload_founders(Ctx*, int, int, const uint64_t*, size_t, size_t, size_t);
int gevo_upload_cv_founders(Ctx*, int, int, int, const uint64_t*, size_t, size_t, size_t);
int gevo_init_gen0(Ctx*, int, size_t, uint32_t, uint8_t*);
int gevo_reproduce(Ctx*, int, const gev_couple*, size_t, uint32_t, const uint32_t*, size_t, size_t, uint8_t*);
int gevo_compute_ad(Ctx*, int, double*, double*, double*, double*);
int gevo_migrate(Ctx*, const gev_move*, size_t);
int gevo_download_haps(Ctx*, int, int, size_t, size_t, uint64_t*, size_t);
int gevo_download_plink_matrix(Ctx*, int, int, size_t, size_t, uint64_t*, size_t);
int gevo_download_intervals(Ctx*, int, int, gev_part*, uint64_t*, size_t*);

int gevo_rank_f64(Ctx*, const double*, size_t, unsigned long long*);
int gevo_random_mate(Ctx*, int, uint32_t, const double*, size_t, gev_couple*, size_t*, size_t*);
int gevo_scale_ad_compute_gef(Ctx*, int, int, const gev_gef_params*, uint32_t, const double*, const double*, const double*, double*, double*, double*, double*, double*, double*);

const char* gev_last_error(void) { return gevo_last_error(); }
int gev_presample(gev_ctx*, int, uint32_t, const uint32_t*, size_t, size_t) { return 0; }      // a head start only: the oracle samples inside reproduce
int gev_rank_f64(gev_ctx* c, const double* x, size_t n, unsigned long long* r) { return gevo_rank_f64((Ctx*)c, x, n, r); }
int gev_random_mate(gev_ctx* c, int p, uint32_t seed, const double* svf, size_t n, gev_couple* out, size_t* nm, size_t* nf) { return gevo_random_mate((Ctx*)c, p, seed, svf, n, out, nm, nf); }
int gev_scale_ad_compute_gef(gev_ctx* c, int p, int ph, const gev_gef_params* par, uint32_t seed, const double* cs, const double* ff, const double* fm,
                             double* a, double* d, double* bv, double* e, double* pe, double* phen)
{ return gevo_scale_ad_compute_gef((Ctx*)c, p, ph, par, seed, cs, ff, fm, a, d, bv, e, pe, phen); }
int gev_create(gev_ctx** out, int, int n_pop, int nchr, int nphen) { return gevo_create((Ctx**)out, n_pop, nchr, nphen); }
int gev_set_rmap(gev_ctx* c, int p, int k, const uint64_t* bp, const double* pr, size_t R, uint64_t d) { return gevo_set_rmap((Ctx*)c, p, k, bp, pr, R, d); }
int gev_set_mutmap(gev_ctx* c, int p, int k, const uint64_t* bp, const double* r, size_t M) { return gevo_set_mutmap((Ctx*)c, p, k, bp, r, M); }
int gev_set_snps(gev_ctx* c, int p, int k, const uint64_t* pos, size_t L) { return gevo_set_snps((Ctx*)c, p, k, pos, L); }
int gev_set_cvs(gev_ctx* c, int p, int ph, int k, const uint64_t* bp, const double* a, const double* d, size_t C, double vd) { return gevo_set_cvs((Ctx*)c, p, ph, k, bp, a, d, C, vd); }
int gev_upload_founders(gev_ctx* c, int p, int k, const uint64_t* b, size_t w, size_t nh, size_t L) { return gevo_upload_founders((Ctx*)c, p, k, b, w, nh, L); }
int gev_upload_cv_founders(gev_ctx* c, int p, int ph, int k, const uint64_t* b, size_t w, size_t nh, size_t C) { return gevo_upload_cv_founders((Ctx*)c, p, ph, k, b, w, nh, C); }
int gev_init_gen0(gev_ctx* c, int p, size_t n, uint32_t s, uint8_t* sex) { return gevo_init_gen0((Ctx*)c, p, n, s, sex); }
int gev_reproduce(gev_ctx* c, int p, const gev_couple* cp, size_t nc, uint32_t s, const uint32_t* ms, size_t nms, size_t n, uint8_t* sex) { return gevo_reproduce((Ctx*)c, p, cp, nc, s, ms, nms, n, sex); }
int gev_compute_ad(gev_ctx* c, int p, double* a, double* d, double* ac, double* dc) { return gevo_compute_ad((Ctx*)c, p, a, d, ac, dc); }
int gev_migrate(gev_ctx* c, const gev_move* m, size_t n) { return gevo_migrate((Ctx*)c, m, n); }
int gev_download_haps(gev_ctx* c, int p, int k, size_t r0, size_t nr, uint64_t* bits, size_t w) { return gevo_download_haps((Ctx*)c, p, k, r0, nr, bits, w); }
int gev_download_plink_matrix(gev_ctx* c, int p, int k, size_t i0, size_t n, uint64_t* bits, size_t w) { return gevo_download_plink_matrix((Ctx*)c, p, k, i0, n, bits, w); }
int gev_download_intervals(gev_ctx* c, int p, int k, gev_part* out, uint64_t* off, size_t* n) { return gevo_download_intervals((Ctx*)c, p, k, out, off, n); }
}
