"""Host-side mirror of the reference's driver for the reproduction hot path.

The reference host is C++ (`class Simulation`, reference src/Simulation.h); INTEGRATION.md shows
the C++ patch.  This module is the same seam in Python for tests and bench.py: it keeps what the
reference keeps on the host -- the `ras_glob_seed()` stream, the couples list, pedigree
bookkeeping -- and forwards the five seam functions to the C-ABI under their reference names.
Nothing here computes genotypes, crossovers, mutations or A/D: that is the library's job.
"""
import numpy as np

from .capi import COUPLE_DTYPE, GevLibrary

M31 = 2147483647


class GlobSeedStream:
    """Simulation::ras_glob_seed (reference src/Simulation.cpp:17-21):
    std::uniform_int_distribution<unsigned>(1,1000000) on the minstd_rand0 `glob_generator`
    seeded with --seed (src/Simulation.cpp:75-76).  libstdc++ down-scaling branch:
    scaling = 2147483645/1000000 = 2147, past = 2147000000, reject engine()-1 >= past."""
    SCALING = 2147483645 // 1000000
    PAST = 1000000 * SCALING

    def __init__(self, seed):
        x = int(seed) % M31
        self.x = x if x else 1

    def draw(self, n):
        """next n seeds (vectorised: engine outputs by jump-ahead, rejections filtered in order)"""
        out = np.empty(0, dtype=np.uint32)
        while len(out) < n:
            m = int((n - len(out)) * 1.001) + 16
            k = np.arange(1, m + 1, dtype=np.uint64)
            xs = (self._powmod(k) * np.uint64(self.x)) % np.uint64(M31)
            ret = xs - np.uint64(1)
            ok = ret < np.uint64(self.PAST)
            vals = (ret[ok] // np.uint64(self.SCALING) + np.uint64(1)).astype(np.uint32)
            need = n - len(out)
            if len(vals) >= need:
                # consume engine outputs only up to the one that produced the last seed we keep
                last = np.flatnonzero(ok)[need - 1]
                self.x = int(xs[last])
                out = np.concatenate([out, vals[:need]])
            else:
                self.x = int(xs[-1])
                out = np.concatenate([out, vals])
        return out

    @staticmethod
    def _powmod(k):
        """16807^k mod (2^31-1), elementwise, by square-and-multiply on uint64 vectors"""
        r = np.ones_like(k)
        b = np.uint64(16807)
        e = k.copy()
        m = np.uint64(M31)
        while e.any():
            odd = (e & np.uint64(1)).astype(bool)
            r[odd] = (r[odd] * b) % m
            b = (b * b) % m
            e >>= np.uint64(1)
        return r


class MinstdStream:
    """std::default_random_engine seeded with the `unsigned` expression the reference passes (seed, seed+1, ...),
    with vectorised draws of the two libstdc++ distributions the mating code uses."""

    def __init__(self, seed):
        x = (int(seed) & 0xFFFFFFFF) % M31
        self.x = x if x else 1

    def _outputs(self, m):
        k = np.arange(1, m + 1, dtype=np.uint64)
        return (GlobSeedStream._powmod(k) * np.uint64(self.x)) % np.uint64(M31)

    def u01(self, n):
        """n draws of uniform_real_distribution<double>(0,1) = generate_canonical<double,53> (2 engine calls each)"""
        xs = self._outputs(2 * n)
        self.x = int(xs[-1]) if n else self.x
        lo = (xs[0::2] - np.uint64(1)).astype(np.float64); hi = (xs[1::2] - np.uint64(1)).astype(np.float64)
        r = (lo + hi * 2147483646.0) / float(2147483646 ** 2)
        return np.where(r >= 1.0, np.nextafter(1.0, 0.0), r)

    def uniform_int(self, n, lo, hi):
        """n draws of uniform_int_distribution<unsigned long>(lo,hi), down-scaling branch (hi-lo < 2147483645)"""
        uerange = int(hi) - int(lo) + 1
        assert 0 < uerange <= 2147483645
        scaling = 2147483645 // uerange; past = uerange * scaling
        out = np.empty(0, dtype=np.uint64)
        while len(out) < n:
            m = int((n - len(out)) * (2147483646.0 / past) * 1.01) + 16
            xs = self._outputs(m)
            ret = xs - np.uint64(1)
            ok = ret < np.uint64(past)
            vals = ret[ok] // np.uint64(scaling) + np.uint64(lo)
            need = n - len(out)
            if len(vals) >= need:
                self.x = int(xs[np.flatnonzero(ok)[need - 1]])
                out = np.concatenate([out, vals[:need]])
            else:
                self.x = int(xs[-1])
                out = np.concatenate([out, vals])
        return out


def random_mate(sex, selection_value_func, pop_size, seed):
    """Simulation::random_mate (reference src/Simulation.cpp:2090-2157), bit-exact: `seed` is the
    ras_glob_seed() value drawn at :2092.  Returns the couples list (one offspring per couple)."""
    n_h = len(sex)
    r = MinstdStream(seed).u01(n_h)                                  # generator(seed), one draw per individual (:2112)
    ok = r < np.asarray(selection_value_func, dtype=np.float64)
    pos_male = np.flatnonzero(ok & (sex == 1)); pos_female = np.flatnonzero(ok & (sex == 2))
    if len(pos_male) == 0 or len(pos_female) == 0:
        raise RuntimeError(f"Error: No one can marry, num_males_mate={len(pos_male)}, num_females_mate={len(pos_female)}")
    i_f = MinstdStream(int(seed) + 1).uniform_int(pop_size, 0, len(pos_male) - 1)      # g_uint_f(seed+1), :2132-2144
    i_m = MinstdStream(int(seed) + 2).uniform_int(pop_size, 0, len(pos_female) - 1)    # g_uint_m(seed+2)
    return couples_array(pos_male[i_f.astype(np.int64)], pos_female[i_m.astype(np.int64)])


def couples_array(pos_male, pos_female, num_offspring=1, inbreed=0):
    c = np.zeros(len(pos_male), dtype=COUPLE_DTYPE)
    c["pos_male"], c["pos_female"], c["num_offspring"], c["inbreed"] = pos_male, pos_female, num_offspring, inbreed
    return c


def synthetic_random_mate(sex, pop_size, rng, out=None):
    """Shape of Simulation::random_mate (src/Simulation.cpp:2090-2157): pop_size couples, one
    offspring each, father drawn uniformly from the males and mother from the females.
    Draws come from `rng` (numpy), NOT from the reference's minstd streams: mating is outside
    the hot path; tests that need the reference's couples take them
    from the golden fixtures."""
    males = np.flatnonzero(sex == 1)
    females = np.flatnonzero(sex == 2)
    if len(males) == 0 or len(females) == 0:
        raise RuntimeError("Error: No one can marry")
    if out is None or len(out) != pop_size:
        out = couples_array(np.zeros(pop_size, dtype=np.uint64), np.zeros(pop_size, dtype=np.uint64))
    out["pos_male"] = males[rng.integers(0, len(males), pop_size)]
    out["pos_female"] = females[rng.integers(0, len(females), pop_size)]
    return out


class SyntheticConfig:
    """SURVEY.md section 8(d) synthetic inputs: 1 chromosome of `chrom_bp` base pairs per chr,
    uniform recombination and mutation maps (rows every `map_step` bp), SNPs and CVs on regular /
    random grids, founders ~ Bernoulli(f_i)."""

    def __init__(self, n_ind, n_loci, nchr=1, chrom_bp=100_000_000, map_step=50_000, rec_per_row=5e-4,
                 mut_per_row=5e-4, n_cv=1000, nphen=1, seed=12345, vd=0.0, with_mutation=True):
        self.n_ind, self.n_loci, self.nchr, self.nphen, self.seed = n_ind, n_loci, nchr, nphen, seed
        self.with_mutation = with_mutation
        rs = np.random.RandomState(seed)
        R = chrom_bp // map_step + 1
        self.rmap_bp = (1000 + map_step * np.arange(R)).astype(np.uint64)
        self.rmap_prob = np.r_[0.0, np.full(R - 1, rec_per_row)]
        self.bp_dist = int(map_step)
        self.mut_bp = self.rmap_bp.copy()
        self.mut_rate = np.r_[0.0, np.full(R - 1, mut_per_row)]
        step = max(chrom_bp // n_loci, 1)
        self.snp_pos = (1000 + step * np.arange(n_loci)).astype(np.uint64)
        self.cv = []
        for p in range(nphen):
            per_chr = []
            for c in range(nchr):
                bp = np.sort(rs.choice(np.arange(1000, 1000 + chrom_bp, 100), size=n_cv, replace=False)).astype(np.uint64)
                per_chr.append((bp, rs.randn(n_cv), rs.randn(n_cv) * (0.3 if vd else 0.0)))
            self.cv.append(per_chr)
        self.vd = vd

    def apply_static(self, ctx, pop=0):
        for c in range(self.nchr):
            ctx.set_rmap(pop, c, self.rmap_bp, self.rmap_prob, self.bp_dist)
            if self.with_mutation:
                ctx.set_mutmap(pop, c, self.mut_bp, self.mut_rate)
            ctx.set_snps(pop, c, self.snp_pos)
            for p in range(self.nphen):
                bp, a, d = self.cv[p][c]
                ctx.set_cvs(pop, p, c, bp, a, d, self.vd)


class Simulation:
    """The seam of reference `class Simulation` (src/Simulation.h:64-144) on top of the C-ABI."""

    def __init__(self, ctx, seed, nchr, has_mutation_map):
        self.ctx, self.nchr, self.has_mut = ctx, nchr, has_mutation_map
        self.glob = GlobSeedStream(seed)             # glob_generator.seed(par._seed), :75-76
        self.couples = {}                            # population[ipop]._couples_info
        self.sex = {}

    def ras_glob_seed(self, n=1):
        return self.glob.draw(n)

    def ras_initial_human_gen0(self, ipop, n_people):          # :3000
        self.sex[ipop] = self.ctx.init_gen0(ipop, n_people, int(self.ras_glob_seed()[0]))
        return True

    def reproduce(self, ipop, gen_num=0, seeds=None, n_people=None):   # :2394 (n_people: known offspring count, skips a host pass)
        c = self.couples[ipop]
        if n_people is None:
            n_people = int(c["num_offspring"][c["inbreed"] == 0].sum())
        if seeds is None:                                       # 1 + n_people*nchr ras_glob_seed() draws (:2398, :2500)
            seeds = self.ras_glob_seed(1 + (n_people * self.nchr if self.has_mut else 0))
        self.sex[ipop] = self.ctx.reproduce(ipop, c, int(seeds[0]), seeds[1:] if self.has_mut else None, n_people=n_people)
        return self.sex[ipop]

    def ras_compute_AD(self, ipop, gen_num=0, per_chr=False):   # :2624
        return self.ctx.compute_ad(ipop, per_chr=per_chr)

    def ras_do_migration(self, moves):                          # :877 (row movement; WHO moves is the caller's)
        self.ctx.migrate(moves)
        return True
