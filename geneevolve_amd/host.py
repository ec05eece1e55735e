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
            xs = (self._powers(m) * np.uint64(self.x)) % np.uint64(M31)
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

    _POW = np.array([16807], dtype=np.uint64)          # 16807^k mod (2^31-1), k = 1.. (a constant of the engine, grown on demand)

    @classmethod
    def _powers(cls, m):
        """16807^k mod (2^31-1) for k = 1..m: table doubled by p[j + len] = p[j] * 16807^len (products < 2^62)"""
        p = cls._POW
        while len(p) < m:
            p = np.concatenate([p, (p * p[-1]) % np.uint64(M31)])
        cls._POW = p
        return p[:m]

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
        return (GlobSeedStream._powers(m) * np.uint64(self.x)) % np.uint64(M31)

    def rewind_u01(self, n_unused):
        """undo the last n_unused draws of the most recent u01() batch (the caller over-drew)"""
        xs, n = self._last_u01
        keep = n - n_unused
        self.x = int(xs[2 * keep - 1]) if keep > 0 else self._last_u01_start

    def u01(self, n):
        """n draws of uniform_real_distribution<double>(0,1) = generate_canonical<double,53> (2 engine calls each)"""
        xs = self._outputs(2 * n)
        self._last_u01 = (xs, n); self._last_u01_start = self.x
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


class GlibcRand:
    """glibc srand(seed) / rand() (TYPE_3 additive feedback generator, stdlib/random_r.c): r[i] = 16807*r[i-1] mod (2^31-1)
    for i < 31, then r[i] = r[i-3] + r[i-31] mod 2^32, the first 310 outputs discarded, rand() = r[i] >> 1.
    Host-side stream for the mating code only (std::random_shuffle in assort_mate); the hot path's rand() chains
    are evaluated on the device."""

    def __init__(self, seed):
        x = int(seed) & 0xFFFFFFFF
        if x == 0:
            x = 1
        w = x - (1 << 32) if x >= (1 << 31) else x           # int32_t word = seed: Schrage's method in signed arithmetic
        r = [w & 0xFFFFFFFF]
        for _ in range(30):
            hi = (abs(w) // 127773) * (1 if w >= 0 else -1)      # C division truncates toward zero
            lo = w - hi * 127773
            w = 16807 * lo - 2836 * hi
            if w < 0:
                w += 2147483647
            r.append(w & 0xFFFFFFFF)
        for i in range(31, 34):
            r.append(r[i - 31])
        for i in range(34, 344):
            r.append((r[i - 31] + r[i - 3]) & 0xFFFFFFFF)
        self.r = r[-31:]

    def rand(self):
        v = (self.r[-31] + self.r[-3]) & 0xFFFFFFFF
        self.r.append(v); self.r.pop(0)
        return v >> 1


def random_shuffle(items, rnd):
    """std::random_shuffle of libstdc++ (bits/stl_algo.h): for i = 1..n-1 swap(i, rnd(i+1)), skipped when equal"""
    for i in range(1, len(items)):
        j = rnd(i + 1)
        if i != j:
            items[i], items[j] = items[j], items[i]


def normal_stream(seed, n):
    """n draws of std::normal_distribution<double>(0,1) on default_random_engine(seed): Marsaglia polar method of
    libstdc++ (bits/random.tcc): x, y = 2*U-1 until 0 < r2 <= 1; returns y*mult first, then the saved x*mult."""
    import math
    out = []
    st = MinstdStream(seed)
    while len(out) < n:
        m = max(16, int((n - len(out)) * 0.7) + 16)
        u = st.u01(2 * m)
        x = 2.0 * u[0::2] - 1.0; y = 2.0 * u[1::2] - 1.0
        r2 = x * x + y * y
        for xi, yi, ri in zip(x.tolist(), y.tolist(), r2.tolist()):
            if ri > 1.0 or ri == 0.0:
                continue
            mult = math.sqrt(-2.0 * math.log(ri) / ri)
            out.append(yi * mult); out.append(xi * mult)
    return np.array(out[:n])         # (the engine state past the n-th draw is never observed: one stream per call)


class NormalEngine:
    """one std::default_random_engine shared by successive std::normal_distribution objects (a fresh distribution per call, as
    in reproduce's common-effect loop, reference src/Simulation.cpp:2417-2429): draw(n) leaves the engine exactly where the
    reference's would stand -- after the candidate pair that produced the last value (a saved second value is dropped)"""

    def __init__(self, seed):
        self.stream = MinstdStream(seed)

    def draw(self, n, stddev=1.0):
        import math
        out = []
        while len(out) < n:
            need_pairs = (n - len(out) + 1) // 2
            m = max(8, int(need_pairs * 1.4) + 8)
            u = self.stream.u01(2 * m)
            x = (2.0 * u[0::2] - 1.0).tolist(); y = (2.0 * u[1::2] - 1.0).tolist()
            used = m
            for j in range(m):
                r2 = x[j] * x[j] + y[j] * y[j]
                if r2 > 1.0 or r2 == 0.0:
                    continue
                mult = math.sqrt(-2.0 * math.log(r2) / r2)
                out.append(y[j] * mult)
                if len(out) < n:
                    out.append(x[j] * mult)
                if len(out) >= n:
                    used = j + 1
                    break
            if used < m:
                self.stream.rewind_u01(2 * (m - used))
        return np.array(out[:n]) * stddev + 0.0


def ras_rank(x):
    """CommFunc::ras_rank (src/CommFunc.cpp:152-161): zero-based rank, ties by index (the O(n^2) loop counts, for element k,
    the strictly smaller elements plus the earlier equal ones) = position in a stable ascending sort"""
    order = np.argsort(np.asarray(x, dtype=np.float64), kind="stable")
    r = np.empty(len(order), dtype=np.uint64)
    r[order] = np.arange(len(order), dtype=np.uint64)
    return r


def ras_rpois(n, lam, seed):
    """RasRandomNumber::ras_rpois (src/RasRandomNumber.cpp:56-66) = std::poisson_distribution<int>(lam) on
    default_random_engine(seed); libstdc++ uses the product-of-uniforms method for mean < 12 (bits/random.tcc)."""
    import math
    if lam >= 12:
        raise NotImplementedError("poisson_distribution with mean >= 12 (libstdc++ rejection branch) is not restated")
    thr = math.exp(-lam)
    st = MinstdStream(seed)
    out = np.empty(n, dtype=np.int64)
    buf = st.u01(int(n * (lam + 1) * 1.3) + 64).tolist(); k = 0
    for i in range(n):
        x, prod = 0, 1.0
        while True:
            if k == len(buf):
                buf = st.u01(max(1024, n)).tolist(); k = 0
            prod *= buf[k]; k += 1; x += 1
            if not prod > thr:
                break
        out[i] = x - 1
    return out


class Pedigree:
    """ID fields of reference `class Human` the mating code reads (src/Population.h:126-137); gen 0: everything = i (:3037-3043)"""
    FIELDS = ("ID", "ID_Father", "ID_Mother", "ID_Fathers_Father", "ID_Fathers_Mother", "ID_Mothers_Father", "ID_Mothers_Mother")

    def __init__(self, n):
        i = np.arange(n, dtype=np.int64)
        for f in self.FIELDS:
            setattr(self, f, i.copy())

    def take(self, idx):
        q = Pedigree(0)
        for f in self.FIELDS:
            setattr(q, f, getattr(self, f)[idx])
        return q

    def append(self, other):
        q = Pedigree(0)
        for f in self.FIELDS:
            setattr(q, f, np.concatenate([getattr(self, f), getattr(other, f)]))
        return q

    def offspring(self, pos_father, pos_mother):
        """pedigree of the next generation in enumeration order (src/Simulation.cpp:2473-2479)"""
        q = Pedigree(len(pos_father))
        q.ID_Father = self.ID[pos_father]; q.ID_Fathers_Mother = self.ID_Mother[pos_father]; q.ID_Fathers_Father = self.ID_Father[pos_father]
        q.ID_Mother = self.ID[pos_mother]; q.ID_Mothers_Mother = self.ID_Mother[pos_mother]; q.ID_Mothers_Father = self.ID_Father[pos_mother]
        return q


def assort_mate(sex, selection_value_func, mating_value, ped, pop_size, mat_cor, seeds, mm_percent=0.0,
                avoid_inbreeding=False, offspring_dist="p", rank=None):
    """Simulation::assort_mate (reference src/Simulation.cpp:2167-2360).  `seeds` = the ras_glob_seed() values in draw order:
    srand seed (:2170), selection generator (:2173), ras_mvnorm (:2268), and -- Poisson offspring numbers only -- ras_rpois
    (:2330).  Returns the couples list.  std::sort's order among EQUAL mating values is unspecified in the reference; a
    stable sort is used here (doubled entries of one individual are identical, so only exact ties between different
    individuals could differ)."""
    n_h = len(sex)
    rnd = GlibcRand(seeds[0])
    u = MinstdStream(seeds[1]).u01(2 * n_h).tolist(); k = 0
    males, females = [], []
    svf = np.asarray(selection_value_func, dtype=np.float64).tolist()
    for i in range(n_h):
        r = u[k]; k += 1
        if r < svf[i]:
            lst = males if sex[i] == 1 else (females if sex[i] == 2 else None)
            if lst is not None:
                lst.append(i)
                r = u[k]; k += 1
                if r < mm_percent:                      # a second spouse (:2196-2198)
                    lst.append(i)
    couples = min(len(males), len(females))
    if couples == 0:
        raise RuntimeError(f"Error: couples=0, num_males_mate={len(males)}, num_females_mate={len(females)}")
    if len(males) > len(females):                       # drop the surplus after a shuffle (:2232-2245)
        random_shuffle(males, lambda m: rnd.rand() % m); males = males[len(males) - len(females):]
    elif len(males) < len(females):
        random_shuffle(females, lambda m: rnd.rand() % m); females = females[len(females) - len(males):]
    mv = np.asarray(mating_value, dtype=np.float64)
    males = np.asarray(males, dtype=np.int64); females = np.asarray(females, dtype=np.int64)
    males = males[np.argsort(mv[males], kind="stable")]; females = females[np.argsort(mv[females], kind="stable")]
    n2 = couples
    # template: ras_mvnorm(n, 0, [[1,c],[c,1]]) = z * chol, chol = LLT upper factor [[1, c], [0, sqrt(1-c*c)]] (:2268, RasRandomNumber.cpp:15-51)
    z = normal_stream(seeds[2], 2 * n2).reshape(n2, 2)
    c = float(mat_cor)
    x11 = 1.0 - (c / 1.0) * (c / 1.0)                  # Eigen's unblocked LLT stops at a non-positive pivot and leaves A(1,1) = 1 in place
    u00, u01_, u11 = 1.0, c / 1.0, (float(np.sqrt(x11)) if x11 > 0 else 1.0)
    t1 = (0.0 + z[:, 0] * u00) + z[:, 1] * 0.0
    t2 = (0.0 + z[:, 0] * u01_) + z[:, 1] * u11
    rank = rank or ras_rank                            # CommFunc::ras_rank: `rank=ctx.rank_f64` evaluates it on the device
    rank_m = rank(t1).astype(np.int64); rank_f = rank(t2).astype(np.int64)
    pos_m = males[rank_m]; pos_f = females[rank_f]
    inbreed = np.zeros(n2, dtype=np.int32)
    if avoid_inbreeding:                                # :2306-2322
        P = ped
        sib = P.ID_Father[pos_m] == P.ID_Father[pos_f]
        cousin = ((P.ID_Fathers_Father[pos_m] == P.ID_Fathers_Father[pos_f]) | (P.ID_Fathers_Father[pos_m] == P.ID_Mothers_Father[pos_f]) |
                  (P.ID_Mothers_Father[pos_m] == P.ID_Fathers_Father[pos_f]) | (P.ID_Mothers_Father[pos_m] == P.ID_Mothers_Father[pos_f]) |
                  (P.ID_Fathers_Mother[pos_m] == P.ID_Fathers_Mother[pos_f]) | (P.ID_Fathers_Mother[pos_m] == P.ID_Mothers_Mother[pos_f]) |
                  (P.ID_Mothers_Mother[pos_m] == P.ID_Fathers_Mother[pos_f]) | (P.ID_Mothers_Mother[pos_m] == P.ID_Mothers_Mother[pos_f]))
        inbreed = (sib | cousin).astype(np.int32)
        can_marry = []                                  # pos_couple_can_marry is only filled in the else branch (:2324-2328)
    else:
        can_marry = list(range(n2))
    n_inbreed = int(inbreed.sum())
    if offspring_dist in ("p", "P"):
        lam = float(pop_size) / (n2 - n_inbreed)
        num = ras_rpois(n2, lam, seeds[3])
    elif offspring_dist in ("f", "F"):
        nf = int(np.floor(float(pop_size) / (n2 - n_inbreed)))
        num = np.full(n2, nf, dtype=np.int64)
        remain = int(pop_size) - nf * (n2 - n_inbreed)
        random_shuffle(can_marry, lambda m: rnd.rand() % m)
        for i in range(remain):
            num[can_marry[i]] += 1                      # (with --avoid_inbreeding the reference indexes an empty vector here)
    else:
        num = np.zeros(n2, dtype=np.int64)
    return couples_array(pos_m, pos_f, num, inbreed)


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


def comm_mean(x):
    """CommFunc::mean (src/CommFunc.cpp:38-45): sequential sum / n"""
    s = 0.0
    for v in np.asarray(x, dtype=np.float64).tolist():
        s += v
    return s / len(x)


def comm_var(x):
    """CommFunc::var (src/CommFunc.cpp:57-68): two passes, sequential sums, n-1"""
    xs = np.asarray(x, dtype=np.float64).tolist()
    if len(xs) <= 1:
        return 0.0
    mu = 0.0
    for v in xs:
        mu += v
    mu /= len(xs)
    s2 = 0.0
    for v in xs:
        s2 += (v - mu) * (v - mu)
    return s2 / (len(xs) - 1)


def ras_combined_variance(phens, gamma, a):
    """Simulation::ras_combined_variance (reference src/Simulation.cpp:3254-3282): s2y - (1 + gamma) * s2x over the phenotype
    values of ALL populations in population order, y = x + a * (2*ipop/(n_pop-1) - 1) -- the factor is an INTEGER expression
    there (-1, 0 or 1; two populations: -1 and +1)"""
    n_pop = len(phens)
    x, y = [], []
    for ipop, ph in enumerate(phens):
        bi = a * float((2 * ipop) // (n_pop - 1) - 1)
        v = np.asarray(ph, dtype=np.float64)
        x.append(v); y.append(v + bi)
    return comm_var(np.concatenate(y)) - (1 + gamma) * comm_var(np.concatenate(x))


def environmental_effects_specific_to_each_population(phens, gamma, x0=10.0, precision=1e-4):
    """Simulation::sim_environmental_effects_specific_to_each_population (:3345-3382) for one phenotype: solves
    ras_combined_variance(a) = 0 by Simulation::NewtonRaphson (:44-63; derivative by central differences, dx = 0.001, :32-36;
    stops on |f(x1)| < precision) and adds a * (2*ipop/(n_pop-1) - 1) to every phenotype value of population ipop (:3285-3297).
    phens: one array per population, changed in place; returns a (0.0 and nothing changed when gamma == 0)."""
    if gamma == 0:
        return 0.0
    f = lambda a: ras_combined_variance(phens, gamma, a)
    dx = 0.001
    x = x0
    for _ in range(1000):                                       # (the reference recurses without a bound)
        fx0 = f(x)
        x1 = x - fx0 / ((f(x + dx) - f(x - dx)) / (2 * dx))
        x = x1
        if abs(f(x1)) < precision:
            break
    else:
        raise RuntimeError("environmental effects: Newton-Raphson did not converge")
    n_pop = len(phens)
    for ipop in range(n_pop):
        phens[ipop] += x * float((2 * ipop) // (n_pop - 1) - 1)
    return x


def selection_func(kind, p1, p2, z):
    """Simulation::ras_selection_func (reference src/Simulation.cpp:3386-3428) for generations >= 1 (generation 0: 1 for all);
    logit and thr; z = selection value standardised to generation 0 (ras_compute_mating_value_selection_value, :3300-3342)"""
    import math
    if kind == "logit":
        out = []
        for v in np.asarray(z, dtype=np.float64).tolist():
            y = math.exp(p1 + p2 * v)
            out.append(y / (1 + y))
        return np.array(out)
    if kind == "thr":
        return np.where(np.asarray(z) <= p2, p1, 1.0)
    if kind == "probit":                      # CommFunc::NormalCDF (src/CommFunc.cpp:257-262): .5*(1+erf((x-mu)/(sqrt(2)*sigma)))
        return np.array([.5 * (1 + math.erf((v - p1) / (math.sqrt(2) * p2))) for v in np.asarray(z, dtype=np.float64).tolist()])
    if kind == "stab":                        # CommFunc::NormalPDF (:266-270): 1/(sqrt(2*pi)*sigma) * exp(-0.5*pow((x-mu)/sigma,2))
        pi = 3.1415926                         # the reference's own constant (src/CommFunc.cpp:4)
        return np.array([1 / (math.sqrt(2.0 * pi) * p2) * math.exp(-0.5 * math.pow((v - p1) / p2, 2)) for v in np.asarray(z, dtype=np.float64).tolist()])
    raise NotImplementedError(kind)


class SampleWithoutReplacement:
    """RasRandomNumber::ras_SampleWithoutReplacement (reference src/RasRandomNumber.cpp:90-120): Knuth's selection sampling on
    a `static` default_random_engine -- it is seeded by the seed of the FIRST call of the process and every later call
    continues that stream (its seed argument is ignored).  One instance = that process-wide static."""

    def __init__(self):
        self.stream = None

    def __call__(self, population_size, sample_size, seed):
        if self.stream is None:
            self.stream = MinstdStream(seed)
        n, N = int(sample_size), int(population_size)
        out, t, m = [], 0, 0
        buf, k = [], 0
        while m < n:
            if k == len(buf):
                buf = self.stream.u01(max(N - t, 16)).tolist(); k = 0      # over-drawn; the unused tail is given back below
            u = buf[k]; k += 1
            if (N - t) * u >= n - m:
                t += 1
            else:
                out.append(t); t += 1; m += 1
        # give back the unused part of the last batch: the engine must stand exactly after the draws consumed
        if k < len(buf):
            unused = len(buf) - k
            self.stream.rewind_u01(unused)
        return np.array(out, dtype=np.int64)


def ras_do_migration(pop_sizes, migration_row, glob_draw, sampler):
    """WHO moves in Simulation::ras_do_migration (reference src/Simulation.cpp:877-937): counts round(m_ij * n_i) (:909), one
    sample per origin over all its emigrants (:921, one ras_glob_seed() each), sorted descending (:922), handed to the
    destinations in ascending j by the reference's loop `while (k < num_move[i][j])` (:930-935, k runs on across j -- kept
    literally).  Returns the moves (src_pop, src_pos, dst_pop) in the append order of :971-981."""
    npop = len(pop_sizes)
    mat = np.asarray(migration_row, dtype=np.float64).reshape(npop, npop)
    num_move = [[0 if i == j else int(np.floor(mat[i][j] * float(pop_sizes[i]) + 0.5)) for j in range(npop)] for i in range(npop)]
    camp = [[[] for _ in range(npop)] for _ in range(npop)]
    for i in range(npop):
        s = sum(num_move[i])
        sample = sorted(sampler(pop_sizes[i], s, int(glob_draw(1)[0])).tolist(), reverse=True)
        k = 0
        for j in range(npop):
            if i == j:
                continue
            lst = [None] * num_move[i][j]; it = 0
            while k < num_move[i][j]:
                lst[it] = sample[k]; k += 1; it += 1
            camp[i][j] = lst
    moves = []
    for i in range(npop):
        for j in range(npop):
            if i != j:
                moves += [(i, int(pos), j) for pos in camp[i][j] if pos is not None]
    return moves


def ras_save_human_info(ped, sex, per_phen, mating_value, selection_value, selection_value_func):
    """bytes of Population::ras_save_human_info's file (reference src/Population.cpp:510-568).  per_phen: one dict per phenotype
    with the vectors additive, dominance, bv, common_sibling, e_noise, parental_effect, phen.  Doubles go through the
    default ostream format = printf %g with 6 significant digits; ids are printed 1-based."""
    hdr = ["ID", "ID_Father", "ID_Mother", "ID_Fathers_Father", "ID_Fathers_Mother", "ID_Mothers_Father", "ID_Mothers_Mother", "sex"]
    for j in range(len(per_phen)):
        hdr += [f"ph{j+1}_{c}" for c in "ADGCEFP"]
    hdr += ["MV", "SV", "SV_f"]
    ids = np.stack([getattr(ped, f) for f in ("ID", "ID_Father", "ID_Mother", "ID_Fathers_Father", "ID_Fathers_Mother", "ID_Mothers_Father", "ID_Mothers_Mother")], axis=1) + 1
    cols = []
    for d in per_phen:
        cols += [d["additive"], d["dominance"], d["bv"], d["common_sibling"], d["e_noise"], d["parental_effect"], d["phen"]]
    cols += [mating_value, selection_value, selection_value_func]
    cols = np.stack([np.asarray(c, dtype=np.float64) for c in cols], axis=1)
    lines = [" ".join(hdr)]
    for i in range(len(sex)):
        lines.append(" ".join([str(int(v)) for v in ids[i]] + [str(int(sex[i]))] + ["%g" % v for v in cols[i]]))
    return ("\n".join(lines) + "\n").encode()


class Simulation:
    """The seam of reference `class Simulation` (src/Simulation.h:64-144) on top of the C-ABI."""

    def __init__(self, ctx, seed, nchr, has_mutation_map, track_pedigree=False):
        self.ctx, self.nchr, self.has_mut, self.track_pedigree = ctx, nchr, has_mutation_map, track_pedigree
        self.glob = GlobSeedStream(seed)             # glob_generator.seed(par._seed), :75-76
        self.couples = {}                            # population[ipop]._couples_info
        self.sex = {}
        self.ped = {}                                # pedigree ids of the current generation (host bookkeeping, :2473-2479)

    def ras_glob_seed(self, n=1):
        return self.glob.draw(n)

    def ras_initial_human_gen0(self, ipop, n_people):          # :3000
        self.sex[ipop] = self.ctx.init_gen0(ipop, n_people, int(self.ras_glob_seed()[0]))
        self.ped[ipop] = Pedigree(n_people)
        return True

    def random_mate(self, ipop, selection_value_func, pop_size):           # :2090 (one ras_glob_seed() draw, :2092)
        self.couples[ipop] = random_mate(self.sex[ipop], selection_value_func, pop_size, int(self.ras_glob_seed()[0]))
        return True

    def random_mate_device(self, ipop, selection_value_func, pop_size, want_couples=True):
        """the same step on the library's side (gev_random_mate): the couples stay there for reproduce(); selection_value_func = None
        when every value is 1"""
        self.couples[ipop], self.num_males_mate, self.num_females_mate = self.ctx.random_mate(ipop, int(self.ras_glob_seed()[0]), selection_value_func, pop_size, want_couples)
        self._device_couples = (ipop, pop_size)
        return True

    def next_generation_rm(self, ipop, pop_size, selection_value_func=None, want_couples=False):
        """random_mate -> reproduce -> ras_compute_AD of sim_next_generation (:1907-1935) as one library call pair; the
        ras_glob_seed() draws of the three are made by the library from glob_generator's state, which is stored back"""
        self.ctx.generation_begin(ipop, self.glob.x, pop_size, selection_value_func)
        r = self.ctx.generation_end(want_couples=want_couples)
        self.glob.x = int(r["glob_state"])
        self.last_seed_reproduce = int(r["seed_reproduce"])
        self.sex[ipop] = r["sex"]
        if want_couples:
            self.couples[ipop] = r["couples"]
            if self.track_pedigree and ipop in self.ped:
                c = r["couples"]
                self.ped[ipop] = self.ped[ipop].offspring(c["pos_male"].astype(np.int64), c["pos_female"].astype(np.int64))
        return r

    def assort_mate(self, ipop, selection_value_func, mating_value, pop_size, mat_cor, mm_percent=0.0,
                    avoid_inbreeding=False, offspring_dist="p", rank=None):   # :2167 (3 draws, +1 for Poisson offspring numbers)
        seeds = [int(x) for x in self.ras_glob_seed(4 if offspring_dist in ("p", "P") else 3)]
        self.couples[ipop] = assort_mate(self.sex[ipop], selection_value_func, mating_value, self.ped[ipop], pop_size, mat_cor, seeds,
                                         mm_percent, avoid_inbreeding, offspring_dist, rank=rank or self.ctx.rank_f64)
        return True

    def reproduce(self, ipop, gen_num=0, seeds=None, n_people=None):   # :2394 (n_people: known offspring count, skips a host pass)
        c = self.couples[ipop]
        dev = getattr(self, "_device_couples", None)
        self._device_couples = None
        if dev is not None and dev[0] == ipop and n_people is None:
            n_people = dev[1]                                     # one child per couple (:2149)
        if n_people is None:
            n_people = int(c["num_offspring"][c["inbreed"] == 0].sum())
        if seeds is None:                                       # 1 + n_people*nchr ras_glob_seed() draws (:2398, :2500)
            seeds = self.ras_glob_seed(1 + (n_people * self.nchr if self.has_mut else 0))
        self.last_seed_reproduce = int(seeds[0])
        self.sex[ipop] = self.ctx.reproduce(ipop, None if (dev is not None and dev[0] == ipop) else c, int(seeds[0]), seeds[1:] if self.has_mut else None, n_people=n_people)
        if self.track_pedigree and ipop in self.ped:            # enumeration order of the couple loop (:2433-2443)
            ok = c["inbreed"] == 0
            rep = c["num_offspring"][ok].astype(np.int64)
            self.ped[ipop] = self.ped[ipop].offspring(np.repeat(c["pos_male"][ok].astype(np.int64), rep), np.repeat(c["pos_female"][ok].astype(np.int64), rep))
        return self.sex[ipop]

    def presample(self, ipop, seeds, n_people):
        """the seeds the next reproduce() will be given (1 + n_people*nchr ras_glob_seed() values, drawn in the reference's
        order) are known before the couples are: let the GPU sample while the host mates"""
        self.ctx.presample(ipop, int(seeds[0]), seeds[1:] if self.has_mut else None, n_people)

    def common_sibling(self, ipop, vc):
        """val_common of the reproduce() call just made (:2417-2429, :2481-2484): per phenotype with vc > 0 one N(0, sqrt(vc)) per
        COUPLE (inbred ones included) from default_random_engine(seed + 1), handed to every child of the couple"""
        c = self.couples[ipop]
        eng = NormalEngine(self.last_seed_reproduce + 1)
        ok = c["inbreed"] == 0
        rep = np.where(ok, c["num_offspring"], 0).astype(np.int64)
        out = []
        for v in vc:
            val = eng.draw(len(c), float(np.sqrt(v))) if v > 0 else np.zeros(len(c))
            out.append(np.repeat(val, rep))
        return out

    def ras_compute_AD(self, ipop, gen_num=0, per_chr=False):   # :2624
        return self.ctx.compute_ad(ipop, per_chr=per_chr)

    def ras_do_migration(self, moves):                          # :877 (row movement; WHO moves is the caller's)
        self.ctx.migrate(moves)
        return True
