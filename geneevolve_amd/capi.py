"""ctypes binding of the C-ABI declared in include/geneevolve_amd.h.

This is the stub a Python host would add; the C++ host binding is shown in INTEGRATION.md.
`GevLibrary(path, prefix)` binds any shared library that exports the ABI under `prefix`
(the product library exports `gev_*`).  No compute happens in Python: every method is one
C call on caller-owned numpy buffers.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.environ.get("GEV_LIBRARY_PATH") or os.path.join(_HERE, "csrc", "libgeneevolve_amd.so")   # (the variable: A/B builds of the same library)


class GevError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


class gev_couple(C.Structure):
    _fields_ = [("pos_male", C.c_uint64), ("pos_female", C.c_uint64), ("inbreed", C.c_int32), ("num_offspring", C.c_int32)]


class gev_part(C.Structure):
    _fields_ = [("st", C.c_uint64), ("en", C.c_uint64), ("hap_index", C.c_uint64), ("root_population", C.c_int32), ("reserved", C.c_int32)]


class gev_move(C.Structure):
    _fields_ = [("src_pop", C.c_int32), ("dst_pop", C.c_int32), ("src_pos", C.c_uint64)]


class gev_gef_params(C.Structure):
    _fields_ = [("va", C.c_double), ("vd", C.c_double), ("ve", C.c_double), ("vf", C.c_double), ("beta", C.c_double),
                ("s2_a_gen0", C.c_double), ("s2_d_gen0", C.c_double), ("gen_num", C.c_int32), ("reserved", C.c_int32)]


class gev_generation_result(C.Structure):
    _fields_ = [("glob_state", C.c_uint32), ("seed_mate", C.c_uint32), ("seed_reproduce", C.c_uint32), ("reserved", C.c_uint32),
                ("num_males_mate", C.c_uint64), ("num_females_mate", C.c_uint64)]


COUPLE_DTYPE = np.dtype([("pos_male", "<u8"), ("pos_female", "<u8"), ("inbreed", "<i4"), ("num_offspring", "<i4")])
PART_DTYPE = np.dtype([("st", "<u8"), ("en", "<u8"), ("hap_index", "<u8"), ("root_population", "<i4"), ("reserved", "<i4")])
MOVE_DTYPE = np.dtype([("src_pop", "<i4"), ("dst_pop", "<i4"), ("src_pos", "<u8")])

# every symbol include/geneevolve_amd.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "last_error", "version", "create", "destroy", "set_rmap", "set_mutmap", "set_snps", "set_cvs",
    "upload_founders", "upload_cv_founders", "synth_founders", "synth_cv_founders", "init_gen0",
    "reproduce", "presample", "compute_ad", "scale_ad_compute_gef", "set_ad", "get_cv_freq", "migrate", "export_size", "export_rows", "remove_rows",
    "import_rows", "download_haps", "download_snp_major", "format_hap_text", "format_bed", "format_vcf_gt", "rank_f64", "download_plink_matrix", "format_ped_text", "download_cv", "download_intervals", "download_mutations",
    "pop_size", "plane_ptr", "reserve", "set_chr_active", "set_dense_state", "materialize", "materialize_pops", "materialize_bed", "stream", "last_reproduce_ms", "set_track_intervals", "set_stitch_mode", "sync", "timing_totals", "stitch_totals", "reproduce_begin", "reproduce_end", "presample_sex", "set_overlap",
    "random_mate", "glob_seeds", "generation_begin", "generation_end", "set_generation_chain", "redo_count", "list_stats", "compute_ad_device", "ad_finish_device",
    "upload_founder_panel", "synth_founder_panel", "set_migrant_rows",
    "dbg_verify_planes", "dbg_prefilter_sweep", "dbg_tables", "dbg_threshold", "dbg_canonical", "dbg_rand", "dbg_sim_loc_rec",
]


def _p(a, t=None):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def _arr(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def words_for(nbits):
    return (int(nbits) + 63) // 64


def pack_rows(bits01):
    """uint8 [rows][L] of 0/1 -> uint64 [rows][ceil(L/64)] in the C-ABI packing."""
    b = np.asarray(bits01, dtype=np.uint8)
    pad = (-b.shape[1]) % 64
    if pad:
        b = np.concatenate([b, np.zeros((b.shape[0], pad), dtype=np.uint8)], axis=1)
    return np.ascontiguousarray(np.packbits(b, axis=1, bitorder="little")).view(np.uint64)


def bytes_to_words(packed_u8, L):
    """np.packbits(..., bitorder='little') bytes [rows][ceil(L/8)] -> uint64 words."""
    rows = packed_u8.shape[0]
    w = words_for(L)
    out = np.zeros((rows, w * 8), dtype=np.uint8)
    out[:, :packed_u8.shape[1]] = packed_u8
    return out.view(np.uint64)


def unpack_rows(words, L):
    return np.unpackbits(np.ascontiguousarray(words).view(np.uint8), axis=1, bitorder="little")[:, :L]


class GevLibrary:
    def __init__(self, path=DEFAULT_LIB, prefix="gev_"):
        if not os.path.exists(path):
            raise GevError(-3, f"native library not found: {path} (build it: python -c 'import __graft_entry__ as g; g.build()')")
        self.path, self.prefix = path, prefix
        if prefix == "gev_" and "torch" not in sys.modules:
            # This image holds two HIP runtimes: /opt/rocm's, which the library links, and the one PyTorch-ROCm bundles.  A process
            # that uses both (the bench and the multi-rank tests hand torch's device buffers to the library) must load torch's first:
            # with the library loaded before `import torch`, the runtime initialised second finds no device (measured on the
            # MI355X box: build() followed by smoke() in one process).  Hosts without torch (C++, the bound program) are not affected.
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        self.lib = C.CDLL(path)
        self.has_device_arg = prefix == "gev_"
        f = self._f("last_error"); f.restype = C.c_char_p; f.argtypes = []

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    def exports(self, name):
        try:
            self._f(name)
            return True
        except AttributeError:
            return False

    def last_error(self):
        s = self._f("last_error")()
        return s.decode() if s else ""

    def check(self, rc):
        if rc != 0:
            raise GevError(rc, self.last_error())

    def create(self, n_pop, nchr, nphen, device=-1):
        return GevContext(self, n_pop, nchr, nphen, device)


class GevContext:
    """One simulation context = one GPU (reference: one Simulation object)."""

    def __init__(self, lib, n_pop, nchr, nphen, device=-1):
        self.L, self.n_pop, self.nchr, self.nphen = lib, n_pop, nchr, nphen
        h = C.c_void_p()
        if lib.has_device_arg:
            rc = lib._f("create")(C.byref(h), C.c_int(device), C.c_int(n_pop), C.c_int(nchr), C.c_int(nphen))
        else:
            rc = lib._f("create")(C.byref(h), C.c_int(n_pop), C.c_int(nchr), C.c_int(nphen))
        lib.check(rc)
        self.h = h
        self._nsnp = {}
        self._ncv = {}

    def close(self):
        if self.h:
            f = self.L._f("destroy"); f.restype = None
            f(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, name, *args):
        self.L.check(self.L._f(name)(self.h, *args))

    # ---- static inputs
    def set_rmap(self, pop, chr, bp, prob, bp_dist):
        bp = _arr(bp, np.uint64); prob = _arr(prob, np.float64)
        self._call("set_rmap", C.c_int(pop), C.c_int(chr), _p(bp), _p(prob), C.c_size_t(len(bp)), C.c_uint64(int(bp_dist)))

    def set_mutmap(self, pop, chr, bp, rate):
        bp = _arr(bp, np.uint64); rate = _arr(rate, np.float64)
        self._call("set_mutmap", C.c_int(pop), C.c_int(chr), _p(bp), _p(rate), C.c_size_t(len(bp)))

    def set_snps(self, pop, chr, pos):
        pos = _arr(pos, np.uint64)
        self._nsnp[(pop, chr)] = len(pos)
        self._call("set_snps", C.c_int(pop), C.c_int(chr), _p(pos), C.c_size_t(len(pos)))

    def set_cvs(self, pop, phen, chr, bp, a, d, vd):
        bp = _arr(bp, np.uint64); a = _arr(a, np.float64); d = _arr(d, np.float64)
        self._ncv[(pop, phen, chr)] = len(bp)
        self._call("set_cvs", C.c_int(pop), C.c_int(phen), C.c_int(chr), _p(bp), _p(a), _p(d), C.c_size_t(len(bp)), C.c_double(vd))

    def upload_founders(self, pop, chr, words, L):
        words = _arr(words, np.uint64)
        self._call("upload_founders", C.c_int(pop), C.c_int(chr), _p(words), C.c_size_t(words.shape[1]), C.c_size_t(words.shape[0]), C.c_size_t(L))

    def upload_cv_founders(self, pop, phen, chr, words, ncv):
        words = _arr(words, np.uint64)
        self._call("upload_cv_founders", C.c_int(pop), C.c_int(phen), C.c_int(chr), _p(words), C.c_size_t(words.shape[1]), C.c_size_t(words.shape[0]), C.c_size_t(ncv))

    def synth_founders(self, pop, chr, nhap, seed):
        self._call("synth_founders", C.c_int(pop), C.c_int(chr), C.c_size_t(nhap), C.c_uint64(seed))

    def upload_founder_panel(self, pop, chr, words, L):
        words = _arr(words, np.uint64)
        self._call("upload_founder_panel", C.c_int(pop), C.c_int(chr), _p(words), C.c_size_t(words.shape[1]), C.c_size_t(words.shape[0]), C.c_size_t(L))

    def synth_founder_panel(self, pop, chr, nhap, seed):
        """read-only founder panel of ROOT population `pop` (immigrants' rows are rebuilt from it, set_migrant_rows(False))"""
        self._call("synth_founder_panel", C.c_int(pop), C.c_int(chr), C.c_size_t(nhap), C.c_uint64(seed))

    def set_migrant_rows(self, on):
        """False: export_rows packs no genotype rows, import_rows rebuilds them from the founder panels"""
        self._call("set_migrant_rows", C.c_int(1 if on else 0))

    def synth_cv_founders(self, pop, phen, chr, nhap, seed):
        self._call("synth_cv_founders", C.c_int(pop), C.c_int(phen), C.c_int(chr), C.c_size_t(nhap), C.c_uint64(seed))

    # ---- generation steps
    def init_gen0(self, pop, n_people, seed_gen0, want_sex=True):
        sex = np.zeros(n_people, dtype=np.uint8) if want_sex else None
        self._call("init_gen0", C.c_int(pop), C.c_size_t(n_people), C.c_uint32(int(seed_gen0)), _p(sex))
        return sex

    def reproduce(self, pop, couples, seed_reproduce, mut_seeds=None, want_sex=True, n_people=None):
        """couples: int array [n,4] (pos_male,pos_female,inbreed,num_offspring) or COUPLE_DTYPE array;
        None: the couples the preceding random_mate() left in the library (n_people = its pop_size)"""
        if couples is None:
            ms = None if mut_seeds is None else _arr(mut_seeds, np.uint32)
            sex = np.zeros(n_people, dtype=np.uint8) if want_sex else None
            self._call("reproduce", C.c_int(pop), None, C.c_size_t(0), C.c_uint32(int(seed_reproduce)),
                       _p(ms), C.c_size_t(0 if ms is None else len(ms)), C.c_size_t(n_people), _p(sex))
            return sex
        if couples.dtype != COUPLE_DTYPE:
            c = np.zeros(len(couples), dtype=COUPLE_DTYPE)
            c["pos_male"], c["pos_female"], c["inbreed"], c["num_offspring"] = couples[:, 0], couples[:, 1], couples[:, 2], couples[:, 3]
            couples = c
        couples = np.ascontiguousarray(couples)
        if n_people is None:
            n_people = int(couples["num_offspring"][couples["inbreed"] == 0].sum())
        ms = None if mut_seeds is None else _arr(mut_seeds, np.uint32)
        sex = np.zeros(n_people, dtype=np.uint8) if want_sex else None
        self._call("reproduce", C.c_int(pop), _p(couples), C.c_size_t(len(couples)), C.c_uint32(int(seed_reproduce)),
                   _p(ms), C.c_size_t(0 if ms is None else len(ms)), C.c_size_t(n_people), _p(sex))
        return sex

    def reproduce_begin(self, pop, couples, seed_reproduce, mut_seeds=None, n_people=None):
        """first half of reproduce(): stage the inputs and enqueue the generation's device work (returns without waiting)"""
        couples = np.ascontiguousarray(couples)
        assert couples.dtype == COUPLE_DTYPE
        if n_people is None:
            n_people = int(couples["num_offspring"][couples["inbreed"] == 0].sum())
        ms = None if mut_seeds is None else _arr(mut_seeds, np.uint32)
        self._call("reproduce_begin", C.c_int(pop), _p(couples), C.c_size_t(len(couples)), C.c_uint32(int(seed_reproduce)),
                   _p(ms), C.c_size_t(0 if ms is None else len(ms)), C.c_size_t(n_people))
        self._pending_people = n_people

    def reproduce_end(self, want_sex=True):
        """second half: wait, redo with larger buffers if needed, publish the generation; returns the sexes"""
        sex = np.zeros(self._pending_people, dtype=np.uint8) if want_sex else None
        self._call("reproduce_end", _p(sex))
        return sex

    def presample_sex(self, pop, n_people):
        """sexes of the generation whose sampling presample() has enqueued (waits for the sampling kernels only)"""
        sex = np.zeros(n_people, dtype=np.uint8)
        self._call("presample_sex", C.c_int(pop), _p(sex), C.c_size_t(n_people))
        return sex

    def presample(self, pop, seed_reproduce, mut_seeds, n_people):
        """head start for the next reproduce() with the same seeds / n_people (returns without waiting)"""
        ms = None if mut_seeds is None else _arr(mut_seeds, np.uint32)
        self._call("presample", C.c_int(pop), C.c_uint32(int(seed_reproduce)), _p(ms), C.c_size_t(0 if ms is None else len(ms)), C.c_size_t(n_people))

    def random_mate(self, pop, seed, selection_value_func, pop_size, want_couples=True):
        """Simulation::random_mate on the library's side -> (couples or None, num_males_mate, num_females_mate)"""
        svf = None if selection_value_func is None else _arr(selection_value_func, np.float64)
        couples = np.zeros(pop_size, dtype=COUPLE_DTYPE) if want_couples else None
        nm, nf = C.c_size_t(), C.c_size_t()
        self._call("random_mate", C.c_int(pop), C.c_uint32(int(seed)), _p(svf), C.c_size_t(pop_size), _p(couples), C.byref(nm), C.byref(nf))
        return couples, nm.value, nf.value

    def glob_seeds(self, engine_state, n, want=True):
        """n Simulation::ras_glob_seed() values from glob_generator's state -> (values or None, state after)"""
        st = C.c_uint32(int(engine_state))
        out = np.zeros(n, dtype=np.uint32) if want else None
        self._call("glob_seeds", C.byref(st), C.c_size_t(n), _p(out))
        return out, st.value

    def generation_begin(self, pop, glob_state, pop_size, selection_value_func=None):
        """random_mate -> reproduce -> ras_compute_AD of one generation, enqueued as one unit (returns without waiting)"""
        svf = None if selection_value_func is None else _arr(selection_value_func, np.float64)
        self._call("generation_begin", C.c_int(pop), C.c_uint32(int(glob_state)), C.c_size_t(pop_size), _p(svf))
        self._pending_people = pop_size

    def generation_end(self, want_couples=False, want_sex=True):
        """-> dict(glob_state, seed_mate, seed_reproduce, num_males_mate, num_females_mate, couples, sex)"""
        res = gev_generation_result()
        couples = np.zeros(self._pending_people, dtype=COUPLE_DTYPE) if want_couples else None
        sex = np.zeros(self._pending_people, dtype=np.uint8) if want_sex else None
        self._call("generation_end", C.byref(res), _p(couples), _p(sex))
        return {"glob_state": res.glob_state, "seed_mate": res.seed_mate, "seed_reproduce": res.seed_reproduce,
                "num_males_mate": res.num_males_mate, "num_females_mate": res.num_females_mate, "couples": couples, "sex": sex}

    def set_generation_chain(self, draws_between):
        """ras_glob_seed() draws the host makes itself between two generation_begin() calls (None: unknown, no head start)"""
        self._call("set_generation_chain", C.c_int(-1 if draws_between is None else int(draws_between)))

    def compute_ad_device(self, pop):
        """gev_compute_ad_device: (device pointer of the per-chromosome additive array, ... of the dominance array, length in doubles)"""
        a, d, n = C.c_void_p(), C.c_void_p(), C.c_size_t()
        self._call("compute_ad_device", C.c_int(pop), C.byref(a), C.byref(d), C.byref(n))
        return int(a.value), int(d.value), int(n.value)

    def ad_finish_device(self, pop, per_chr=True):
        """gev_ad_finish_device -> (additive, dominance, additive_chr, dominance_chr) after the caller's in-place all-reduce"""
        n = self.pop_size(pop)
        add = np.zeros((n, self.nphen)); dom = np.zeros((n, self.nphen))
        addc = np.zeros((n, self.nchr, self.nphen)) if per_chr else None; domc = np.zeros((n, self.nchr, self.nphen)) if per_chr else None
        self._call("ad_finish_device", C.c_int(pop), _p(add), _p(dom), _p(addc) if per_chr else None, _p(domc) if per_chr else None)
        return add, dom, addc, domc

    def list_stats(self, pop=0, chrom=0):
        """gev_list_stats as a dict"""
        out = (C.c_ulonglong * 10)()
        self._call("list_stats", C.c_int(pop), C.c_int(chrom), out)
        keys = ("rebuilds", "interval_entries_used", "mutation_entries_used", "interval_capacity", "mutation_capacity", "ranges_per_row", "interval_entries_last_generation", "mutation_entries_last_generation", "compactions", "reserved")
        return dict(zip(keys, (int(v) for v in out)))

    def redo_count(self):
        n = C.c_ulonglong()
        self._call("redo_count", C.byref(n))
        return n.value

    def compute_ad(self, pop, per_chr=True):
        n = self.pop_size(pop)
        add = np.zeros((n, self.nphen)); dom = np.zeros((n, self.nphen))
        addc = np.zeros((n, self.nchr, self.nphen)) if per_chr else None
        domc = np.zeros((n, self.nchr, self.nphen)) if per_chr else None
        self._call("compute_ad", C.c_int(pop), _p(add), _p(dom), _p(addc), _p(domc))
        return add, dom, addc, domc

    def scale_ad_compute_gef(self, pop, phen, gen_num, seed, va, vd, ve, vf, beta, s2_a_gen0, s2_d_gen0,
                             common_sibling=None, f_father=None, f_mother=None):
        """Simulation::ras_scale_AD_compute_GEF (reference src/Simulation.cpp:3075): -> dict of n-vectors"""
        n = self.pop_size(pop)
        par = gev_gef_params(va, vd, ve, vf, beta, s2_a_gen0, s2_d_gen0, gen_num, 0)
        arr = lambda x: None if x is None else _arr(x, np.float64)
        cs, ff, fm = arr(common_sibling), arr(f_father), arr(f_mother)
        out = {k: np.zeros(n) for k in ("additive", "dominance", "bv", "e_noise", "parental_effect", "phen")}
        self._call("scale_ad_compute_gef", C.c_int(pop), C.c_int(phen), C.byref(par), C.c_uint32(int(seed)), _p(cs), _p(ff), _p(fm),
                   _p(out["additive"]), _p(out["dominance"]), _p(out["bv"]), _p(out["e_noise"]), _p(out["parental_effect"]), _p(out["phen"]))
        return out

    def set_ad(self, pop, additive, dominance):
        """raw A/D totals of the current generation from the host (locus-split populations: the all-reduced sums)"""
        a = _arr(additive, np.float64); d = _arr(dominance, np.float64)
        self._call("set_ad", C.c_int(pop), _p(a), _p(d))

    def get_cv_freq(self, pop, phen, chr):
        ncv = self._ncv[(pop, phen, chr)]
        f = np.zeros(ncv)
        self._call("get_cv_freq", C.c_int(pop), C.c_int(phen), C.c_int(chr), _p(f), C.c_size_t(ncv))
        return f

    def migrate(self, moves):
        """moves: list/array of (src_pop, src_pos, dst_pop) in reference append order"""
        m = np.zeros(len(moves), dtype=MOVE_DTYPE)
        for i, (sp, pos, dp) in enumerate(moves):
            m[i] = (sp, dp, pos)
        self._call("migrate", _p(m), C.c_size_t(len(m)))

    # ---- downloads
    def pop_size(self, pop):
        n = C.c_size_t()
        self._call("pop_size", C.c_int(pop), C.byref(n))
        return n.value

    def download_haps(self, pop, chr, row_begin=0, n_rows=None):
        L = self._nsnp[(pop, chr)]
        if n_rows is None:
            n_rows = 2 * self.pop_size(pop) - row_begin
        w = words_for(L)
        out = np.zeros((n_rows, w), dtype=np.uint64)
        self._call("download_haps", C.c_int(pop), C.c_int(chr), C.c_size_t(row_begin), C.c_size_t(n_rows), _p(out), C.c_size_t(w))
        return out

    def download_snp_major(self, pop, chr, snp_begin=0, n_snps=None):
        L = self._nsnp[(pop, chr)]
        n_snps = L - snp_begin if n_snps is None else n_snps
        w = words_for(2 * self.pop_size(pop))
        out = np.zeros((n_snps, w), dtype=np.uint64)
        self._call("download_snp_major", C.c_int(pop), C.c_int(chr), C.c_size_t(snp_begin), C.c_size_t(n_snps), _p(out), C.c_size_t(w))
        return out

    def format_hap_text(self, pop, chr, snp_begin=0, n_snps=None):
        """bytes of the reference's .hap file (format_hap::write_hap) for the given SNP lines"""
        L = self._nsnp[(pop, chr)]
        n_snps = L - snp_begin if n_snps is None else n_snps
        nb = n_snps * (4 * self.pop_size(pop) + 1)
        out = np.zeros(nb, dtype=np.uint8)
        self._call("format_hap_text", C.c_int(pop), C.c_int(chr), C.c_size_t(snp_begin), C.c_size_t(n_snps), _p(out), C.c_size_t(nb))
        return out

    def format_bed(self, pop, chr, snp_begin=0, n_snps=None):
        L = self._nsnp[(pop, chr)]
        n_snps = L - snp_begin if n_snps is None else n_snps
        nb = n_snps * ((self.pop_size(pop) + 3) // 4)
        out = np.zeros(nb, dtype=np.uint8)
        self._call("format_bed", C.c_int(pop), C.c_int(chr), C.c_size_t(snp_begin), C.c_size_t(n_snps), _p(out), C.c_size_t(nb))
        return out

    def format_vcf_gt(self, pop, chr, snp_begin=0, n_snps=None):
        """sample columns ("\\ta|b" per individual + newline) of the reference's VCF data lines (format_vcf::write_vcf_file)"""
        L = self._nsnp[(pop, chr)]
        n_snps = L - snp_begin if n_snps is None else n_snps
        nb = n_snps * (4 * self.pop_size(pop) + 1)
        out = np.zeros(nb, dtype=np.uint8)
        self._call("format_vcf_gt", C.c_int(pop), C.c_int(chr), C.c_size_t(snp_begin), C.c_size_t(n_snps), _p(out), C.c_size_t(nb))
        return out

    def rank_f64(self, x):
        """CommFunc::ras_rank (zero-based, ties by index)"""
        x = _arr(x, np.float64)
        out = np.zeros(len(x), dtype=np.uint64)
        self._call("rank_f64", _p(x), C.c_size_t(len(x)), _p(out))
        return out

    def download_plink_matrix(self, pop, chr, ind_begin=0, n_ind=None):
        """matrix_plink_ped rows (bit 2*snp + hap) of ras_convert_interval_to_format_plink"""
        L = self._nsnp[(pop, chr)]
        n_ind = self.pop_size(pop) - ind_begin if n_ind is None else n_ind
        w = words_for(2 * L)
        out = np.zeros((n_ind, w), dtype=np.uint64)
        self._call("download_plink_matrix", C.c_int(pop), C.c_int(chr), C.c_size_t(ind_begin), C.c_size_t(n_ind), _p(out), C.c_size_t(w))
        return out

    def format_ped_text(self, pop, chr, al0=None, al1=None, ind_begin=0, n_ind=None):
        """genotype columns of the reference's .ped lines (format_plink::write_ped_map; write_ped01_map when al0 is None)"""
        L = self._nsnp[(pop, chr)]
        n_ind = self.pop_size(pop) - ind_begin if n_ind is None else n_ind
        if al0 is not None:
            al0 = _arr(al0, np.uint8); al1 = _arr(al1, np.uint8)
            if len(al0) != L or len(al1) != L:
                raise ValueError("format_ped_text: one allele letter per SNP expected")
        nb = n_ind * (4 * L + 1)
        out = np.zeros(nb, dtype=np.uint8)
        self._call("format_ped_text", C.c_int(pop), C.c_int(chr), C.c_size_t(ind_begin), C.c_size_t(n_ind), _p(al0), _p(al1), _p(out), C.c_size_t(nb))
        return out

    def download_cv(self, pop, phen, chr):
        ncv = self._ncv[(pop, phen, chr)]
        w = words_for(ncv)
        out = np.zeros((2 * self.pop_size(pop), w), dtype=np.uint64)
        self._call("download_cv", C.c_int(pop), C.c_int(phen), C.c_int(chr), _p(out), C.c_size_t(w))
        return out

    def download_intervals(self, pop, chr):
        n = C.c_size_t()
        self._call("download_intervals", C.c_int(pop), C.c_int(chr), None, None, C.byref(n))
        parts = np.zeros(n.value, dtype=PART_DTYPE)
        off = np.zeros(2 * self.pop_size(pop) + 1, dtype=np.uint64)
        self._call("download_intervals", C.c_int(pop), C.c_int(chr), _p(parts), _p(off), C.byref(n))
        return parts, off

    def download_mutations(self, pop, chr):
        n = C.c_size_t()
        self._call("download_mutations", C.c_int(pop), C.c_int(chr), None, None, C.byref(n))
        muts = np.zeros(n.value, dtype=np.uint64)
        off = np.zeros(2 * self.pop_size(pop) + 1, dtype=np.uint64)
        self._call("download_mutations", C.c_int(pop), C.c_int(chr), _p(muts), _p(off), C.byref(n))
        return muts, off

    # ---- product-only entry points
    def reserve(self, pop, max_people):
        self._call("reserve", C.c_int(pop), C.c_size_t(max_people))

    def plane_ptr(self, pop, chr):
        """(pool pointer, unit bytes, number of haplotype slots, device pointer of the (slot, segment) -> unit table, segments per row)"""
        p = C.c_void_p(); s = C.c_size_t(); n = C.c_size_t(); t = C.c_void_p(); g = C.c_uint32()
        self._call("plane_ptr", C.c_int(pop), C.c_int(chr), C.byref(p), C.byref(s), C.byref(n), C.byref(t), C.byref(g))
        return p.value, s.value, n.value, t.value, g.value

    def stitch_totals(self):
        """(bytes written by the dense stitch, bytes of the rows of the generations produced, segments written, segments) over all reproduce calls"""
        w = C.c_ulonglong(); t = C.c_ulonglong(); sw = C.c_ulonglong(); st = C.c_ulonglong()
        self._call("stitch_totals", C.byref(w), C.byref(t), C.byref(sw), C.byref(st))
        return w.value, t.value, sw.value, st.value

    def stream(self):
        p = C.c_void_p()
        self._call("stream", C.byref(p))
        return p.value

    def last_reproduce_ms(self):
        ms = (C.c_float * 4)()
        self._call("last_reproduce_ms", ms)
        return [float(x) for x in ms]

    def sync(self):
        self._call("sync")

    def timing_totals(self):
        ms = (C.c_double * 4)(); n = C.c_ulonglong()
        self._call("timing_totals", ms, C.byref(n))
        return [float(x) for x in ms], n.value

    def set_chr_active(self, chr, active):
        self._call("set_chr_active", C.c_int(chr), C.c_int(1 if active else 0))

    def set_dense_state(self, on):
        self._call("set_dense_state", C.c_int(1 if on else 0))

    def materialize(self, pop, chr, founder_tile, row_begin=0, n_rows=None, snp_begin=0, n_snps=None):
        """genotype tile from the interval state; founder_tile: uint64 [n_founder_haps][words] holding SNPs [snp_begin, +n_snps)"""
        L = self._nsnp[(pop, chr)]
        n_snps = L - snp_begin if n_snps is None else n_snps
        n_rows = 2 * self.pop_size(pop) - row_begin if n_rows is None else n_rows
        ft = np.ascontiguousarray(founder_tile, dtype=np.uint64)
        w = words_for(n_snps)
        out = np.zeros((n_rows, w), dtype=np.uint64)
        self._call("materialize", C.c_int(pop), C.c_int(chr), C.c_size_t(row_begin), C.c_size_t(n_rows), C.c_size_t(snp_begin), C.c_size_t(n_snps),
                   _p(ft), C.c_size_t(ft.shape[1]), C.c_size_t(ft.shape[0]), _p(out), C.c_size_t(w))
        return out

    def materialize_pops(self, pop, chr, founder_tiles, row_begin=0, n_rows=None, snp_begin=0, n_snps=None):
        """as materialize(), with one founder tile per root population (list of uint64 [n_founder_haps][words])"""
        L = self._nsnp[(pop, chr)]
        n_snps = L - snp_begin if n_snps is None else n_snps
        n_rows = 2 * self.pop_size(pop) - row_begin if n_rows is None else n_rows
        tiles = [np.ascontiguousarray(t, dtype=np.uint64) for t in founder_tiles]
        w = words_for(n_snps)
        ptrs = (C.c_void_p * len(tiles))(*[t.ctypes.data for t in tiles])
        strides = (C.c_size_t * len(tiles))(*[t.shape[1] for t in tiles])
        rows = (C.c_size_t * len(tiles))(*[t.shape[0] for t in tiles])
        out = np.zeros((n_rows, w), dtype=np.uint64)
        self._call("materialize_pops", C.c_int(pop), C.c_int(chr), C.c_size_t(row_begin), C.c_size_t(n_rows), C.c_size_t(snp_begin), C.c_size_t(n_snps),
                   ptrs, strides, rows, _p(out), C.c_size_t(w))
        return out

    def materialize_bed(self, pop, chr, founder_tiles, snp_begin=0, n_snps=None):
        """PLINK .bed body of a SNP range from the interval state + one founder tile per root population"""
        L = self._nsnp[(pop, chr)]
        n_snps = L - snp_begin if n_snps is None else n_snps
        tiles = [np.ascontiguousarray(t, dtype=np.uint64) for t in founder_tiles]
        ptrs = (C.c_void_p * len(tiles))(*[t.ctypes.data for t in tiles])
        strides = (C.c_size_t * len(tiles))(*[t.shape[1] for t in tiles])
        rows = (C.c_size_t * len(tiles))(*[t.shape[0] for t in tiles])
        nb = n_snps * ((self.pop_size(pop) + 3) // 4)
        out = np.zeros(nb, dtype=np.uint8)
        self._call("materialize_bed", C.c_int(pop), C.c_int(chr), C.c_size_t(snp_begin), C.c_size_t(n_snps), ptrs, strides, rows, _p(out), C.c_size_t(nb))
        return out

    def dbg_verify_planes(self, pop, chr, founder_seeds):
        """(mismatching plane words, bad parts) of the device-side full comparison plane == materialise(intervals, synthetic founders);
        founder_seeds: the gev_synth_founders seed of every ROOT population's panel (a scalar for a single-population context)"""
        seeds = np.atleast_1d(np.asarray(founder_seeds, dtype=np.uint64))
        assert len(seeds) == self.n_pop
        a, b = C.c_ulonglong(0), C.c_ulonglong(0)
        self._call("dbg_verify_planes", C.c_int(pop), C.c_int(chr), _p(seeds), C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def dbg_prefilter_sweep(self, x_begin, x_end):
        out = (C.c_ulonglong * 3)()
        self._call("dbg_prefilter_sweep", C.c_uint32(int(x_begin)), C.c_uint32(int(x_end)), out)
        return int(out[0]), int(out[1]), int(out[2])

    def set_overlap(self, on):
        """True (default) / False / 2 (sampling-only overlap), or None: decide from two timed serialised generations"""
        self._call("set_overlap", C.c_int(-1 if on is None else int(on)))

    def set_stitch_mode(self, mode):
        self._call("set_stitch_mode", C.c_int(mode))

    def set_track_intervals(self, on):
        self._call("set_track_intervals", C.c_int(1 if on else 0))

    def export_size(self, pop, positions):
        pos = _arr(positions, np.uint64); b = C.c_size_t()
        self._call("export_size", C.c_int(pop), _p(pos), C.c_size_t(len(pos)), C.byref(b))
        return b.value

    def export_rows(self, pop, positions, device_ptr, nbytes):
        pos = _arr(positions, np.uint64)
        self._call("export_rows", C.c_int(pop), _p(pos), C.c_size_t(len(pos)), C.c_void_p(device_ptr), C.c_size_t(nbytes))

    def remove_rows(self, pop, positions):
        pos = _arr(positions, np.uint64)
        self._call("remove_rows", C.c_int(pop), _p(pos), C.c_size_t(len(pos)))

    def import_rows(self, pop, device_ptr, nbytes, n):
        self._call("import_rows", C.c_int(pop), C.c_void_p(device_ptr), C.c_size_t(nbytes), C.c_size_t(n))
