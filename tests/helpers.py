"""Shared test plumbing: load a golden fixture, feed its static inputs through the C-ABI of
any library (oracle `gevo_*` or product `gev_*`), replay generations, compare state."""
import hashlib
import os

import numpy as np

from geneevolve_amd.capi import bytes_to_words, words_for
from tests.synth import synth_packed

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        fx = {k: z[k] for k in z.files}
    fx["fixture_name"] = np.array(name)
    return fx


def bed_golden():
    """PLINK .bed bodies packed from the unmodified reference's own .ped01 output files (tests/golden/make_bed_golden.py)"""
    with np.load(os.path.join(GOLDEN, "bed_from_ref_ped01.npz")) as z:
        return {k: z[k] for k in z.files}


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def find_gen0_seeds(fx, oracle_lib):
    """The harness dumps the ras_glob_seed() stream gen 0 consumes; the seed of each
    population's ras_initial_human_gen0 is the entry that reproduces its gen-0 sexes."""
    from oracle import oracle_api
    seeds = []
    for ip in range(int(fx["n_pop"])):
        sex = fx[f"g0_pop{ip}_sex"]
        found = None
        for s in fx["globseq"]:
            r = oracle_api.kat_rand(oracle_lib, int(s), len(sex))
            if np.array_equal((r % 2 + 1).astype(np.uint8), sex):
                found = int(s); break
        assert found is not None, "gen-0 seed not identifiable"
        seeds.append(found)
    return seeds


def setup_static(ctx, fx, only_chr=None, snp_founders=True, panels=False):
    """ras_init_parameters equivalent: maps, SNP/CV grids, founder panels.  only_chr: the chromosomes whose genotype / CV
    inputs this context is given (locus-split populations); the others get their maps only."""
    n_pop, nchr, nphen = int(fx["n_pop"]), int(fx["nchr"]), int(fx["nphen"])
    for ip in range(n_pop):
        pre = f"pop{ip}_"
        nh = int(fx[pre + "n_founder_hap"])
        for ic in range(nchr):
            ctx.set_rmap(ip, ic, fx[f"{pre}chr{ic}_rmap_bp"], fx[f"{pre}chr{ic}_rmap_prob"], int(fx[f"{pre}chr{ic}_bp_dist"]))
            if int(fx[pre + "has_mut"]):
                ctx.set_mutmap(ip, ic, fx[f"{pre}chr{ic}_mut_bp"], fx[f"{pre}chr{ic}_mut_rate"])
            if only_chr is not None and ic not in only_chr:
                continue
            pos = fx[f"{pre}chr{ic}_snp_pos"]
            ctx.set_snps(ip, ic, pos)
            if not snp_founders:                     # plane-less contexts keep no genotype matrix: founder tiles are passed to gev_materialize
                pass
            else:
                words = (bytes_to_words(fx[f"{pre}chr{ic}_founders"], len(pos)) if f"{pre}chr{ic}_founders" in fx
                         else synth_packed(int(fx[f"{pre}chr{ic}_founders_synth_seed"]), nh, len(pos)))
                ctx.upload_founders(ip, ic, words, len(pos))
                if panels:                           # migrants travel as lists: every root population's panel is kept for good
                    ctx.upload_founder_panel(ip, ic, words, len(pos))
            for iph in range(nphen):
                k = f"{pre}ph{iph}_chr{ic}_"
                ctx.set_cvs(ip, iph, ic, fx[k + "cv_bp"], fx[k + "cv_a"], fx[k + "cv_d"], float(fx[f"{pre}ph{iph}_vd"]))
                ncv = len(fx[k + "cv_bp"])
                if k + "cv_val" in fx:
                    ctx.upload_cv_founders(ip, iph, ic, bytes_to_words(fx[k + "cv_val"], ncv), ncv)
                else:
                    ctx.upload_cv_founders(ip, iph, ic, synth_packed(int(fx[k + "cv_val_synth_seed"]), nh, ncv), ncv)


def derive_moves(fx, g):
    """WHO moved in generation g as (src_pop, src_pos, dst_pop) in the reference's append
    order; recorded by tests/golden/make_golden.py from the reference run."""
    return [tuple(int(x) for x in m) for m in fx[f"g{g}_moves"]]


def compare_lists(ctx, fx, key_prefix, ip, nchr, label, chrs=None):
    for ic in (range(nchr) if chrs is None else chrs):
        parts, off = ctx.download_intervals(ip, ic)
        muts, moff = ctx.download_mutations(ip, ic)
        got = np.stack([parts["st"].astype(np.int64), parts["en"].astype(np.int64), parts["hap_index"].astype(np.int64),
                        parts["root_population"].astype(np.int64)], axis=1).reshape(-1, 4)
        k = f"{key_prefix}chr{ic}_"
        if k + "parts" in fx:
            assert np.array_equal(off, fx[k + "part_off"]), f"{label}: interval offsets differ (chr {ic})"
            assert np.array_equal(got, fx[k + "parts"]), f"{label}: interval lists differ (chr {ic})"
            assert np.array_equal(moff, fx[k + "mut_off"]), f"{label}: mutation offsets differ (chr {ic})"
            assert np.array_equal(muts, fx[k + "muts"]), f"{label}: mutation lists differ (chr {ic})"
        else:
            assert np.array_equal(sha(off), fx[k + "part_off_sha"]), f"{label}: interval offsets hash (chr {ic})"
            assert np.array_equal(sha(got), fx[k + "parts_sha"]), f"{label}: interval lists hash (chr {ic})"
            assert np.array_equal(sha(moff), fx[k + "mut_off_sha"]), f"{label}: mutation offsets hash (chr {ic})"
            assert np.array_equal(sha(muts), fx[k + "muts_sha"]), f"{label}: mutation lists hash (chr {ic})"


def compare_dense(ctx, fx, g, ip, nchr, label, chrs=None):
    checked = 0
    for ic in (range(nchr) if chrs is None else chrs):
        L = len(fx[f"pop{ip}_chr{ic}_snp_pos"])
        k = f"g{g}_pop{ip}_chr{ic}_dense"
        if k in fx or k + "_sha" in fx:
            words = ctx.download_haps(ip, ic)
            packed = np.ascontiguousarray(words).view(np.uint8)[:, :(L + 7) // 8]
            if k in fx:
                assert np.array_equal(packed, fx[k]), f"{label}: dense genotype matrix differs (gen {g} chr {ic})"
            else:
                assert np.array_equal(packed[:8], fx[k + "_head"]), f"{label}: dense head differs (gen {g} chr {ic})"
                assert np.array_equal(sha(packed), fx[k + "_sha"]), f"{label}: dense hash differs (gen {g} chr {ic})"
            checked += 1
    return checked


def bits_equal(a, b):
    """bit-exact comparison of float64 arrays (NaN-safe, distinguishes -0.0)"""
    return np.array_equal(np.ascontiguousarray(a).view(np.uint64), np.ascontiguousarray(b).view(np.uint64))


def check_cvval_dump(ctx, fx, ip, nchr, nphen, label):
    """the reference's --debug dump of the CV genotypes ras_find_cv resolved (src/Simulation.cpp:2665-2683), written inside the
    LAST generation's ras_compute_AD (before migration): per individual "c0 c1 " per CV in FILE order; every phenotype writes the
    same file, so it holds the last phenotype"""
    from geneevolve_amd.capi import unpack_rows
    for ic in range(nchr):
        k = f"cvvalfile_pop{ip}_chr{ic}_sha"
        if k in fx:
            ncv = len(fx[f"pop{ip}_ph{nphen-1}_chr{ic}_cv_bp"])
            bits = unpack_rows(ctx.download_cv(ip, nphen - 1, ic), ncv)          # [2n][ncv]
            n = bits.shape[0] // 2
            inter = np.empty((n, 2 * ncv), dtype=np.uint8); inter[:, 0::2] = bits[0::2]; inter[:, 1::2] = bits[1::2]
            txt = "".join(" ".join(str(int(v)) for v in row) + " \n" for row in inter).encode() if ncv else b"\n" * n
            assert np.array_equal(np.frombuffer(hashlib.sha256(txt).digest(), dtype=np.uint8), fx[k]), f"{label}: CV genotypes differ from the reference's .cvval dump (pop {ip} chr {ic})"


def replay_case(lib, fx, gen0_seeds, label, device=-1, float_exact=True, check_lists=True, max_gen=None, check_gef=False, gef_rtol=0.0):
    """Run the whole fixture through `lib` and compare every dumped quantity."""
    n_pop, nchr, nphen, ngen = int(fx["n_pop"]), int(fx["nchr"]), int(fx["nphen"]), int(fx["n_gen"])
    if max_gen:
        ngen = min(ngen, max_gen)
    ctx = lib.create(n_pop, nchr, nphen, device) if lib.has_device_arg else lib.create(n_pop, nchr, nphen)
    setup_static(ctx, fx)

    def cmp_float(a, b, what):
        if float_exact:
            assert bits_equal(a, b), f"{label}: {what} not bit-identical (max abs diff {np.max(np.abs(a-b))})"
        else:
            assert np.allclose(a, b, rtol=0, atol=1e-6), f"{label}: {what} differs by more than 1e-6"

    for ip in range(n_pop):
        n0 = len(fx[f"g0_pop{ip}_sex"])
        sex = ctx.init_gen0(ip, n0, gen0_seeds[ip])
        assert np.array_equal(sex, fx[f"g0_pop{ip}_sex"]), f"{label}: gen-0 sex differs"
    for ip in range(n_pop):
        _, _, addc, domc = ctx.compute_ad(ip)
        cmp_float(addc, fx[f"g0_pop{ip}_add_chr"], "gen-0 additive_chr")
        cmp_float(domc, fx[f"g0_pop{ip}_dom_chr"], "gen-0 dominance_chr")
        if check_lists:
            compare_lists(ctx, fx, f"g0_pop{ip}_", ip, nchr, label + " gen0")
        compare_dense(ctx, fx, 0, ip, nchr, label)
    n_dense = 0
    for g in range(1, ngen + 1):
        for ip in range(n_pop):
            pre = f"g{g}_pop{ip}_"
            ms = fx[pre + "mut_seeds"]
            sex = ctx.reproduce(ip, fx[pre + "couples"], int(fx[pre + "seed_reproduce"]), ms if len(ms) else None)
            assert np.array_equal(sex, fx[pre + "sex"]), f"{label}: offspring sex differs at gen {g} pop {ip}"
            add, dom, addc, domc = ctx.compute_ad(ip)
            cmp_float(addc, fx[pre + "add_chr"], f"additive_chr gen {g}")
            cmp_float(domc, fx[pre + "dom_chr"], f"dominance_chr gen {g}")
            cmp_float(add, fx[pre + "additive"], f"additive gen {g}")
            cmp_float(dom, fx[pre + "dominance"], f"dominance gen {g}")
            if check_lists:
                compare_lists(ctx, fx, pre, ip, nchr, f"{label} gen {g} pop {ip}")
            if g == int(fx["n_gen"]):
                check_cvval_dump(ctx, fx, ip, nchr, nphen, label)
            if check_gef:                      # ras_scale_AD_compute_GEF follows ras_compute_AD (src/Simulation.cpp:1943-1946)
                for iph in range(nphen):
                    k = f"g{g}_pop{ip}_ph{iph}_gef_"
                    s2a, s2d, va, vd, ve, vf, beta = [float(x) for x in fx[k + "par"]]
                    gi, want = fx[k + "in"], fx[k + "out"]
                    got = ctx.scale_ad_compute_gef(ip, iph, g, int(fx[k + "seed"]), va, vd, ve, vf, beta, s2a, s2d,
                                                   common_sibling=gi[:, 0], f_father=gi[:, 1], f_mother=gi[:, 2])
                    for j, nm in enumerate(("additive", "dominance", "bv", "e_noise", "parental_effect", "phen")):
                        if gef_rtol == 0.0:
                            assert bits_equal(got[nm], want[:, j]), f"{label}: scaled {nm} not bit-identical (gen {g} phen {iph}), max diff {np.max(np.abs(got[nm]-want[:, j]))}"
                        else:
                            assert np.allclose(got[nm], want[:, j], rtol=gef_rtol, atol=1e-12), f"{label}: scaled {nm} differs by more than {gef_rtol} relative (gen {g} phen {iph}): max abs diff {np.max(np.abs(got[nm]-want[:, j]))}"
        if f"g{g}_moves" in fx:
            ctx.migrate(derive_moves(fx, g))
            for ip in range(n_pop):
                assert ctx.pop_size(ip) == len(fx[f"g{g}_pop{ip}_postmig_sex"])
                if check_lists:
                    compare_lists(ctx, fx, f"g{g}_pop{ip}_postmig_", ip, nchr, f"{label} gen {g} post-migration pop {ip}")
        for ip in range(n_pop):
            n_dense += compare_dense(ctx, fx, g, ip, nchr, label)
    # K8: genotype tiles rebuilt from the interval state + founder tiles of every root population == the dense matrix
    if lib.exports("materialize_pops") and all(f"pop{ip}_chr0_founders" in fx for ip in range(n_pop)):
        from geneevolve_amd.capi import pack_rows, unpack_rows
        for ic in range(nchr):
            L = len(fx[f"pop0_chr{ic}_snp_pos"])
            for (s0, ns) in ((0, L), (L // 3, min(L - L // 3, 77))):
                tiles = [pack_rows(unpack_rows(bytes_to_words(fx[f"pop{ip}_chr{ic}_founders"], L), L)[:, s0:s0 + ns]) for ip in range(n_pop)]
                for ip in range(n_pop):
                    want = pack_rows(unpack_rows(ctx.download_haps(ip, ic), L)[:, s0:s0 + ns])
                    assert np.array_equal(ctx.materialize_pops(ip, ic, tiles, 0, None, s0, ns), want), f"{label}: materialised tile differs (pop {ip} chr {ic} SNPs {s0}+{ns})"
    # the reference's own .hap output file of the last generation (format_hap::write_hap)
    if ngen == int(fx["n_gen"]) and lib.exports("format_hap_text"):
        for ip in range(n_pop):
            for ic in range(nchr):
                k = f"hapfile_pop{ip}_chr{ic}_"
                if k + "sha" in fx:
                    txt = ctx.format_hap_text(ip, ic)
                    assert len(txt) == int(fx[k + "size"]), f"{label}: .hap size"
                    assert np.array_equal(txt[:4096], fx[k + "head"]), f"{label}: .hap head differs"
                    assert np.array_equal(sha(txt), fx[k + "sha"]), f"{label}: .hap text differs from the reference's file (pop {ip} chr {ic})"
    # the reference's own .vcf output of the last generation (format_vcf::write_vcf_file): sample columns of every data line
    if ngen == int(fx["n_gen"]) and lib.exports("format_vcf_gt"):
        for ip in range(n_pop):
            for ic in range(nchr):
                k = f"vcffile_pop{ip}_chr{ic}_"
                if k + "gt_sha" in fx:
                    assert ctx.pop_size(ip) == int(fx[k + "n_samples"])
                    assert np.array_equal(sha(ctx.format_vcf_gt(ip, ic)), fx[k + "gt_sha"]), f"{label}: VCF sample columns differ from the reference's file (pop {ip} chr {ic})"
    # PLINK .bed: the reference has no writer, but its own .ped01 file holds the same genotypes as text; the golden bytes are that
    # file packed per the PLINK specification.  Both device paths must reproduce them: the resident planes (gev_format_bed) and
    # the interval state + founder tiles (gev_materialize_bed, the output path of plane-less contexts, BASELINE config 5).
    if ngen == int(fx["n_gen"]) and lib.exports("format_bed"):
        bed = bed_golden()
        name = str(fx["fixture_name"])
        from geneevolve_amd.capi import unpack_rows
        for ip in range(n_pop):
            for ic in range(nchr):
                k = f"{name}_pop{ip}_chr{ic}_bed"
                if k not in bed:
                    continue
                L = len(fx[f"pop{ip}_chr{ic}_snp_pos"])
                assert tuple(bed[k[:-3] + "shape"]) == (ctx.pop_size(ip), L)
                assert np.array_equal(ctx.format_bed(ip, ic), bed[k].ravel()), f"{label}: .bed differs from the reference's .ped01 genotypes (pop {ip} chr {ic})"
                if lib.exports("materialize_bed") and all(f"pop{jp}_chr{ic}_founders" in fx for jp in range(n_pop)):
                    tiles = [bytes_to_words(fx[f"pop{jp}_chr{ic}_founders"], L) for jp in range(n_pop)]
                    assert np.array_equal(ctx.materialize_bed(ip, ic, tiles), bed[k].ravel()), f"{label}: .bed from the interval state differs from the reference's .ped01 genotypes (pop {ip} chr {ic})"
    # the reference's own .ped files of the last generation (format_plink::write_ped_map / write_ped01_map)
    if ngen == int(fx["n_gen"]) and lib.exports("format_ped_text"):
        for ip in range(n_pop):
            for ic in range(nchr):
                check_ped_files(ctx, fx, ngen, ip, ic, label)
    ctx.close()
    return n_dense


def interleave_haps(bits, L):
    """numpy statement of matrix_plink_ped (src/Simulation.cpp:1335-1362): bit 2*ii+ihap of row ih, from the hap-major matrix"""
    u = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, :L]
    n = u.shape[0] // 2
    m = np.zeros((n, 2 * L), dtype=np.uint8)
    m[:, 0::2] = u[0::2]; m[:, 1::2] = u[1::2]
    w = (2 * L + 63) // 64
    out = np.zeros((n, w * 64), dtype=np.uint8); out[:, :2 * L] = m
    return np.packbits(out, axis=1, bitorder="little").view(np.uint64)


def check_ped_files(ctx, fx, ngen, ip, ic, label):
    L = len(fx[f"pop{ip}_chr{ic}_snp_pos"])
    n = ctx.pop_size(ip)
    assert np.array_equal(ctx.download_plink_matrix(ip, ic), interleave_haps(ctx.download_haps(ip, ic), L)), f"{label}: matrix_plink_ped differs (pop {ip} chr {ic})"
    for tag in ("ped", "ped01"):
        k = f"{tag}file_pop{ip}_chr{ic}_"
        if k + "sha" not in fx:
            continue
        al0 = fx[f"pop{ip}_chr{ic}_al0"] if tag == "ped" else None
        al1 = fx[f"pop{ip}_chr{ic}_al1"] if tag == "ped" else None
        txt = ctx.format_ped_text(ip, ic, al0, al1)
        assert np.array_equal(sha(txt), fx[k + "geno_sha"]), f"{label}: .{tag} genotype columns differ from the reference's file (pop {ip} chr {ic})"
        # whole file = the host's six id columns (FID IID PID MID sex phen, src/Simulation.cpp:1391-1402) + the device text
        pre = f"g{ngen}_pop{ip}_postmig_" if f"g{ngen}_pop{ip}_postmig_ids" in fx else f"g{ngen}_pop{ip}_"
        ids, sex = fx[pre + "ids"], fx[pre + "sex"]
        cols = np.stack([ids[:, 1] + 1, ids[:, 0] + 1, ids[:, 1] + 1, ids[:, 2] + 1, sex.astype(np.int64), np.full(n, -9)], axis=1)
        assert np.array_equal(cols, fx[k + "ids"]), f"{label}: .{tag} id columns"
        lines = txt.reshape(n, 4 * L + 1)
        h = hashlib.sha256()
        for i in range(n):
            h.update(" ".join(str(int(v)) for v in cols[i]).encode()); h.update(lines[i].tobytes())
        assert np.array_equal(np.frombuffer(h.digest(), dtype=np.uint8), fx[k + "sha"]), f"{label}: .{tag} file differs from the reference's (pop {ip} chr {ic})"


# ---------------------------------------------------------------------------------------------------------------
# closed loop: the whole generation loop of the reference from --seed alone (host mirror + library), single population
# ---------------------------------------------------------------------------------------------------------------
from geneevolve_amd.host import comm_mean, comm_var, selection_func  # noqa: E402  (host mirror of CommFunc::mean/var, ras_selection_func)


def closed_loop_case(lib, fx, label, device=-1, exact=True, plane_less=False, at_end=None, info_texts=None, mate="host"):
    """Simulation::run for a single-population fixture, driven from the seed ALONE: ras_glob_seed() stream, gen-0 founders,
    ras_compute_AD, ras_scale_AD_compute_GEF (every phenotype, parental effect with the adjusted beta), mating / selection
    values, random_mate or assort_mate (device rank), reproduce -- every generation's couples, sexes, pedigree, raw A/D,
    phenotypes, next-generation mating inputs and the reference's .info files are compared with what the reference did.
    exact=False (device phenotype scaling is within 1e-12, not bit-exact) relaxes the float comparisons only.
    mate (random-mating fixtures): "host" = the host mirror of Simulation::random_mate; "device" = gev_random_mate, the couples stay
    in the library for gev_reproduce; "fused" = gev_generation_begin/_end (random_mate -> reproduce -> ras_compute_AD in one piece,
    the ras_glob_seed() draws of the three made by the library from glob_generator's state)."""
    from geneevolve_amd.host import Simulation, ras_save_human_info
    assert int(fx["n_pop"]) == 1
    nchr, nphen, ngen, rm = int(fx["nchr"]), int(fx["nphen"]), int(fx["n_gen"]), bool(int(fx["pop0_rm"]))
    ctx = lib.create(1, nchr, nphen, device) if lib.has_device_arg else lib.create(1, nchr, nphen)
    if plane_less:                                   # BASELINE config 5's mode: interval state only, genotypes materialised on demand
        ctx.set_dense_state(False)
    setup_static(ctx, fx, snp_founders=not plane_less)
    var = [[float(v) for v in fx[f"pop0_ph{p}_var"]] for p in range(nphen)]       # va, vd, ve, vf
    vc = [float(fx[f"pop0_ph{p}_vc"]) if f"pop0_ph{p}_vc" in fx else 0.0 for p in range(nphen)]
    beta = [1.0] * nphen                                                          # parameters.cpp default; adjusted after generation 0
    extra = [str(x) for x in fx["args_extra"]]
    mm = float(extra[extra.index("--MM") + 1]) if "--MM" in extra else 0.0
    avoid = "--avoid_inbreeding" in extra
    vt_type = int(extra[extra.index("--vt_type") + 1]) if "--vt_type" in extra else 1     # parameters.cpp default 1
    handed_down = "phen" if vt_type == 1 else "parental_effect"                           # what _Pop_info_prev_gen hands to the children's F (:3123-3132)
    has_mut = bool(int(fx["pop0_has_mut"]))
    sim = Simulation(ctx, int(fx["seed"]), nchr, has_mut, track_pedigree=True)

    def close(a, b, what):
        if exact:
            assert bits_equal(a, b), f"{label}: {what} not bit-identical (max abs diff {np.max(np.abs(np.asarray(a) - np.asarray(b)))})"
        else:
            assert np.allclose(a, b, rtol=1e-9, atol=1e-12), f"{label}: {what} differs (max abs diff {np.max(np.abs(np.asarray(a) - np.asarray(b)))})"

    def scale(g, s2, prev_phen):
        """ras_scale_AD_compute_GEF for every phenotype (:1938-1946), one ras_glob_seed() each"""
        n = len(sim.sex[0]); outs = []
        common = common_gen0 if g == 0 else (sim.common_sibling(0, vc) if any(v > 0 for v in vc) else [np.zeros(n)] * nphen)
        for p in range(nphen):
            va, vd, ve, vf = var[p]
            seed = int(sim.ras_glob_seed()[0])
            if g > 0:
                assert seed == int(fx[f"g{g}_pop0_ph{p}_gef_seed"]), f"{label}: the ras_glob_seed() stream is out of step at generation {g}"
            ff = prev_phen[p][sim.ped[0].ID_Father] if g > 0 else np.zeros(n)       # vt_type 1: the parents' phenotypes (:3123-3127);
            fm = prev_phen[p][sim.ped[0].ID_Mother] if g > 0 else np.zeros(n)       # vt_type 2: their parental effects (:3128-3132)
            if g > 0:
                assert int(fx[f"g{g}_pop0_ph{p}_gef_vt"]) == vt_type
                assert bits_equal(np.stack([ff, fm], axis=1), fx[f"g{g}_pop0_ph{p}_gef_in"][:, 1:3]) or not exact, f"{label}: parental inputs, phenotype {p} generation {g}"
                assert bits_equal(common[p], fx[f"g{g}_pop0_ph{p}_gef_in"][:, 0]), f"{label}: common sibling effect, phenotype {p} generation {g}"
            o = ctx.scale_ad_compute_gef(0, p, g, seed, va, vd, ve, vf, beta[p], s2[p][0], s2[p][1], common_sibling=common[p], f_father=ff, f_mother=fm)
            o["common_sibling"] = common[p]
            if g > 0:
                close(o["phen"], fx[f"g{g}_pop0_ph{p}_gef_out"][:, 5], f"phenotype {p} generation {g}")
            outs.append(o)
        return outs

    def values(outs):
        mv = np.zeros(len(sim.sex[0])); sv = np.zeros(len(sim.sex[0]))
        for p, o in enumerate(outs):                                             # --omega / --lambda, default 1 (:3311-3320)
            omega = float(fx[f"pop0_ph{p}_omega"]) if f"pop0_ph{p}_omega" in fx else 1.0
            lam = float(fx[f"pop0_ph{p}_lambda"]) if f"pop0_ph{p}_lambda" in fx else 1.0
            mv = mv + omega * o["phen"]; sv = sv + lam * o["phen"]
        return mv, sv

    def check_info_file(g, outs, mv, z, svf):
        """Population::ras_save_human_info: the reference's .info.popK.genG.txt, byte for byte (exact builds only)"""
        if info_texts is not None:                    # the caller compares the texts of two builds field by field
            info_texts.append(ras_save_human_info(sim.ped[0], sim.sex[0], outs, mv, z, svf))
        if exact and f"infofile_pop0_gen{g}_sha" in fx:
            txt = ras_save_human_info(sim.ped[0], sim.sex[0], outs, mv, z, svf)
            assert np.array_equal(np.frombuffer(hashlib.sha256(txt).digest(), dtype=np.uint8), fx[f"infofile_pop0_gen{g}_sha"]), f"{label}: .info file of generation {g} differs from the reference's"

    sim.ras_initial_human_gen0(0, len(fx["g0_pop0_sex"]))                       # ras_init_generation0 (:529)
    assert np.array_equal(sim.sex[0], fx["g0_pop0_sex"]), f"{label}: gen-0 sex"
    from geneevolve_amd.host import NormalEngine
    common_gen0 = []                                                             # common effect of generation 0 (:3053-3066): one ras_glob_seed() per phenotype with vc > 0
    for p in range(nphen):
        n0 = len(sim.sex[0])
        common_gen0.append(NormalEngine(int(sim.ras_glob_seed()[0])).draw(n0, float(np.sqrt(vc[p]))) if vc[p] > 0 else np.zeros(n0))
    add, dom, _, _ = ctx.compute_ad(0)
    s2 = [(comm_var(add[:, p]), comm_var(dom[:, p])) for p in range(nphen)]      # _var_a_gen0 / _var_d_gen0 (:557-561)
    outs = scale(0, s2, None)
    mv, sv = values(outs)
    sv_mean, sv_var = comm_mean(sv), comm_var(sv)                               # standardised to generation 0 (:3326-3330)
    z = (sv - sv_mean) / np.sqrt(sv_var) if sv_var > 0 else sv - sv_mean
    svf = np.ones(len(mv))                                                      # generation 0: everybody may marry (:3388)
    check_info_file(0, outs, mv, z, svf)
    for p in range(nphen):                                                      # "adjust beta" (:648-657): on var(P), or (vt_type 2) on var(F) when that is > 0
        if vt_type == 1:
            beta[p] = float(np.sqrt(var[p][3] / (2 * comm_var(outs[p]["phen"]))))
        elif vt_type == 2 and comm_var(outs[p]["parental_effect"]) > 0:
            beta[p] = float(np.sqrt(var[p][3] / (2 * comm_var(outs[p]["parental_effect"]))))
    for g in range(1, ngen + 1):
        pop_size, mat_cor, dist, func, p1, p2 = str(fx["pop0_popinfo"][g - 1]).split()
        k = f"g{g}_pop0_mate_"
        close(svf, fx[k + "svf"], f"selection function values entering generation {g}")
        fused = False
        if rm:
            # selection_value_func NULL = "every value is 1" (the draws are then skipped): exercised on even generations
            svf_arg = None if (g % 2 == 0 and np.all(np.asarray(svf) == 1.0)) else svf
            if mate == "host":
                sim.random_mate(0, svf, int(pop_size))
            elif mate == "device":
                sim.random_mate_device(0, svf_arg, int(pop_size))
                assert sim.num_males_mate == int(np.sum(fx[k + "sex"] == 1) if np.all(np.asarray(svf) == 1.0) else sim.num_males_mate)
            else:
                fused = True
                prev_phen = [o[handed_down] for o in outs]
                res = sim.next_generation_rm(0, int(pop_size), svf_arg, want_couples=True)
                assert int(res["seed_mate"]) == int(fx[k + "seed"]) and int(res["seed_reproduce"]) == int(fx[f"g{g}_pop0_seed_reproduce"]), f"{label}: seeds drawn by the library, generation {g}"
        else:
            close(mv, fx[k + "am_mv"], f"mating values entering generation {g}")
            sim.assort_mate(0, svf, mv, int(pop_size), float(mat_cor), mm_percent=mm, avoid_inbreeding=avoid, offspring_dist=dist,
                            rank=ctx.rank_f64 if lib.exports("rank_f64") else None)
        c, want = sim.couples[0], fx[f"g{g}_pop0_couples"]
        assert len(c) == len(want) and np.array_equal(c["pos_male"].astype(np.int64), want[:, 0]) and np.array_equal(c["pos_female"].astype(np.int64), want[:, 1]) \
            and np.array_equal(c["inbreed"], want[:, 2]) and np.array_equal(c["num_offspring"], want[:, 3]), f"{label}: couples of generation {g}"
        if not fused:
            prev_phen = [o[handed_down] for o in outs]
            sim.reproduce(0, g)
        assert np.array_equal(sim.sex[0], fx[f"g{g}_pop0_sex"]), f"{label}: sex generation {g}"
        ped = sim.ped[0]
        assert np.array_equal(np.stack([ped.ID, ped.ID_Father, ped.ID_Mother], axis=1), fx[f"g{g}_pop0_ids"]), f"{label}: pedigree generation {g}"
        add, dom, _, _ = ctx.compute_ad(0)
        assert bits_equal(add, fx[f"g{g}_pop0_additive"]) and bits_equal(dom, fx[f"g{g}_pop0_dominance"]), f"{label}: raw A/D generation {g}"
        outs = scale(g, s2, prev_phen)
        mv, sv = values(outs)
        z = (sv - sv_mean) / np.sqrt(sv_var) if sv_var > 0 else sv - sv_mean
        svf = selection_func(func, float(p1), float(p2), z)
        check_info_file(g, outs, mv, z, svf)
    if not plane_less:
        compare_dense(ctx, fx, ngen, 0, nchr, label)
    if at_end is not None:
        at_end(ctx, sim)
    ctx.close()


def closed_loop_migration_case(lib, fx, label, device=-1, exact=True, mate="host"):
    """the same closed loop for SEVERAL populations with migration (fixture mig2, BASELINE config 3's shape): per generation
    and population random_mate -> reproduce -> ras_compute_AD -> ras_scale_AD_compute_GEF, then mating/selection values, then
    ras_do_migration -- WHO moves is restated on the host (selection sampling on the reference's process-wide static engine),
    the rows move inside the library -- all from --seed alone; post-migration populations and the .info files of both
    populations are compared with the reference's."""
    from geneevolve_amd.host import SampleWithoutReplacement, Simulation, ras_do_migration, ras_save_human_info, environmental_effects_specific_to_each_population
    n_pop, nchr, nphen, ngen = int(fx["n_pop"]), int(fx["nchr"]), int(fx["nphen"]), int(fx["n_gen"])
    assert nphen == 1 and all(int(fx[f"pop{ip}_rm"]) == 1 for ip in range(n_pop))
    extra = [str(x) for x in fx["args_extra"]]
    gamma = float(extra[extra.index("--gamma") + 1]) if "--gamma" in extra else 0.0      # environmental effects specific to each population (:3345)
    ctx = lib.create(n_pop, nchr, nphen, device) if lib.has_device_arg else lib.create(n_pop, nchr, nphen)
    setup_static(ctx, fx)
    sim = Simulation(ctx, int(fx["seed"]), nchr, bool(int(fx["pop0_has_mut"])), track_pedigree=True)
    sampler = SampleWithoutReplacement()
    var = [[float(v) for v in fx[f"pop{ip}_ph0_var"]] for ip in range(n_pop)]
    assert all(v[3] == 0.0 for v in var)

    def close(a, b, what):
        if exact:
            assert bits_equal(a, b), f"{label}: {what} not bit-identical"
        else:
            assert np.allclose(a, b, rtol=1e-9, atol=1e-12), f"{label}: {what} differs"

    rec = [dict() for _ in range(n_pop)]        # per population: outs (phenotype components), mv, z, svf of the CURRENT individuals
    s2, sv0 = [None] * n_pop, [None] * n_pop

    def scale(ip, g):
        n = len(sim.sex[ip]); va, vd, ve, vf = var[ip]
        seed = int(sim.ras_glob_seed()[0])
        if g > 0:
            assert seed == int(fx[f"g{g}_pop{ip}_ph0_gef_seed"]), f"{label}: ras_glob_seed() stream out of step (gen {g} pop {ip})"
        o = ctx.scale_ad_compute_gef(ip, 0, g, seed, va, vd, ve, vf, 1.0, s2[ip][0], s2[ip][1], common_sibling=np.zeros(n), f_father=np.zeros(n), f_mother=np.zeros(n))
        o["common_sibling"] = np.zeros(n)
        if g > 0:
            close(o["phen"], fx[f"g{g}_pop{ip}_ph0_gef_out"][:, 5], f"phenotypes gen {g} pop {ip}")
        return o

    def values(ip, g, o, func=None):
        mv = 0.0 + 1.0 * o["phen"]; sv = 0.0 + 1.0 * o["phen"]
        if g == 0:
            sv0[ip] = (comm_mean(sv), comm_var(sv))
        z = (sv - sv0[ip][0]) / np.sqrt(sv0[ip][1]) if sv0[ip][1] > 0 else sv - sv0[ip][0]
        svf = np.ones(len(mv)) if g == 0 else selection_func(func[0], func[1], func[2], z)
        rec[ip] = {"out": o, "mv": mv, "z": z, "svf": svf}

    def info_file(ip, g):
        if exact and f"infofile_pop{ip}_gen{g}_sha" in fx:
            r = rec[ip]
            txt = ras_save_human_info(sim.ped[ip], sim.sex[ip], [r["out"]], r["mv"], r["z"], r["svf"])
            assert np.array_equal(np.frombuffer(hashlib.sha256(txt).digest(), dtype=np.uint8), fx[f"infofile_pop{ip}_gen{g}_sha"]), f"{label}: .info file gen {g} pop {ip}"

    for ip in range(n_pop):                                                      # ras_init_generation0
        sim.ras_initial_human_gen0(ip, len(fx[f"g0_pop{ip}_sex"]))
        assert np.array_equal(sim.sex[ip], fx[f"g0_pop{ip}_sex"])
        add, dom, _, _ = ctx.compute_ad(ip)
        s2[ip] = (comm_var(add[:, 0]), comm_var(dom[:, 0]))
        rec[ip]["o"] = scale(ip, 0)
    environmental_effects_specific_to_each_population([rec[ip]["o"]["phen"] for ip in range(n_pop)], gamma)   # (:578)
    for ip in range(n_pop):
        values(ip, 0, rec[ip]["o"])
    for ip in range(n_pop):
        info_file(ip, 0)
    for g in range(1, ngen + 1):
        funcs = []
        for ip in range(n_pop):
            pop_size, mat_cor, dist, func, p1, p2 = str(fx[f"pop{ip}_popinfo"][g - 1]).split()
            funcs.append((func, float(p1), float(p2)))
            close(rec[ip]["svf"], fx[f"g{g}_pop{ip}_mate_svf"], f"selection function values entering gen {g} pop {ip}")
            if mate == "host":
                sim.random_mate(ip, rec[ip]["svf"], int(pop_size))
            elif mate == "device":                       # the library mates on the sexes that followed the migrants
                sim.random_mate_device(ip, rec[ip]["svf"], int(pop_size))
            else:
                sim.next_generation_rm(ip, int(pop_size), rec[ip]["svf"], want_couples=True)
            c, want = sim.couples[ip], fx[f"g{g}_pop{ip}_couples"]
            assert np.array_equal(c["pos_male"].astype(np.int64), want[:, 0]) and np.array_equal(c["pos_female"].astype(np.int64), want[:, 1]), f"{label}: couples gen {g} pop {ip}"
            if mate != "fused":
                sim.reproduce(ip, g)
            assert np.array_equal(sim.sex[ip], fx[f"g{g}_pop{ip}_sex"]), f"{label}: sex gen {g} pop {ip}"
            add, dom, _, _ = ctx.compute_ad(ip)
            assert bits_equal(add, fx[f"g{g}_pop{ip}_additive"]) and bits_equal(dom, fx[f"g{g}_pop{ip}_dominance"]), f"{label}: raw A/D gen {g} pop {ip}"
            rec[ip]["o"] = scale(ip, g)
        environmental_effects_specific_to_each_population([rec[ip]["o"]["phen"] for ip in range(n_pop)], gamma)   # (:1976)
        for ip in range(n_pop):
            values(ip, g, rec[ip]["o"], funcs[ip])
        # ---- ras_do_migration(gen_num - 1)
        moves = ras_do_migration([len(sim.sex[ip]) for ip in range(n_pop)], fx["migration_mat_gen"][g - 1], sim.ras_glob_seed, sampler)
        assert moves == derive_moves(fx, g), f"{label}: WHO migrates in generation {g}"
        sim.ras_do_migration(moves)
        # the host's Human records follow (erase at the origin, append at the destination in move order)
        gone = [np.zeros(len(sim.sex[ip]), dtype=bool) for ip in range(n_pop)]
        for sp, pos, dp in moves:
            gone[sp][pos] = True
        old = [(sim.sex[ip], sim.ped[ip], rec[ip]) for ip in range(n_pop)]
        for ip in range(n_pop):
            keep = np.flatnonzero(~gone[ip])
            sx, pd, r = old[ip]
            sex_new = [sx[keep]]; ped_new = pd.take(keep)
            cols = {k: [r[k][keep]] for k in ("mv", "z", "svf")}
            out_new = {k: [v[keep]] for k, v in r["out"].items()}
            for sp, pos, dp in moves:
                if dp != ip:
                    continue
                osx, opd, orr = old[sp]
                sex_new.append(osx[pos:pos + 1]); ped_new = ped_new.append(opd.take(np.array([pos])))
                for k in cols:
                    cols[k].append(orr[k][pos:pos + 1])
                for k in out_new:
                    out_new[k].append(orr["out"][k][pos:pos + 1])
            sim.sex[ip] = np.concatenate(sex_new); sim.ped[ip] = ped_new
            rec[ip] = {k: np.concatenate(v) for k, v in cols.items()}
            rec[ip]["out"] = {k: np.concatenate(v) for k, v in out_new.items()}
            assert ctx.pop_size(ip) == len(sim.sex[ip])
            assert np.array_equal(sim.sex[ip], fx[f"g{g}_pop{ip}_postmig_sex"]), f"{label}: post-migration sex gen {g} pop {ip}"
            assert np.array_equal(np.stack([sim.ped[ip].ID, sim.ped[ip].ID_Father, sim.ped[ip].ID_Mother], axis=1), fx[f"g{g}_pop{ip}_postmig_ids"]), f"{label}: post-migration ids gen {g} pop {ip}"
        for ip in range(n_pop):
            info_file(ip, g)
        for ip in range(n_pop):
            compare_dense(ctx, fx, g, ip, nchr, label)
    ctx.close()
