"""GPU parity tests (pytest -m gpu): the HIP library, called through the C-ABI, against
 (1) the golden fixtures generated from the real reference (bit-exact, incl. FP64 A/D),
 (2) the CPU oracle on seeded synthetic inputs,
 (3) size-independent properties at larger sizes."""
import ctypes as C
import gzip
import os

import numpy as np
import pytest

from geneevolve_amd import capi
from geneevolve_amd.host import GlobSeedStream, Simulation, SyntheticConfig, synthetic_random_mate
from oracle import oracle_api
from tests import helpers
from tests.synth import synth_packed

pytestmark = pytest.mark.gpu


def kat_lines(tag):
    with gzip.open(os.path.join(helpers.GOLDEN, "kat.txt.gz"), "rt") as f:
        return [l.split() for l in f if l.startswith(tag + " ")]


def test_native_library_is_the_hip_build(gpu_lib):
    assert gpu_lib.path.endswith("geneevolve_amd/csrc/libgeneevolve_amd.so")
    v = gpu_lib.lib.gev_version; v.restype = C.c_char_p
    assert b"gfx950" in v()


def test_device_glibc_rand_matches_oracle(gpu_lib, oracle_lib):
    ctx = gpu_lib.create(1, 1, 1)
    rs = np.random.RandomState(3)
    for seed in [0, 1, 2, 12345, 2147483646, 2147483647, 2147483648, 4294967295] + [int(x) for x in rs.randint(0, 2**32, 20, dtype=np.uint64)]:
        out = np.zeros(200, dtype=np.int32)
        gpu_lib.check(gpu_lib.lib.gev_dbg_rand(ctx.h, C.c_uint32(seed), C.c_uint32(200), out.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(out, oracle_api.kat_rand(oracle_lib, seed, 200)), f"srand({seed})"
    ctx.close()


def test_device_sim_loc_rec_matches_reference_vectors(gpu_lib):
    m = kat_lines("LOCMAP")[0]
    R, bp0, dist = int(m[1]), int(m[2]), int(m[3])
    prob = np.array([float.fromhex(x) for x in m[4:]])
    bp = (bp0 + dist * np.arange(R)).astype(np.uint64)
    ctx = gpu_lib.create(1, 1, 1)
    ctx.set_rmap(0, 0, bp, prob, dist)
    for t in kat_lines("LOC"):
        seed = int(t[1]); nxt = [int(t[2]), int(t[3])]; n = int(t[4])
        want = np.array([int(x) for x in t[5:5 + n]], dtype=np.uint64)
        locs = np.zeros(512, dtype=np.uint64); k = C.c_uint32(); nx = (C.c_int * 2)()
        gpu_lib.check(gpu_lib.lib.gev_dbg_sim_loc_rec(ctx.h, 0, 0, C.c_uint32(seed), locs.ctypes.data_as(C.c_void_p), C.c_uint32(512), C.byref(k), nx))
        assert k.value == n - 2, f"seed {seed}: crossover count"
        assert np.array_equal(locs[:k.value], want[1:-1]), f"seed {seed}: breakpoints"
        assert [nx[0], nx[1]] == nxt, f"seed {seed}: rand() state after the call"
    ctx.close()


def test_fp64_prefilter_of_the_sampling_scan_is_a_superset_for_every_engine_state(gpu_lib):
    """the batched sampling kernels test most draws with one FP64 FMA instead of an exact modular multiply; the exact test runs
    only for the draws the prefilter flags.  EXHAUSTIVE over the engine's whole state space (x = 1 .. 2^31-2) and all 32
    per-lane multipliers (6.9e10 cases): the prefilter's 20-bit fraction never deviates from the exact one by more than the
    bound its threshold allows for, so no draw the exact test would accept can be missed, whatever the map's probabilities."""
    g = gpu_lib.create(1, 1, 1)
    bad, dpos, dneg = g.dbg_prefilter_sweep(1, 2147483647)
    assert bad == 0 and dpos <= 2 and dneg <= 1, (bad, dpos, dneg)
    g.close()


def test_synth_founders_match_specification(gpu_lib):
    cfg = SyntheticConfig(64, 1000, n_cv=77)
    ctx = gpu_lib.create(1, 1, 1)
    cfg.apply_static(ctx)
    ctx.synth_founders(0, 0, 128, 4242)
    ctx.synth_cv_founders(0, 0, 0, 128, 777)
    ctx.init_gen0(0, 64, 5)
    got = ctx.download_haps(0, 0)
    want = synth_packed(4242, 128, 1000)
    assert np.array_equal(got, want)
    gotcv = ctx.download_cv(0, 0, 0)
    assert np.array_equal(gotcv, synth_packed(777, 128, 77))
    ctx.close()


@pytest.mark.parametrize("case", ["ex1sub", "ex1mut", "dense", "mig2", "syn1k", "am1", "am2", "sel1", "vc1", "ex1full", "mig3c", "c4mini", "vt2", "vcf1", "gam2", "om1"])
def test_gpu_replays_reference_generations_bit_exact(gpu_lib, oracle_lib, case):
    fx = helpers.load_fixture(case)
    seeds = helpers.find_gen0_seeds(fx, oracle_lib)
    n_dense = helpers.replay_case(gpu_lib, fx, seeds, f"gpu/{case}")
    assert n_dense >= 1


def run_pair(gpu_lib, oracle_lib, cfg, n_gen, seed, check_every=1, stitch_mode=0, family=None, at_end=None):
    """same seeded synthetic scenario through the HIP library and the oracle"""
    g = gpu_lib.create(1, cfg.nchr, cfg.nphen)
    g.set_stitch_mode(stitch_mode)
    o = oracle_lib.create(1, cfg.nchr, cfg.nphen)
    cfg.apply_static(g); cfg.apply_static(o)
    nh = 2 * cfg.n_ind
    for c in range(cfg.nchr):
        g.synth_founders(0, c, nh, cfg.seed + c)
        o.upload_founders(0, c, synth_packed(cfg.seed + c, nh, cfg.n_loci), cfg.n_loci)
        for p in range(cfg.nphen):
            ncv = len(cfg.cv[p][c][0])
            g.synth_cv_founders(0, p, c, nh, cfg.seed + 100 + 7 * p + c)
            o.upload_cv_founders(0, p, c, synth_packed(cfg.seed + 100 + 7 * p + c, nh, ncv), ncv)
    sg = Simulation(g, seed, cfg.nchr, cfg.with_mutation)
    so = Simulation(o, seed, cfg.nchr, cfg.with_mutation)
    sg.ras_initial_human_gen0(0, cfg.n_ind); so.ras_initial_human_gen0(0, cfg.n_ind)
    assert np.array_equal(sg.sex[0], so.sex[0])
    rng = np.random.default_rng(seed)
    for gen in range(1, n_gen + 1):
        couples = synthetic_random_mate(sg.sex[0], cfg.n_ind, rng)
        if family:                                   # few couples with `family` offspring each: many gametes per parent
            couples = couples[:max(cfg.n_ind // family, 1)]
            couples["num_offspring"] = family
            couples["num_offspring"][0] += cfg.n_ind - family * len(couples)
        sg.couples[0] = couples; so.couples[0] = couples
        sx_g = sg.reproduce(0, gen); sx_o = so.reproduce(0, gen)
        assert np.array_equal(sx_g, sx_o), f"sex differs at generation {gen}"
        ag = sg.ras_compute_AD(0, gen, per_chr=True); ao = so.ras_compute_AD(0, gen, per_chr=True)
        for x, y, nm in zip(ag, ao, ("additive", "dominance", "additive_chr", "dominance_chr")):
            assert helpers.bits_equal(x, y), f"{nm} not bit-identical at generation {gen}"
        if gen % check_every == 0 or gen == n_gen:
            for c in range(cfg.nchr):
                assert np.array_equal(g.download_haps(0, c), o.download_haps(0, c)), f"dense genotypes differ (gen {gen} chr {c})"
                pg, og = g.download_intervals(0, c); po, oo = o.download_intervals(0, c)
                assert np.array_equal(og, oo) and np.array_equal(pg, po), f"interval lists differ (gen {gen} chr {c})"
                mg, mog = g.download_mutations(0, c); mo, moo = o.download_mutations(0, c)
                assert np.array_equal(mog, moo) and np.array_equal(mg, mo), f"mutation lists differ (gen {gen} chr {c})"
                for p in range(cfg.nphen):
                    assert np.array_equal(g.download_cv(0, p, c), o.download_cv(0, p, c))
                    assert helpers.bits_equal(g.get_cv_freq(0, p, c), o.get_cv_freq(0, p, c))
    if at_end is not None:
        at_end(g)
    g.close(); o.close()


def test_gpu_vs_oracle_task_parallel_mode(gpu_lib, oracle_lib):
    # hot maps: ~10 crossovers and ~10 mutations per gamete, 2 chromosomes, 2 phenotypes, dominance on
    cfg = SyntheticConfig(300, 5000, nchr=2, chrom_bp=2_000_000, map_step=1000, rec_per_row=5e-3, mut_per_row=5e-3,
                          n_cv=300, nphen=2, seed=11, vd=0.4)
    run_pair(gpu_lib, oracle_lib, cfg, n_gen=5, seed=2024)


@pytest.mark.parametrize("mode", [1])
def test_alternative_stitch_kernels_give_the_same_state(gpu_lib, oracle_lib, mode):
    """the production stitch is k_stitch_segments (mode 0, every other test: one workgroup per entry of the list of segments to
    write); the gamete-major k_stitch_rows (1) finds its segments and sources itself -- an independent formulation of the same rule"""
    cfg = SyntheticConfig(300, 5000, nchr=2, chrom_bp=2_000_000, map_step=1000, rec_per_row=5e-3, mut_per_row=5e-3, n_cv=100, seed=12)
    run_pair(gpu_lib, oracle_lib, cfg, n_gen=3, seed=31, stitch_mode=mode)
    # many boundaries per row and > 256 boundaries for single gametes (the LDS list of the parent-major kernels overflows)
    cfg = SyntheticConfig(40, 40000, nchr=1, chrom_bp=4_000_000, map_step=1000, rec_per_row=0.1, mut_per_row=0.05, n_cv=200, seed=3, vd=0.1)
    run_pair(gpu_lib, oracle_lib, cfg, n_gen=2, seed=7, stitch_mode=mode)


def _read_device_u32(ptr, n):
    """copy n uint32 from a device pointer (the library's slot -> row table)"""
    hip = C.CDLL("libamdhip64.so")
    out = np.empty(n, dtype=np.uint32)
    rc = hip.hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(ptr), C.c_size_t(4 * n), C.c_int(2))      # hipMemcpyDeviceToHost
    assert rc == 0, f"hipMemcpy failed ({rc})"
    return out


@pytest.mark.parametrize("max_ranges,arena,literal,lanes", [(32, 0, 0, 8), (32, 0, 1, 1), (32, 0, 0, 1), (1, 0, 0, 8), (5, 0, 1, 1), (32, 6, 0, 8), (7, 3, 0, 8)])
def test_lists_as_shared_pieces_per_position_range(gpu_lib, oracle_lib, monkeypatch, max_ranges, arena, literal, lanes):
    """the mutation / interval lists live as pieces per position range, shared between parent and offspring where a gamete has
    no crossover and no new mutation (csrc/gev_lists.h); the whole lists every other function reads are made from them on demand.
    Compared with the oracle's lists every generation (or every third: generations in between are then never materialised):
    32 ranges (default), ONE range (= a piece is a whole list), 5 and 7 ranges (boundaries and mutations on range edges shift),
    and arenas of 6 / 3 entries per row, which fill up every few generations and are compacted (pieces nobody names are
    dropped, the sharing is kept).  Hot maps (many crossovers and new mutations per row: every range has events, inserts inside
    ranges) and cold maps (most ranges name the parent's piece)."""
    monkeypatch.setenv("GEV_LIST_SEGS", str(max_ranges))
    monkeypatch.setenv("GEV_LP_LITERAL", str(literal))       # interval pieces built by recombine's statements one by one (1) or by their closed form (0, default)
    monkeypatch.setenv("GEV_LP_LANES", str(lanes))           # ... by a group of eight lanes per piece (default) or by one lane
    if arena:
        monkeypatch.setenv("GEV_LIST_ARENA", str(arena))
    seen = {}

    def stats(g):
        seen.update(g.list_stats(0, 0))
    cfg = SyntheticConfig(200, 5000, nchr=2, chrom_bp=2_000_000, map_step=1000, rec_per_row=5e-3, mut_per_row=8e-3, n_cv=100, seed=61)
    run_pair(gpu_lib, oracle_lib, cfg, n_gen=6, seed=62, at_end=stats)
    assert 1 <= seen["ranges_per_row"] <= max_ranges and (max_ranges != 32 or seen["ranges_per_row"] == 31) and seen["rebuilds"] == 1   # 2 Mb in 2^16-bp ranges: 31 of them
    cfg = SyntheticConfig(200, 5000, nchr=1, chrom_bp=2_000_000, map_step=1000, rec_per_row=2e-4, mut_per_row=2e-3, n_cv=100, seed=63)
    run_pair(gpu_lib, oracle_lib, cfg, n_gen=12, seed=64, check_every=3, at_end=stats)
    # a map whose rows are 3 bp apart: breakpoints of different rows coincide all the time (zero-length parts, parts that start
    # exactly at a breakpoint -- the equality cases of recombine's comparisons)
    cfg = SyntheticConfig(60, 600, nchr=1, chrom_bp=30000, map_step=3, rec_per_row=6e-4, mut_per_row=4e-4, n_cv=40, seed=65)
    run_pair(gpu_lib, oracle_lib, cfg, n_gen=8, seed=66)
    if arena:
        assert seen["compactions"] >= 2, seen
    elif max_ranges == 32:
        # cold maps: ~0.4 crossovers and ~4 new mutations per row and generation: far fewer new interval entries than rows x ranges
        assert seen["interval_entries_last_generation"] < 400 * 31 // 4, seen


@pytest.mark.parametrize("mode,alias,seg_chunks", [(0, 1, 16), (1, 1, 16), (0, 0, 16), (1, 0, 64), (0, 1, 1024), (0, 1, 4)])
def test_segments_without_a_crossover_share_the_parental_unit(gpu_lib, oracle_lib, monkeypatch, mode, alias, seg_chunks):
    """Cold maps: ~0.3 crossovers per gamete.  A row is kept as segments; in a segment that contains none of its boundaries an
    offspring gamete is one parental haplotype unchanged (Simulation::recombine copies the parent's parts between two crossovers,
    src/Simulation.cpp:2939-2946, and returns the parental Hap when there is none, :2910), so the offspring's table entry names
    the parent's unit and nothing is copied; after several generations units are shared along chains of ancestors.  Everything
    observable must equal the oracle in both stitch kernels, for tiny segments (16 chunks = 2048 loci: 15 segments per row, many
    boundaries on segment edges), whole-row segments and 64-byte segments (60 per row, the 64-segment limit), and equal the run that writes every segment
    (GEV_ALIAS_ROWS=0)."""
    monkeypatch.setenv("GEV_ALIAS_ROWS", str(alias)); monkeypatch.setenv("GEV_SEG_CHUNKS", str(seg_chunks))
    cfg = SyntheticConfig(400, 30000, nchr=3, chrom_bp=3_000_000, map_step=1000, rec_per_row=1e-4, mut_per_row=3e-4, n_cv=120, nphen=2, seed=41, vd=0.3)
    g = gpu_lib.create(1, cfg.nchr, cfg.nphen); g.set_stitch_mode(mode)
    o = oracle_lib.create(1, cfg.nchr, cfg.nphen)
    cfg.apply_static(g); cfg.apply_static(o)
    nh = 2 * cfg.n_ind
    for c in range(cfg.nchr):
        g.synth_founders(0, c, nh, cfg.seed + c); o.upload_founders(0, c, synth_packed(cfg.seed + c, nh, cfg.n_loci), cfg.n_loci)
        for p in range(cfg.nphen):
            ncv = len(cfg.cv[p][c][0])
            g.synth_cv_founders(0, p, c, nh, cfg.seed + 100 + 7 * p + c); o.upload_cv_founders(0, p, c, synth_packed(cfg.seed + 100 + 7 * p + c, nh, ncv), ncv)
    sg = Simulation(g, 77, cfg.nchr, True); so = Simulation(o, 77, cfg.nchr, True)
    sg.ras_initial_human_gen0(0, cfg.n_ind); so.ras_initial_human_gen0(0, cfg.n_ind)
    rng = np.random.default_rng(5)
    sizes = [400, 400, 520, 520, 300, 300, 300, 300]          # growth (the pool is reallocated) and shrinkage on the way
    for gen, n in enumerate(sizes, 1):
        couples = synthetic_random_mate(sg.sex[0], n, rng)
        sg.couples[0] = couples; so.couples[0] = couples
        before = []
        for c in range(cfg.nchr):
            _, _, n_slots, tab, nseg = g.plane_ptr(0, c)
            before.append(_read_device_u32(tab, n_slots * nseg))
        _, _, sw0, st0 = g.stitch_totals()
        assert np.array_equal(sg.reproduce(0, gen, n_people=n), so.reproduce(0, gen, n_people=n)), f"sex differs at generation {gen}"
        bw1, bt1, sw1, st1 = g.stitch_totals()
        ag = sg.ras_compute_AD(0, gen, per_chr=True); ao = so.ras_compute_AD(0, gen, per_chr=True)
        for x, y in zip(ag, ao):
            assert helpers.bits_equal(x, y), f"A/D not bit-identical at generation {gen}"
        n_shared = n_units = 0
        for c in range(cfg.nchr):
            _, unit_bytes, n_slots, tab, nseg = g.plane_ptr(0, c)
            assert n_slots == 2 * n and unit_bytes == 16 * min(seg_chunks, 256) and nseg == -(-3840 // unit_bytes)      # rows of 30000 bits = 3840 bytes = 240 chunks (a longer segment shrinks to 256 chunks)
            after = _read_device_u32(tab, n_slots * nseg)
            inherited = np.isin(after, before[c])
            fresh = after[~inherited]
            assert len(np.unique(fresh)) == len(fresh), "two written segments were given the same pool unit"
            n_shared += int(inherited.sum()); n_units += len(after)
            assert np.array_equal(g.download_haps(0, c), o.download_haps(0, c)), f"dense genotypes differ (gen {gen} chr {c})"
            pg, og = g.download_intervals(0, c); po, oo = o.download_intervals(0, c)
            assert np.array_equal(og, oo) and np.array_equal(pg, po)
            mg, mog = g.download_mutations(0, c); mo, moo = o.download_mutations(0, c)
            assert np.array_equal(mog, moo) and np.array_equal(mg, mo)
            for ph in range(cfg.nphen):
                assert np.array_equal(g.download_cv(0, ph, c), o.download_cv(0, ph, c))
        assert st1 - st0 == n_units
        assert (st1 - st0) - (sw1 - sw0) == n_shared, "segments reported as not written != table entries that name a parental unit"
        if alias:
            assert n_shared > 0.5 * n_units, "expected most segments to be boundary-free with this map"
        else:
            assert n_shared == 0
    assert 0 < bw1 <= bt1
    assert g.dbg_verify_planes(0, 1, [cfg.seed + 1]) == (0, 0)
    g.close(); o.close()


@pytest.mark.parametrize("setting", ["auto", "8", "5"])
def test_stitch_occupancy_settings_do_not_change_results(gpu_lib, oracle_lib, monkeypatch, setting):
    """GEV_STITCH_WG_PER_CU limits the dense stitch's workgroups per CU (dynamic LDS padding); `auto` switches between 8, 7 and 6
    while it measures.  Rows of 64 KiB (the long-row kernel), 16 generations so that the tuner walks through every candidate."""
    monkeypatch.setenv("GEV_STITCH_WG_PER_CU", setting)
    cfg = SyntheticConfig(150, 524288, nchr=1, chrom_bp=60_000_000, map_step=50_000, rec_per_row=8e-4, mut_per_row=5e-4, n_cv=64, seed=51)
    run_pair(gpu_lib, oracle_lib, cfg, n_gen=16 if setting == "auto" else 3, seed=52, check_every=8)


def test_population_without_any_crossover(gpu_lib, oracle_lib):
    """recombination probability 0 everywhere: every offspring haplotype is a parental one, the dense stitch has nothing to copy"""
    cfg = SyntheticConfig(200, 10000, nchr=2, chrom_bp=1_000_000, map_step=1000, rec_per_row=0.0, mut_per_row=1e-3, n_cv=50, seed=43)
    run_pair(gpu_lib, oracle_lib, cfg, n_gen=4, seed=9)
    g = gpu_lib.create(1, cfg.nchr, cfg.nphen)
    cfg.apply_static(g)
    for c in range(cfg.nchr):
        g.synth_founders(0, c, 2 * cfg.n_ind, cfg.seed + c); g.synth_cv_founders(0, 0, c, 2 * cfg.n_ind, cfg.seed + 100 + c)
    sg = Simulation(g, 9, cfg.nchr, True)
    sg.ras_initial_human_gen0(0, cfg.n_ind)
    rng = np.random.default_rng(9)
    for gen in range(1, 4):
        sg.couples[0] = synthetic_random_mate(sg.sex[0], cfg.n_ind, rng)
        sg.reproduce(0, gen)
    bw, bt, sw, st = g.stitch_totals()
    assert (bw, sw) == (0, 0) and st == 3 * 2 * cfg.n_ind * cfg.nchr and bt == st * 1280          # one segment per row of 10000 loci = 1280 bytes
    g.close()


def test_large_families_many_gametes_per_parent(gpu_lib, oracle_lib):
    # 40 offspring per couple: 40 gametes per parent -> several PM_GMAX batches in the parent-major kernel
    cfg = SyntheticConfig(240, 20000, nchr=1, chrom_bp=2_000_000, map_step=1000, rec_per_row=4e-3, mut_per_row=2e-3, n_cv=50, seed=13)
    run_pair(gpu_lib, oracle_lib, cfg, n_gen=3, seed=32, family=40)


@pytest.mark.parametrize("wg", [1, 0])
def test_gpu_vs_oracle_serial_chain_mode(gpu_lib, oracle_lib, monkeypatch, wg):
    """no mutation map: gamete seeds chain through crossover counts (Example1-style; src/Simulation.cpp:2447-2455).  wg = 1: a workgroup
    per link of the chain (k_rec_chain_wg: one wave seeds glibc's generator while eight scan the map), 0: the one-wave form.  Cold
    and hot maps (more hits than a scanning wave's list holds: the link falls back to the one-wave walk; more than GEV_BK_CAP
    crossovers per gamete: overflow records), maps longer than one 2048-row round, thresholds in LDS and (many rows) in global memory."""
    monkeypatch.setenv("GEV_CHAIN_WG", str(wg))
    cfg = SyntheticConfig(200, 3000, nchr=3, chrom_bp=1_000_000, map_step=1000, rec_per_row=3e-3, n_cv=100, seed=5, with_mutation=False)   # 3 x 1001 rows: thresholds from global memory
    run_pair(gpu_lib, oracle_lib, cfg, n_gen=4, seed=99)
    cfg = SyntheticConfig(60, 3000, nchr=1, chrom_bp=1_000_000, map_step=400, rec_per_row=3e-3, with_mutation=False, n_cv=100, seed=6)      # 2501 rows: two rounds
    run_pair(gpu_lib, oracle_lib, cfg, n_gen=2, seed=98)
    cfg = SyntheticConfig(60, 3000, nchr=1, chrom_bp=1_000_000, map_step=1000, rec_per_row=2e-3, with_mutation=False, n_cv=100, seed=8)     # 1001 rows: thresholds in LDS
    run_pair(gpu_lib, oracle_lib, cfg, n_gen=3, seed=96)
    cfg = SyntheticConfig(30, 3000, nchr=2, chrom_bp=400_000, map_step=1000, rec_per_row=0.4, with_mutation=False, n_cv=100, seed=7)       # ~160 crossovers per gamete
    run_pair(gpu_lib, oracle_lib, cfg, n_gen=2, seed=97)


def test_serial_chain_refuses_sizes_that_would_run_for_minutes(gpu_lib, monkeypatch):
    """GEV_EUNSUPPORTED with a reference-style message above GEV_CHAIN_MAX_TASKS (offspring x chromosomes without a mutation map)"""
    from geneevolve_amd.capi import GevError
    monkeypatch.setenv("GEV_CHAIN_MAX_TASKS", "100")
    cfg = SyntheticConfig(120, 3000, nchr=1, chrom_bp=1_000_000, map_step=2000, rec_per_row=8e-3, with_mutation=False, n_cv=50, seed=5)
    g = gpu_lib.create(1, 1, 1)
    cfg.apply_static(g)
    g.synth_founders(0, 0, 240, 5); g.synth_cv_founders(0, 0, 0, 240, 6)
    sg = Simulation(g, 3, 1, False)
    sg.ras_initial_human_gen0(0, 120)
    sg.couples[0] = synthetic_random_mate(sg.sex[0], 120, np.random.default_rng(1))
    with pytest.raises(GevError) as e:
        sg.reproduce(0, 1)
    assert e.value.code == -5 and "serial rand() chain" in str(e.value)
    monkeypatch.setenv("GEV_CHAIN_MAX_TASKS", "1000")
    sg.reproduce(0, 1)
    g.close()


def test_gpu_vs_oracle_many_crossovers_per_gamete(gpu_lib, oracle_lib):
    # > 64 rand() outputs per srand (second output block) and > STITCH_KMAX boundaries per row
    cfg = SyntheticConfig(40, 40000, nchr=1, chrom_bp=4_000_000, map_step=1000, rec_per_row=0.1, mut_per_row=0.05, n_cv=200, seed=3, vd=0.1)
    run_pair(gpu_lib, oracle_lib, cfg, n_gen=3, seed=7)


def test_config1_shape_against_oracle(gpu_lib, oracle_lib):
    # BASELINE config 1 shape: 1k individuals x 10k SNPs, 1 chr of 100 Mb, 2001 map rows, 1000 CVs
    cfg = SyntheticConfig(1000, 10000, seed=12345)
    run_pair(gpu_lib, oracle_lib, cfg, n_gen=10, seed=12345, check_every=5)


def test_edge_cases(gpu_lib, oracle_lib):
    cfg = SyntheticConfig(8, 130, nchr=1, chrom_bp=100_000, map_step=1000, rec_per_row=0.02, mut_per_row=0.02, n_cv=3, seed=9)
    g = gpu_lib.create(1, 1, 1); o = oracle_lib.create(1, 1, 1)
    for ctx in (g, o):
        cfg.apply_static(ctx)
    g.synth_founders(0, 0, 16, 1); o.upload_founders(0, 0, synth_packed(1, 16, 130), 130)
    g.synth_cv_founders(0, 0, 0, 16, 2); o.upload_cv_founders(0, 0, 0, synth_packed(2, 16, 3), 3)
    g.init_gen0(0, 8, 77); o.init_gen0(0, 8, 77)
    # ragged families: zero-offspring couples, inbred couples skipped, one big family, population shrinks then grows
    couples = np.array([[0, 1, 0, 0], [2, 3, 1, 5], [4, 5, 0, 7], [6, 7, 0, 1], [1, 0, 0, 2]])
    gs = GlobSeedStream(5).draw(1 + 10)
    assert np.array_equal(g.reproduce(0, couples, gs[0], gs[1:]), o.reproduce(0, couples, gs[0], gs[1:]))
    assert g.pop_size(0) == 10
    assert np.array_equal(g.download_haps(0, 0), o.download_haps(0, 0))
    couples = np.array([[i % 10, (i * 3 + 1) % 10, 0, 3] for i in range(12)])
    gs = GlobSeedStream(6).draw(1 + 36)
    assert np.array_equal(g.reproduce(0, couples, gs[0], gs[1:]), o.reproduce(0, couples, gs[0], gs[1:]))
    assert np.array_equal(g.download_haps(0, 0), o.download_haps(0, 0))
    # error behaviour: wrong n_people / out-of-range positions / bad seeds count are refused, state untouched
    with pytest.raises(capi.GevError):
        g.reproduce(0, np.array([[0, 99, 0, 1]]), 1, np.array([1], dtype=np.uint32))
    with pytest.raises(capi.GevError):
        g.reproduce(0, np.array([[0, 1, 0, 2]]), 1, np.array([1], dtype=np.uint32))
    assert g.pop_size(0) == 36
    g.close(); o.close()


def test_full_row_properties_at_scale(gpu_lib):
    """size-independent properties where the oracle would be too slow: 20k individuals x 200k loci.
    With recombination and mutation switched off every offspring haplotype must equal one of its
    parent's two haplotypes (k = 0 path), and allele counts are conserved under 2 offspring/couple."""
    n, L = 20000, 200000
    cfg = SyntheticConfig(n, L, rec_per_row=0.0, mut_per_row=0.0, n_cv=100, seed=1)
    g = gpu_lib.create(1, 1, 1)
    cfg.apply_static(g)
    g.synth_founders(0, 0, 2 * n, 1); g.synth_cv_founders(0, 0, 0, 2 * n, 2)
    sex = g.init_gen0(0, n, 1)
    parents = g.download_haps(0, 0, 0, 2000)
    rng = np.random.default_rng(1)
    fa = rng.integers(0, 1000, n); mo = rng.integers(0, 1000, n)
    c = np.stack([fa, mo, np.zeros(n, dtype=np.int64), np.ones(n, dtype=np.int64)], axis=1)
    seeds = GlobSeedStream(3).draw(1 + n)
    g.reproduce(0, c, seeds[0], seeds[1:])
    kids = g.download_haps(0, 0, 0, 4000)
    for i in range(2000):
        for s, par in ((0, fa[i]), (1, mo[i])):
            row = kids[2 * i + s]
            assert np.array_equal(row, parents[2 * par]) or np.array_equal(row, parents[2 * par + 1])
    g.close()


def test_two_rank_migration_on_gpu_matches_reference_fixture():
    """two processes share the GPU; packed records are produced/consumed by the HIP library
    (gev_export_rows / gev_remove_rows / gev_import_rows) and exchanged with all_to_all over gloo"""
    from tests import dist_worker
    res = dist_worker.launch("gpu")
    for r, msg in res:
        assert msg == "ok", f"rank {r}: {msg}"


def test_two_rank_migration_fixture_with_the_per_haplotype_effect_lookup(monkeypatch):
    """the same run with GEV_AD_RP_FAST=0: k_ad_accumulate (one a / d lookup per haplotype and CV in global memory) instead of the
    piecewise kernel that position-ordered CV files get"""
    monkeypatch.setenv("GEV_AD_RP_FAST", "0")
    from tests import dist_worker
    res = dist_worker.launch("gpu")
    for r, msg in res:
        assert msg == "ok", f"rank {r}: {msg}"


def test_locus_split_population_on_gpu_matches_the_unsplit_reference_run():
    """gev_set_chr_active: one population split along chromosomes over two processes (BASELINE config 4's layout; both share
    the test box's GPU, gloo carries the per-chromosome A/D all-reduce).  Each context is given genotype / CV inputs of its
    own chromosomes only; sex (the seed chain spans all chromosomes), per-chromosome and total A/D, dense genotypes and
    interval offsets equal the reference's unsplit run (fixture ex1mut) bit for bit."""
    from tests import dist_worker
    res = dist_worker.launch("gpu", target=dist_worker.run_locus_split)
    for r, msg in res:
        assert msg == "ok", f"rank {r}: {msg}"


def test_two_rank_migration_beyond_fixture_size_with_full_plane_verification():
    """see tests/dist_worker.py:run_migration_at_scale"""
    from tests import dist_worker
    res = dist_worker.launch("gpu", world=2, target=dist_worker.run_migration_at_scale)
    for r, msg in res:
        assert msg == "ok", f"rank {r}: {msg}"


def test_two_rank_migration_with_migrants_as_lists_matches_reference_fixture():
    """gev_set_migrant_rows(0) on the reference's `mig2` run: the records carry no genotype rows, the importing rank rebuilds them
    from the founder panels of both root populations (gev_upload_founder_panel) by the rule of
    Simulation::ras_convert_interval_to_hap_matrix (src/Simulation.cpp:1198-1211); couples, sexes, A/D, lists and the dense
    matrices after every migration step equal the reference's, as with rows in the payload"""
    from tests import dist_worker
    res = dist_worker.launch("gpu_lists")
    for r, msg in res:
        assert msg == "ok", f"rank {r}: {msg}"


def test_two_rank_migration_as_lists_beyond_fixture_size_with_full_plane_verification():
    """tests/dist_worker.py:run_migration_at_scale with the migrants travelling as lists: after three exchanges (5 % of each
    population, both directions, so parts of both root populations come back) and one more generation every word of the resident
    planes equals the materialised interval state; the oracle in each rank's shadow moves whole rows"""
    from tests import dist_worker
    res = dist_worker.launch("gpu_lists", world=2, target=dist_worker.run_migration_at_scale)
    for r, msg in res:
        assert msg == "ok", f"rank {r}: {msg}"


def test_config4_in_miniature_with_migrants_as_lists():
    """fixture `c4mini` (22 autosomes of the reference's own map, assortative mating, mutation, two populations with migration,
    each split 11 | 11 chromosomes over two processes): the exchange between the processes that hold the same chromosomes carries
    lists only, every shard rebuilds the immigrants' rows of its chromosomes from the panels of both root populations"""
    from tests import dist_worker
    res = dist_worker.launch("gpu_lists", world=4, target=dist_worker.run_split_migration_c4mini)
    for r, msg in res:
        assert msg == "ok", f"rank {r}: {msg}"


def test_migrants_as_lists_payload_size_and_refusals(gpu_lib):
    """one context, two populations: the lists-only record is smaller by exactly the rows; an import without the root population's
    panel, or with the other end's setting, is refused; with the panel the imported rows equal the exported individuals' rows"""
    from geneevolve_amd.capi import GevError
    hip = C.CDLL("libamdhip64.so.7")                        # the HIP runtime the library itself is linked to (already loaded with it): a plain device buffer
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipFree.argtypes = [C.c_void_p]
    n, L = 64, 5000
    cfg = SyntheticConfig(n, L, chrom_bp=2_000_000, map_step=1000, rec_per_row=1e-3, mut_per_row=1e-3, n_cv=16, seed=5)
    g = gpu_lib.create(2, 1, 1, 0)
    for p in range(2):
        cfg.apply_static(g, p)
        g.synth_founders(p, 0, 2 * n, 40 + p); g.synth_cv_founders(p, 0, 0, 2 * n, 50 + p)
    seeds = GlobSeedStream(9)
    sex = [g.init_gen0(p, n, int(seeds.draw(1)[0])) for p in range(2)]
    rng = np.random.default_rng(0)
    for gen in range(3):
        for p in range(2):
            sd = seeds.draw(1 + n)
            sex[p] = g.reproduce(p, synthetic_random_mate(sex[p], n, rng), int(sd[0]), sd[1:])
    who = np.array([3, 17, 40], dtype=np.uint64)
    want = g.download_haps(0, 0)[np.stack([2 * who, 2 * who + 1], 1).ravel().astype(np.int64)]
    full = g.export_size(0, who)
    g.set_migrant_rows(False)
    small = g.export_size(0, who)
    stride = -(-L // 1024) * 128                                                  # bytes of a flat row: multiple of 128
    assert full - small == 2 * len(who) * stride, (full, small, stride)
    dbuf = C.c_void_p()
    assert hip.hipMalloc(C.byref(dbuf), small) == 0 and dbuf.value
    buf = dbuf.value
    g.export_rows(0, who, buf, small)
    with pytest.raises(GevError, match="founder panel is not held here"):
        g.import_rows(1, buf, small, len(who))
    g.close()
    # again with the panels: rows come out as they went in; with the other end's setting the record's size gives it away
    g = gpu_lib.create(2, 1, 1, 0)
    for p in range(2):
        cfg.apply_static(g, p)
        g.synth_founders(p, 0, 2 * n, 40 + p); g.synth_cv_founders(p, 0, 0, 2 * n, 50 + p); g.synth_founder_panel(p, 0, 2 * n, 40 + p)
    seeds = GlobSeedStream(9)
    sex = [g.init_gen0(p, n, int(seeds.draw(1)[0])) for p in range(2)]
    rng = np.random.default_rng(0)
    for gen in range(3):
        for p in range(2):
            sd = seeds.draw(1 + n)
            sex[p] = g.reproduce(p, synthetic_random_mate(sex[p], n, rng), int(sd[0]), sd[1:])
    assert np.array_equal(g.download_haps(0, 0)[np.stack([2 * who, 2 * who + 1], 1).ravel().astype(np.int64)], want)
    g.set_migrant_rows(False)
    g.export_rows(0, who, buf, small)
    g.set_migrant_rows(True)
    with pytest.raises(GevError, match="expected from its header"):
        g.import_rows(1, buf, small, len(who))
    g.set_migrant_rows(False)
    g.import_rows(1, buf, small, len(who))
    assert g.pop_size(1) == n + len(who)
    got = g.download_haps(1, 0)
    assert np.array_equal(got[2 * n:], want), "rebuilt rows (mutations applied) != the emigrants' rows"
    assert g.dbg_verify_planes(1, 0, [40, 41]) == (0, 0)
    g.close()
    hip.hipFree(dbuf)


def test_rccl_backend_single_rank_collectives_drive_the_device_buffer_branch():
    """The test box has ONE GPU, so RCCL cannot carry a two-rank exchange here; a single-rank "nccl" process group still runs the
    collectives of the device-buffer branch for real (all_to_all_single of the size table on a device tensor, all_reduce of the
    per-chromosome A/D on the GPU, the fences between torch's stream and the library's streams).  With one rank nobody can
    emigrate, so the payload exchange is the empty case; the population must come out untouched.  Runs in a fresh process, as
    a rank of a real job would (tests/rccl_single_rank.py)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_single_rank.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl single rank ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_two_populations_times_two_chromosome_shards_with_migration_on_gpu():
    """BASELINE config 4's layout: 2 populations x 2 chromosome shards = 4 processes driving the HIP library (they share the test
    box's GPU; gloo carries the A/D all-reduce and the migration all-to-all).  Split contexts exchange migrants (records hold
    the active chromosomes only) and scale phenotypes from the all-reduced totals (gev_set_ad): everything equals the real
    reference's unsplit run (fixture mig3c)."""
    from tests import dist_worker
    res = dist_worker.launch("gpu", world=4, target=dist_worker.run_split_migration)
    for r, msg in res:
        assert msg == "ok", f"rank {r}: {msg}"


def test_config4_in_miniature_on_gpu():
    """composite of BASELINE config 4 at fixture scale (c4mini, from the real reference): 22 autosomes on the reference's own genetic
    map, assortative mating (couples from the fixture), mutation map, two populations with migration, each population split
    11 | 11 chromosomes over two processes"""
    from tests import dist_worker
    res = dist_worker.launch("gpu", world=4, target=dist_worker.run_split_migration_c4mini)
    for r, msg in res:
        assert msg == "ok", f"rank {r}: {msg}"


def test_baseline_config2_full_size_dense_state_equals_interval_state(gpu_lib):
    """BASELINE config 2 at FULL size (100k individuals x 1M SNPs; the oracle would need hours):
    after three generations the dense genotype rows produced by the stitch kernel must equal what the
    reference's materialisation rule (ras_convert_interval_to_hap_matrix, src/Simulation.cpp:1186-1230)
    gives from the device's own ancestry intervals + mutation sets + the founder panel -- checked
    for EVERY row on the device (gev_dbg_verify_planes) and, including the mutation overlay, on sampled rows whose
    founder rows are regenerated independently by tests/synth.py."""
    from tests.synth import synth_bits
    n, L = 100_000, 1_000_000
    cfg = SyntheticConfig(n, L, seed=12345)
    g = gpu_lib.create(1, 1, 1)
    cfg.apply_static(g)
    g.synth_founders(0, 0, 2 * n, 4711); g.synth_cv_founders(0, 0, 0, 2 * n, 4712)
    sim = Simulation(g, 2024, 1, True)
    sim.ras_initial_human_gen0(0, n)
    rng = np.random.default_rng(5)
    for gen in (1, 2, 3):
        sim.couples[0] = synthetic_random_mate(sim.sex[0], n, rng)
        sim.reproduce(0, gen)
        add, dom, _, _ = sim.ras_compute_AD(0, gen)
        assert np.isfinite(add).all() and add.std() > 0
    # FULL coverage on the device: every one of the 200 000 x 31 250 plane words (dense stitch, from breakpoints) against the
    # materialisation rule applied to the interval state (interval kernel) and the regenerated founder panel
    assert g.dbg_verify_planes(0, 0, 4711) == (0, 0)
    assert g.dbg_verify_planes(0, 0, 4712)[0] > 10 ** 9          # negative control: a different founder panel does not match
    parts, off = g.download_intervals(0, 0)
    muts, moff = g.download_mutations(0, 0)
    assert off[-1] > 2 * n * 1.5                                  # recombination happened
    assert moff[-1] > n * 0.5                                     # mutations accumulated
    pos = cfg.snp_pos
    bp0, bpe = int(cfg.rmap_bp[0]), int(cfg.rmap_bp[-1])
    rows = rng.choice(2 * n, size=24, replace=False)
    for r in rows:
        got = capi.unpack_rows(g.download_haps(0, 0, int(r), 1), L)[0]
        want = np.zeros(L, dtype=np.uint8)
        cover = np.zeros(L, dtype=bool)
        for p in parts[int(off[r]):int(off[r + 1])]:
            lo, hi = np.searchsorted(pos, int(p["st"])), np.searchsorted(pos, int(p["en"]))
            f = synth_bits(4711, 1, L, row_begin=int(p["hap_index"]))[0]
            want[lo:hi] = f[lo:hi]; cover[lo:hi] = True
        assert cover[np.searchsorted(pos, bp0):np.searchsorted(pos, bpe)].all()
        flipped = set()
        for x in muts[int(moff[r]):int(moff[r + 1])]:
            j = int(np.searchsorted(pos, int(x)))
            if j < L and pos[j] == x and j not in flipped:       # set semantics: one flip per position however often it was hit
                want[j] = 1 - want[j]; flipped.add(j)
        assert np.array_equal(got, want), f"row {r}: dense state != interval state"
    g.close()


def test_three_hundred_generations_of_shared_rows_dense_state_equals_interval_state(gpu_lib):
    """A long run at a size where the stitch overlaps the next generation's kernels (20k individuals x 256k SNPs, no synchronisation
    between generations): rows are handed down along chains of crossover-free gametes for many generations, the row pool is
    rebuilt 300 times, the lists pass the length where the fill kernels switch to eight lanes per row, and the population size
    wobbles (pool growth).  Every word of the final genotype rows must equal the materialised interval state, the list invariants
    must hold, and a founder haplotype block that survived must still be there bit for bit."""
    L = 262_144
    cfg = SyntheticConfig(20_000, L, chrom_bp=100_000_000, n_cv=200, seed=71)
    g = gpu_lib.create(1, 1, 1)
    cfg.apply_static(g)
    n = 20_000
    g.synth_founders(0, 0, 2 * n, 8001); g.synth_cv_founders(0, 0, 0, 2 * n, 8002)
    sim = Simulation(g, 4242, 1, True)
    sim.ras_initial_human_gen0(0, n)
    rng = np.random.default_rng(17)
    for gen in range(1, 301):
        n = 20_000 + (gen % 7) * 500 if gen % 50 else 24_000            # a few growth steps beyond the current capacity
        sim.couples[0] = synthetic_random_mate(sim.sex[0], n, rng)
        sim.reproduce(0, gen, n_people=n)
        if gen % 100 == 0:
            assert g.dbg_verify_planes(0, 0, 8001) == (0, 0), f"dense state != interval state after {gen} generations"
    copied, total, _, _ = g.stitch_totals()
    assert 0.04 < copied / total < 0.30                                   # sixteen 2 KiB segments per row, a sixteenth of a Morgan each: 1 - e^-0.0625 = 0.06 of them hold a boundary (four 8 KiB segments: 0.22)
    parts, off = g.download_intervals(0, 0)
    muts, moff = g.download_mutations(0, 0)
    assert len(off) == 2 * n + 1 and off[-1] == len(parts) and moff[-1] == len(muts)
    st = parts["st"].astype(np.int64); en = parts["en"].astype(np.int64)
    first = off[:-1].astype(np.int64); last = off[1:].astype(np.int64) - 1
    assert (st[first] == int(cfg.rmap_bp[0])).all() and (en[last] == int(cfg.rmap_bp[-1])).all()      # every row tiles the map range
    inner = np.ones(len(parts), dtype=bool); inner[last] = False
    assert (en[inner] == st[1:][inner[:-1]]).all()
    assert off[-1] / (2 * n) > 50                                          # ~ one new part per row and generation: the eight-lane fill form was in use
    for r in range(0, 2 * n, 4001):
        m = muts[int(moff[r]):int(moff[r + 1])]
        assert (np.diff(m.astype(np.int64)) >= 0).all()
    g.close()


@pytest.mark.parametrize("head_start", ["0", "1"])
def test_three_hundred_one_call_generations_dense_state_equals_interval_state(gpu_lib, monkeypatch, head_start):
    """bench.py's loop -- gev_generation_begin / gev_generation_end with the head start across generations, the next generation
    handed over before this one's A/D is read -- for 300 generations at a size where the streams overlap (20k individuals x 256k
    SNPs), with list arenas small enough to be compacted many times on the way and a population size that changes now and then
    (a head start made for another size is dropped).  Every word of the genotype rows must equal the materialised interval state
    at generations 100, 200 and 300, the lists must tile the map, A/D must be finite and vary."""
    monkeypatch.setenv("GEV_LIST_ARENA", "48"); monkeypatch.setenv("GEV_HEAD_START", head_start)
    L = 262_144
    cfg = SyntheticConfig(20_000, L, chrom_bp=100_000_000, n_cv=200, seed=72)
    g = gpu_lib.create(1, 1, 1)
    cfg.apply_static(g)
    n = 20_000
    g.synth_founders(0, 0, 2 * n, 9001); g.synth_cv_founders(0, 0, 0, 2 * n, 9002)
    sim = Simulation(g, 777, 1, True)
    sim.ras_initial_human_gen0(0, n)
    g.set_generation_chain(0)
    state = sim.glob.x
    size = lambda gen: 20_000 if gen % 40 else 21_000 + 100 * (gen // 40)
    gen, pending = 0, False
    while gen < 300:
        if not pending:
            g.generation_begin(0, state, size(gen + 1))
        r = g.generation_end(); pending = False; gen += 1
        state = int(r["glob_state"])
        verify = gen % 100 == 0
        if gen < 300 and not verify:                               # the next generation is in flight while this one's A/D is read
            g.generation_begin(0, state, size(gen + 1)); pending = True
        ad = g.compute_ad(0, per_chr=False)
        assert np.isfinite(ad[0]).all() and np.var(ad[0]) > 0, gen
        if verify:                                                 # (not allowed while a generation is pending)
            assert g.dbg_verify_planes(0, 0, 9001) == (0, 0), f"dense state != interval state after {gen} generations"
    st = g.list_stats(0, 0)
    assert st["compactions"] >= 3, st
    parts, off = g.download_intervals(0, 0)
    n = g.pop_size(0)
    assert len(off) == 2 * n + 1 and off[-1] == len(parts)
    pst = parts["st"].astype(np.int64); pen = parts["en"].astype(np.int64)
    first = off[:-1].astype(np.int64); last = off[1:].astype(np.int64) - 1
    assert (pst[first] == int(cfg.rmap_bp[0])).all() and (pen[last] == int(cfg.rmap_bp[-1])).all()
    inner = np.ones(len(parts), dtype=bool); inner[last] = False
    assert (pen[inner] == pst[1:][inner[:-1]]).all()
    g.close()


def test_population_growth_without_intermediate_sync(gpu_lib, oracle_lib):
    """The population grows every generation (capacity growth reallocates and copies the CURRENT planes) while the previous
    generation's dense stitch may still be running on the library's second stream: nothing between the generations waits
    for it (no genotype download), rows are 500 KB so a stitch takes milliseconds.  Sex, A/D, interval and mutation lists
    must equal the oracle's, and every word of the final planes must equal the materialised interval state."""
    L = 4_000_000
    sizes = [3000, 4200, 5900, 8200]
    cfg = SyntheticConfig(sizes[0], L, n_cv=64, seed=21)
    g = gpu_lib.create(1, 1, 1); o = oracle_lib.create(1, 1, 1)
    cfg.apply_static(g); cfg.apply_static(o)
    g.synth_founders(0, 0, 2 * sizes[0], 77)                     # the oracle never materialises genotypes here: no SNP panel
    g.synth_cv_founders(0, 0, 0, 2 * sizes[0], 78); o.upload_cv_founders(0, 0, 0, synth_packed(78, 2 * sizes[0], 64), 64)
    sg = Simulation(g, 909, 1, True); so = Simulation(o, 909, 1, True)
    sg.ras_initial_human_gen0(0, sizes[0]); so.ras_initial_human_gen0(0, sizes[0])
    rng = np.random.default_rng(4)
    for gen, n in enumerate(sizes[1:], 1):
        c = synthetic_random_mate(sg.sex[0], n, rng)
        sg.couples[0] = c; so.couples[0] = c
        assert np.array_equal(sg.reproduce(0, gen), so.reproduce(0, gen)), f"sex differs at generation {gen}"
        ag = sg.ras_compute_AD(0, gen); ao = so.ras_compute_AD(0, gen)
        assert helpers.bits_equal(ag[0], ao[0]), f"additive values differ at generation {gen}"
    pg, og = g.download_intervals(0, 0); po, oo = o.download_intervals(0, 0)
    assert np.array_equal(og, oo) and np.array_equal(pg, po)
    mg, mog = g.download_mutations(0, 0); mo, moo = o.download_mutations(0, 0)
    assert np.array_equal(mog, moo) and np.array_equal(mg, mo)
    assert g.dbg_verify_planes(0, 0, 77) == (0, 0)
    g.close(); o.close()


def test_baseline_config4_per_gpu_shard_full_size_dense_state_equals_interval_state(gpu_lib):
    """BASELINE config 4's per-GPU share at FULL size: 125k individuals x 11 chromosomes x 227k SNPs = 156 GB of resident planes
    (two sets), all chromosomes stitched by ONE launch per generation.  After two generations every word of all eleven planes
    (2.75 M rows x 7 094 words) must equal the materialised interval state (gev_dbg_verify_planes); A/D must be finite and vary."""
    n, L, nchr = 125_000, 227_000, 11
    cfg = SyntheticConfig(n, L, nchr=nchr, seed=12345)
    g = gpu_lib.create(1, nchr, 1)
    cfg.apply_static(g)
    for c in range(nchr):
        g.synth_founders(0, c, 2 * n, 3000 + c); g.synth_cv_founders(0, 0, c, 2 * n, 4000 + c)
    sim = Simulation(g, 404, nchr, True)
    sim.ras_initial_human_gen0(0, n)
    rng = np.random.default_rng(6)
    for gen in (1, 2):
        sim.couples[0] = synthetic_random_mate(sim.sex[0], n, rng)
        sim.reproduce(0, gen)
        add, _, addc, _ = sim.ras_compute_AD(0, gen, per_chr=True)
        assert np.isfinite(add).all() and add.std() > 0 and all(addc[:, c, 0].std() > 0 for c in range(nchr))
    for c in range(nchr):
        assert g.dbg_verify_planes(0, c, 3000 + c) == (0, 0), f"chromosome {c}: dense state != interval state"
    assert g.dbg_verify_planes(0, 3, 3004)[0] > 10 ** 8              # negative control: chromosome 3 against chromosome 4's founders
    g.close()


def test_baseline_config5_per_gpu_shape_full_size_tiles_from_intervals(gpu_lib):
    """BASELINE config 5's per-GPU shape at FULL size: 125k individuals x 10M loci, no resident genotype planes (the matrix would
    need 312 GB per generation): interval state + tiled materialisation.  The interval state does not depend on the SNP grid,
    so a second, DENSE context that holds only a 16 384-SNP window of the same chromosome (same maps, CVs, seeds, couples) evolves
    the same population: after two generations the tile gev_materialize assembles for that window from the plane-less context's
    intervals (all 250 000 haplotype rows) must equal the dense context's resident rows, and gev_materialize_bed its .bed."""
    n, L, s0, ns = 125_000, 10_000_000, 6_000_000, 16_384
    big = SyntheticConfig(n, L, seed=12345)
    sparse = gpu_lib.create(1, 1, 1); dense = gpu_lib.create(1, 1, 1)
    sparse.set_dense_state(False)
    big.apply_static(sparse)
    big.apply_static(dense)
    dense.set_snps(0, 0, big.snp_pos[s0:s0 + ns])                  # the window only
    dense.synth_founders(0, 0, 2 * n, 808)                         # founder alleles of the window, generated on the device
    for ctx in (sparse, dense):
        ctx.synth_cv_founders(0, 0, 0, 2 * n, 809)
    sims = [Simulation(ctx, 31, 1, True) for ctx in (sparse, dense)]
    for sm in sims:
        sm.ras_initial_human_gen0(0, n)
    tile = dense.download_haps(0, 0)                                # generation 0 = the founder panel (row = founder haplotype): the tile gev_materialize is given
    rng = np.random.default_rng(8)
    for gen in (1, 2):
        c = synthetic_random_mate(sims[0].sex[0], n, rng)
        for sm in sims:
            sm.couples[0] = c
        assert np.array_equal(sims[0].reproduce(0, gen), sims[1].reproduce(0, gen))
        a0 = sims[0].ras_compute_AD(0, gen); a1 = sims[1].ras_compute_AD(0, gen)
        assert helpers.bits_equal(a0[0], a1[0])
    got = sparse.materialize(0, 0, tile, 0, 2 * n, s0, ns)
    want = dense.download_haps(0, 0)
    assert np.array_equal(got, want), "tile assembled from the interval state != the dense context's rows"
    assert np.array_equal(sparse.materialize_bed(0, 0, [tile], s0, ns), dense.format_bed(0, 0))
    parts, off = sparse.download_intervals(0, 0)
    assert off[-1] > 2 * n * 1.5
    sparse.close(); dense.close()


def test_work_table_ring_wraps_around(oracle_lib):
    """the per-generation work tables (ChrWork / CvWork / AdWork) are staged through a ring of pinned host memory; with the ring
    shrunk to a few entries it wraps every other generation (a wrap waits for the stream before slots are reused): 14 generations
    of a 3-chromosome, 2-phenotype population against the oracle.  Runs in its own process (the ring size is read once)."""
    import subprocess
    import sys
    code = (
        "import numpy as np\n"
        "from geneevolve_amd.capi import GevLibrary\n"
        "from oracle import oracle_api\n"
        "from geneevolve_amd.host import SyntheticConfig\n"
        "from tests.test_gpu_parity import run_pair\n"
        "cfg = SyntheticConfig(150, 3000, nchr=3, chrom_bp=1_000_000, map_step=1000, rec_per_row=3e-3, mut_per_row=3e-3, n_cv=60, nphen=2, seed=19, vd=0.2)\n"
        "run_pair(GevLibrary(), oracle_api.load(), cfg, n_gen=14, seed=5, check_every=7)\n"
        "print('ring ok')\n")
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=root, env=dict(os.environ, GEV_TABLE_RING_BYTES="2048", PYTHONPATH=root))
    assert r.returncode == 0 and "ring ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_cpp_host_drives_the_c_abi_like_the_python_host(gpu_lib):
    """tools/host_demo.cpp (C++ host, geneevolve_amd/host/gev_host.hpp, no Python) and the ctypes host
    run the same scenario through the same C-ABI: identical genotype / A / sex checksums."""
    import subprocess
    exe = os.path.join(os.path.dirname(helpers.GOLDEN), "..", "tools", "host_demo")
    N, L, G, R, C = 500, 5000, 4, 201, 64
    out = subprocess.run([exe, str(N), str(L), str(G)], capture_output=True, text=True, check=True).stdout
    want = dict(l.split() for l in out.strip().splitlines())
    # same scenario with gev_presample: the host draws [mate seed][reproduce seed][mutation seeds] -- the reference's order --
    # before it forms the couples, the GPU samples meanwhile: identical results
    out2 = subprocess.run([exe, str(N), str(L), str(G), "1"], capture_output=True, text=True, check=True).stdout
    assert dict(l.split() for l in out2.strip().splitlines()) == want

    def fnv(b):
        h = 1469598103934665603
        for x in bytes(b):
            h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return "%016x" % h
    g = gpu_lib.create(1, 1, 1)
    bp = (1000 + 10000 * np.arange(R)).astype(np.uint64)
    prob = np.r_[0.0, np.full(R - 1, 5e-3)]
    g.set_rmap(0, 0, bp, prob, 10000); g.set_mutmap(0, 0, bp, prob)
    g.set_snps(0, 0, (1000 + (2000000 // L) * np.arange(L)).astype(np.uint64))
    a = np.array([float((i * 37) % 11) - 5.0 for i in range(C)])
    g.set_cvs(0, 0, 0, (1500 + 31000 * np.arange(C)).astype(np.uint64), a, np.zeros(C), 0.0)
    g.synth_founders(0, 0, 2 * N, 77); g.synth_cv_founders(0, 0, 0, 2 * N, 78)
    sim = Simulation(g, 12345, 1, True)
    sim.ras_initial_human_gen0(0, N)
    from geneevolve_amd.host import random_mate
    for gen in range(1, G + 1):
        c = random_mate(sim.sex[0], np.ones(len(sim.sex[0])), N, int(sim.ras_glob_seed()[0]))     # same draws as gev::random_mate
        sim.couples[0] = c
        seeds = sim.ras_glob_seed(1 + N)
        sim.sex[0] = g.reproduce(0, c, int(seeds[0]), seeds[1:])
        add, dom, _, _ = g.compute_ad(0, per_chr=False)
    assert fnv(np.ascontiguousarray(g.download_haps(0, 0)).tobytes()) == want["HAPS"]
    assert fnv(np.ascontiguousarray(add).tobytes()) == want["ADD"]
    assert fnv(sim.sex[0].tobytes()) == want["SEX"]
    g.close()


def test_inbred_couples_zero_row_mutation_map_and_heterogeneous_chromosomes(gpu_lib, oracle_lib):
    """3 chromosomes with different lengths / map densities / SNP counts; chromosome 1 has an EMPTY mutation map
    (the reference still draws its ras_glob_seed and calls srand for it, src/Simulation.cpp:2459-2462, 2509);
    couples flagged inbreed are skipped (:2440); 2 phenotypes with different CV counts per chromosome."""
    rs = np.random.RandomState(21)
    nchr, nphen, n = 3, 2, 150
    g = gpu_lib.create(1, nchr, nphen); o = oracle_lib.create(1, nchr, nphen)
    Ls = [3000, 777, 64]
    for c in range(nchr):
        R = [301, 41, 9][c]; step = [1000, 5000, 200][c]
        bp = (500 + step * np.arange(R)).astype(np.uint64)
        prob = np.r_[0.0, rs.uniform(0, [4e-3, 3e-2, 0.2][c], R - 1)]
        pos = np.sort(rs.randint(0, int(bp[-1]) + 3 * step, Ls[c])).astype(np.uint64)          # duplicates and loci outside the map range
        rate = np.r_[0.0, rs.uniform(0, 0.02, R - 1)]
        for ctx in (g, o):
            ctx.set_rmap(0, c, bp, prob, step)
            if c == 1:
                ctx.set_mutmap(0, c, np.empty(0, dtype=np.uint64), np.empty(0))
            else:
                ctx.set_mutmap(0, c, bp, rate)
            ctx.set_snps(0, c, pos)
        g.synth_founders(0, c, 2 * n, 50 + c); o.upload_founders(0, c, synth_packed(50 + c, 2 * n, Ls[c]), Ls[c])
        for p in range(nphen):
            C = [[40, 7, 3], [11, 0, 5]][p][c]
            cvbp = rs.randint(int(bp[0]) - 100, int(bp[-1]) + 100, C).astype(np.uint64)              # unsorted, some outside the range
            a, d = rs.randn(C), rs.randn(C)
            for ctx in (g, o):
                ctx.set_cvs(0, p, c, cvbp, a, d, [0.3, 0.0][p])
            g.synth_cv_founders(0, p, c, 2 * n, 900 + 10 * p + c)
            o.upload_cv_founders(0, p, c, synth_packed(900 + 10 * p + c, 2 * n, C) if C else np.zeros((2 * n, 1), dtype=np.uint64), C)
    sg = Simulation(g, 5, nchr, True); so = Simulation(o, 5, nchr, True)
    sg.ras_initial_human_gen0(0, n); so.ras_initial_human_gen0(0, n)
    rng = np.random.default_rng(3)
    for gen in range(1, 5):
        c = synthetic_random_mate(sg.sex[0], n, rng)
        c["num_offspring"] = rng.integers(0, 4, n)
        c["inbreed"] = rng.integers(0, 5, n) == 0
        if c["num_offspring"][c["inbreed"] == 0].sum() == 0:
            c["num_offspring"][0] = 2; c["inbreed"][0] = 0
        sg.couples[0] = c; so.couples[0] = c.copy()
        assert np.array_equal(sg.reproduce(0, gen), so.reproduce(0, gen)), f"sex gen {gen}"
        for x, y in zip(sg.ras_compute_AD(0, gen, per_chr=True), so.ras_compute_AD(0, gen, per_chr=True)):
            assert helpers.bits_equal(x, y), f"A/D gen {gen}"
        for k in range(nchr):
            assert np.array_equal(g.download_haps(0, k), o.download_haps(0, k)), f"dense gen {gen} chr {k}"
            pg, og = g.download_intervals(0, k); po, oo = o.download_intervals(0, k)
            assert np.array_equal(og, oo) and np.array_equal(pg, po)
            mg, mog = g.download_mutations(0, k); mo, moo = o.download_mutations(0, k)
            assert np.array_equal(mog, moo) and np.array_equal(mg, mo)
            for p in range(nphen):
                assert np.array_equal(g.download_cv(0, p, k), o.download_cv(0, p, k))
    g.close(); o.close()


def test_unsupported_inputs_are_refused_not_miscomputed(gpu_lib):
    g = gpu_lib.create(1, 1, 1)
    bp = np.array([0, 1000, 1500, 3000], dtype=np.uint64)
    with pytest.raises(capi.GevError) as e:                       # rows closer than bp_dist: the reference would emit unsorted breakpoints
        g.set_rmap(0, 0, bp, np.zeros(4), 1000)
    assert e.value.code == -5
    with pytest.raises(capi.GevError) as e:
        g.set_snps(0, 0, np.array([5, 4, 6], dtype=np.uint64))    # unsorted SNP grid
    assert e.value.code == -5
    with pytest.raises(capi.GevError) as e:
        g.set_rmap(0, 0, bp[:2], np.zeros(2), 0)                  # rand() % 0
    assert e.value.code == -1
    with pytest.raises(capi.GevError) as e:                       # call order
        g.reproduce(0, np.array([[0, 1, 0, 1]]), 1, None)
    assert e.value.code == -2
    g.close()


@pytest.mark.parametrize("case", ["dense", "ex1sub", "vt2"])
def test_gpu_scale_ad_compute_gef_matches_reference(gpu_lib, oracle_lib, case):
    """SURVEY 8(f) row 1 on the device: phenotype floats within 1e-12 relative of the reference's hex-dumped values
    (north star: 1e-6); e comes from the parallel polar-method normal stream, var(e) from parallel sums."""
    fx = helpers.load_fixture(case)
    seeds = helpers.find_gen0_seeds(fx, oracle_lib)
    helpers.replay_case(gpu_lib, fx, seeds, f"gpu-gef/{case}", check_lists=False, check_gef=True, gef_rtol=1e-12)


def test_snp_major_and_bed_packing_match_their_definition(gpu_lib):
    """K9: SNP-major transpose / PLINK .bed against numpy on the downloaded haplotype-major matrix (mutations applied).
    The reference has no .bed writer: checked by definition (parity unpinned); the .hap TEXT is pinned on the
    reference's own output files in the golden replays."""
    cfg = SyntheticConfig(333, 5000, chrom_bp=2_000_000, map_step=1000, rec_per_row=2e-3, mut_per_row=0.05, n_cv=10, seed=4)
    g = gpu_lib.create(1, 1, 1)
    cfg.apply_static(g)
    g.synth_founders(0, 0, 666, 9); g.synth_cv_founders(0, 0, 0, 666, 10)
    sim = Simulation(g, 77, 1, True)
    sim.ras_initial_human_gen0(0, 333)
    rng = np.random.default_rng(0)
    for gen in (1, 2, 3):
        sim.couples[0] = synthetic_random_mate(sim.sex[0], 333, rng)
        sim.reproduce(0, gen)
    L, n = 5000, 333
    H = capi.unpack_rows(g.download_haps(0, 0), L)                       # [666][5000]
    for (s0, ns) in ((0, L), (70, 999), (4937, 63)):
        S = capi.unpack_rows(g.download_snp_major(0, 0, s0, ns), 2 * n)  # [ns][666]
        assert np.array_equal(S, H[:, s0:s0 + ns].T), (s0, ns)
        bed = g.format_bed(0, 0, s0, ns).reshape(ns, (n + 3) // 4)
        a, b = H[0::2, s0:s0 + ns].T, H[1::2, s0:s0 + ns].T             # [ns][n]
        code = np.where((a & b) == 1, 0, np.where((a ^ b) == 1, 2, 3)).astype(np.uint8)
        pad = (-n) % 4
        code = np.concatenate([code, np.zeros((ns, pad), dtype=np.uint8)], axis=1).reshape(ns, -1, 4)
        want = code[:, :, 0] | (code[:, :, 1] << 2) | (code[:, :, 2] << 4) | (code[:, :, 3] << 6)
        assert np.array_equal(bed, want), (s0, ns)
        txt = g.format_hap_text(0, 0, s0, ns).reshape(ns, 4 * n + 1)
        assert (txt[:, -1] == ord("\n")).all() and (txt[:, 1::2][:, :2 * n] == ord(" ")).all()
        assert np.array_equal(txt[:, 0:4 * n:2] - ord("0"), H[:, s0:s0 + ns].T)
        gt = g.format_vcf_gt(0, 0, s0, ns).reshape(ns, 4 * n + 1)            # "\ta|b" per individual (format_vcf.cpp:55-59)
        assert (gt[:, -1] == ord("\n")).all() and (gt[:, 0:4 * n:4] == ord("\t")).all() and (gt[:, 2:4 * n:4] == ord("|")).all()
        assert np.array_equal(gt[:, 1:4 * n:4] - ord("0"), H[0::2, s0:s0 + ns].T) and np.array_equal(gt[:, 3:4 * n:4] - ord("0"), H[1::2, s0:s0 + ns].T)
    # PLINK individual-major siblings (ranges that start / end inside the population; L = 5000 is not a multiple of 32 or 64)
    full = helpers.interleave_haps(g.download_haps(0, 0), L)
    al0 = np.frombuffer(b"ACGT", dtype=np.uint8)[np.arange(L) % 4]; al1 = np.frombuffer(b"TGCA", dtype=np.uint8)[(np.arange(L) // 3) % 4]
    for (i0, ni) in ((0, n), (1, 7), (330, 3), (5, 0)):
        assert np.array_equal(g.download_plink_matrix(0, 0, i0, ni), full[i0:i0 + ni]), (i0, ni)
        for letters in (False, True):
            txt = g.format_ped_text(0, 0, al0 if letters else None, al1 if letters else None, i0, ni).reshape(ni, 4 * L + 1)
            assert (txt[:, -1] == ord("\n")).all() and (txt[:, 0:4 * L:2] == ord(" ")).all()
            for hp in (0, 1):
                bits = H[2 * i0 + hp:2 * (i0 + ni):2, :]
                want = np.where(bits == 1, al1, al0) if letters else bits + ord("0")
                assert np.array_equal(txt[:, 1 + 2 * hp:4 * L:4], want), (i0, ni, letters, hp)
    with pytest.raises(capi.GevError):
        g.format_ped_text(0, 0, None, None, n - 1, 2)
    g.close()


def test_randomised_small_configurations_against_oracle(gpu_lib, oracle_lib):
    """40 random tiny scenarios (1-3 chromosomes, 2-60 individuals, 1-300 loci, 2-40 map rows, hot maps, ragged families,
    with/without mutation map, both stitch kernels): every quantity bit-identical to the oracle."""
    rs = np.random.RandomState(20241004)
    for trial in range(40):
        nchr = int(rs.randint(1, 4)); nphen = int(rs.randint(1, 3)); n = int(rs.randint(2, 61))
        has_mut = bool(rs.randint(0, 2))
        g = gpu_lib.create(1, nchr, nphen); o = oracle_lib.create(1, nchr, nphen)
        g.set_stitch_mode(int(rs.randint(0, 2)))
        for c in range(nchr):
            R = int(rs.randint(2, 41)); step = int(rs.choice([1, 3, 10, 1000]))
            bp = (int(rs.randint(0, 50)) + step * np.arange(R) * int(rs.randint(1, 4))).astype(np.uint64)
            prob = np.r_[0.0, rs.uniform(0, rs.choice([0.01, 0.3, 1.0]), R - 1)]
            L = int(rs.randint(1, 301))
            pos = np.sort(rs.randint(0, int(bp[-1]) + 2 * step + 2, L)).astype(np.uint64)
            rate = np.r_[0.0, rs.uniform(0, rs.choice([0.01, 0.5]), R - 1)]
            for ctx in (g, o):
                ctx.set_rmap(0, c, bp, prob, step)
                if has_mut:
                    ctx.set_mutmap(0, c, bp, rate)
                ctx.set_snps(0, c, pos)
            F = synth_packed(1000 * trial + c, 2 * n, L)
            g.upload_founders(0, c, F, L); o.upload_founders(0, c, F, L)
            for p in range(nphen):
                C = int(rs.randint(0, 40))
                cvbp = rs.randint(0, int(bp[-1]) + 5, C).astype(np.uint64)
                a, d = rs.randn(C), rs.randn(C)
                vd = float(rs.choice([0.0, 0.5, -1.0]))
                V = synth_packed(7000 * trial + 10 * p + c, 2 * n, C) if C else np.zeros((2 * n, 1), dtype=np.uint64)
                for ctx in (g, o):
                    ctx.set_cvs(0, p, c, cvbp, a, d, vd)
                    ctx.upload_cv_founders(0, p, c, V, C)
        seed0 = int(rs.randint(1, 1000001))
        assert np.array_equal(g.init_gen0(0, n, seed0), o.init_gen0(0, n, seed0))
        npop = n
        for gen in range(1, 4):
            ncp = int(rs.randint(1, 2 * npop + 1))
            c = np.stack([rs.randint(0, npop, ncp), rs.randint(0, npop, ncp), (rs.rand(ncp) < 0.15).astype(np.int64), rs.randint(0, 4, ncp)], axis=1)
            if c[c[:, 2] == 0, 3].sum() == 0:
                c[0, 2] = 0; c[0, 3] = 1
            n_off = int(c[c[:, 2] == 0, 3].sum())
            sd = int(rs.randint(1, 1000001)); ms = rs.randint(1, 1000001, n_off * nchr).astype(np.uint32) if has_mut else None
            assert np.array_equal(g.reproduce(0, c, sd, ms), o.reproduce(0, c, sd, ms)), f"trial {trial} gen {gen}: sex"
            for x, y in zip(g.compute_ad(0), o.compute_ad(0)):
                assert helpers.bits_equal(x, y), f"trial {trial} gen {gen}: A/D"
            for k in range(nchr):
                assert np.array_equal(g.download_haps(0, k), o.download_haps(0, k)), f"trial {trial} gen {gen} chr {k}: dense"
                pg, og = g.download_intervals(0, k); po, oo = o.download_intervals(0, k)
                assert np.array_equal(og, oo) and np.array_equal(pg, po), f"trial {trial} gen {gen} chr {k}: intervals"
                mg, mog = g.download_mutations(0, k); mo, moo = o.download_mutations(0, k)
                assert np.array_equal(mog, moo) and np.array_equal(mg, mo), f"trial {trial} gen {gen} chr {k}: mutations"
                for p in range(nphen):
                    assert np.array_equal(g.download_cv(0, p, k), o.download_cv(0, p, k)), f"trial {trial} gen {gen} chr {k}: CV matrix"
                assert np.array_equal(g.format_hap_text(0, k), o.format_hap_text(0, k))
                assert np.array_equal(g.format_vcf_gt(0, k), o.format_vcf_gt(0, k)) and np.array_equal(g.format_ped_text(0, k), o.format_ped_text(0, k))
                assert np.array_equal(g.download_plink_matrix(0, k), o.download_plink_matrix(0, k))
            npop = n_off
        g.close(); o.close()


def test_whole_genome_22_chromosomes_on_the_reference_genetic_map(gpu_lib, oracle_lib):
    """BASELINE config 4's map: Recom_Map.zip:Recom.Map.b37.50KbDiff (22 autosomes, 55 657 rows, 35.9 Morgans) with a
    mutation map on the same rows; synthetic founders; ragged families.  GPU == oracle on every chromosome."""
    z = np.load(os.path.join(helpers.GOLDEN, "recom_map_b37_50kb.npz"))
    nchr, n, L, C = 22, 60, 400, 20
    g = gpu_lib.create(1, nchr, 1); o = oracle_lib.create(1, nchr, 1)
    rs = np.random.RandomState(8)
    for c in range(nchr):
        bp = z["bp"][z["chr"] == c + 1]; cM = z["cM"][z["chr"] == c + 1]
        prob = np.zeros(len(bp)); prob[1:] = (cM[1:] - cM[:-1]) * .01          # Population::ras_compute_recom_prob
        pos = np.sort(rs.randint(int(bp[0]), int(bp[-1]), L)).astype(np.uint64)
        cvbp = np.sort(rs.randint(int(bp[0]), int(bp[-1]), C)).astype(np.uint64)
        a, d = rs.randn(C), rs.randn(C) * 0.2
        F = synth_packed(300 + c, 2 * n, L); V = synth_packed(600 + c, 2 * n, C)
        for ctx in (g, o):
            ctx.set_rmap(0, c, bp, prob, int(bp[1] - bp[0]))
            ctx.set_mutmap(0, c, bp, np.r_[0.0, np.full(len(bp) - 1, 2e-4)])
            ctx.set_snps(0, c, pos)
            ctx.set_cvs(0, 0, c, cvbp, a, d, -1.0)
            ctx.upload_founders(0, c, F, L); ctx.upload_cv_founders(0, 0, c, V, C)
    sg = Simulation(g, 31, nchr, True); so = Simulation(o, 31, nchr, True)
    sg.ras_initial_human_gen0(0, n); so.ras_initial_human_gen0(0, n)
    rng = np.random.default_rng(2)
    for gen in (1, 2, 3):
        c = synthetic_random_mate(sg.sex[0], n // 2, rng); c["num_offspring"] = 2
        sg.couples[0] = c; so.couples[0] = c.copy()
        assert np.array_equal(sg.reproduce(0, gen), so.reproduce(0, gen))
        for x, y in zip(sg.ras_compute_AD(0, gen, per_chr=True), so.ras_compute_AD(0, gen, per_chr=True)):
            assert helpers.bits_equal(x, y), f"A/D gen {gen}"
        for k in range(nchr):
            assert np.array_equal(g.download_haps(0, k), o.download_haps(0, k)), f"dense gen {gen} chr {k + 1}"
            pg, og = g.download_intervals(0, k); po, oo = o.download_intervals(0, k)
            assert np.array_equal(og, oo) and np.array_equal(pg, po)
    total_parts = sum(len(g.download_intervals(0, k)[0]) for k in range(nchr))
    assert total_parts > 2 * n * nchr * 1.5          # ~36 crossovers per gamete set per generation
    g.close(); o.close()


def test_device_rank_matches_reference_and_oracle(gpu_lib, oracle_lib):
    """gev_rank_f64 == CommFunc::ras_rank: the reference's vectors, then random vectors with many ties against the oracle's
    literal O(n^2) loop, then a 200k-element vector against the stable-sort identity (the reference would need 2e10 compares)"""
    from geneevolve_amd.host import ras_rank
    g = gpu_lib.create(1, 1, 1); o = oracle_lib.create(1, 1, 1)
    with gzip.open(os.path.join(helpers.GOLDEN, "kat.txt.gz"), "rt") as f:
        rows = [l.split() for l in f if l.startswith("RANK ")]
    for t in rows:
        n = int(t[1]); x = np.array([float.fromhex(v) for v in t[2:2 + n]]); want = np.array([int(v) for v in t[2 + n:2 + 2 * n]], dtype=np.uint64)
        assert np.array_equal(g.rank_f64(x), want)
    rng = np.random.default_rng(3)
    for n in (0, 1, 2, 255, 256, 257, 1023, 1024, 1025, 5000):
        x = np.round(rng.standard_normal(n) * 4) / 4
        assert np.array_equal(g.rank_f64(x), o.rank_f64(x)), n
    x = rng.standard_normal(200_000); x[0:199_990:7] = x[3:199_993:7]            # exact ties between different positions
    r = g.rank_f64(x)
    assert np.array_equal(r, ras_rank(x)) and np.array_equal(np.sort(r), np.arange(len(x), dtype=np.uint64))
    # the reference's comparisons, literally: NaNs, both zeros, infinities, denormals, ties -- against the oracle's O(n^2) pair loop
    special = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 5e-324, -5e-324, 1.0, -1.0, np.nan, 0.0, -0.0, 1.7976931348623157e308, 2.0, 2.0])
    for n in (15, 400, 3000):
        x = special[rng.integers(0, len(special), n)]
        assert np.array_equal(g.rank_f64(x), o.rank_f64(x)), f"special values, n={n}"
    # config-4 size and beyond: 10^6 values (the reference's loop would need 5e11 compares); timed for profiles/
    import time
    x = rng.standard_normal(1_000_000); x[::5] = x[1::5]
    g.rank_f64(x[:1000])
    t0 = time.perf_counter(); r = g.rank_f64(x); dt = time.perf_counter() - t0
    assert np.array_equal(r, ras_rank(x))
    print(f"gev_rank_f64: 10^6 doubles in {dt * 1e3:.2f} ms (host to host)")
    g.close(); o.close()


def test_presample_gives_the_same_generation_and_falls_back_on_any_mismatch(gpu_lib, oracle_lib):
    """gev_presample only moves the sampling kernels ahead of the couples: states with / without it are identical to the
    oracle's; a presample whose seeds, offspring count or population do not match the following gev_reproduce is ignored."""
    cfg = SyntheticConfig(90, 700, nchr=2, chrom_bp=500_000, map_step=5000, rec_per_row=0.02, mut_per_row=0.03, n_cv=30, seed=8)
    ctxs = []
    for lib in (gpu_lib, gpu_lib, oracle_lib):
        g = lib.create(2, 2, 1)
        for pop in (0, 1):
            cfg.apply_static(g, pop)
            for c in range(2):
                if lib is gpu_lib:
                    g.synth_founders(pop, c, 180, 11 + c + 7 * pop); g.synth_cv_founders(pop, 0, c, 180, 52 + c + 7 * pop)
                else:
                    g.upload_founders(pop, c, synth_packed(11 + c + 7 * pop, 180, 700), 700)
                    g.upload_cv_founders(pop, 0, c, synth_packed(52 + c + 7 * pop, 180, 30), 30)
        ctxs.append(g)
    sims = [Simulation(g, 99, 2, True) for g in ctxs]
    for s in sims:
        s.ras_initial_human_gen0(0, 90); s.ras_initial_human_gen0(1, 90)
    rng = np.random.default_rng(1)
    a, b, o = sims
    for gen in range(1, 7):
        n_off = int(rng.integers(60, 120))
        for pop in (0, 1):
            couples = synthetic_random_mate(a.sex[pop], n_off, rng)
            seeds = a.ras_glob_seed(1 + n_off * 2); b.ras_glob_seed(1 + n_off * 2); o.ras_glob_seed(1 + n_off * 2)
            # a: presample with the right inputs (odd generations) or with wrong ones that must be ignored (even)
            if gen % 2:
                a.presample(pop, seeds, n_off)
            elif gen == 2:
                wrong = seeds.copy(); wrong[5] ^= 1
                a.presample(pop, wrong, n_off)                         # one mutation seed differs
            elif gen == 4:
                a.presample(pop, seeds[:1 + (n_off - 1) * 2], n_off - 1)    # other offspring count
            else:
                a.presample(1 - pop, seeds, n_off)                     # other population
            for s in sims:
                s.couples[pop] = couples
                s.reproduce(pop, gen, seeds=seeds, n_people=n_off)
            assert np.array_equal(a.sex[pop], o.sex[pop]) and np.array_equal(b.sex[pop], o.sex[pop])
            xa, xb, xo = a.ras_compute_AD(pop), b.ras_compute_AD(pop), o.ras_compute_AD(pop)
            assert helpers.bits_equal(xa[0], xo[0]) and helpers.bits_equal(xb[0], xo[0]), (gen, pop)
            for c in range(2):
                want = ctxs[2].download_haps(pop, c)
                assert np.array_equal(ctxs[0].download_haps(pop, c), want) and np.array_equal(ctxs[1].download_haps(pop, c), want), (gen, pop, c)
                pa, oa = ctxs[0].download_intervals(pop, c); po, oo = ctxs[2].download_intervals(pop, c)
                assert np.array_equal(oa, oo) and np.array_equal(pa, po)
                ma, moa = ctxs[0].download_mutations(pop, c); mo, moo = ctxs[2].download_mutations(pop, c)
                assert np.array_equal(moa, moo) and np.array_equal(ma, mo)
    with pytest.raises(capi.GevError):
        ctxs[0].presample(0, 1, np.zeros(3, dtype=np.uint32), 90)      # wrong seed count
    for g in ctxs:
        g.close()


def test_reproduce_in_two_halves_with_early_sexes_gives_the_same_generations(gpu_lib, oracle_lib):
    """gev_reproduce_begin / gev_reproduce_end with the head start and gev_presample_sex -- the pipelined host loop of bench.py,
    where the NEXT generation's couples are formed between the two halves from the sexes the sampling already produced -- against
    the oracle driven the plain way with the same couples; any other call between the two halves is refused."""
    cfg = SyntheticConfig(300, 20000, nchr=2, chrom_bp=4_000_000, map_step=2000, rec_per_row=6e-4, mut_per_row=8e-4, n_cv=80, seed=81)
    g = gpu_lib.create(1, cfg.nchr, cfg.nphen); o = oracle_lib.create(1, cfg.nchr, cfg.nphen)
    cfg.apply_static(g); cfg.apply_static(o)
    nh = 2 * cfg.n_ind
    for c in range(cfg.nchr):
        g.synth_founders(0, c, nh, cfg.seed + c); o.upload_founders(0, c, synth_packed(cfg.seed + c, nh, cfg.n_loci), cfg.n_loci)
        g.synth_cv_founders(0, 0, c, nh, cfg.seed + 100 + c); o.upload_cv_founders(0, 0, c, synth_packed(cfg.seed + 100 + c, nh, 80), 80)
    sg = Simulation(g, 5, cfg.nchr, True); so = Simulation(o, 5, cfg.nchr, True)
    sg.ras_initial_human_gen0(0, cfg.n_ind); so.ras_initial_human_gen0(0, cfg.n_ind)
    rng = np.random.default_rng(3)
    n, n_seeds = cfg.n_ind, 1 + cfg.n_ind * cfg.nchr
    seeds = sg.ras_glob_seed(n_seeds)
    sg.presample(0, seeds, n)
    couples = synthetic_random_mate(sg.sex[0], n, rng)
    for gen in range(1, 7):
        sex_early = g.presample_sex(0, n)
        seeds_next = sg.ras_glob_seed(n_seeds)
        g.reproduce_begin(0, couples, int(seeds[0]), seeds[1:], n_people=n)
        with pytest.raises(capi.GevError):
            g.download_cv(0, 0, 0)                                              # nothing else is allowed between the halves ...
        if gen % 2:
            sg.presample(0, seeds_next, n)                                      # ... but the head start of the next generation (bench.py's order)
        next_couples = synthetic_random_mate(sex_early, n, rng)              # what bench.py does here
        sex = g.reproduce_end(want_sex=bool(gen % 3))
        assert sex is None or np.array_equal(sex, sex_early)
        so.couples[0] = couples
        assert np.array_equal(so.reproduce(0, gen, seeds=seeds, n_people=n), sex_early), f"sex differs at generation {gen}"
        if not gen % 2:
            sg.presample(0, seeds_next, n)                                      # or after the generation is complete
        seeds = seeds_next
        ag = g.compute_ad(0); ao = o.compute_ad(0)
        for x, y in zip(ag, ao):
            assert helpers.bits_equal(x, y), f"A/D not bit-identical at generation {gen}"
        for c in range(cfg.nchr):
            assert np.array_equal(g.download_haps(0, c), o.download_haps(0, c)), f"dense genotypes differ (gen {gen} chr {c})"
            pg, og = g.download_intervals(0, c); po, oo = o.download_intervals(0, c)
            assert np.array_equal(og, oo) and np.array_equal(pg, po)
        couples = next_couples
    with pytest.raises(capi.GevError):
        g.reproduce_end()                                                     # nothing pending
    g.close(); o.close()


def slice_bits(packed, L, s0, ns):
    """columns [s0, s0+ns) of a bit matrix packed in uint64 words -> packed again (bit j = column s0 + j)"""
    u = capi.unpack_rows(packed, L)[:, s0:s0 + ns]
    return capi.pack_rows(u)


def test_materialize_tiles_from_intervals_and_the_plane_less_mode(gpu_lib, oracle_lib):
    """K8 / BASELINE config 5's mode: (1) on a dense context gev_materialize (intervals + mutation sets + a founder
    tile) reproduces the resident planes tile by tile; (2) a context WITHOUT resident planes (gev_set_dense_state 0) run through
    the same generations gives the same sex, A/D, intervals and mutation sets, and its materialised tiles equal the dense
    context's genotypes and the oracle's statement of ras_convert_interval_to_hap_matrix on the same tile."""
    cfg = SyntheticConfig(150, 3001, nchr=2, chrom_bp=1_000_000, map_step=10_000, rec_per_row=0.02, mut_per_row=0.05, n_cv=40, seed=21)
    nh, L = 300, 3001
    ctxs = []
    for kind in ("dense", "sparse", "oracle"):
        lib = oracle_lib if kind == "oracle" else gpu_lib
        g = lib.create(1, 2, 1)
        if kind == "sparse":
            g.set_dense_state(False)
        cfg.apply_static(g)
        for c in range(2):
            if kind == "dense":
                g.synth_founders(0, c, nh, 300 + c)
            elif kind == "oracle":
                g.upload_founders(0, c, synth_packed(300 + c, nh, L), L)
            if kind == "oracle":
                g.upload_cv_founders(0, 0, c, synth_packed(400 + c, nh, 40), 40)
            else:
                g.synth_cv_founders(0, 0, c, nh, 400 + c)
        ctxs.append(g)
    dense, sparse, orc = ctxs
    with pytest.raises(capi.GevError):
        sparse.synth_founders(0, 0, nh, 1)                         # no resident planes in this mode
    sims = [Simulation(g, 5, 2, True) for g in ctxs]
    for s in sims:
        s.ras_initial_human_gen0(0, 150)
    founders = [synth_packed(300 + c, nh, L) for c in range(2)]
    rng = np.random.default_rng(9)
    for gen in range(1, 6):
        n_off = int(rng.integers(100, 200))
        couples = synthetic_random_mate(sims[0].sex[0], n_off, rng)
        seeds = sims[0].ras_glob_seed(1 + 2 * n_off)
        for s in sims[1:]:
            s.ras_glob_seed(1 + 2 * n_off)
        sex = []
        for s in sims:
            s.couples[0] = couples
            sex.append(s.reproduce(0, gen, seeds=seeds, n_people=n_off))
        assert np.array_equal(sex[0], sex[2]) and np.array_equal(sex[1], sex[2])
        ad = [s.ras_compute_AD(0, gen, per_chr=True) for s in sims]
        for x, y, z in zip(*ad):
            assert helpers.bits_equal(x, z) and helpers.bits_equal(y, z), f"A/D gen {gen}"
        for c in range(2):
            full = dense.download_haps(0, c)
            assert np.array_equal(full, orc.download_haps(0, c))
            ps, osx = sparse.download_intervals(0, c); po, oo = orc.download_intervals(0, c)
            assert np.array_equal(osx, oo) and np.array_equal(ps, po)
            ms, mos = sparse.download_mutations(0, c); mo, moo = orc.download_mutations(0, c)
            assert np.array_equal(mos, moo) and np.array_equal(ms, mo)
            for (s0, ns, r0, nr) in ((0, L, 0, 2 * n_off), (64, 1000, 3, 57), (2990, 11, 2 * n_off - 1, 1), (31, 33, 0, 10)):
                tile = slice_bits(founders[c], L, s0, ns)
                want = slice_bits(full[r0:r0 + nr], L, s0, ns)
                for g in (dense, sparse, orc):
                    assert np.array_equal(g.materialize(0, c, tile, r0, nr, s0, ns), want), (gen, c, s0, ns, r0, nr)
            for (s0, ns) in ((0, L), (100, 640), (2999, 2)):                      # .bed straight from the interval state == the dense context's
                tiles = [slice_bits(founders[c], L, s0, ns)]
                want = dense.format_bed(0, c, s0, ns)
                assert np.array_equal(sparse.materialize_bed(0, c, tiles, s0, ns), want) and np.array_equal(dense.materialize_bed(0, c, tiles, s0, ns), want), (gen, c, s0, ns)
            with pytest.raises(capi.GevError):
                sparse.download_haps(0, c)
            with pytest.raises(capi.GevError):
                sparse.materialize(0, c, slice_bits(founders[c], L, 0, 64)[:10], 0, 2 * n_off, 0, 64)     # founder tile too short: hap_index not in range
    for g in ctxs:
        g.close()


def test_long_run_exercises_list_growth_and_redo_paths(gpu_lib, oracle_lib):
    """60 generations with ~5 crossovers and ~3 mutations per gamete per generation: the interval and mutation lists outgrow
    their buffers several times (capacity guess too small -> status flag -> grow -> the small work is enqueued again; outgrown
    buffers parked and freed later), with gev_presample on every other generation.  State compared with the oracle every 10."""
    cfg = SyntheticConfig(120, 1500, nchr=2, chrom_bp=1_000_000, map_step=5_000, rec_per_row=0.025, mut_per_row=0.015, n_cv=25, seed=33)
    g = gpu_lib.create(1, 2, 1); o = oracle_lib.create(1, 2, 1)
    cfg.apply_static(g); cfg.apply_static(o)
    for c in range(2):
        g.synth_founders(0, c, 240, 70 + c); o.upload_founders(0, c, synth_packed(70 + c, 240, 1500), 1500)
        g.synth_cv_founders(0, 0, c, 240, 80 + c); o.upload_cv_founders(0, 0, c, synth_packed(80 + c, 240, 25), 25)
    sg, so = Simulation(g, 3, 2, True), Simulation(o, 3, 2, True)
    sg.ras_initial_human_gen0(0, 120); so.ras_initial_human_gen0(0, 120)
    rng = np.random.default_rng(4)
    for gen in range(1, 61):
        n_off = 120 if gen % 7 else 150                      # population size changes now and then
        couples = synthetic_random_mate(sg.sex[0], n_off, rng)
        seeds = sg.ras_glob_seed(1 + 2 * n_off); so.ras_glob_seed(1 + 2 * n_off)
        if gen % 2:
            sg.presample(0, seeds, n_off)
        sg.couples[0] = couples; so.couples[0] = couples
        a = sg.reproduce(0, gen, seeds=seeds, n_people=n_off); b = so.reproduce(0, gen, seeds=seeds, n_people=n_off)
        assert np.array_equal(a, b), f"sex gen {gen}"
        xa, xo = sg.ras_compute_AD(0, gen), so.ras_compute_AD(0, gen)
        assert helpers.bits_equal(xa[0], xo[0]), f"A gen {gen}"
        if gen % 10 == 0:
            for c in range(2):
                assert np.array_equal(g.download_haps(0, c), o.download_haps(0, c)), f"dense gen {gen} chr {c}"
                pg, og = g.download_intervals(0, c); po, oo = o.download_intervals(0, c)
                assert np.array_equal(og, oo) and np.array_equal(pg, po), f"intervals gen {gen} chr {c}"
                mg, mog = g.download_mutations(0, c); mo, moo = o.download_mutations(0, c)
                assert np.array_equal(mog, moo) and np.array_equal(mg, mo), f"mutations gen {gen} chr {c}"
    pg, og = g.download_intervals(0, 0)
    assert og[-1] > 240 * 60                                   # lists really grew far beyond the initial headroom
    g.close(); o.close()


def test_pipelined_host_loop_through_list_growth_and_redo_paths(gpu_lib, oracle_lib):
    """bench.py's order of calls -- gev_presample_sex, gev_reproduce_begin, the head start of the NEXT generation, the next couples on
    the host, gev_reproduce_end -- for 40 generations in which the lists outgrow their buffers several times: a generation is redone
    inside gev_reproduce_end while the next head start is already queued (it is then sampled again), and the population size
    changes (a head start for another size does not match and is dropped).  Against the oracle driven the plain way."""
    cfg = SyntheticConfig(120, 1500, nchr=2, chrom_bp=1_000_000, map_step=5_000, rec_per_row=0.025, mut_per_row=0.015, n_cv=25, seed=35)
    g = gpu_lib.create(1, 2, 1); o = oracle_lib.create(1, 2, 1)
    cfg.apply_static(g); cfg.apply_static(o)
    for c in range(2):
        g.synth_founders(0, c, 240, 70 + c); o.upload_founders(0, c, synth_packed(70 + c, 240, 1500), 1500)
        g.synth_cv_founders(0, 0, c, 240, 80 + c); o.upload_cv_founders(0, 0, c, synth_packed(80 + c, 240, 25), 25)
    sg, so = Simulation(g, 3, 2, True), Simulation(o, 3, 2, True)
    sg.ras_initial_human_gen0(0, 120); so.ras_initial_human_gen0(0, 120)
    rng = np.random.default_rng(6)
    size = lambda gen: 120 if gen % 9 else 150
    seeds = sg.ras_glob_seed(1 + 2 * size(1))
    sg.presample(0, seeds, size(1))
    couples = synthetic_random_mate(sg.sex[0], size(1), rng)
    for gen in range(1, 41):
        n = size(gen)
        sex_early = g.presample_sex(0, n)
        seeds_next = sg.ras_glob_seed(1 + 2 * size(gen + 1))
        g.reproduce_begin(0, couples, int(seeds[0]), seeds[1:], n_people=n)
        sg.presample(0, seeds_next, size(gen + 1))
        next_couples = synthetic_random_mate(sex_early, size(gen + 1), rng)
        sex = g.reproduce_end()
        assert np.array_equal(sex, sex_early)
        so.couples[0] = couples
        assert np.array_equal(so.reproduce(0, gen, seeds=seeds, n_people=n), sex), f"sex gen {gen}"
        xa, xo = g.compute_ad(0), o.compute_ad(0)
        assert helpers.bits_equal(xa[0], xo[0]), f"A gen {gen}"
        if gen % 10 == 0:
            for c in range(2):
                assert np.array_equal(g.download_haps(0, c), o.download_haps(0, c)), f"dense gen {gen} chr {c}"
                pg, og = g.download_intervals(0, c); po, oo = o.download_intervals(0, c)
                assert np.array_equal(og, oo) and np.array_equal(pg, po), f"intervals gen {gen} chr {c}"
                mg, mog = g.download_mutations(0, c); mo, moo = o.download_mutations(0, c)
                assert np.array_equal(mog, moo) and np.array_equal(mg, mo), f"mutations gen {gen} chr {c}"
        couples, seeds = next_couples, seeds_next
    g.close(); o.close()


@pytest.mark.parametrize("case", ["am1", "am2", "ex1sub", "ex1mut", "dense", "syn1k", "sel1", "vc1", "ex1full", "vt2", "om1"])
def test_closed_loop_from_the_seed_alone_on_gpu(gpu_lib, case):
    """the same closed loop on the HIP library (A/D, intervals, genotypes, couples, sexes bit-exact; scaled phenotypes and what
    derives from them within 1e-9 relative: the device's parallel variance / log are not bit-identical to libm's)"""
    helpers.closed_loop_case(gpu_lib, helpers.load_fixture(case), f"gpu/{case}", device=0, exact=False)


def test_config5_in_miniature_plane_less_selection_and_bed_output(gpu_lib):
    """composite of BASELINE config 5 at fixture scale (syn1k from the real reference: logit selection on the phenotype, random
    mating, mutation map, 10 generations): the WHOLE run is driven from --seed alone on a PLANE-LESS context (interval state
    only, gev_set_dense_state 0) -- couples, sexes, pedigree, raw A/D and phenotypes are checked against the reference every
    generation -- and at the end the genotype matrix and the PLINK .bed are assembled from the interval state in SNP tiles:
    the matrix must have the reference's own hash, the .bed must equal the dense context's gev_format_bed."""
    from tests.synth import synth_packed
    from geneevolve_amd.capi import pack_rows, unpack_rows
    fx = helpers.load_fixture("syn1k")
    L = len(fx["pop0_chr0_snp_pos"]); nh = int(fx["pop0_n_founder_hap"]); ngen = int(fx["n_gen"])
    founders = synth_packed(int(fx["pop0_chr0_founders_synth_seed"]), nh, L)
    got = {}

    def dense_end(ctx, sim):
        got["bed"] = ctx.format_bed(0, 0)

    def sparse_end(ctx, sim):
        n = ctx.pop_size(0)
        rows = []
        beds = []
        for s0 in range(0, L, 2048):                                 # SNP tiles: the full matrix is never resident
            ns = min(2048, L - s0)
            tile = pack_rows(unpack_rows(founders, L)[:, s0:s0 + ns])
            rows.append(unpack_rows(ctx.materialize(0, 0, tile, 0, 2 * n, s0, ns), ns))
            beds.append(ctx.materialize_bed(0, 0, [tile], s0, ns))
        full = np.concatenate(rows, axis=1)
        packed = np.packbits(full, axis=1, bitorder="little")
        assert np.array_equal(helpers.sha(packed), fx[f"g{ngen}_pop0_chr0_dense_sha"]), "genotype matrix assembled from the interval state != the reference's"
        got["bed_tiles"] = np.concatenate(beds)
        with pytest.raises(capi.GevError):
            ctx.download_haps(0, 0)                                  # no resident planes in this mode

    helpers.closed_loop_case(gpu_lib, fx, "gpu/config5-mini dense", device=0, exact=False, at_end=dense_end)
    helpers.closed_loop_case(gpu_lib, fx, "gpu/config5-mini plane-less", device=0, exact=False, plane_less=True, at_end=sparse_end)
    assert np.array_equal(got["bed_tiles"], got["bed"]), ".bed from the interval state != .bed of the resident planes"


def test_info_files_with_device_phenotype_scaling(gpu_lib, oracle_lib):
    """gev_scale_ad_compute_gef agrees with the reference to ~1e-12 relative, not bit for bit (device log(), parallel variance).
    What that means for the reference's own output: the .info.popK.genG.txt text (6 significant digits) produced from the
    DEVICE's phenotype values is compared field by field with the text of the bit-exact oracle build (which equals the
    reference's files byte for byte, tests/test_oracle_golden.py).  Reported: how many of the fields differ."""
    total = differing = 0
    for case in ("dense", "am1", "vc1", "syn1k"):
        fx = helpers.load_fixture(case)
        t_gpu, t_ref = [], []
        helpers.closed_loop_case(gpu_lib, fx, f"gpu/{case}", device=0, exact=False, info_texts=t_gpu)
        helpers.closed_loop_case(oracle_lib, fx, f"oracle/{case}", exact=True, info_texts=t_ref)
        assert len(t_gpu) == len(t_ref) == int(fx["n_gen"]) + 1
        for a, b in zip(t_gpu, t_ref):
            la, lb = a.decode().splitlines(), b.decode().splitlines()
            assert len(la) == len(lb)
            for x, y in zip(la, lb):
                fa, fb = x.split(), y.split()
                assert len(fa) == len(fb)
                total += len(fa)
                for u, v in zip(fa, fb):
                    if u != v:
                        differing += 1
                        assert abs(float(u) - float(v)) <= 2e-6 * max(abs(float(v)), 1e-300) + 1e-12, f"{case}: field {u} vs {v}"
    print(f".info files from device phenotype scaling: {differing} of {total} text fields differ from the reference's (last printed digit)")
    assert differing <= total * 1e-3


@pytest.mark.parametrize("case", ["mig2", "gam2"])
def test_closed_loop_two_populations_with_migration_on_gpu(gpu_lib, case):
    helpers.closed_loop_migration_case(gpu_lib, helpers.load_fixture(case), f"gpu/{case}", device=0, exact=False)


def test_nan_effect_sizes_are_reported_like_the_reference(gpu_lib):
    """Simulation::ras_compute_AD returns false with "Error: A or D is nan for human ..." (src/Simulation.cpp:2716-2720);
    gev_compute_ad returns GEV_ENAN with the same line instead of handing NaNs back"""
    cfg = SyntheticConfig(64, 500, chrom_bp=200_000, map_step=10_000, n_cv=12, seed=2)
    g = gpu_lib.create(1, 1, 1)
    cfg.apply_static(g)
    bp, a, d = cfg.cv[0][0]
    a = a.copy(); a[5] = np.nan
    g.set_cvs(0, 0, 0, bp, a, d, 0.0)
    g.synth_founders(0, 0, 128, 5); g.synth_cv_founders(0, 0, 0, 128, 6)
    g.init_gen0(0, 64, 11)
    with pytest.raises(capi.GevError) as e:
        g.compute_ad(0)
    assert "A or D is nan for human" in str(e.value)
    g.close()


# ---------------------------------------------------------------------------------------------------------------
# round 3: Simulation::random_mate and Simulation::ras_glob_seed on the device, a whole generation in one call
# ---------------------------------------------------------------------------------------------------------------
def test_device_glob_seeds_equal_the_host_stream(gpu_lib):
    """gev_glob_seeds (jump-ahead candidates, rejections ranked out by two passes) == the sequential ras_glob_seed() stream, values
    and engine state behind them, for counts below / at / above one block of candidates and at config-2's size"""
    g = gpu_lib.create(1, 1, 1)
    for seed, n in ((12345, 1), (1, 2), (2147483646, 4095), (7, 4096), (99, 4097), (987654321, 100002), (31337, 1400003)):
        s = GlobSeedStream(seed); st0 = s.x
        want = s.draw(n)
        got, st = g.glob_seeds(st0, n)
        assert np.array_equal(got, want), (seed, n)
        assert st == s.x, (seed, n)
    _, st = g.glob_seeds(5, 1000, want=False)
    s = GlobSeedStream(5); s.draw(1000)
    assert st == s.x
    with pytest.raises(capi.GevError):
        g.glob_seeds(0, 10)                                   # not a state of minstd_rand0
    g.close()


@pytest.mark.parametrize("case", ["ex1mut", "dense", "syn1k", "sel1", "mig2", "mig3c"])
def test_device_random_mate_reproduces_the_reference_couples(gpu_lib, oracle_lib, case):
    """gev_random_mate against every --RM fixture: the reference's own seed (:2092), sexes and selection_value_func values go in,
    its couples must come out; the couples then stay on the device (gev_reproduce with couples == NULL) and the generation's sexes
    and A/D must be the reference's too.  mig2: after gev_migrate the sexes the library mates on followed the migrants."""
    fx = helpers.load_fixture(case)
    n_pop, nchr, nphen, ngen = int(fx["n_pop"]), int(fx["nchr"]), int(fx["nphen"]), int(fx["n_gen"])
    ctx = gpu_lib.create(n_pop, nchr, nphen, 0)
    helpers.setup_static(ctx, fx)
    for ip, seed in enumerate(helpers.find_gen0_seeds(fx, oracle_lib)):
        ctx.init_gen0(ip, len(fx[f"g0_pop{ip}_sex"]), seed)
    for g in range(1, ngen + 1):
        for ip in range(n_pop):
            pre = f"g{g}_pop{ip}_"
            assert int(fx[pre + "mate_rm"]) == 1
            svf, n = fx[pre + "mate_svf"], int(fx[pre + "mate_popsize"])
            svf_arg = None if (g % 2 == 0 and np.all(svf == 1.0)) else svf
            couples, nm, nf = ctx.random_mate(ip, int(fx[pre + "mate_seed"]), svf_arg, n)
            want = fx[pre + "couples"]
            assert np.array_equal(couples["pos_male"].astype(np.int64), want[:, 0]) and np.array_equal(couples["pos_female"].astype(np.int64), want[:, 1]), f"{case}: couples gen {g} pop {ip}"
            assert np.array_equal(couples["inbreed"], want[:, 2]) and np.array_equal(couples["num_offspring"], want[:, 3])
            if np.all(svf == 1.0):
                sx = fx[pre + "mate_sex"]
                assert (nm, nf) == (int(np.sum(sx == 1)), int(np.sum(sx == 2)))
            ms = fx[pre + "mut_seeds"]
            sex = ctx.reproduce(ip, None, int(fx[pre + "seed_reproduce"]), ms if len(ms) else None, n_people=n)
            assert np.array_equal(sex, fx[pre + "sex"]), f"{case}: sex gen {g} pop {ip}"
            add, dom, _, _ = ctx.compute_ad(ip)
            assert helpers.bits_equal(add, fx[pre + "additive"]) and helpers.bits_equal(dom, fx[pre + "dominance"]), f"{case}: A/D gen {g} pop {ip}"
        if f"g{g}_moves" in fx:
            ctx.migrate(helpers.derive_moves(fx, g))
    with pytest.raises(capi.GevError):
        ctx.reproduce(0, None, 1, None, n_people=5)            # no couples are waiting on the device
    ctx.close()


@pytest.mark.parametrize("mate", ["device", "fused"])
@pytest.mark.parametrize("case", ["ex1mut", "dense", "syn1k", "sel1", "vt2"])
def test_closed_loop_from_the_seed_alone_with_device_random_mate(gpu_lib, case, mate):
    """the whole run from --seed alone with mating on the device; "fused": gev_generation_begin/_end draw the generation's
    ras_glob_seed() values themselves (the host's engine state goes in, the state behind the draws comes back) -- the seeds,
    couples, sexes, pedigree, A/D and phenotypes are the reference's"""
    helpers.closed_loop_case(gpu_lib, helpers.load_fixture(case), f"gpu/{case}/{mate}", device=0, exact=False, mate=mate)


@pytest.mark.parametrize("mate", ["device", "fused"])
@pytest.mark.parametrize("case", ["mig2", "gam2"])
def test_closed_loop_two_populations_with_migration_and_device_random_mate(gpu_lib, case, mate):
    helpers.closed_loop_migration_case(gpu_lib, helpers.load_fixture(case), f"gpu/{case}/{mate}", device=0, exact=False, mate=mate)


def _pair(gpu_lib, oracle_lib, cfg, n0, seed_f=70):
    g = gpu_lib.create(1, cfg.nchr, cfg.nphen); o = oracle_lib.create(1, cfg.nchr, cfg.nphen)
    cfg.apply_static(g); cfg.apply_static(o)
    for c in range(cfg.nchr):
        g.synth_founders(0, c, 2 * n0, seed_f + c); o.upload_founders(0, c, synth_packed(seed_f + c, 2 * n0, cfg.n_loci), cfg.n_loci)
        ncv = len(cfg.cv[0][c][0])
        g.synth_cv_founders(0, 0, c, 2 * n0, seed_f + 10 + c); o.upload_cv_founders(0, 0, c, synth_packed(seed_f + 10 + c, 2 * n0, ncv), ncv)
    return g, o


def _same_state(g, o, nchr, what):
    for c in range(nchr):
        assert np.array_equal(g.download_haps(0, c), o.download_haps(0, c)), f"dense {what} chr {c}"
        pg, og = g.download_intervals(0, c); po, oo = o.download_intervals(0, c)
        assert np.array_equal(og, oo) and np.array_equal(pg, po), f"intervals {what} chr {c}"
        mg, mog = g.download_mutations(0, c); mo, moo = o.download_mutations(0, c)
        assert np.array_equal(mog, moo) and np.array_equal(mg, mo), f"mutations {what} chr {c}"


@pytest.mark.parametrize("with_mut", [True, False])
def test_whole_generations_on_the_device_against_the_oracle_through_redo_paths(gpu_lib, oracle_lib, monkeypatch, with_mut):
    """gev_generation_begin/_end for 30 generations against the oracle's sequential statement of random_mate -> reproduce ->
    ras_compute_AD, with selection values that exclude part of the population, a changing population size, and buffers that are
    too small on purpose (tiny overflow regions, no list headroom): generations are enqueued again inside _end -- seeds and couples
    are drawn again on the device from the retained engine state -- and must come out the same.  with_mut = False: without a
    mutation map reproduce draws ONE ras_glob_seed() value and the rand() chain runs through every gamete (serial-chain mode)."""
    monkeypatch.setenv("GEV_OVF_CAP", "8"); monkeypatch.setenv("GEV_LIST_HEADROOM", "0")
    cfg = SyntheticConfig(150, 1500, nchr=2, chrom_bp=1_000_000, map_step=5_000, rec_per_row=0.03, mut_per_row=0.02, n_cv=25, seed=37, with_mutation=with_mut)
    g, o = _pair(gpu_lib, oracle_lib, cfg, 150)
    sg, so = Simulation(g, 8, 2, with_mut), Simulation(o, 8, 2, with_mut)
    sg.ras_initial_human_gen0(0, 150); so.ras_initial_human_gen0(0, 150)
    rng = np.random.default_rng(12)
    for gen in range(1, (31 if with_mut else 9)):
        n = 150 if gen % 6 else 190
        svf = None if gen % 3 == 0 else rng.uniform(0.2, 1.4, len(sg.sex[0]))
        ra = sg.next_generation_rm(0, n, svf, want_couples=True); rb = so.next_generation_rm(0, n, svf, want_couples=True)
        for k in ("glob_state", "seed_mate", "seed_reproduce", "num_males_mate", "num_females_mate"):
            assert ra[k] == rb[k], (gen, k, ra[k], rb[k])
        assert np.array_equal(ra["couples"], rb["couples"]), f"couples gen {gen}"
        assert np.array_equal(ra["sex"], rb["sex"]), f"sex gen {gen}"
        assert sg.glob.x == so.glob.x
        xa, xo = g.compute_ad(0), o.compute_ad(0)
        for x, y in zip(xa, xo):
            assert helpers.bits_equal(x, y), f"A/D gen {gen}"
        if gen % 10 == 0 or not with_mut:
            _same_state(g, o, 2, f"gen {gen}")
    if with_mut:
        assert g.redo_count() >= 1, "the undersized buffers were meant to force generations to be enqueued again"
    g.close(); o.close()


def test_pipelined_host_loop_survives_a_redo_while_the_next_head_start_is_queued(gpu_lib, oracle_lib, monkeypatch):
    """bench.py's --host-mating order of calls with buffers that are too small on purpose: gev_reproduce_end enqueues generation g
    again (larger record regions) while the head start of g+1 -- sampled with the old regions -- is already queued; the following
    gev_presample_sex must sample again from the retained inputs instead of failing (round-2 advisor finding)."""
    monkeypatch.setenv("GEV_OVF_CAP", "8"); monkeypatch.setenv("GEV_LIST_HEADROOM", "0")
    cfg = SyntheticConfig(120, 1500, nchr=2, chrom_bp=1_000_000, map_step=5_000, rec_per_row=0.03, mut_per_row=0.02, n_cv=25, seed=35)
    g, o = _pair(gpu_lib, oracle_lib, cfg, 120)
    sg, so = Simulation(g, 3, 2, True), Simulation(o, 3, 2, True)
    sg.ras_initial_human_gen0(0, 120); so.ras_initial_human_gen0(0, 120)
    rng = np.random.default_rng(6)
    n = 120
    seeds = sg.ras_glob_seed(1 + 2 * n)
    sg.presample(0, seeds, n)
    couples = synthetic_random_mate(sg.sex[0], n, rng)
    for gen in range(1, 25):
        sex_early = g.presample_sex(0, n)
        seeds_next = sg.ras_glob_seed(1 + 2 * n)
        g.reproduce_begin(0, couples, int(seeds[0]), seeds[1:], n_people=n)
        sg.presample(0, seeds_next, n)
        next_couples = synthetic_random_mate(sex_early, n, rng)
        sex = g.reproduce_end()
        assert np.array_equal(sex, sex_early)
        so.couples[0] = couples
        assert np.array_equal(so.reproduce(0, gen, seeds=seeds, n_people=n), sex), f"sex gen {gen}"
        xa, xo = g.compute_ad(0), o.compute_ad(0)
        assert helpers.bits_equal(xa[0], xo[0]), f"A gen {gen}"
        if gen % 8 == 0:
            _same_state(g, o, 2, f"gen {gen}")
        couples, seeds = next_couples, seeds_next
    assert g.redo_count() >= 1
    g.close(); o.close()


def test_no_one_can_marry_is_reported_like_the_reference(gpu_lib, oracle_lib):
    """Simulation::random_mate prints "Error: No one can marry, ..." and returns false (src/Simulation.cpp:2125-2129):
    GEV_ENOMATE with that line from gev_random_mate and from gev_generation_end, nothing is published, the population lives on"""
    cfg = SyntheticConfig(64, 500, chrom_bp=200_000, map_step=10_000, n_cv=12, seed=2)
    g, o = _pair(gpu_lib, oracle_lib, cfg, 64)
    sg, so = Simulation(g, 4, 1, True), Simulation(o, 4, 1, True)
    sg.ras_initial_human_gen0(0, 64); so.ras_initial_human_gen0(0, 64)
    nobody = np.zeros(64)
    only_f = np.where(sg.sex[0] == 2, 1.0, 0.0)
    for lib_ctx in (g, o):
        with pytest.raises(capi.GevError) as e:
            lib_ctx.random_mate(0, 77, nobody, 64)
        assert e.value.code == -6 and "No one can marry, num_males_mate=0, num_females_mate=0" in str(e.value)
        lib_ctx.generation_begin(0, 4242, 64, only_f)
        with pytest.raises(capi.GevError) as e:
            lib_ctx.generation_end()
        assert e.value.code == -6 and f"num_males_mate=0, num_females_mate={int(np.sum(sg.sex[0] == 2))}" in str(e.value)
        assert lib_ctx.pop_size(0) == 64
    ra = sg.next_generation_rm(0, 80, None, want_couples=True); rb = so.next_generation_rm(0, 80, None, want_couples=True)
    assert np.array_equal(ra["couples"], rb["couples"]) and np.array_equal(ra["sex"], rb["sex"])
    _same_state(g, o, 1, "after the refused generations")
    g.close(); o.close()


def test_mutations_on_cv_positions_flip_the_resolved_allele_once_and_are_inherited(gpu_lib, oracle_lib):
    """ras_find_cv reads !founder where the CV position is in the covering part's mutation_pos (src/Simulation.cpp:2770-2775); the
    device keeps the RESOLVED alleles in the CV plane, inherited by the crossover pattern; a new mutation on a CV position flips the
    allele unless the position already is in the inherited mutation set (looked up in the parental list).  A tiny coordinate range makes most mutations land on a CV position (duplicate CV positions,
    CVs outside the map range, positions hit twice): CV matrix, allele frequencies and A/D against the oracle for 10 generations,
    with migrations between two populations in the middle."""
    rs = np.random.RandomState(5)
    bp = (7 + 10 * np.arange(21)).astype(np.uint64)
    prob = np.r_[0.0, np.full(20, 0.15)]; rate = np.r_[0.0, np.full(20, 0.35)]
    pos = np.sort(rs.randint(0, 230, 150)).astype(np.uint64)
    n = 90
    g = gpu_lib.create(2, 1, 2); o = oracle_lib.create(2, 1, 2)
    for ctx in (g, o):
        for pop in (0, 1):
            ctx.set_rmap(pop, 0, bp, prob, 10); ctx.set_mutmap(pop, 0, bp, rate); ctx.set_snps(pop, 0, pos)
    for p in range(2):
        cvbp = rs.randint(0, 225, 70).astype(np.uint64)                      # unsorted, with duplicates, some outside [bp0, bp_end)
        a, d = rs.randn(70), rs.randn(70)
        for pop in (0, 1):
            V = synth_packed(40 + 2 * p + pop, 2 * n, 70)
            for ctx in (g, o):
                ctx.set_cvs(pop, p, 0, cvbp, a + pop, d, 0.5 if p else 0.0)
                ctx.upload_cv_founders(pop, p, 0, V, 70)
    for pop in (0, 1):
        F = synth_packed(30 + pop, 2 * n, 150)
        g.upload_founders(pop, 0, F, 150); o.upload_founders(pop, 0, F, 150)
    sg, so = Simulation(g, 77, 1, True), Simulation(o, 77, 1, True)
    for pop in (0, 1):
        sg.ras_initial_human_gen0(pop, n); so.ras_initial_human_gen0(pop, n)
    for gen in range(1, 11):
        for pop in (0, 1):
            npop = g.pop_size(pop)
            ra = sg.next_generation_rm(pop, npop + (3 if gen % 2 else -2), want_couples=True); rb = so.next_generation_rm(pop, npop + (3 if gen % 2 else -2), want_couples=True)
            assert np.array_equal(ra["couples"], rb["couples"]) and np.array_equal(ra["sex"], rb["sex"]), (gen, pop)
            for x, y in zip(g.compute_ad(pop), o.compute_ad(pop)):
                assert helpers.bits_equal(x, y), f"A/D gen {gen} pop {pop}"
            for p in range(2):
                assert np.array_equal(g.download_cv(pop, p, 0), o.download_cv(pop, p, 0)), f"CV matrix gen {gen} pop {pop} phen {p}"
                assert helpers.bits_equal(g.get_cv_freq(pop, p, 0), o.get_cv_freq(pop, p, 0))
        if gen in (4, 7):
            moves = [(0, 5, 1), (0, 17, 1), (1, 3, 0), (1, 40, 0), (1, 41, 0)]
            g.migrate(moves); o.migrate(moves)
    mg, _ = g.download_mutations(0, 0)
    cv_all = set(int(x) for x in cvbp)
    assert sum(int(x) in cv_all for x in mg) > 50, "the scenario was meant to put many mutations on CV positions"
    g.close(); o.close()


@pytest.mark.parametrize("n_cv", [300, 1000])
def test_ad_with_root_population_specific_effects_in_position_ordered_cv_files(gpu_lib, oracle_lib, n_cv):
    """Three populations whose CV files (in position order) carry DIFFERENT a and d: after migration a haplotype's effects are its
    root population's (src/Simulation.cpp:2695-2696).  k_ad_accumulate_rp reads the two root-population bit sub-rows of the CV
    plane and the per-population a / d of 128 CVs at a time; 300 CVs = rows of 10 words (4-byte loads, a ragged last piece), 1000
    CVs = rows of 32 words (16-byte loads, eight pieces).  Two phenotypes (vd = 0 and vd = 0.5), migrations in both directions:
    A/D of every population bit for bit against the oracle, whose statements are the reference's."""
    rs = np.random.RandomState(11)
    R = 41
    bp = (100 + 5000 * np.arange(R)).astype(np.uint64)
    prob = np.r_[0.0, np.full(R - 1, 0.04)]; rate = np.r_[0.0, np.full(R - 1, 0.05)]
    L = 2000
    pos = np.sort(rs.randint(0, 5000 * R + 500, L)).astype(np.uint64)
    n, npop = 120, 3
    g = gpu_lib.create(npop, 1, 2); o = oracle_lib.create(npop, 1, 2)
    for ctx in (g, o):
        for pop in range(npop):
            ctx.set_rmap(pop, 0, bp, prob, 5000); ctx.set_mutmap(pop, 0, bp, rate); ctx.set_snps(pop, 0, pos)
    for p in range(2):
        cvbp = np.sort(rs.randint(0, 5000 * R + 400, n_cv)).astype(np.uint64)   # position order, duplicates, a few outside [bp0, bp_end)
        a, d = rs.randn(n_cv), rs.randn(n_cv)
        for pop in range(npop):
            V = synth_packed(60 + 3 * p + pop, 2 * n, n_cv)
            for ctx in (g, o):
                ctx.set_cvs(pop, p, 0, cvbp, a * (1 + 0.5 * pop) + 0.25 * pop, d - 0.125 * pop, 0.5 if p else 0.0)
                ctx.upload_cv_founders(pop, p, 0, V, n_cv)
    for pop in range(npop):
        F = synth_packed(70 + pop, 2 * n, L)
        g.upload_founders(pop, 0, F, L); o.upload_founders(pop, 0, F, L)
    sg, so = Simulation(g, 31, 1, True), Simulation(o, 31, 1, True)
    for pop in range(npop):
        sg.ras_initial_human_gen0(pop, n); so.ras_initial_human_gen0(pop, n)
    for gen in range(1, 8):
        for pop in range(npop):
            npp = g.pop_size(pop)
            ra = sg.next_generation_rm(pop, npp, want_couples=True); rb = so.next_generation_rm(pop, npp, want_couples=True)
            assert np.array_equal(ra["couples"], rb["couples"]) and np.array_equal(ra["sex"], rb["sex"]), (gen, pop)
            for x, y in zip(g.compute_ad(pop), o.compute_ad(pop)):
                assert helpers.bits_equal(x, y), f"A/D gen {gen} pop {pop}"
        if gen in (2, 4, 5):
            moves = [(0, 5, 1), (0, 17, 2), (0, 60, 1), (1, 3, 0), (1, 40, 2), (1, 41, 0), (2, 7, 0), (2, 8, 1)]
            g.migrate(moves); o.migrate(moves)
    parts, _ = g.download_intervals(0, 0)
    assert len(set(int(x) for x in parts["root_population"])) == 3, "population 0 was meant to hold parts of all three root populations"
    g.close(); o.close()


@pytest.mark.parametrize("head_start,tight", [("0", False), ("1", False), ("0", True), ("1", True)])
def test_head_start_across_generations_is_only_a_schedule(gpu_lib, oracle_lib, monkeypatch, head_start, tight):
    """gev_set_generation_chain: the library draws the next generation's seeds from the PREDICTED glob_generator state and samples
    ahead.  Generations whose gev_generation_begin arrives with the predicted state use the head start, the others (the host drew a
    different number of values in between, another size, a redo in between) sample again -- states are the oracle's either way.
    GEV_HEAD_START=1: the head start is enqueued in front of the generation's own work and the next generation waits for its seeds
    and its sampling separately.  tight: buffers too small on purpose, generations are enqueued again while a head start is queued."""
    monkeypatch.setenv("GEV_HEAD_START", head_start)
    if tight:
        monkeypatch.setenv("GEV_OVF_CAP", "8"); monkeypatch.setenv("GEV_LIST_HEADROOM", "0")
    cfg = SyntheticConfig(200, 3000, nchr=2, chrom_bp=2_000_000, map_step=10_000, rec_per_row=0.01, mut_per_row=0.01, n_cv=40, seed=19)
    g, o = _pair(gpu_lib, oracle_lib, cfg, 200)
    sg, so = Simulation(g, 21, 2, True), Simulation(o, 21, 2, True)
    sg.ras_initial_human_gen0(0, 200); so.ras_initial_human_gen0(0, 200)
    g.set_generation_chain(2)                                   # "I draw two values between generations" ...
    rng = np.random.default_rng(5)
    for gen in range(1, 15):
        n = 200 if gen != 7 else 230                              # (a size the head start was not made for)
        # selection values on some generations, and people who leave between two generations (positions are no longer rows)
        svf = rng.uniform(0.3, 1.5, len(sg.sex[0])) if gen in (4, 10) else None
        ra = sg.next_generation_rm(0, n, svf, want_couples=True); rb = so.next_generation_rm(0, n, svf, want_couples=True)
        assert ra["glob_state"] == rb["glob_state"] and np.array_equal(ra["couples"], rb["couples"]) and np.array_equal(ra["sex"], rb["sex"]), gen
        assert ra["num_males_mate"] == rb["num_males_mate"] and ra["num_females_mate"] == rb["num_females_mate"] and ra["seed_mate"] == rb["seed_mate"], gen
        if gen in (5, 11):
            gone = [3, 10, 57, 199]
            g.remove_rows(0, gone); o.remove_rows(0, gone)
            sg.sex[0] = np.delete(sg.sex[0], gone); so.sex[0] = np.delete(so.sex[0], gone)
        for x, y in zip(g.compute_ad(0), o.compute_ad(0)):
            assert helpers.bits_equal(x, y), f"A/D gen {gen}"
        k = 2 if gen % 3 else 5                                   # ... and keep the promise two times out of three
        sg.ras_glob_seed(k); so.ras_glob_seed(k)
        if gen % 4 == 0:
            _same_state(g, o, 2, f"gen {gen}")
    # the published generation's A/D can be read while the next generation is in flight (bench.py's order: hand over first)
    want = o.compute_ad(0, per_chr=False)
    g.generation_begin(0, sg.glob.x, 200)
    got = g.compute_ad(0, per_chr=False)
    assert helpers.bits_equal(got[0], want[0]) and helpers.bits_equal(got[1], want[1])
    with pytest.raises(capi.GevError):
        g.compute_ad(0, per_chr=True)                         # needs the device-side arrays: not while a generation is pending
    r = g.generation_end(want_couples=True); sg.glob.x = r["glob_state"]; sg.sex[0] = r["sex"]
    rb = so.next_generation_rm(0, 200, want_couples=True)
    assert np.array_equal(r["couples"], rb["couples"]) and np.array_equal(r["sex"], rb["sex"])
    g.set_generation_chain(None)
    ra = sg.next_generation_rm(0, 200, want_couples=True); rb = so.next_generation_rm(0, 200, want_couples=True)
    assert np.array_equal(ra["couples"], rb["couples"]) and np.array_equal(ra["sex"], rb["sex"])
    _same_state(g, o, 2, "end")
    if tight:
        assert g.redo_count() >= 1, "the undersized buffers were meant to force generations to be enqueued again"
    g.close(); o.close()
