"""CPU tests: pin the oracle (oracle/gev_oracle.cpp) against golden vectors produced by the
REAL reference built from source (tests/golden/make_golden.py).  Bit-exact everywhere,
including the FP64 A/D values (hex-dumped by the harness)."""
import ctypes
import gzip
import os

import numpy as np
import pytest

from geneevolve_amd.capi import PART_DTYPE
from oracle import oracle_api
from tests import helpers

KAT = os.path.join(helpers.GOLDEN, "kat.txt.gz")


def kat_lines(tag):
    with gzip.open(KAT, "rt") as f:
        return [l.split() for l in f if l.startswith(tag + " ")]


def test_glibc_rand_restatement_matches_reference_process(oracle_lib):
    rows = kat_lines("RAND")
    assert len(rows) >= 10
    for t in rows:
        seed = int(t[1]); want = np.array([int(x) for x in t[2:]], dtype=np.int32)
        assert np.array_equal(oracle_api.kat_rand(oracle_lib, seed, len(want)), want), f"srand({seed})"


def test_glibc_rand_restatement_matches_live_libc(oracle_lib):
    libc = ctypes.CDLL("libc.so.6")
    rs = np.random.RandomState(1)
    for seed in [0, 1, 2**31 - 1, 2**31, 2**32 - 1] + [int(x) for x in rs.randint(0, 2**32, size=50, dtype=np.uint64)]:
        libc.srand(ctypes.c_uint(seed))
        want = np.array([libc.rand() for _ in range(100)], dtype=np.int32)
        assert np.array_equal(oracle_api.kat_rand(oracle_lib, seed, 100), want), f"srand({seed})"


def test_minstd_and_canonical(oracle_lib):
    for t in kat_lines("MINSTD"):
        want = np.array([int(x) for x in t[2:]], dtype=np.uint64)
        assert np.array_equal(oracle_api.kat_minstd(oracle_lib, int(t[1]), len(want)), want)
    for t in kat_lines("U01"):
        want = np.array([float.fromhex(x) for x in t[2:]])
        assert helpers.bits_equal(oracle_api.kat_u01(oracle_lib, int(t[1]), len(want)), want)


def test_uniform_int_all_branches(oracle_lib):
    rows = kat_lines("UINT")
    assert len(rows) >= 20
    for t in rows:
        seed, lo, hi = int(t[1]), int(t[2]), int(t[3])
        want = np.array([int(x) for x in t[4:]], dtype=np.uint64)
        assert np.array_equal(oracle_api.kat_uint(oracle_lib, seed + 1, lo, hi, len(want)), want), (seed, lo, hi)
    for t in kat_lines("GLOB"):      # ras_glob_seed stream
        want = np.array([int(x) for x in t[2:]], dtype=np.uint64)
        assert np.array_equal(oracle_api.kat_uint(oracle_lib, int(t[1]), 1, 1000000, len(want)), want)


def test_normal_distribution(oracle_lib):
    for t in kat_lines("NORMAL"):
        want = np.array([float.fromhex(x) for x in t[3:]])
        got = oracle_api.kat_normal(oracle_lib, int(t[1]), float.fromhex(t[2]), len(want))
        # log/sqrt come from libm on both sides in this container: bit-exact here, 1e-15 elsewhere
        assert np.allclose(got, want, rtol=1e-15, atol=0)


def test_sim_loc_rec_direct_calls(oracle_lib):
    m = kat_lines("LOCMAP")[0]
    R, bp0, dist = int(m[1]), int(m[2]), int(m[3])
    prob = np.array([float.fromhex(x) for x in m[4:]])
    bp = (bp0 + dist * np.arange(R)).astype(np.uint64)
    rows = kat_lines("LOC")
    assert len(rows) == 400
    n_last_row_hits = 0
    for t in rows:
        seed = int(t[1]); nxt = [int(t[2]), int(t[3])]; n = int(t[4])
        want = np.array([int(x) for x in t[5:5 + n]], dtype=np.uint64)
        locs, nx = oracle_api.kat_sim_loc_rec(oracle_lib, bp, prob, dist, seed)
        assert np.array_equal(locs, want), f"seed {seed}"
        assert list(nx) == nxt, f"rand() state after ras_sim_loc_rec, seed {seed}"
        n_last_row_hits += int(len(want) > 2 and want[-2] >= want[-1])
    assert n_last_row_hits > 50      # the "hit on the last map row" quirk is exercised


def test_recombine_direct_calls(oracle_lib):
    ins = kat_lines("RECIN"); pin = kat_lines("RECP"); pout = kat_lines("RECO")
    assert len(ins) == 300
    by_t_in, by_t_out = {}, {}
    for t in pin:
        by_t_in.setdefault(int(t[1]), []).append(t)
    for t in pout:
        by_t_out.setdefault(int(t[1]), []).append(t)
    for t in ins:
        k = int(t[1]); start = int(t[2]); n = int(t[3]); locs = [int(x) for x in t[4:4 + n]]
        rows = by_t_in[k]
        parts = np.zeros(len(rows), dtype=PART_DTYPE); mc = []; mu = []; hap_n = [0, 0]
        for i, r in enumerate(rows):
            hap_n[int(r[2])] += 1
            parts[i] = (int(r[3]), int(r[4]), int(r[5]), int(r[6]), 0)
            nm = int(r[7]); mc.append(nm); mu += [int(x) for x in r[8:8 + nm]]
        op, omc, om = oracle_api.kat_recombine(oracle_lib, parts, mc, mu, hap_n, start, locs)
        want = by_t_out.get(k, [])
        assert len(op) == len(want), f"trial {k}: part count"
        wm = []
        for i, r in enumerate(want):
            assert (int(op[i]["st"]), int(op[i]["en"]), int(op[i]["hap_index"]), int(op[i]["root_population"])) == (int(r[2]), int(r[3]), int(r[4]), int(r[5])), f"trial {k} part {i}"
            assert int(omc[i]) == int(r[6])
            wm += [int(x) for x in r[7:7 + int(r[6])]]
        assert list(om) == wm, f"trial {k}: mutation_pos"


@pytest.mark.parametrize("case", ["ex1sub", "ex1mut", "dense", "mig2", "syn1k", "am1", "am2", "sel1", "vc1", "ex1full", "mig3c", "c4mini", "vt2", "vcf1", "gam2", "om1"])
def test_oracle_replays_reference_generations(oracle_lib, case):
    fx = helpers.load_fixture(case)
    seeds = helpers.find_gen0_seeds(fx, oracle_lib)
    n_dense = helpers.replay_case(oracle_lib, fx, seeds, f"oracle/{case}")
    assert n_dense >= 1


@pytest.mark.parametrize("case", ["dense", "ex1sub", "mig2", "mig3c", "c4mini", "vt2"])
def test_oracle_scale_ad_compute_gef_matches_reference(oracle_lib, case):
    """ras_scale_AD_compute_GEF (src/Simulation.cpp:3075-3206) incl. libstdc++ normal_distribution."""
    fx = helpers.load_fixture(case)
    seeds = helpers.find_gen0_seeds(fx, oracle_lib)
    checked = helpers.replay_case(oracle_lib, fx, seeds, f"oracle-gef/{case}", check_lists=False, check_gef=True)
    assert checked >= 0


def test_rank_matches_reference_vectors(oracle_lib):
    """CommFunc::ras_rank vectors (ties, signed zeros, huge values): the oracle's literal loop and the host mirror's stable-sort form"""
    from geneevolve_amd.host import ras_rank
    rows = kat_lines("RANK")
    assert len(rows) == 3
    o = oracle_lib.create(1, 1, 1)
    for t in rows:
        n = int(t[1]); x = np.array([float.fromhex(v) for v in t[2:2 + n]]); want = np.array([int(v) for v in t[2 + n:2 + 2 * n]], dtype=np.uint64)
        assert np.array_equal(o.rank_f64(x), want)
        assert np.array_equal(ras_rank(x), want)
    o.close()


@pytest.mark.parametrize("case", ["am1", "am2", "ex1sub", "ex1mut", "dense", "syn1k", "sel1", "vc1", "ex1full", "vt2", "om1"])
def test_closed_loop_from_the_seed_alone(oracle_lib, case):
    """the reference's whole generation loop (assortative mating, inbreeding avoidance, Poisson / fixed families, logit
    selection) re-driven from --seed by the host mirror on top of the C-ABI (oracle build): bit-identical at every step"""
    helpers.closed_loop_case(oracle_lib, helpers.load_fixture(case), f"oracle/{case}")


@pytest.mark.parametrize("case", ["mig2", "gam2"])
def test_closed_loop_two_populations_with_migration(oracle_lib, case):
    """BASELINE config 3's shape end to end from the seed alone: two populations, random mating, mutation, migration decided by
    the host restatement of ras_do_migration (selection sampling on the process-wide static engine), rows moved by the library.
    gam2: with --gamma, the environmental effects specific to each population (src/Simulation.cpp:3345-3382: Newton-Raphson on the
    combined variance) shift the phenotypes that the logit selection function and the .info files see"""
    helpers.closed_loop_migration_case(oracle_lib, helpers.load_fixture(case), f"oracle/{case}")


@pytest.mark.parametrize("mate", ["device", "fused"])
@pytest.mark.parametrize("case", ["ex1mut", "dense", "syn1k", "sel1", "vt2"])
def test_closed_loop_with_the_library_side_random_mate(oracle_lib, case, mate):
    """pins the oracle's restatement of Simulation::random_mate (src/Simulation.cpp:2090-2157) and of ras_glob_seed (:17-21) behind
    gevo_random_mate / gevo_generation_begin: the reference's couples, seeds, sexes, A/D and .info files of every --RM fixture"""
    helpers.closed_loop_case(oracle_lib, helpers.load_fixture(case), f"oracle/{case}/{mate}", mate=mate)


@pytest.mark.parametrize("mate", ["device", "fused"])
@pytest.mark.parametrize("case", ["mig2", "gam2"])
def test_closed_loop_two_populations_with_migration_and_library_side_random_mate(oracle_lib, case, mate):
    helpers.closed_loop_migration_case(oracle_lib, helpers.load_fixture(case), f"oracle/{case}/{mate}", mate=mate)


def test_oracle_glob_seeds_equal_the_host_stream(oracle_lib):
    from geneevolve_amd.host import GlobSeedStream
    o = oracle_lib.create(1, 1, 1)
    for seed, n in ((12345, 1), (1, 5), (2147483646, 4097), (987654321, 100001)):
        g = GlobSeedStream(seed); st0 = g.x
        want = g.draw(n)
        got, st = o.glob_seeds(st0, n)
        assert np.array_equal(got, want) and st == g.x
    o.close()
