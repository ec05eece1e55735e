"""CPU tests of the host-side mirror (geneevolve_amd/host.py)."""
import gzip
import os

import numpy as np

from geneevolve_amd.host import GlobSeedStream, SyntheticConfig, synthetic_random_mate
from oracle import oracle_api
from tests import helpers


def test_glob_seed_stream_matches_reference_vectors():
    with gzip.open(os.path.join(helpers.GOLDEN, "kat.txt.gz"), "rt") as f:
        rows = [l.split() for l in f if l.startswith("GLOB ")]
    assert rows
    for t in rows:
        want = np.array([int(x) for x in t[2:]], dtype=np.uint32)
        g = GlobSeedStream(int(t[1]))
        got = np.concatenate([g.draw(5), g.draw(len(want) - 5)])      # split draws keep the engine state
        assert np.array_equal(got, want)


def test_glob_seed_stream_long_run_with_rejections(oracle_lib):
    n = 200000                                                   # ~45 rejections expected
    want = oracle_api.kat_uint(oracle_lib, 12345, 1, 1000000, n).astype(np.uint32)
    g = GlobSeedStream(12345)
    got = np.concatenate([g.draw(70000), g.draw(1), g.draw(n - 70001)])
    assert np.array_equal(got, want)


def test_fixture_gen0_seed_is_first_glob_draw(oracle_lib):
    fx = helpers.load_fixture("ex1sub")
    seeds = helpers.find_gen0_seeds(fx, oracle_lib)
    assert seeds[0] == int(GlobSeedStream(int(fx["seed"])).draw(1)[0])


def test_synthetic_random_mate_shapes():
    rng = np.random.default_rng(0)
    sex = rng.integers(1, 3, 1000).astype(np.uint8)
    c = synthetic_random_mate(sex, 1000, rng)
    assert len(c) == 1000 and (sex[c["pos_male"]] == 1).all() and (sex[c["pos_female"]] == 2).all()
    cfg = SyntheticConfig(100, 1000, n_cv=50)
    assert len(cfg.rmap_bp) == 2001 and cfg.rmap_prob[0] == 0 and len(cfg.snp_pos) == 1000


def test_random_mate_restatement_reproduces_reference_couples():
    """Simulation::random_mate (src/Simulation.cpp:2090-2157) restated in the host mirror, checked
    against the couples the real reference formed (all --RM fixtures; syn1k has a non-constant selection function)."""
    from geneevolve_amd.host import random_mate
    n_checked = 0
    for case in ("syn1k", "ex1mut", "dense", "mig2"):
        fx = helpers.load_fixture(case)
        for g in range(1, int(fx["n_gen"]) + 1):
            for ip in range(int(fx["n_pop"])):
                k = f"g{g}_pop{ip}_mate_"
                assert int(fx[k + "rm"]) == 1
                c = random_mate(fx[k + "sex"], fx[k + "svf"], int(fx[k + "popsize"]), int(fx[k + "seed"]))
                want = fx[f"g{g}_pop{ip}_couples"]
                assert np.array_equal(c["pos_male"].astype(np.int64), want[:, 0]) and np.array_equal(c["pos_female"].astype(np.int64), want[:, 1]), (case, g, ip)
                assert (c["num_offspring"] == want[:, 3]).all() and (c["inbreed"] == want[:, 2]).all()
                n_checked += 1
    assert n_checked >= 20


def test_host_glibc_rand_and_normal_stream_match_reference_vectors(oracle_lib):
    """the host-side streams assort_mate needs, against the reference's own vectors (kat.txt.gz RAND / NORMAL lines)"""
    from geneevolve_amd.host import GlibcRand, normal_stream
    with gzip.open(os.path.join(helpers.GOLDEN, "kat.txt.gz"), "rt") as f:
        lines = [l.split() for l in f]
    n_r = n_n = 0
    for t in lines:
        if t[0] == "RAND":
            g = GlibcRand(int(t[1])); want = [int(x) for x in t[2:]]
            assert [g.rand() for _ in want] == want; n_r += 1
        elif t[0] == "NORMAL":
            sd = float.fromhex(t[2]); want = np.array([float.fromhex(x) for x in t[3:]])
            got = normal_stream(int(t[1]), len(want)) * sd + 0.0      # __ret * stddev + mean
            assert np.allclose(got, want, rtol=1e-15, atol=0), (t[1], np.max(np.abs(got - want))); n_n += 1
    assert n_r >= 3 and n_n >= 1


def test_assort_mate_restatement_reproduces_reference_couples():
    """Simulation::assort_mate (src/Simulation.cpp:2167-2360) restated in the host mirror -- selection
    draws, second spouses (--MM), surplus removal by std::random_shuffle on glibc rand(), the bivariate-normal rank
    template (ras_mvnorm + ras_rank), inbreeding avoidance, Poisson / fixed offspring numbers -- against the couples
    the real reference formed."""
    from geneevolve_amd.host import Pedigree, assort_mate
    n_checked = 0
    for case in ("am1", "am2", "ex1sub"):
        fx = helpers.load_fixture(case)
        for g in range(1, int(fx["n_gen"]) + 1):
            k = f"g{g}_pop0_mate_"
            assert int(fx[k + "rm"]) == 0
            matcor, mm, avoid = fx[k + "am_par"]
            ped = Pedigree(len(fx[k + "sex"]))
            (ped.ID_Father, ped.ID_Fathers_Father, ped.ID_Fathers_Mother, ped.ID_Mothers_Father, ped.ID_Mothers_Mother) = fx[k + "am_ped"].T
            c = assort_mate(fx[k + "sex"], fx[k + "svf"], fx[k + "am_mv"], ped, int(fx[k + "popsize"]), matcor, [int(s) for s in fx[k + "am_seeds"]],
                            mm_percent=mm, avoid_inbreeding=bool(avoid), offspring_dist=bytes(fx[k + "am_dist"]).decode())
            want = fx[f"g{g}_pop0_couples"]
            assert len(c) == len(want), (case, g, len(c), len(want))
            assert np.array_equal(c["pos_male"].astype(np.int64), want[:, 0]) and np.array_equal(c["pos_female"].astype(np.int64), want[:, 1]), (case, g)
            assert np.array_equal(c["inbreed"], want[:, 2]) and np.array_equal(c["num_offspring"], want[:, 3]), (case, g)
            n_checked += 1
    assert n_checked >= 12
