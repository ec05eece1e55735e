#!/usr/bin/env python3
"""BASELINE config 4's per-GPU share for real: ONE of the two contexts that hold a 125k-individual population of 22 chromosomes
(227k SNPs each, 5M in all) -- this one owns chromosomes 0-10 (gev_set_chr_active), evaluates the seed chain of all 22 and
everything else for its 11.  Prints ms per generation.  usage: python tools/shard22.py [n_ind] [n_loci_per_chr] [generations]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geneevolve_amd.capi import GevLibrary                                    # noqa: E402
from geneevolve_amd.host import Simulation, SyntheticConfig, synthetic_random_mate   # noqa: E402

n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000, int(sys.argv[2]) if len(sys.argv) > 2 else 227_000
gens = int(sys.argv[3]) if len(sys.argv) > 3 else 12
nchr, mine = 22, range(0, 11)
cfg = SyntheticConfig(n, L, nchr=nchr, n_cv=1000, seed=12345)
ctx = GevLibrary().create(1, nchr, 1, 0)
for c in range(nchr):
    ctx.set_rmap(0, c, cfg.rmap_bp, cfg.rmap_prob, cfg.bp_dist)
    ctx.set_mutmap(0, c, cfg.mut_bp, cfg.mut_rate)
    ctx.set_chr_active(c, c in mine)
    if c in mine:
        ctx.set_snps(0, c, cfg.snp_pos)
        bp, a, d = cfg.cv[0][c]
        ctx.set_cvs(0, 0, c, bp, a, d, 0.0)
        ctx.synth_founders(0, c, 2 * n, 1000 + c); ctx.synth_cv_founders(0, 0, c, 2 * n, 2000 + c)
sim = Simulation(ctx, 1, nchr, True)
sim.ras_initial_human_gen0(0, n)
rng = np.random.default_rng(0)
seeds = [sim.ras_glob_seed(1 + n * nchr) for _ in range(gens)]
times = []
for g in range(gens):
    t0 = time.perf_counter()
    sim.couples[0] = synthetic_random_mate(sim.sex[0], n, rng)
    sim.reproduce(0, g + 1, seeds=seeds[g], n_people=n)
    if g + 1 < gens:
        sim.presample(0, seeds[g + 1], n)
    add, dom, addc, domc = sim.ras_compute_AD(0, g + 1, per_chr=True)
    times.append((time.perf_counter() - t0) * 1e3)
ctx.sync()
assert not addc[:, 11:, :].any() and addc[:, :11, :].any()
print(f"{n} individuals x 22 chromosomes, 11 active ({11 * L} of {22 * L} SNPs here): ms per generation {[round(t, 1) for t in times]}; "
      f"steady {np.mean(times[4:]):.1f} ms = {1e3 / np.mean(times[4:]):.1f} generations/s")
