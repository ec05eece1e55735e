#!/usr/bin/env python3
"""Rate of K8 (genotype tiles from the interval state, gev_materialize_bed) on a plane-less context of config-2 size:
loci x individuals per second, to set next to the reference's ras_convert_interval_to_hap_matrix (about 8e7 loci*ind/s at
11 parts per haplotype).  usage: python tools/k8_rate.py [n_ind] [n_loci] [tile_snps] [generations]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geneevolve_amd.capi import GevLibrary                                    # noqa: E402
from geneevolve_amd.host import Simulation, SyntheticConfig, synthetic_random_mate   # noqa: E402

n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000, int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
tile, gens = int(sys.argv[3]) if len(sys.argv) > 3 else 16384, int(sys.argv[4]) if len(sys.argv) > 4 else 10
cfg = SyntheticConfig(n, L, n_cv=1000, seed=12345)
ctx = GevLibrary().create(1, 1, 1, 0)
ctx.set_dense_state(False)
cfg.apply_static(ctx)
ctx.synth_cv_founders(0, 0, 0, 2 * n, 77)
sim = Simulation(ctx, 1, 1, True)
sim.ras_initial_human_gen0(0, n)
rng = np.random.default_rng(0)
for g in range(1, gens + 1):
    sim.couples[0] = synthetic_random_mate(sim.sex[0], n, rng)
    sim.reproduce(0, g, n_people=n)
_, off = ctx.download_intervals(0, 0)
founders = np.random.default_rng(1).integers(0, 2**63, size=(2 * n, (tile + 63) // 64), dtype=np.int64).astype(np.uint64)   # any founder tile will do for a rate
ctx.materialize_bed(0, 0, [founders], 0, tile)                                # warm-up (allocations)
t0 = time.perf_counter()
reps = 3
for r in range(reps):
    ctx.materialize_bed(0, 0, [founders], (r * tile) % (L - tile), tile)
dt = (time.perf_counter() - t0) / reps
print(f"{n} individuals, {off[-1] / (2 * n):.1f} parts per haplotype after {gens} generations: tile of {tile} SNPs x {2*n} haplotypes -> .bed in {dt*1e3:.1f} ms "
      f"(founder tile upload + tile + transpose + pack + download) = {n * tile / dt:.3e} loci*individuals/s")
