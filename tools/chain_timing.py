#!/usr/bin/env python3
"""Serial-chain mode (no mutation map: src/Simulation.cpp:2447-2455): time of one generation at 100k individuals x 1 chromosome
(2001 map rows) with the workgroup-per-link kernel (k_rec_chain_wg) and with the one-wave form (GEV_CHAIN_WG=0), plus the refusal
above GEV_CHAIN_MAX_TASKS.  usage: python3 tools/chain_timing.py [n_individuals]  -> one JSON line"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(n, wg):
    import numpy as np
    from geneevolve_amd.capi import GevLibrary, GevError
    from geneevolve_amd.host import Simulation, SyntheticConfig, synthetic_random_mate
    lib = GevLibrary()
    cfg = SyntheticConfig(n, 100_000, nchr=1, n_cv=1000, seed=12345, with_mutation=False)
    ctx = lib.create(1, 1, 1, 0)
    cfg.apply_static(ctx)
    ctx.synth_founders(0, 0, 2 * n, 1000); ctx.synth_cv_founders(0, 0, 0, 2 * n, 2000)
    sim = Simulation(ctx, 12345, 1, False)
    sim.ras_initial_human_gen0(0, n)
    rng = np.random.default_rng(0)
    times = []
    for g in range(1, 4):
        sim.couples[0] = synthetic_random_mate(sim.sex[0], n, rng)
        t0 = time.perf_counter()
        sim.reproduce(0, g)
        times.append(time.perf_counter() - t0)
    refused = None
    os.environ["GEV_CHAIN_MAX_TASKS"] = str(n // 2)
    try:
        sim.couples[0] = synthetic_random_mate(sim.sex[0], n, rng)
        sim.reproduce(0, 4)
    except GevError as e:
        refused = str(e)
    ctx.close()
    print(json.dumps({"form": "workgroup per link (k_rec_chain_wg)" if wg else "one wave per link (k_rec_chain, GEV_CHAIN_WG=0)", "n_individuals": n, "gametes": 2 * n,
                      "s_per_generation": times, "us_per_gamete": min(times) / (2 * n) * 1e6, "refusal_above_GEV_CHAIN_MAX_TASKS": refused}))


if __name__ == "__main__":
    if len(sys.argv) > 2:
        one(int(sys.argv[1]), sys.argv[2] == "1")
    else:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
        out = []
        for wg in ("1", "0"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), str(n), wg], env=dict(os.environ, GEV_CHAIN_WG=wg), capture_output=True, text=True)
            out.append(json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-800:]})
        print(json.dumps({"workload": f"{n} individuals x 1 chromosome of 100 Mb, 2001 map rows, NO mutation map (serial rand() chain), 100 000 SNPs", "runs": out}))
