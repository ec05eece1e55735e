"""Where a generation's time goes right after a migration step, on ONE GPU and in one process: population 0 of a two-population
context exports k individuals, removes them and imports them again (the records take the same path as between two GPUs), then
runs the next generation.  Prints the host time of every call, the redo count and the list statistics.
    python tools/migration_probe.py [--n-ind 50000] [--k 500] [--rows]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geneevolve_amd.capi import GevLibrary  # noqa: E402
from geneevolve_amd.host import Simulation, SyntheticConfig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n-ind", type=int, default=50_000)
ap.add_argument("--n-loci", type=int, default=1_000_000)
ap.add_argument("--k", type=int, default=500)
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--rows", action="store_true")
ap.add_argument("--no-migration", action="store_true")
ap.add_argument("--distinct-effects", action="store_true", help="population 1 gets other CV effects than population 0: A/D looks a, d up by root population")
args = ap.parse_args()
lib = GevLibrary()
cfg = SyntheticConfig(args.n_ind, args.n_loci, nchr=1, n_cv=1000, seed=12345, map_step=50_000, rec_per_row=5e-4, mut_per_row=5e-4)
ctx = lib.create(2, 1, 1, 0)
for p in range(2):
    cfg.apply_static(ctx, p)
if args.distinct_effects:
    bp, a, d = cfg.cv[0][0]
    ctx.set_cvs(1, 0, 0, bp, a * 1.5 + 0.1, d, cfg.vd)
ctx.reserve(0, args.n_ind + args.k)                  # room for the immigrants (they are appended before the emigrants' slots are reused)
ctx.synth_founders(0, 0, 2 * args.n_ind, 1000); ctx.synth_cv_founders(0, 0, 0, 2 * args.n_ind, 2000)
if not args.rows:
    ctx.synth_founder_panel(0, 0, 2 * args.n_ind, 1000)
    ctx.set_migrant_rows(False)
sim = Simulation(ctx, 12345, 1, True)
sim.ras_initial_human_gen0(0, args.n_ind)
ctx.set_generation_chain(0)
rng = np.random.default_rng(0)
for i in range(args.steps):
    t = [time.perf_counter()]
    ctx.generation_begin(0, sim.glob.x, args.n_ind, None); t.append(time.perf_counter())
    r = ctx.generation_end(want_couples=False, want_sex=True); t.append(time.perf_counter())
    sim.glob.x = int(r["glob_state"])
    ctx.compute_ad(0); t.append(time.perf_counter())
    names = ["begin", "end", "ad"]
    if not args.no_migration:
        who = np.sort(rng.choice(args.n_ind, size=args.k, replace=False))[::-1].astype(np.uint64)
        nb = ctx.export_size(0, who); t.append(time.perf_counter())
        buf = torch.empty(nb, dtype=torch.uint8, device="cuda:0"); torch.cuda.synchronize(); t.append(time.perf_counter())
        ctx.export_rows(0, who, buf.data_ptr(), nb); t.append(time.perf_counter())
        ctx.remove_rows(0, who); t.append(time.perf_counter())
        ctx.import_rows(0, buf.data_ptr(), nb, len(who)); t.append(time.perf_counter())
        names += ["export_size", "alloc", "export", "remove", "import"]
    d = np.diff(t) * 1e3
    print(f"gen {i + 1}: " + "  ".join(f"{n} {x:.3f}" for n, x in zip(names, d)) + f"  | payload {0 if args.no_migration else nb} B  redo {ctx.redo_count()}", flush=True)
print(ctx.list_stats(0, 0))
