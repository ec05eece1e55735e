#!/usr/bin/env python3
"""Time the drop-in PROGRAM itself: the reference's own command-line host (Main, parameters, readers, mating, phenotype scaling,
per-generation text dump) bound to libgeneevolve_amd.so (oracle/_ref/GeneEvolve_gpu, built by integration/build_gpu_cli.py) next
to the unmodified reference (oracle/_ref/GeneEvolve_ref), on a config-2-shaped input with a bounded founder panel:

    N individuals (default 100 000), 1 chromosome of 100 Mb, 2001 recombination / mutation map rows (5e-4 per row),
    1000 CVs, `--n-snps` SNPs (default 1000: the text panel is 2 bytes per haplotype per SNP), random mating, 3 generations.

Phases are measured from OUTSIDE, without touching either program: the reference prints one marker line per phase of
Simulation::sim_next_generation (src/Simulation.cpp:1890-2082: "random mating", "reproducing", "computing additive and
dominance components", "creating phenotypes", ..., "saving human info") and ends every line with std::endl (a flush), so the
arrival times of those lines on a pipe bracket the phases.

    python tools/cli_timing.py [--exe gpu|ref|both] [--n-ind N] [--n-snps S] [--gens G] [--assortative]

Prints one JSON object (per program: seconds per generation and per phase, averaged over generations >= 1).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MARKERS = {
    "random mating": "mating", "assortative mating": "mating",
    "reproducing": "reproduce (Simulation::reproduce)",
    "computing additive and dominance components": "ras_compute_AD",
    "creating phenotypes": "phenotypes (ras_scale_AD_compute_GEF + variances)",
    "environmental effects specific to each population": "environmental effects",
    "computing mating value and selection value": "mating / selection values",
    "migration": "migration",
    "saving human info": "ras_save_human_info (+ summary)",
    "memory used": "end of generation",
}


def write_inputs(wd, n, nsnp, gens, assortative, seed=12345):
    from bench import write_bits_text
    from geneevolve_amd.host import SyntheticConfig
    from tests.synth import synth_bits
    cfg = SyntheticConfig(n, nsnp, seed=seed)
    j = lambda f: os.path.join(wd, f)
    write_bits_text(j("ref.hap"), np.ascontiguousarray(synth_bits(1, 2 * n, nsnp).T))              # .hap is SNP-major
    with open(j("ref.legend"), "w") as f:
        f.write("id pos al0 al1\n" + "".join(f"rs{i+1} {int(p)} A C\n" for i, p in enumerate(cfg.snp_pos)))
    with open(j("ref.indv"), "w") as f:
        f.write("".join(f"id{i+1}\n" for i in range(n)))
    with open(j("hapaddr.txt"), "w") as f:
        f.write("chr hap legend sample\n" + f"1 {j('ref.hap')} {j('ref.legend')} {j('ref.indv')}\n")
    cM = 1e-6 * (cfg.rmap_bp - cfg.rmap_bp[0]).astype(np.float64)
    with open(j("rmap.txt"), "w") as f:
        f.write("chr bp cM\n" + "".join(f"1 {int(b)} {float(c)!r}\n" for b, c in zip(cfg.rmap_bp, cM)))
    with open(j("mmap.txt"), "w") as f:
        f.write("chr bp mutation_rate\n" + "".join(f"1 {int(b)} {float(r)!r}\n" for b, r in zip(cfg.mut_bp, cfg.mut_rate)))
    bp, a, d = cfg.cv[0][0]
    with open(j("cvinfo.txt"), "w") as f:
        f.write("chr pos a d\n" + "".join(f"1 {int(b)} {float(x)!r} {float(y)!r}\n" for b, x, y in zip(bp, a, d)))
    write_bits_text(j("cv.hap"), np.ascontiguousarray(synth_bits(2, 2 * n, len(bp)).T))
    with open(j("cvaddr.txt"), "w") as f:
        f.write(f"1 {j('cv.hap')}\n")
    with open(j("popinfo.txt"), "w") as f:
        f.write("pop_size mat_cor offspring_dist selection_func selection_func_par1 selection_func_par2\n" +
                (f"{n} 0.4 p thr 1 1\n" if assortative else f"{n} 0 p thr 1 1\n") * gens)
    args = ["--file_gen_info", j("popinfo.txt"), "--file_hap_name", j("hapaddr.txt"), "--file_recom_map", j("rmap.txt"),
            "--file_mutation_map", j("mmap.txt"), "--file_cv_info", j("cvinfo.txt"), "--file_cvs", j("cvaddr.txt"),
            "--va", "0.5", "--vd", "0", "--ve", "0.5", "--seed", str(seed), "--prefix", j("out")]
    if not assortative:
        args.append("--RM")
    return args


def run_timed(exe, args, env=None):
    """run the program, timestamp every stdout line; returns (per-generation list of {phase: seconds}, startup seconds, total)"""
    t0 = time.perf_counter()
    p = subprocess.Popen([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, bufsize=1, env=env)
    gens, cur, phase, t_phase, startup = [], None, None, None, None
    tail = []
    for line in p.stdout:
        now = time.perf_counter()
        s = line.strip()
        tail.append(s); tail = tail[-15:]
        if s.startswith("Generation") or s.startswith("generation"):      # a new generation block begins
            pass
        key = MARKERS.get(s)
        if key is None:
            continue
        if phase is not None:
            cur[phase] = cur.get(phase, 0.0) + (now - t_phase)
        if key == "mating":
            if startup is None:
                startup = now - t0
            cur = {}
            gens.append(cur)
        if cur is None:                                                     # markers of generation 0 (before the first mating)
            phase = None
            continue
        if key == "end of generation":
            phase = None
        else:
            phase, t_phase = key, now
    rc = p.wait()
    if rc != 0:
        raise RuntimeError(f"{exe} exited with {rc}:\n" + "\n".join(tail))
    return gens, startup, time.perf_counter() - t0


def summarise(gens, startup, total):
    keys = []
    for g in gens:
        for k in g:
            if k not in keys:
                keys.append(k)
    per_phase = {k: float(np.mean([g.get(k, 0.0) for g in gens])) for k in keys}
    return {"generations_timed": len(gens), "startup_s (readers, generation 0)": startup, "total_s": total,
            "s_per_generation": float(sum(per_phase.values())), "phase_s": per_phase}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--exe", default="both", choices=["gpu", "ref", "both"])
    ap.add_argument("--n-ind", type=int, default=100_000)
    ap.add_argument("--n-snps", type=int, default=1000)
    ap.add_argument("--gens", type=int, default=3)
    ap.add_argument("--ref-n-ind", type=int, default=0, help="run the unmodified reference on this many individuals instead (its cost is ~27 s/generation at 100k)")
    ap.add_argument("--assortative", action="store_true", help="mat_cor 0.4 (assort_mate: O(n^2) CommFunc::ras_rank on the host, gev_rank_f64 in the bound program)")
    args = ap.parse_args()
    out = {"workload": f"{args.n_ind} individuals, 1 chromosome of 100 Mb, 2001 map rows, 1000 CVs, {args.n_snps} SNPs, "
                       f"{'assortative (mat_cor 0.4)' if args.assortative else 'random'} mating, {args.gens} generations",
           "host_cores_used": "reference: 1 (single threaded); bound program: 1 + up to 7 helper threads inside the glue (record allocation / release, .info formatting)"}
    with tempfile.TemporaryDirectory(prefix="gev_cli_") as wd:
        t0 = time.perf_counter()
        a = write_inputs(wd, args.n_ind, args.n_snps, args.gens, args.assortative)
        out["input_files_written_s"] = time.perf_counter() - t0
        if args.exe in ("gpu", "both"):
            out["GeneEvolve_gpu"] = summarise(*run_timed(os.path.join(ROOT, "oracle", "_ref", "GeneEvolve_gpu"), a))
            if os.environ.get("GEV_CLI_ALSO_DEVICE_GEF"):
                out["GeneEvolve_gpu (GEV_GEF_DEVICE=1)"] = summarise(*run_timed(os.path.join(ROOT, "oracle", "_ref", "GeneEvolve_gpu"), a, dict(os.environ, GEV_GEF_DEVICE="1")))
        if args.exe in ("ref", "both"):
            if args.ref_n_ind and args.ref_n_ind != args.n_ind:
                with tempfile.TemporaryDirectory(prefix="gev_cli_ref_") as wd2:
                    a2 = write_inputs(wd2, args.ref_n_ind, args.n_snps, args.gens, args.assortative)
                    r = summarise(*run_timed(os.path.join(ROOT, "oracle", "_ref", "GeneEvolve_ref"), a2))
                r["n_individuals"] = args.ref_n_ind
            else:
                r = summarise(*run_timed(os.path.join(ROOT, "oracle", "_ref", "GeneEvolve_ref"), a))
                r["n_individuals"] = args.n_ind
            out["GeneEvolve_ref (unmodified reference)"] = r
    print(json.dumps(out))


if __name__ == "__main__":
    main()
