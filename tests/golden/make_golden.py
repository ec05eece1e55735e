#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REAL reference.

Runs only in the build container (needs /root/reference): builds oracle/_ref with
oracle/Makefile.ref (the unmodified reference + oracle/ref_harness.cpp), writes small input
files in the reference's own formats, runs the harness and packs its
lossless dumps into tests/golden/*.npz.  The fixtures are data only: inputs as arrays and the
reference's outputs (couples, ras_glob_seed values, sex, interval lists, mutation lists,
raw A/D, dense haplotype matrices).

    python tests/golden/make_golden.py            # regenerate everything

Cases
  kat.txt.gz   RNG known-answer vectors + direct ras_sim_loc_rec / recombine calls
  ex1sub       Examples.zip:Example1 inputs (first 300 founders), 3 chr, assortative mating,
               Poisson family sizes, NO mutation map  -> serial rand() chain mode
  ex1full      Example1.sh at full size (2000 founders, pop 3000, 10 generations), hashes
  ex1mut       same inputs, --RM, mutation map at 2e-3/row -> task-parallel mode
  dense        tiny synthetic chromosome with SNPs every 7 bp and a hot mutation map so that
               mutations land on SNPs and CVs; 2 phenotypes, unsorted CV file order, vd>0
  mig2         two populations with different founder panels / CV effects and migration
  vc1          common sibling environment (--vc), two phenotypes
  sel1         probit / stabilising selection functions, random mating
  syn1k        BASELINE config-1 shape: 1000 ind x 10000 SNPs, 1 chr of 100 Mb, 10 gen, mutation
"""
import gzip
import hashlib
import os
import shutil
import subprocess
import sys
import zipfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
ORACLE = os.path.join(ROOT, "oracle")
HARNESS = os.path.join(ORACLE, "_ref", "ref_harness")
WORK = "/tmp/gev_golden_work"

sys.path.insert(0, ROOT)
from tests.synth import synth_bits, synth_thresholds  # noqa: E402  (shared deterministic generator)


def sh(cmd, **kw):
    subprocess.run(cmd, check=True, **kw)


# ----------------------------------------------------------------------------- input writers
def write_hap(path, bits):  # bits [nhap][L]; file is SNP-major text (format_hap.cpp:62-121)
    with open(path, "w") as f:
        for i in range(bits.shape[1]):
            f.write(" ".join("1" if b else "0" for b in bits[:, i]) + "\n")


def legend_alleles(n):
    """allele letters of the synthetic legends (they only matter to the PLINK .ped writer)"""
    i = np.arange(n)
    al0 = np.frombuffer(b"ACGT", dtype=np.uint8)[i % 4]
    al1 = np.frombuffer(b"CGTA", dtype=np.uint8)[(i + i // 4) % 4]
    return al0, al1


def write_legend(path, pos):
    al0, al1 = legend_alleles(len(pos))
    with open(path, "w") as f:
        f.write("id pos al0 al1\n")
        for i, p in enumerate(pos):
            f.write(f"rs{i+1} {int(p)} {chr(al0[i])} {chr(al1[i])}\n")


def write_vcf(path, chrom, founders, pos, prefix):
    """phased biallelic VCF of a founder panel [nhap][L] with the same sample ids / alleles as the .indv / .legend files"""
    al0, al1 = legend_alleles(len(pos))
    n = founders.shape[0] // 2
    F = founders.astype(np.uint8)
    with open(path, "w") as f:
        f.write("##fileformat=VCFv4.1\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n")
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"{prefix}{i+1}" for i in range(n)) + "\n")
        for j, p in enumerate(pos):
            gts = "\t".join(f"{F[2*i, j]}|{F[2*i+1, j]}" for i in range(n))
            f.write(f"{chrom}\t{int(p)}\trs{j+1}\t{chr(al0[j])}\t{chr(al1[j])}\t.\tPASS\t.\tGT\t{gts}\n")


def write_indv(path, n, prefix="id"):
    with open(path, "w") as f:
        for i in range(n):
            f.write(f"{prefix}{i+1}\n")


def write_map(path, header, chrs_rows):  # chrs_rows: list of (chr_label, bp[], val[])
    with open(path, "w") as f:
        f.write(header + "\n")
        for c, bp, val in chrs_rows:
            for b, v in zip(bp, val):
                f.write(f"{c} {int(b)} {float(v)!r}\n")


def write_cvinfo(path, chrs_rows):  # (chr, bp[], a[], d[])
    with open(path, "w") as f:
        f.write("chr pos a d\n")
        for c, bp, a, d in chrs_rows:
            for b, x, y in zip(bp, a, d):
                f.write(f"{c} {int(b)} {float(x)!r} {float(y)!r}\n")


def write_popinfo(path, rows):
    with open(path, "w") as f:
        f.write("pop_size mat_cor offspring_dist selection_func selection_func_par1 selection_func_par2\n")
        for r in rows:
            f.write(r + "\n")


# ----------------------------------------------------------------------------- dump parser
def hexf(s):
    return float.fromhex(s)


def parse_dump(path):
    """-> dict of sections of one generation dump written by oracle/ref_harness.cpp"""
    out = {"couples": {}, "seed": {}, "mutseeds": {}, "offspring": {}, "ad": {}, "dense": {},
           "premig": {}, "postmig": {}, "post": {}}
    cur = None
    with open(path) as f:
        for line in f:
            t = line.split()
            k = t[0]
            if k == "GEN":
                out["gen"] = int(t[1]); out["npop"] = int(t[3])
            elif k == "COUPLES":
                cur = out["couples"].setdefault(int(t[2]), [])
            elif k == "C":
                cur.append([int(x) for x in t[1:5]])
            elif k == "SEED_REPRODUCE":
                out["seed"][int(t[2])] = int(t[3])
            elif k == "MUTSEEDS":
                cur = out["mutseeds"].setdefault(int(t[2]), [])
            elif k == "S":
                cur.append(int(t[1]))
            elif k in ("OFFSPRING", "POSTMIG", "POST"):
                cur = {"n": int(t[4]), "nchr": int(t[6]), "H": [], "P": []}
                out[k.lower()][int(t[2])] = cur
            elif k == "H":
                cur["H"].append([int(x) for x in t[1:6]])
            elif k == "P":
                cur["P"].append([int(x) for x in t[1:]])
            elif k == "AD":
                cur = {"n": int(t[4]), "nchr": int(t[6]), "nphen": int(t[8]), "A": []}
                out["ad"][int(t[2])] = cur
            elif k == "A":
                cur["A"].append((int(t[1]), int(t[2]), [hexf(x) for x in t[3:]]))
            elif k == "DENSE":
                cur = []
                out["dense"][(int(t[2]), int(t[4]))] = cur
            elif k == "D":
                cur.append(t[1])
            elif k == "PREMIG":
                cur = out["premig"].setdefault(int(t[2]), [])
            elif k == "I":
                cur.append((int(t[1]), hexf(t[2])))
            elif k == "POSTFP":
                cur = out.setdefault("postfp", {}).setdefault(int(t[2]), [])
            elif k == "J":
                cur.append((int(t[1]), hexf(t[2])))
            elif k == "MATE":
                cur = {"pop": int(t[2]), "rm": int(t[4]), "seed": int(t[6]), "popsize": int(t[8]), "MS": []}
                out.setdefault("mate", []).append(cur)
            elif k == "MS":
                cur["MS"].append((int(t[1]), hexf(t[2])))
            elif k == "MATEP":
                cur["am"] = {"matcor": hexf(t[4]), "mm": hexf(t[6]), "avoid": int(t[8]), "dist": t[10], "seeds": [int(t[12]), int(t[13]), int(t[14])], "MV": []}
            elif k == "MV":
                cur["am"]["MV"].append((hexf(t[1]),) + tuple(int(x) for x in t[2:7]))
            elif k == "GEF":
                cur = {"pop": int(t[2]), "phen": int(t[4]), "seed": int(t[6]), "par": [hexf(t[i]) for i in (8, 10, 12, 14, 16, 18, 20)], "vt": int(t[22]), "GI": [], "GO": []}
                out.setdefault("gef", []).append(cur)
            elif k == "GI":
                cur["GI"].append([hexf(x) for x in t[2:5]])
            elif k == "GO":
                cur["GO"].append([hexf(x) for x in t[2:8]])
            elif k == "GLOBSEQ":
                out["globseq"] = [int(x) for x in t[1:]]
    return out


def pack_humans(sec, prefix, arrs):
    """interval + mutation lists of one population -> CSR arrays keyed by `prefix`"""
    n, nchr = sec["n"], sec["nchr"]
    H = np.array(sec["H"], dtype=np.int64).reshape(n, 5)
    arrs[prefix + "sex"] = H[:, 1].astype(np.uint8)
    arrs[prefix + "ids"] = H[:, 2:5].astype(np.int64)
    parts = [[[] for _ in range(2 * n)] for _ in range(nchr)]
    for p in sec["P"]:
        ih, c, hp = p[0], p[1], p[2]
        parts[c][2 * ih + hp].append(p[3:])
    for c in range(nchr):
        off = [0]; rows = []; moff = [0]; muts = []
        for r in range(2 * n):
            rowm = []
            for q in parts[c][r]:
                rows.append(q[:4])
                rowm.extend(q[5:5 + q[4]])
            off.append(len(rows))
            muts.extend(sorted(rowm))
            moff.append(len(muts))
        arrs[f"{prefix}chr{c}_part_off"] = np.array(off, dtype=np.uint64)
        arrs[f"{prefix}chr{c}_parts"] = np.array(rows, dtype=np.int64).reshape(-1, 4)
        arrs[f"{prefix}chr{c}_mut_off"] = np.array(moff, dtype=np.uint64)
        arrs[f"{prefix}chr{c}_muts"] = np.array(muts, dtype=np.uint64)


def pack_ad(sec, prefix, arrs, totals=True):
    n, nchr, nphen = sec["n"], sec["nchr"], sec["nphen"]
    add = np.zeros((n, nphen)); dom = np.zeros((n, nphen))
    addc = np.zeros((n, nchr, nphen)); domc = np.zeros((n, nchr, nphen))
    for ih, p, v in sec["A"]:
        add[ih, p], dom[ih, p] = v[0], v[1]
        for c in range(nchr):
            addc[ih, c, p], domc[ih, c, p] = v[3 + 2 * c], v[4 + 2 * c]
    if totals:
        arrs[prefix + "additive"] = add; arrs[prefix + "dominance"] = dom
    arrs[prefix + "add_chr"] = addc; arrs[prefix + "dom_chr"] = domc


def pack_dense(rows, L):
    a = (np.frombuffer("".join(rows).encode(), dtype=np.uint8) - 48).reshape(len(rows), L)
    return np.packbits(a, axis=1, bitorder="little")


# ----------------------------------------------------------------------------- case runner
class Case:
    """Static inputs of one population; knows how to write the reference's input files."""

    def __init__(self, name):
        self.name = name
        self.pops = []        # list of dicts
        self.args_extra = []

    def add_pop(self, **kw):
        self.pops.append(kw)


def run_case(case, seed, dense_gens, out_name=None, hash_only_dense=False, keep_parts_gens=None, vcf=False):
    wd = os.path.join(WORK, case.name)
    shutil.rmtree(wd, ignore_errors=True)
    os.makedirs(wd)
    args = []
    arrs = {"n_pop": np.int64(len(case.pops)), "seed": np.int64(seed)}
    for ip, P in enumerate(case.pops):
        chrs = P["chrs"]                      # list of chr labels
        nchr = len(chrs)
        pre = f"pop{ip}_"
        arrs["nchr"] = np.int64(nchr); arrs["nphen"] = np.int64(len(P["phens"]))
        # founder panels
        with open(os.path.join(wd, f"p{ip}.hapaddr.txt"), "w") as f:
            f.write("chr hap legend sample\n")
            for ic, c in enumerate(chrs):
                base = os.path.join(wd, f"p{ip}.chr{c}")
                write_hap(base + ".hap", P["founders"][ic]); write_legend(base + ".legend", P["snp_pos"][ic])
                write_indv(base + ".indv", P["founders"][ic].shape[0] // 2, prefix=f"p{ip}i")
                f.write(f"{c} {base}.hap {base}.legend {base}.indv\n")
                arrs[f"{pre}chr{ic}_snp_pos"] = np.asarray(P["snp_pos"][ic], dtype=np.uint64)
                if P.get("founders_synth_seed") is not None:     # regenerated by tests/synth.py, not stored
                    arrs[f"{pre}chr{ic}_founders_synth_seed"] = np.uint64(P["founders_synth_seed"])
                else:
                    arrs[f"{pre}chr{ic}_founders"] = np.packbits(P["founders"][ic].astype(np.uint8), axis=1, bitorder="little")
        arrs[f"{pre}n_founder_hap"] = np.int64(P["founders"][0].shape[0])
        # maps
        write_map(os.path.join(wd, f"p{ip}.rmap.txt"), "chr bp cM", [(c, P["rmap_bp"][ic], P["rmap_cM"][ic]) for ic, c in enumerate(chrs)])
        for ic in range(nchr):
            bp = np.asarray(P["rmap_bp"][ic], dtype=np.uint64); cM = np.asarray(P["rmap_cM"][ic], dtype=np.float64)
            prob = np.zeros(len(cM)); prob[1:] = (cM[1:] - cM[:-1]) * .01      # Population.cpp:471-480
            arrs[f"{pre}chr{ic}_rmap_bp"] = bp; arrs[f"{pre}chr{ic}_rmap_prob"] = prob; arrs[f"{pre}chr{ic}_rmap_cM"] = cM   # cM: the file's own column (tests/cli_inputs.py rewrites the file)
            arrs[f"{pre}chr{ic}_label"] = np.int64(chrs[ic])
            arrs[f"{pre}chr{ic}_bp_dist"] = np.uint64(bp[1] - bp[0])           # Population.cpp:396-397
        a = ["--file_gen_info", os.path.join(wd, f"p{ip}.popinfo.txt"), "--file_hap_name", os.path.join(wd, f"p{ip}.hapaddr.txt"),
             "--file_recom_map", os.path.join(wd, f"p{ip}.rmap.txt")]
        write_popinfo(os.path.join(wd, f"p{ip}.popinfo.txt"), P["popinfo"])
        arrs[f"{pre}popinfo"] = np.array(P["popinfo"])         # rows of --file_gen_info: pop_size mat_cor offspring_dist selection_func par1 par2
        arrs[f"{pre}rm"] = np.int64(1 if P.get("RM") else 0)
        arrs[f"{pre}has_mut"] = np.int64(1 if P.get("mut_bp") is not None else 0)
        if P.get("mut_bp") is not None:
            write_map(os.path.join(wd, f"p{ip}.mmap.txt"), "chr bp mutation_rate", [(c, P["mut_bp"][ic], P["mut_rate"][ic]) for ic, c in enumerate(chrs)])
            a += ["--file_mutation_map", os.path.join(wd, f"p{ip}.mmap.txt")]
            for ic in range(nchr):
                r = np.asarray(P["mut_rate"][ic], dtype=np.float64).copy(); r[(r < 0) | (r > 1)] = 0   # Population.cpp:458
                arrs[f"{pre}chr{ic}_mut_bp"] = np.asarray(P["mut_bp"][ic], dtype=np.uint64); arrs[f"{pre}chr{ic}_mut_rate"] = r
        if P.get("RM"):
            a += ["--RM"]
        for iph, ph in enumerate(P["phens"]):
            write_cvinfo(os.path.join(wd, f"p{ip}.ph{iph}.cvinfo.txt"), [(c, ph["bp"][ic], ph["a"][ic], ph["d"][ic]) for ic, c in enumerate(chrs)])
            with open(os.path.join(wd, f"p{ip}.ph{iph}.cvaddr.txt"), "w") as f:
                for ic, c in enumerate(chrs):
                    fn = os.path.join(wd, f"p{ip}.ph{iph}.chr{c}.cv.hap")
                    write_hap(fn, ph["val"][ic]); f.write(f"{c} {fn}\n")
                    arrs[f"{pre}ph{iph}_chr{ic}_cv_bp"] = np.asarray(ph["bp"][ic], dtype=np.uint64)
                    arrs[f"{pre}ph{iph}_chr{ic}_cv_a"] = np.asarray(ph["a"][ic], dtype=np.float64)
                    arrs[f"{pre}ph{iph}_chr{ic}_cv_d"] = np.asarray(ph["d"][ic], dtype=np.float64)
                    if ph.get("val_synth_seed") is not None:
                        arrs[f"{pre}ph{iph}_chr{ic}_cv_val_synth_seed"] = np.uint64(ph["val_synth_seed"])
                    else:
                        arrs[f"{pre}ph{iph}_chr{ic}_cv_val"] = np.packbits(ph["val"][ic].astype(np.uint8), axis=1, bitorder="little")
            a += ["--file_cv_info", os.path.join(wd, f"p{ip}.ph{iph}.cvinfo.txt"), "--file_cvs", os.path.join(wd, f"p{ip}.ph{iph}.cvaddr.txt")]
            arrs[f"{pre}ph{iph}_vd"] = np.float64(ph.get("vd", -1.0))
            arrs[f"{pre}ph{iph}_var"] = np.array([ph.get(k, dflt) for k, dflt in (("va", -1.0), ("vd", -1.0), ("ve", 1.0), ("vf", 0.0))])   # CLI defaults: parameters.cpp
            arrs[f"{pre}ph{iph}_vc"] = np.float64(ph.get("vc", 0.0))
            arrs[f"{pre}ph{iph}_omega"] = np.float64(ph.get("omega", 1.0)); arrs[f"{pre}ph{iph}_lambda"] = np.float64(ph.get("lambda", 1.0))   # coefficients of the mating / selection value (:3311-3320)
        for key in ("va", "vd", "ve", "vf", "vc", "omega", "lambda"):
            for ph in P["phens"]:
                if key in ph:
                    a += [f"--{key}", repr(float(ph[key]))]
        if ip > 0:
            args.append("--next_population")
        args += a
    args += ["--seed", str(seed), "--prefix", os.path.join(wd, "out")] + case.args_extra
    out_gens = getattr(case, "output_generations", None)           # --file_output_generations: genotype output at these generations only (:2059, :137)
    if out_gens:
        with open(os.path.join(wd, "outgens.txt"), "w") as f:
            f.write("".join(f"{g}\n" for g in out_gens))
        args += ["--file_output_generations", os.path.join(wd, "outgens.txt")]
        arrs["output_generations"] = np.array(out_gens, dtype=np.int64)
    arrs["args_extra"] = np.array(case.args_extra if case.args_extra else [""])
    if "--file_migration" in case.args_extra:                      # one row per generation, n_pop^2 row-major entries (:839-896)
        arrs["migration_mat_gen"] = np.loadtxt(case.args_extra[case.args_extra.index("--file_migration") + 1], ndmin=2)
    env = dict(os.environ, GEV_DUMP=os.path.join(wd, "d"), GEV_DUMP_DENSE="1",
               GEV_DENSE_GENS=",".join(str(g) for g in sorted(set(dense_gens) | {0})))
    with open(os.path.join(wd, "log.txt"), "w") as log:
        sh([HARNESS] + args, env=env, stdout=log, stderr=subprocess.STDOUT)
    # validate the harness driver against the stock CLI on this very input
    wd2 = os.path.join(wd, "cli"); os.makedirs(wd2)
    args2 = [x if x != os.path.join(wd, "out") else os.path.join(wd2, "out") for x in args] + ["--out_hap", "--out_interval", "--debug"]
    with open(os.path.join(wd2, "log.txt"), "w") as log:
        sh([os.path.join(ORACLE, "_ref", "GeneEvolve_ref")] + args2, stdout=log, stderr=subprocess.STDOUT)
    ngen = len(case.pops[0]["popinfo"])
    for g in range(ngen + 1):
        for ip in range(len(case.pops)):
            fa = os.path.join(wd, f"out.info.pop{ip+1}.gen{g}.txt"); fb = os.path.join(wd2, f"out.info.pop{ip+1}.gen{g}.txt")
            assert open(fa, "rb").read() == open(fb, "rb").read(), f"harness != CLI at gen {g} pop {ip+1}"
    arrs["n_gen"] = np.int64(ngen)
    for ip in range(len(case.pops)):                # the run's .summary table (Simulation::ras_save_summary, src/Simulation.cpp:782-834), stock CLI
        raw = open(os.path.join(wd2, f"out.pop{ip+1}.summary"), "rb").read()
        arrs[f"summaryfile_pop{ip}_sha"] = np.frombuffer(hashlib.sha256(raw).digest(), dtype=np.uint8)
    # the reference's per-generation text dump (Population::ras_save_human_info, src/Population.cpp:510-568)
    for g in range(ngen + 1):
        for ip in range(len(case.pops)):
            raw = open(os.path.join(wd, f"out.info.pop{ip+1}.gen{g}.txt"), "rb").read()
            arrs[f"infofile_pop{ip}_gen{g}_sha"] = np.frombuffer(hashlib.sha256(raw).digest(), dtype=np.uint8)
    # the reference's own .hap text of the last generation (format_hap::write_hap, src/format_hap.cpp:6-30): hash + head
    for ip in range(len(case.pops)):
        for ic, c in enumerate(case.pops[ip]["chrs"]):
            raw = open(os.path.join(wd2, f"out.pop{ip+1}.gen{ngen}.chr{c}.hap"), "rb").read()
            arrs[f"hapfile_pop{ip}_chr{ic}_sha"] = np.frombuffer(hashlib.sha256(raw).digest(), dtype=np.uint8)
            arrs[f"hapfile_pop{ip}_chr{ic}_size"] = np.int64(len(raw))
            arrs[f"hapfile_pop{ip}_chr{ic}_head"] = np.frombuffer(raw[:4096], dtype=np.uint8)
            raw = open(os.path.join(wd2, f"out.pop{ip+1}.gen{ngen}.chr{c}.cvval"), "rb").read()    # --debug dump of ras_find_cv's CV genotypes at the last generation
            arrs[f"cvvalfile_pop{ip}_chr{ic}_sha"] = np.frombuffer(hashlib.sha256(raw).digest(), dtype=np.uint8)   # (:2665-2683; every phenotype writes the same file name: the last one stays)
            raw = open(os.path.join(wd2, f"out.pop{ip+1}.gen{ngen}.chr{c}.int"), "rb").read()      # --out_interval (:1582-1633)
            arrs[f"intfile_pop{ip}_chr{ic}_sha"] = np.frombuffer(hashlib.sha256(raw).digest(), dtype=np.uint8)
            for g in (out_gens or []):               # genotype files written in the middle of the run
                if g != ngen:
                    for ext in ("hap", "int"):
                        raw = open(os.path.join(wd2, f"out.pop{ip+1}.gen{g}.chr{c}.{ext}"), "rb").read()
                        arrs[f"{ext}file_g{g}_pop{ip}_chr{ic}_sha"] = np.frombuffer(hashlib.sha256(raw).digest(), dtype=np.uint8)
    # the reference's PLINK text of the last generation (format_plink::write_ped_map / write_ped01_map, src/format_plink.cpp):
    # both flags write the same <prefix>.ped, so one CLI run each.  Stored: hash of the whole file, hash of the genotype
    # columns alone (each line after its six id columns = what gev_format_ped_text produces) and the id columns.
    for tag, flag in (("ped", "--out_plink"), ("ped01", "--out_plink01")):
        wd3 = os.path.join(wd, "cli_" + tag); os.makedirs(wd3)
        args3 = [x if x != os.path.join(wd, "out") else os.path.join(wd3, "out") for x in args] + [flag]
        with open(os.path.join(wd3, "log.txt"), "w") as log:
            sh([os.path.join(ORACLE, "_ref", "GeneEvolve_ref")] + args3, stdout=log, stderr=subprocess.STDOUT)
        for ip in range(len(case.pops)):
            for ic, c in enumerate(case.pops[ip]["chrs"]):
                raw = open(os.path.join(wd3, f"out.pop{ip+1}.gen{ngen}.chr{c}.ped"), "rb").read()
                geno, ids = hashlib.sha256(), []
                for line in raw.split(b"\n")[:-1]:
                    tok = line.split(b" ", 6)
                    ids.append([int(t) for t in tok[:6]])
                    geno.update(b" " + tok[6] + b"\n")
                k = f"{tag}file_pop{ip}_chr{ic}_"
                arrs[k + "sha"] = np.frombuffer(hashlib.sha256(raw).digest(), dtype=np.uint8)
                arrs[k + "size"] = np.int64(len(raw))
                arrs[k + "geno_sha"] = np.frombuffer(geno.digest(), dtype=np.uint8)
                arrs[k + "ids"] = np.array(ids, dtype=np.int64)
                if tag == "ped":
                    al0, al1 = legend_alleles(len(case.pops[ip]["snp_pos"][ic]))
                    arrs[f"pop{ip}_chr{ic}_al0"] = al0; arrs[f"pop{ip}_chr{ic}_al1"] = al1

    # the reference's VCF output (format_vcf::write_vcf_file, src/format_vcf.cpp:44-61): it only exists for a VCF reference
    # panel, so the same founders are also written as phased VCF and the stock CLI is run on them with --out_vcf; the
    # run is the same simulation (identical .info files are asserted).  Stored: hash of the GT columns of every data line
    # ("\ta|b" per individual + newline = what gev_format_vcf_gt produces), hash of the whole data section.
    # Built with its own flags (-O3) by g++ 11 the reference crashes on every VCF panel: format_vcf::read_vcf_header_sample
    # (src/format_vcf.cpp:367-389) has no return statement.  oracle/_ref/GeneEvolve_ref_vcf is the same program with that one
    # return supplied (oracle/ref_vcf_return.cpp); it is used here and nowhere else.
    if vcf:
        wd4 = os.path.join(wd, "cli_vcf"); os.makedirs(wd4)
        # vcf == "full": the VCF-panel siblings of the output paths as well (ras_write_vcf_to_hap_legend_sample, ras_write_vcf_to_plink_format)
        args4 = [x if x != os.path.join(wd, "out") else os.path.join(wd4, "out") for x in args] + ["--out_vcf"] + (["--out_hap", "--out_plink", "--out_interval"] if vcf == "full" else [])
        for ip, P in enumerate(case.pops):
            addr = os.path.join(wd4, f"p{ip}.vcfaddr.txt")
            with open(addr, "w") as f:
                f.write("chr vcf\n")
                for ic, c in enumerate(P["chrs"]):
                    fn = os.path.join(wd4, f"p{ip}.chr{c}.vcf")
                    write_vcf(fn, c, P["founders"][ic], P["snp_pos"][ic], f"p{ip}i")
                    f.write(f"{c} {fn}\n")
            k = args4.index(os.path.join(wd, f"p{ip}.hapaddr.txt"))
            args4[k - 1] = "--file_ref_vcf"; args4[k] = addr
        with open(os.path.join(wd4, "log.txt"), "w") as log:
            sh([os.path.join(ORACLE, "_ref", "GeneEvolve_ref_vcf")] + args4, stdout=log, stderr=subprocess.STDOUT)
        for g in range(ngen + 1):
            for ip in range(len(case.pops)):
                fa = os.path.join(wd, f"out.info.pop{ip+1}.gen{g}.txt"); fb = os.path.join(wd4, f"out.info.pop{ip+1}.gen{g}.txt")
                assert open(fa, "rb").read() == open(fb, "rb").read(), f"VCF-reference run != hap-reference run at gen {g} pop {ip+1}"
        for ip in range(len(case.pops)):
            for ic, c in enumerate(case.pops[ip]["chrs"]):
                raw = open(os.path.join(wd4, f"out.pop{ip+1}.gen{ngen}.chr{c}.vcf"), "rb").read()
                gt, body, samples = hashlib.sha256(), hashlib.sha256(), None
                for line in raw.split(b"\n")[:-1]:
                    if line.startswith(b"##"):
                        continue
                    if line.startswith(b"#"):
                        samples = line.split(b"\t")[9:]
                        continue
                    tok = line.split(b"\t", 9)
                    gt.update(b"\t" + tok[9] + b"\n"); body.update(line + b"\n")
                if vcf == "full":
                    for ext in ("hap", "ped", "map", "int"):
                        rawx = open(os.path.join(wd4, f"out.pop{ip+1}.gen{ngen}.chr{c}.{ext}"), "rb").read()
                        arrs[f"vcfrun_{ext}file_pop{ip}_chr{ic}_sha"] = np.frombuffer(hashlib.sha256(rawx).digest(), dtype=np.uint8)
                k = f"vcffile_pop{ip}_chr{ic}_"
                arrs[k + "gt_sha"] = np.frombuffer(gt.digest(), dtype=np.uint8)
                arrs[k + "body_sha"] = np.frombuffer(body.digest(), dtype=np.uint8)
                arrs[k + "n_samples"] = np.int64(len(samples))
                arrs[k + "first_sample"] = np.frombuffer(samples[0], dtype=np.uint8)

    # gen 0
    d0 = parse_dump(os.path.join(wd, "d.gen0.txt"))
    arrs["globseq"] = np.array(d0["globseq"], dtype=np.uint32)
    for ip in range(len(case.pops)):
        pack_humans(d0["post"][ip], f"g0_pop{ip}_", arrs)
        pack_ad(d0["ad"][ip], f"g0_pop{ip}_", arrs, totals=False)
    for (ip, ic), rows in d0["dense"].items():
        L = len(case.pops[ip]["snp_pos"][ic])
        store_dense(arrs, f"g0_pop{ip}_chr{ic}_dense", pack_dense(rows, L), hash_only_dense)
    # generations
    for g in range(1, ngen + 1):
        d = parse_dump(os.path.join(wd, f"d.gen{g}.txt"))
        for ip in range(len(case.pops)):
            pre = f"g{g}_pop{ip}_"
            arrs[pre + "couples"] = np.array(d["couples"][ip], dtype=np.int64).reshape(-1, 4)
            arrs[pre + "seed_reproduce"] = np.uint32(d["seed"][ip])
            arrs[pre + "mut_seeds"] = np.array(d["mutseeds"][ip], dtype=np.uint32)
            tmp = {}
            pack_humans(d["offspring"][ip], pre, tmp)
            if keep_parts_gens is not None and g not in keep_parts_gens:   # keep only hashes of the big lists
                for k in list(tmp):
                    if "_part" in k or "_mut" in k:
                        tmp[k + "_sha"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(tmp.pop(k)).tobytes()).digest(), dtype=np.uint8)
            arrs.update(tmp)
            pack_ad(d["ad"][ip], pre, arrs)
            if d["premig"]:
                pack_humans(d["postmig"][ip], pre + "postmig_", arrs)
        for ma in d.get("mate", []):      # mating inputs
            k = f"g{g}_pop{ma['pop']}_mate_"
            arrs[k + "rm"] = np.int64(ma["rm"]); arrs[k + "seed"] = np.uint32(ma["seed"]); arrs[k + "popsize"] = np.int64(ma["popsize"])
            arrs[k + "sex"] = np.array([m[0] for m in ma["MS"]], dtype=np.uint8); arrs[k + "svf"] = np.array([m[1] for m in ma["MS"]])
            if "am" in ma:                # assort_mate: parameters, the 4 glob seeds in draw order, mating values, pedigree ids
                am = ma["am"]
                arrs[k + "am_par"] = np.array([am["matcor"], am["mm"], float(am["avoid"])])
                arrs[k + "am_dist"] = np.frombuffer(am["dist"].encode(), dtype=np.uint8)
                arrs[k + "am_seeds"] = np.array([ma["seed"]] + am["seeds"], dtype=np.uint32)
                arrs[k + "am_mv"] = np.array([m[0] for m in am["MV"]])
                arrs[k + "am_ped"] = np.array([m[1:] for m in am["MV"]], dtype=np.int64)      # ID_Father, FF, FM, MF, MM
        for ge in d.get("gef", []):       # ras_scale_AD_compute_GEF inputs / outputs
            k = f"g{g}_pop{ge['pop']}_ph{ge['phen']}_gef_"
            arrs[k + "seed"] = np.uint32(ge["seed"]); arrs[k + "par"] = np.array(ge["par"]); arrs[k + "vt"] = np.int64(ge["vt"])
            arrs[k + "in"] = np.array(ge["GI"]).reshape(-1, 3); arrs[k + "out"] = np.array(ge["GO"]).reshape(-1, 6)
        if d["premig"]:
            # WHO moved (the host's decision in ras_do_migration, src/Simulation.cpp:899-937), recovered by
            # matching (ID, phenotype) fingerprints; listed in the reference's append order (:971-981):
            # destination blocks are contiguous tails of the post-migration vectors.
            npop = len(case.pops)
            where = {}
            for ip in range(npop):
                for pos, fp in enumerate(d["premig"][ip]):
                    assert (ip, fp) not in where
                    where[(ip, fp)] = pos
            tails = {}
            for j in range(npop):
                t = []
                for pos, fp in enumerate(d["postfp"][j]):
                    src = [i for i in range(npop) if i != j and (i, fp) in where]
                    own = (j, fp) in where
                    assert own != bool(src) and len(src) <= 1, "fingerprint collision"
                    if src:
                        t.append((src[0], where[(src[0], fp)]))
                    else:
                        assert not t, "stayers must precede migrants"
                tails[j] = t
            moves = []
            for i in range(npop):
                for j in range(npop):
                    if i != j:
                        moves += [(i, pos, j) for (si, pos) in tails[j] if si == i]
            arrs[f"g{g}_moves"] = np.array(moves, dtype=np.int64).reshape(-1, 3)
        if g in dense_gens:
            for (ip, ic), rows in d["dense"].items():
                L = len(case.pops[ip]["snp_pos"][ic])
                store_dense(arrs, f"g{g}_pop{ip}_chr{ic}_dense", pack_dense(rows, L), hash_only_dense)
    out = os.path.join(HERE, (out_name or case.name) + ".npz")
    np.savez_compressed(out, **arrs)
    print(f"wrote {out}: {os.path.getsize(out)/1e6:.2f} MB")


def store_dense(arrs, key, packed, hash_only):
    if hash_only:
        arrs[key + "_sha"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(packed).tobytes()).digest(), dtype=np.uint8)
        arrs[key + "_head"] = packed[:8].copy()
    else:
        arrs[key] = packed


# ----------------------------------------------------------------------------- the cases
def read_hap_text(path, nhap_keep):
    rows = []
    with open(path) as f:
        for line in f:
            rows.append([line[2 * i] == "1" for i in range(nhap_keep)])
    return np.array(rows, dtype=np.uint8).T.copy()        # [hap][snp]


def example1_inputs(n_founder_ind=300):
    ex = os.path.join(WORK, "Examples")
    if not os.path.isdir(ex):
        with zipfile.ZipFile(os.path.join(REF, "Examples.zip")) as z:
            z.extractall(WORK)
    chrs = [1, 2, 3]
    nh = 2 * n_founder_ind
    founders, snp_pos = [], []
    for c in chrs:
        founders.append(read_hap_text(os.path.join(ex, f"ref.chr{c}.hap"), nh))
        snp_pos.append(np.loadtxt(os.path.join(ex, f"ref.chr{c}.legend"), skiprows=1, usecols=1, dtype=np.float64).astype(np.uint64))
    rm = np.loadtxt(os.path.join(ex, "Recom.Map.b37.50KbDiff"), skiprows=1)
    rmap_bp = [rm[rm[:, 0] == c, 1].astype(np.uint64) for c in chrs]
    rmap_cM = [rm[rm[:, 0] == c, 2] for c in chrs]
    cvi = np.loadtxt(os.path.join(ex, "cv.info"), skiprows=1)
    ph = {"bp": [cvi[cvi[:, 0] == c, 1].astype(np.uint64) for c in chrs], "a": [cvi[cvi[:, 0] == c, 2] for c in chrs],
          "d": [cvi[cvi[:, 0] == c, 3] for c in chrs], "val": [read_hap_text(os.path.join(ex, f"cv.chr{c}.hap"), nh) for c in chrs]}
    return dict(chrs=chrs, founders=founders, snp_pos=snp_pos, rmap_bp=rmap_bp, rmap_cM=rmap_cM, phens=[ph])


def genetic_map_fixture():
    """Recom_Map.zip:Recom.Map.b37.50KbDiff (a DATA file of the reference: 22 autosomes, 55 657 rows at 50 kb) as arrays --
    the genetic map BASELINE config 4 names."""
    d = os.path.join(WORK, "rmap")
    with zipfile.ZipFile(os.path.join(REF, "Recom_Map.zip")) as z:
        z.extract("Recom.Map.b37.50KbDiff", d)
    m = np.loadtxt(os.path.join(d, "Recom.Map.b37.50KbDiff"), skiprows=1)
    out = os.path.join(HERE, "recom_map_b37_50kb.npz")
    np.savez_compressed(out, chr=m[:, 0].astype(np.uint8), bp=m[:, 1].astype(np.uint64), cM=m[:, 2])
    print(f"wrote {out}: {os.path.getsize(out)/1e6:.2f} MB")


def extra_cases(which):
    """Cases added after round 1; each has its own RandomState so that it can be regenerated alone:
        python tests/golden/make_golden.py mig3c c4mini
      mig3c   two populations x THREE chromosomes, mutation map, migration every generation, parental effect (vf > 0): the
              fixture of the locus-split + migration test (2 populations x 2 chromosome shards = 4 ranks)
      c4mini  BASELINE config 4 in miniature: two populations x the 22 autosomes of Recom.Map.b37.50KbDiff (the reference's
              own map file), assortative mating (mat_cor 0.4, Poisson family sizes), mutation map, migration
      vcf1    VCF reference panels, two populations with migration and mutation: .vcf, .hap, .ped/.map and .int files of the
              VCF-panel run (GeneEvolve_ref_vcf)
      om1     --omega / --lambda on two phenotypes under assortative mating, --file_output_generations (genotype files of generation 2
              and of the last one)
      gam2    --gamma: environmental effects specific to each population, two populations with migration and logit selection
      vt2     --vt_type 2: the parental effect of a child is beta * (the parents' PARENTAL EFFECTS, not their phenotypes)
              (src/Simulation.cpp:3128-3131), beta adjusted on var(F) after generation 0 (:653-657); two phenotypes (vf > 0 and
              vf = 0), random mating with a logit selection function, mutation map"""
    if "om1" in which:
        # --omega / --lambda (the coefficients of the phenotypes in the mating and the selection value, :3311-3320) on two phenotypes
        # under assortative mating with a logit selection function, and --file_output_generations: genotype files written in the
        # MIDDLE of the run (generation 2) as well as at its end
        rs = np.random.RandomState(4242)
        R = 101
        rbp = (1000 + 1000 * np.arange(R)).astype(np.uint64)
        rcM = np.cumsum(np.r_[0.0, np.full(R - 1, 1.0)])
        snp = np.arange(1500, 100000, 400).astype(np.uint64)
        nf = 240
        founders = (rs.rand(nf, len(snp)) < 0.3).astype(np.uint8)
        cvbp = np.sort(rs.choice(np.arange(1100, 100000, 50), size=80, replace=False)).astype(np.uint64)
        ph_a = {"bp": [cvbp], "a": [rs.randn(80)], "d": [np.zeros(80)], "val": [(rs.rand(nf, 80) < 0.4).astype(np.uint8)], "va": 0.6, "ve": 0.4, "omega": 0.7, "lambda": 0.25}
        ph_b = {"bp": [cvbp], "a": [rs.randn(80)], "d": [np.zeros(80)], "val": [(rs.rand(nf, 80) < 0.4).astype(np.uint8)], "va": 0.5, "ve": 0.5, "omega": -0.3, "lambda": 1.5}
        c = Case("om1")
        c.add_pop(chrs=[1], founders=[founders], snp_pos=[snp], rmap_bp=[rbp], rmap_cM=[rcM], phens=[ph_a, ph_b],
                  mut_bp=[rbp], mut_rate=[np.full(R, 0.01)], popinfo=["130 0.4 p logit 1 1", "125 0.3 f logit 0.5 1", "128 0.5 p logit 1 1", "120 0.2 p thr 1 1"])
        c.output_generations = [2, 4]
        run_case(c, 90210, dense_gens={4})
    if "gam2" in which:
        # mig2's shape with --gamma: environmental effects specific to each population (src/Simulation.cpp:3345-3382: Newton-Raphson
        # on the combined variance of all populations' phenotype values, then -a / +a added per population), logit selection so that
        # the shifted phenotypes decide who marries
        rs = np.random.RandomState(808)
        R = 201
        rbp = (1000 + 100 * np.arange(R)).astype(np.uint64)
        rcM = np.cumsum(np.r_[0.0, np.full(R - 1, 0.5)])
        snp = np.arange(950, 21100, 17).astype(np.uint64)
        cvbp = np.sort(rs.choice(np.arange(1000, 21000), size=60, replace=False)).astype(np.uint64)
        c = Case("gam2")
        for ip in range(2):
            f2 = (rs.rand(160, len(snp)) < rs.uniform(0.05, 0.5, len(snp))).astype(np.uint8)
            ph = {"bp": [cvbp], "a": [rs.randn(60)], "d": [rs.randn(60) * 0.2], "val": [(rs.rand(160, 60) < 0.4).astype(np.uint8)], "va": 0.5, "ve": 0.5}
            c.add_pop(chrs=[1], founders=[f2], snp_pos=[snp], rmap_bp=[rbp], rmap_cM=[rcM], phens=[ph], RM=True,
                      mut_bp=[rbp], mut_rate=[np.full(R, 0.01)], popinfo=["100 0 p logit 1 1", "110 0 p logit 0.5 1", "100 0 p thr 1 1", "105 0 p logit 1 1"])
        with open(os.path.join(WORK, "mig_gam2.txt"), "w") as f:
            for g in range(4):
                f.write("0.9 0.1 0.15 0.85\n")
        c.args_extra = ["--file_migration", os.path.join(WORK, "mig_gam2.txt"), "--gamma", "0.35"]
        run_case(c, 6021, dense_gens={4})
    if "mig3c" in which:
        rs = np.random.RandomState(555)
        R = 151
        rbp = (1000 + 100 * np.arange(R)).astype(np.uint64)
        rcM = np.cumsum(np.r_[0.0, np.full(R - 1, 0.8)])
        snp = [np.arange(900 + 3 * k, 16200, 11).astype(np.uint64) for k in range(3)]
        cvbp = [np.sort(rs.choice(np.arange(1000, 16000), size=40, replace=False)).astype(np.uint64) for _ in range(3)]
        c = Case("mig3c")
        for ip, n0 in enumerate((100, 90)):
            nf = 2 * max(n0, 100)
            founders = [(rs.rand(nf, len(snp[k])) < rs.uniform(0.05, 0.5, len(snp[k]))).astype(np.uint8) for k in range(3)]
            ph = {"bp": cvbp, "a": [rs.randn(40) for _ in range(3)], "d": [rs.randn(40) * 0.2 for _ in range(3)],
                  "val": [(rs.rand(nf, 40) < 0.4).astype(np.uint8) for _ in range(3)], "va": 0.5, "vd": 0.1, "ve": 0.3, "vf": 0.1}
            c.add_pop(chrs=[1, 2, 3], founders=founders, snp_pos=snp, rmap_bp=[rbp] * 3, rmap_cM=[rcM] * 3, phens=[ph], RM=True,
                      mut_bp=[rbp] * 3, mut_rate=[np.full(R, 0.01)] * 3, popinfo=[f"{n0} 0 p thr 1 1"] * 4)
        with open(os.path.join(WORK, "mig3c.txt"), "w") as f:
            for g in range(4):
                f.write("0.9 0.1 0.15 0.85\n")
        c.args_extra = ["--file_migration", os.path.join(WORK, "mig3c.txt")]
        run_case(c, 60606, dense_gens={2, 4})
    if "vt2" in which:
        rs = np.random.RandomState(222)
        R = 121
        rbp = (1000 + 500 * np.arange(R)).astype(np.uint64)
        rcM = np.cumsum(np.r_[0.0, np.full(R - 1, 0.9)])
        snp = np.arange(1200, 61000, 130).astype(np.uint64)
        nf = 260
        founders = (rs.rand(nf, len(snp)) < rs.uniform(0.05, 0.5, len(snp))).astype(np.uint8)
        cvbp = np.sort(rs.choice(np.arange(1100, 60900, 20), size=70, replace=False)).astype(np.uint64)
        phens = [{"bp": [cvbp], "a": [rs.randn(70)], "d": [rs.randn(70) * 0.2], "val": [(rs.rand(nf, 70) < 0.35).astype(np.uint8)],
                  "va": 0.45, "vd": 0.05, "ve": 0.3, "vf": 0.2},
                 {"bp": [cvbp], "a": [rs.randn(70)], "d": [np.zeros(70)], "val": [(rs.rand(nf, 70) < 0.35).astype(np.uint8)],
                  "va": 0.6, "vd": 0.0, "ve": 0.4, "vf": 0.0}]
        c = Case("vt2")
        c.add_pop(chrs=[1], founders=[founders], snp_pos=[snp], rmap_bp=[rbp], rmap_cM=[rcM], phens=phens, RM=True,
                  mut_bp=[rbp], mut_rate=[np.full(R, 0.01)], popinfo=["130 0 p logit 1 1", "125 0 p thr 1 1", "120 0 p logit 0.5 1", "126 0 p logit 1 1", "118 0 p thr 1 1"])
        c.args_extra = ["--vt_type", "2"]
        run_case(c, 20202, dense_gens={5})
    if "vcf1" in which:
        # VCF reference panels (--file_ref_vcf) for two populations that exchange migrants: every output format the VCF-panel run
        # can write (--out_vcf --out_hap --out_plink --out_interval), from oracle/_ref/GeneEvolve_ref_vcf (the reference with the one
        # missing `return` of format_vcf::read_vcf_header_sample supplied, oracle/ref_vcf_return.cpp)
        rs = np.random.RandomState(808)
        R = 161
        rbp = (1000 + 100 * np.arange(R)).astype(np.uint64)
        rcM = np.cumsum(np.r_[0.0, np.full(R - 1, 0.9)])
        snp = np.arange(950, 17100, 9).astype(np.uint64)                       # some SNPs outside [bp0, bpEnd)
        cvbp = np.sort(rs.choice(np.arange(1000, 17000), size=50, replace=False)).astype(np.uint64)
        cvbp[:15] = np.sort(rs.choice(snp, 15, replace=False))
        cvbp = np.sort(cvbp)
        c = Case("vcf1")
        for ip, n0 in enumerate((90, 80)):
            nf = 200
            f2 = (rs.rand(nf, len(snp)) < rs.uniform(0.05, 0.5, len(snp))).astype(np.uint8)
            ph = {"bp": [cvbp], "a": [rs.randn(50)], "d": [rs.randn(50) * 0.2], "val": [(rs.rand(nf, 50) < 0.4).astype(np.uint8)], "va": 0.5, "vd": 0.1, "ve": 0.4}
            c.add_pop(chrs=[1], founders=[f2], snp_pos=[snp], rmap_bp=[rbp], rmap_cM=[rcM], phens=[ph], RM=True,
                      mut_bp=[rbp], mut_rate=[np.full(R, 0.02)], popinfo=[f"{n0} 0 p thr 1 1"] * 4)
        with open(os.path.join(WORK, "vcf1.txt"), "w") as f:
            for g in range(4):
                f.write("0.85 0.15 0.2 0.8\n")
        c.args_extra = ["--file_migration", os.path.join(WORK, "vcf1.txt")]
        run_case(c, 31415, dense_gens={4}, vcf="full")
    if "c4mini" in which:
        rs = np.random.RandomState(444)
        d = os.path.join(WORK, "rmap")
        with zipfile.ZipFile(os.path.join(REF, "Recom_Map.zip")) as z:
            z.extract("Recom.Map.b37.50KbDiff", d)
        m = np.loadtxt(os.path.join(d, "Recom.Map.b37.50KbDiff"), skiprows=1)
        chrs = list(range(1, 23))
        rbp = [m[m[:, 0] == k, 1].astype(np.uint64) for k in chrs]
        rcM = [m[m[:, 0] == k, 2] for k in chrs]
        L, C = 300, 15
        snp = [np.sort(rs.randint(int(b[0]), int(b[-1]), L)).astype(np.uint64) for b in rbp]
        cvbp = [np.sort(rs.randint(int(b[0]), int(b[-1]), C)).astype(np.uint64) for b in rbp]
        c = Case("c4mini")
        for ip, n0 in enumerate((70, 60)):
            nf = 2 * 70
            founders = [(rs.rand(nf, L) < rs.uniform(0.05, 0.5, L)).astype(np.uint8) for _ in chrs]
            ph = {"bp": cvbp, "a": [rs.randn(C) for _ in chrs], "d": [np.zeros(C) for _ in chrs],
                  "val": [(rs.rand(nf, C) < 0.4).astype(np.uint8) for _ in chrs], "va": 0.6, "ve": 0.4}
            c.add_pop(chrs=chrs, founders=founders, snp_pos=snp, rmap_bp=rbp, rmap_cM=rcM, phens=[ph],
                      mut_bp=rbp, mut_rate=[np.r_[0.0, np.full(len(b) - 1, 2e-4)] for b in rbp],
                      popinfo=[f"{n0} 0.4 p logit 1 1"] * 3)
        with open(os.path.join(WORK, "c4mini.txt"), "w") as f:
            for g in range(3):
                f.write("0.9 0.1 0.1 0.9\n")
        c.args_extra = ["--file_migration", os.path.join(WORK, "c4mini.txt")]
        run_case(c, 40404, dense_gens={3})


def main():
    os.makedirs(WORK, exist_ok=True)
    if len(sys.argv) > 1:                      # only the named later cases (the round-1 cases below share one RandomState and are made together)
        sh(["make", "-f", "Makefile.ref", "-j8"], cwd=ORACLE, stdout=subprocess.DEVNULL)
        extra_cases(set(sys.argv[1:]))
        return
    genetic_map_fixture()
    sh(["make", "-f", "Makefile.ref", "-j8"], cwd=ORACLE, stdout=subprocess.DEVNULL)
    # ---- KAT
    kat = os.path.join(WORK, "kat.txt")
    sh([HARNESS, "KAT", kat])
    with open(kat, "rb") as f, gzip.GzipFile(os.path.join(HERE, "kat.txt.gz"), "wb", mtime=0) as g:
        g.write(f.read())
    print("wrote kat.txt.gz")

    # ---- ex1sub: literal Example1 inputs (subset of founders), assortative, no mutation
    base = example1_inputs(300)
    c = Case("ex1sub")
    c.add_pop(popinfo=["300 0 p thr 1 1"] * 4, **base)
    run_case(c, 12345, dense_gens={1, 4})

    # ---- ex1full: Example1.sh at its own size -- all 2000 founders, 3 chromosomes x 1000 SNPs, 3000 individuals x 10 generations,
    #      assortative mating with mat_cor 0, no mutation map (serial rand() chain).  Hashes of the big lists only.
    full = example1_inputs(2000)
    c = Case("ex1full")
    c.add_pop(popinfo=["3000 0 p thr 1 1"] * 10, **full)
    run_case(c, 12345, dense_gens={1, 10}, hash_only_dense=True, keep_parts_gens={1})

    # ---- ex1mut: same inputs, random mating + hot mutation map
    c = Case("ex1mut")
    mut_bp = base["rmap_bp"]; mut_rate = [np.full(len(b), 2e-3) for b in mut_bp]
    mut_rate[1][5] = 1.5; mut_rate[1][6] = -0.5          # out-of-range rates are zeroed by the reader
    c.add_pop(popinfo=["300 0 p thr 1 1"] * 4, RM=True, mut_bp=mut_bp, mut_rate=mut_rate, **base)
    run_case(c, 777, dense_gens={1, 4})

    # ---- dense: mutations land on SNPs and CVs; 2 phenotypes; unsorted CV order; vd > 0
    rs = np.random.RandomState(2024)
    R = 201
    rbp = (1000 + 100 * np.arange(R)).astype(np.uint64)
    rcM = np.cumsum(np.r_[0.0, np.full(R - 1, 1.0)])                    # 1 cM/row -> prob 0.01
    snp = np.arange(900, 21200, 7).astype(np.uint64)                    # some SNPs outside [bp0, bpEnd)
    nf = 200
    founders = (rs.rand(nf, len(snp)) < rs.uniform(0.05, 0.5, len(snp))).astype(np.uint8)
    phens = []
    for k in range(2):
        cvbp = rs.choice(np.arange(950, 21100), size=150, replace=False).astype(np.uint64)   # file order unsorted
        cvbp[:40] = rs.choice(snp, 40, replace=False)
        phens.append({"bp": [cvbp], "a": [rs.randn(150)], "d": [rs.randn(150) * 0.3],
                      "val": [(rs.rand(nf, 150) < 0.3).astype(np.uint8)], "vd": 0.2 if k == 0 else 0.0, "va": 0.5, "ve": 0.3, "vf": 0.15 if k == 0 else 0.0})
    c = Case("dense")
    c.add_pop(chrs=[1], founders=[founders], snp_pos=[snp], rmap_bp=[rbp], rmap_cM=[rcM], phens=phens, RM=True,
              mut_bp=[rbp], mut_rate=[np.full(R, 0.02)], popinfo=["120 0 p thr 1 1"] * 6)
    run_case(c, 4242, dense_gens={1, 2, 3, 4, 5, 6}, vcf=True)

    # ---- mig2: two populations + migration
    c = Case("mig2")
    cvbp = np.sort(rs.choice(np.arange(1000, 21000), size=60, replace=False)).astype(np.uint64)   # shared CV grid
    for ip in range(2):
        f2 = (rs.rand(160, len(snp)) < rs.uniform(0.05, 0.5, len(snp))).astype(np.uint8)
        ph = {"bp": [cvbp], "a": [rs.randn(60)], "d": [rs.randn(60) * 0.2], "val": [(rs.rand(160, 60) < 0.4).astype(np.uint8)]}
        c.add_pop(chrs=[1], founders=[f2], snp_pos=[snp], rmap_bp=[rbp], rmap_cM=[rcM], phens=[ph], RM=True,
                  mut_bp=[rbp], mut_rate=[np.full(R, 0.01)], popinfo=["100 0 p thr 1 1"] * 4)
    with open(os.path.join(WORK, "mig.txt"), "w") as f:
        for g in range(4):
            f.write("0.9 0.1 0.2 0.8\n")
    c.args_extra = ["--file_migration", os.path.join(WORK, "mig.txt")]
    run_case(c, 9001, dense_gens={1, 2, 3, 4})

    # ---- am1: assortative mating variants: mat_cor != 0, two spouses (--MM), inbreeding avoidance,
    #      Poisson and fixed offspring numbers, logit selection; small single chromosome
    rs = np.random.RandomState(77)
    R = 101
    rbp = (1000 + 1000 * np.arange(R)).astype(np.uint64)
    rcM = np.cumsum(np.r_[0.0, np.full(R - 1, 1.0)])
    snp = np.arange(1500, 100000, 400).astype(np.uint64)
    nf = 240
    founders = (rs.rand(nf, len(snp)) < 0.3).astype(np.uint8)
    cvbp = np.sort(rs.choice(np.arange(1100, 100000, 50), size=80, replace=False)).astype(np.uint64)
    ph = {"bp": [cvbp], "a": [rs.randn(80)], "d": [np.zeros(80)], "val": [(rs.rand(nf, 80) < 0.4).astype(np.uint8)], "va": 0.6, "ve": 0.4}
    c = Case("am1")
    c.add_pop(chrs=[1], founders=[founders], snp_pos=[snp], rmap_bp=[rbp], rmap_cM=[rcM], phens=[ph],
              popinfo=["130 0.4 p logit 1 1", "140 0.3 p logit 1 1", "120 -0.5 p logit 0.5 1", "125 0.8 p thr 1 1", "110 0.2 p logit 1 1"])
    c.args_extra = ["--MM", "0.15", "--avoid_inbreeding"]
    run_case(c, 31337, dense_gens={5})
    c = Case("am2")
    c.add_pop(chrs=[1], founders=[founders], snp_pos=[snp], rmap_bp=[rbp], rmap_cM=[rcM], phens=[ph],
              popinfo=["130 0.4 f logit 1 1", "141 0.3 f logit 1 1", "120 0 f logit 0.5 1", "125 1 f thr 1 1"])
    run_case(c, 4711, dense_gens={4})

    # ---- vc1: common (sibling) environment, two phenotypes sharing one normal stream, families of several children
    ph_a = dict(ph); ph_a.update(vc=0.15)
    ph_b = {"bp": [cvbp], "a": [rs.randn(80)], "d": [np.zeros(80)], "val": [(rs.rand(nf, 80) < 0.4).astype(np.uint8)], "va": 0.5, "ve": 0.3, "vc": 0.2}
    c = Case("vc1")
    c.add_pop(chrs=[1], founders=[founders], snp_pos=[snp], rmap_bp=[rbp], rmap_cM=[rcM], phens=[ph_a, ph_b],
              popinfo=["131 0.3 p logit 1 1", "120 0.2 f logit 1 1", "125 0.4 p thr 1 1"])
    run_case(c, 1618, dense_gens={3})

    # ---- sel1: probit and stabilising selection (CommFunc::NormalCDF / NormalPDF), random mating, small
    c = Case("sel1")
    c.add_pop(chrs=[1], founders=[founders], snp_pos=[snp], rmap_bp=[rbp], rmap_cM=[rcM], phens=[ph], RM=True,
              popinfo=["130 0 p probit 0 1", "125 0 p stab 0 1", "120 0 p probit -0.5 0.8", "128 0 p stab 0.3 1.5"])
    run_case(c, 2718, dense_gens={4})

    # ---- syn1k: config-1 shape, deterministic synthetic founders (tests/synth.py), hashes only
    L, N0 = 10000, 1000
    snp = (1000 + 10000 * np.arange(L)).astype(np.uint64)
    R = 2001
    rbp = (1000 + 50000 * np.arange(R)).astype(np.uint64)
    rcM = 1e-6 * (rbp - rbp[0]).astype(np.float64)                         # 1 cM/Mb -> 5e-4 per 50 kb row
    founders = synth_bits(12345, 2 * N0, L)
    cvbp = np.sort(rs.choice(np.arange(1000, 100001000, 100), size=1000, replace=False)).astype(np.uint64)
    ph = {"bp": [cvbp], "a": [rs.randn(1000)], "d": [np.zeros(1000)], "val": [synth_bits(999, 2 * N0, 1000)], "val_synth_seed": 999,
          "va": 0.5, "vd": 0.0, "ve": 0.5}
    c = Case("syn1k")
    c.add_pop(chrs=[1], founders=[founders], founders_synth_seed=12345, snp_pos=[snp], rmap_bp=[rbp], rmap_cM=[rcM], phens=[ph], RM=True,
              mut_bp=[rbp], mut_rate=[np.r_[0.0, np.full(R - 1, 5e-4)]], popinfo=["1000 0 p logit 0 1"] * 10)
    run_case(c, 12345, dense_gens={1, 5, 10}, hash_only_dense=True, keep_parts_gens={1, 10})
    extra_cases({"mig3c", "c4mini"})


if __name__ == "__main__":
    main()
