"""Write the input FILES of the reference's command-line program from a golden fixture (tests/golden/*.npz) -- the same text
files tests/golden/make_golden.py gave the reference when it produced that fixture (numbers are written with repr(), which
round-trips doubles exactly) -- and return the argument list.  Used to run oracle/_ref/GeneEvolve_gpu (the reference's own
host bound to the library, integration/) on the GPU box, where neither the reference nor make_golden's work directory exist."""
import os

import numpy as np

from geneevolve_amd.capi import unpack_rows, bytes_to_words
from tests.synth import synth_packed


def _legend_alleles(n):
    i = np.arange(n)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[i % 4], np.frombuffer(b"CGTA", dtype=np.uint8)[(i + i // 4) % 4]


def _write_hap(path, bits):                      # [nhap][L] -> SNP-major text
    with open(path, "w") as f:
        for i in range(bits.shape[1]):
            f.write(" ".join("1" if b else "0" for b in bits[:, i]) + "\n")


def _write_vcf(path, chrom, bits, pos, prefix):
    """phased biallelic VCF of a founder panel [nhap][L]: the file tests/golden/make_golden.py:write_vcf gave the reference"""
    al0, al1 = _legend_alleles(len(pos))
    n = bits.shape[0] // 2
    F = np.asarray(bits, dtype=np.uint8)
    with open(path, "w") as f:
        f.write("##fileformat=VCFv4.1\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n")
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"{prefix}{i+1}" for i in range(n)) + "\n")
        for j, p in enumerate(pos):
            gts = "\t".join(f"{F[2*i, j]}|{F[2*i+1, j]}" for i in range(n))
            f.write(f"{chrom}\t{int(p)}\trs{j+1}\t{chr(al0[j])}\t{chr(al1[j])}\t.\tPASS\t.\tGT\t{gts}\n")


def write_inputs_from_fixture(fx, wd, prefix="out", vcf_panel=False):
    """vcf_panel: the founder panels as phased VCF files behind --file_ref_vcf instead of hap/legend/indv behind --file_hap_name"""
    os.makedirs(wd, exist_ok=True)
    n_pop, nchr, nphen = int(fx["n_pop"]), int(fx["nchr"]), int(fx["nphen"])
    args = []
    for ip in range(n_pop):
        pre = f"pop{ip}_"
        nh = int(fx[pre + "n_founder_hap"])
        labels = [int(fx[f"{pre}chr{ic}_label"]) for ic in range(nchr)]
        with open(os.path.join(wd, f"p{ip}.hapaddr.txt"), "w") as f:
            f.write("chr hap legend sample\n")
            for ic, c in enumerate(labels):
                base = os.path.join(wd, f"p{ip}.chr{c}")
                pos = fx[f"{pre}chr{ic}_snp_pos"]; L = len(pos)
                if f"{pre}chr{ic}_founders" in fx:
                    bits = unpack_rows(bytes_to_words(fx[f"{pre}chr{ic}_founders"], L), L)
                else:
                    bits = unpack_rows(synth_packed(int(fx[f"{pre}chr{ic}_founders_synth_seed"]), nh, L), L)
                _write_hap(base + ".hap", bits)
                al0, al1 = _legend_alleles(L)
                with open(base + ".legend", "w") as g:
                    g.write("id pos al0 al1\n")
                    for i, p in enumerate(pos):
                        g.write(f"rs{i+1} {int(p)} {chr(al0[i])} {chr(al1[i])}\n")
                with open(base + ".indv", "w") as g:
                    for i in range(nh // 2):
                        g.write(f"p{ip}i{i+1}\n")
                f.write(f"{c} {base}.hap {base}.legend {base}.indv\n")
                if vcf_panel:
                    _write_vcf(base + ".vcf", c, bits, pos, f"p{ip}i")
        if vcf_panel:
            with open(os.path.join(wd, f"p{ip}.vcfaddr.txt"), "w") as f:
                f.write("chr vcf\n")
                for c in labels:
                    f.write(f"{c} {os.path.join(wd, f'p{ip}.chr{c}')}.vcf\n")
        with open(os.path.join(wd, f"p{ip}.rmap.txt"), "w") as f:
            f.write("chr bp cM\n")
            for ic, c in enumerate(labels):
                for b, v in zip(fx[f"{pre}chr{ic}_rmap_bp"], fx[f"{pre}chr{ic}_rmap_cM"]):
                    f.write(f"{c} {int(b)} {float(v)!r}\n")
        with open(os.path.join(wd, f"p{ip}.popinfo.txt"), "w") as f:
            f.write("pop_size mat_cor offspring_dist selection_func selection_func_par1 selection_func_par2\n")
            for r in fx[pre + "popinfo"]:
                f.write(str(r) + "\n")
        a = ["--file_gen_info", os.path.join(wd, f"p{ip}.popinfo.txt")] + \
            (["--file_ref_vcf", os.path.join(wd, f"p{ip}.vcfaddr.txt")] if vcf_panel else ["--file_hap_name", os.path.join(wd, f"p{ip}.hapaddr.txt")]) + \
            ["--file_recom_map", os.path.join(wd, f"p{ip}.rmap.txt")]
        if int(fx[pre + "has_mut"]):
            with open(os.path.join(wd, f"p{ip}.mmap.txt"), "w") as f:
                f.write("chr bp mutation_rate\n")
                for ic, c in enumerate(labels):
                    for b, v in zip(fx[f"{pre}chr{ic}_mut_bp"], fx[f"{pre}chr{ic}_mut_rate"]):
                        f.write(f"{c} {int(b)} {float(v)!r}\n")
            a += ["--file_mutation_map", os.path.join(wd, f"p{ip}.mmap.txt")]
        if int(fx[pre + "rm"]):
            a += ["--RM"]
        for iph in range(nphen):
            with open(os.path.join(wd, f"p{ip}.ph{iph}.cvinfo.txt"), "w") as f:
                f.write("chr pos a d\n")
                for ic, c in enumerate(labels):
                    k = f"{pre}ph{iph}_chr{ic}_"
                    for b, x, y in zip(fx[k + "cv_bp"], fx[k + "cv_a"], fx[k + "cv_d"]):
                        f.write(f"{c} {int(b)} {float(x)!r} {float(y)!r}\n")
            with open(os.path.join(wd, f"p{ip}.ph{iph}.cvaddr.txt"), "w") as f:
                for ic, c in enumerate(labels):
                    k = f"{pre}ph{iph}_chr{ic}_"
                    ncv = len(fx[k + "cv_bp"])
                    if k + "cv_val" in fx:
                        val = unpack_rows(bytes_to_words(fx[k + "cv_val"], ncv), ncv)
                    else:
                        val = unpack_rows(synth_packed(int(fx[k + "cv_val_synth_seed"]), nh, ncv), ncv)
                    fn = os.path.join(wd, f"p{ip}.ph{iph}.chr{c}.cv.hap")
                    _write_hap(fn, val)
                    f.write(f"{c} {fn}\n")
            a += ["--file_cv_info", os.path.join(wd, f"p{ip}.ph{iph}.cvinfo.txt"), "--file_cvs", os.path.join(wd, f"p{ip}.ph{iph}.cvaddr.txt")]
        for j, key in enumerate(("va", "vd", "ve", "vf")):
            for iph in range(nphen):
                a += [f"--{key}", repr(float(fx[f"{pre}ph{iph}_var"][j]))]
        for iph in range(nphen):
            a += ["--vc", repr(float(fx[f"{pre}ph{iph}_vc"]))]
        for key in ("omega", "lambda"):
            if f"{pre}ph0_{key}" in fx:
                for iph in range(nphen):
                    a += [f"--{key}", repr(float(fx[f"{pre}ph{iph}_{key}"]))]
        if ip > 0:
            args.append("--next_population")
        args += a
    extra = [str(x) for x in fx["args_extra"] if str(x)]
    if "--file_migration" in extra:
        mp = os.path.join(wd, "mig.txt")
        with open(mp, "w") as f:
            for row in fx["migration_mat_gen"]:
                f.write(" ".join(repr(float(v)) for v in row) + "\n")
        extra[extra.index("--file_migration") + 1] = mp
    args += ["--seed", str(int(fx["seed"])), "--prefix", os.path.join(wd, prefix)] + extra
    if "output_generations" in fx:
        og = os.path.join(wd, "outgens.txt")
        with open(og, "w") as f:
            f.write("".join(f"{int(g)}\n" for g in fx["output_generations"]))
        args += ["--file_output_generations", og]
    return args
