#!/usr/bin/env python3
"""PLINK .bed bodies derived from the UNMODIFIED reference's own output.

The reference has no .bed writer (BASELINE config 5 asks for "PLINK bit-packed output"), but it writes the same genotypes as
text: --out_plink01 -> <prefix>.popP.genG.chrC.ped with alleles "0"/"1" (format_plink::write_ped01_map,
reference src/format_plink.cpp:77-141).  This script runs oracle/_ref/GeneEvolve_ref (built from /root/reference by
oracle/Makefile.ref) on the input files of the golden fixtures that hold a .ped01 hash, checks that the file it gets is the very
file the fixture pinned (sha256), and packs its genotype columns per the PLINK 1 binary specification (SNP-major, 2 bits per
individual, low bits first; A1 = allele "1": 00 = 1/1, 10 = heterozygous, 11 = 0/0, 01 = missing (never produced); pad bits 0;
without the 3 magic bytes).  The result, tests/golden/bed_from_ref_ped01.npz, is data: what gev_format_bed /
gev_materialize_bed must reproduce.

    python tests/golden/make_bed_golden.py        (in the build container; needs /root/reference via oracle/_ref/GeneEvolve_ref)
"""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import helpers                                   # noqa: E402
from tests.cli_inputs import write_inputs_from_fixture      # noqa: E402


def ped01_to_bed(raw):
    lines = raw.decode().splitlines()
    rows = [l.split() for l in lines]
    n = len(rows)
    L = (len(rows[0]) - 6) // 2
    g = np.array([[int(v) for v in r[6:]] for r in rows], dtype=np.uint8).reshape(n, L, 2)      # [individual][snp][hap]
    ones = g.sum(axis=2)                                     # number of "1" alleles
    code = np.where(ones == 2, 0, np.where(ones == 1, 2, 3)).astype(np.uint8)                   # 00 / 10 / 11
    bpl = (n + 3) // 4
    padded = np.zeros((bpl * 4, L), dtype=np.uint8); padded[:n] = code
    q = padded.reshape(bpl, 4, L)
    packed = (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8)      # [byte][snp]
    return np.ascontiguousarray(packed.T), n, L              # SNP-major


def main():
    exe = os.path.join(ROOT, "oracle", "_ref", "GeneEvolve_ref")
    out = {}
    for case in ("dense", "mig2"):
        fx = helpers.load_fixture(case)
        ngen = int(fx["n_gen"])
        with tempfile.TemporaryDirectory(prefix="gev_bed_") as wd:
            args = write_inputs_from_fixture(fx, wd)
            subprocess.run([exe] + args + ["--out_plink01"], check=True, stdout=subprocess.DEVNULL)
            for ip in range(int(fx["n_pop"])):
                for ic in range(int(fx["nchr"])):
                    lab = int(fx[f"pop{ip}_chr{ic}_label"])
                    raw = open(os.path.join(wd, f"out.pop{ip+1}.gen{ngen}.chr{lab}.ped"), "rb").read()
                    assert np.array_equal(np.frombuffer(hashlib.sha256(raw).digest(), dtype=np.uint8), fx[f"ped01file_pop{ip}_chr{ic}_sha"]), \
                        f"{case}: this run's .ped01 differs from the pinned one"
                    bed, n, L = ped01_to_bed(raw)
                    out[f"{case}_pop{ip}_chr{ic}_bed"] = bed
                    out[f"{case}_pop{ip}_chr{ic}_shape"] = np.array([n, L])
                    print(case, ip, ic, "individuals", n, "SNPs", L, "bed bytes", bed.size)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "bed_from_ref_ped01.npz"), **out)


if __name__ == "__main__":
    main()
