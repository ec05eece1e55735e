"""The drop-in for real: the reference's OWN command-line program (Main.cpp, parameters, all file readers and writers, mating,
phenotype scaling, migration decisions, summaries -- compiled from the reference tree, unmodified) with the six call sites of
INTEGRATION.md rerouted through the C-ABI (integration/gev_glue.cpp, integration/build_gpu_cli.py), run on the same input
files the unmodified reference was given when the golden fixtures were made.  Every per-generation .info file and the .hap, .ped (both
variants) and .int files of the last generation must equal the unmodified reference's byte for byte (sha256 in the fixtures).

  GeneEvolve_glue_on_oracle : linked to the CPU oracle  -> checks the glue and the edit script without a GPU
  GeneEvolve_gpu            : linked to libgeneevolve_amd.so (the product)  -> the GPU test
Both binaries are built in the container (where the reference tree is) and travel with oracle/_ref/."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from tests import helpers
from tests.cli_inputs import write_inputs_from_fixture

REFDIR = os.path.join(os.path.dirname(helpers.GOLDEN), "..", "oracle", "_ref")
CASES = ["dense", "am1", "am2", "mig2", "ex1mut", "ex1sub", "sel1", "vc1", "ex1full", "vt2", "gam2", "om1"]


def run_cli(exe, case, tmp_path):
    fx = helpers.load_fixture(case)
    wd = str(tmp_path / case)
    args = write_inputs_from_fixture(fx, wd)
    r = subprocess.run([exe] + args + ["--out_hap", "--out_interval", "--out_plink"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, f"{case}: exit {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}"
    ngen = int(fx["n_gen"])

    def same(path, key):
        raw = open(path, "rb").read()
        return np.array_equal(np.frombuffer(hashlib.sha256(raw).digest(), dtype=np.uint8), fx[key])
    for ip in range(int(fx["n_pop"])):
        assert same(os.path.join(wd, f"out.pop{ip+1}.summary"), f"summaryfile_pop{ip}_sha"), f"{case}: .summary file of population {ip+1} differs"
        for ic in range(int(fx["nchr"])):
            base = os.path.join(wd, f"out.pop{ip+1}.gen{ngen}.chr{int(fx[f'pop{ip}_chr{ic}_label'])}")
            assert same(base + ".int", f"intfile_pop{ip}_chr{ic}_sha"), f"{case}: .int (--out_interval) file differs (pop {ip+1} chr index {ic})"
            assert same(base + ".ped", f"pedfile_pop{ip}_chr{ic}_sha"), f"{case}: .ped (--out_plink) file differs (pop {ip+1} chr index {ic})"
            for g in (fx["output_generations"] if "output_generations" in fx else []):      # --file_output_generations: files written in the middle of the run
                if int(g) != ngen:
                    mid = os.path.join(wd, f"out.pop{ip+1}.gen{int(g)}.chr{int(fx[f'pop{ip}_chr{ic}_label'])}")
                    assert same(mid + ".hap", f"hapfile_g{int(g)}_pop{ip}_chr{ic}_sha"), f"{case}: .hap file of generation {int(g)} differs"
                    assert same(mid + ".int", f"intfile_g{int(g)}_pop{ip}_chr{ic}_sha"), f"{case}: .int file of generation {int(g)} differs"
    if case in ("dense", "mig2"):               # both PLINK flags write <prefix>.ped: the 0/1 variant needs its own run
        wd01 = str(tmp_path / (case + "_01"))
        r = subprocess.run([exe] + write_inputs_from_fixture(fx, wd01) + ["--out_plink01"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0
        for ip in range(int(fx["n_pop"])):
            for ic in range(int(fx["nchr"])):
                raw = open(os.path.join(wd01, f"out.pop{ip+1}.gen{ngen}.chr{int(fx[f'pop{ip}_chr{ic}_label'])}.ped"), "rb").read()
                assert np.array_equal(np.frombuffer(hashlib.sha256(raw).digest(), dtype=np.uint8), fx[f"ped01file_pop{ip}_chr{ic}_sha"]), f"{case}: .ped (--out_plink01) differs"
    for g in range(ngen + 1):
        for ip in range(int(fx["n_pop"])):
            raw = open(os.path.join(wd, f"out.info.pop{ip+1}.gen{g}.txt"), "rb").read()
            assert np.array_equal(np.frombuffer(hashlib.sha256(raw).digest(), dtype=np.uint8), fx[f"infofile_pop{ip}_gen{g}_sha"]), \
                f"{case}: .info file of generation {g} population {ip+1} differs from the unmodified reference's"
    for ip in range(int(fx["n_pop"])):
        for ic in range(int(fx["nchr"])):
            lab = int(fx[f"pop{ip}_chr{ic}_label"])
            raw = open(os.path.join(wd, f"out.pop{ip+1}.gen{ngen}.chr{lab}.hap"), "rb").read()
            assert np.array_equal(np.frombuffer(hashlib.sha256(raw).digest(), dtype=np.uint8), fx[f"hapfile_pop{ip}_chr{ic}_sha"]), \
                f"{case}: .hap file of population {ip+1} chromosome {lab} differs from the unmodified reference's"


def run_cli_vcf_panel(exe, case, tmp_path):
    """VCF reference panels (--file_ref_vcf): ras_write_vcf_to_vcf_format / _to_hap_legend_sample / _to_plink_format and their
    converters ras_convert_interval_from_vcf_to_{vcf_structure, hap_matrix, plink} (src/Simulation.cpp:1477-1571, 1690-1758,
    1838-1881) bound to the library: .vcf data lines, .hap, .ped, .map, .int and every .info file equal GeneEvolve_ref_vcf's
    (the reference with the `return` its format_vcf::read_vcf_header_sample lacks supplied -- the unmodified build crashes on every
    VCF panel, tests/golden/make_golden.py).  The ##fileDate line of the .vcf header is today's date and is not compared."""
    fx = helpers.load_fixture(case)
    wd = str(tmp_path / case)
    full = f"vcfrun_hapfile_pop0_chr0_sha" in fx
    args = write_inputs_from_fixture(fx, wd, vcf_panel=True)
    r = subprocess.run([exe] + args + ["--out_vcf"] + (["--out_hap", "--out_plink", "--out_interval"] if full else []), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, f"{case}: exit {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}"
    ngen = int(fx["n_gen"])
    digest = lambda raw: np.frombuffer(hashlib.sha256(raw).digest(), dtype=np.uint8)
    for g in range(ngen + 1):
        for ip in range(int(fx["n_pop"])):
            raw = open(os.path.join(wd, f"out.info.pop{ip+1}.gen{g}.txt"), "rb").read()
            assert np.array_equal(digest(raw), fx[f"infofile_pop{ip}_gen{g}_sha"]), f"{case}: .info file of generation {g} population {ip+1} differs"
    for ip in range(int(fx["n_pop"])):
        for ic in range(int(fx["nchr"])):
            base = os.path.join(wd, f"out.pop{ip+1}.gen{ngen}.chr{int(fx[f'pop{ip}_chr{ic}_label'])}")
            body, samples = hashlib.sha256(), None
            for line in open(base + ".vcf", "rb").read().split(b"\n")[:-1]:
                if line.startswith(b"##"):
                    continue
                if line.startswith(b"#"):
                    samples = line.split(b"\t")[9:]
                    continue
                body.update(line + b"\n")
            k = f"vcffile_pop{ip}_chr{ic}_"
            assert len(samples) == int(fx[k + "n_samples"]) and samples[0] == fx[k + "first_sample"].tobytes(), f"{case}: .vcf sample columns (pop {ip+1})"
            assert np.array_equal(np.frombuffer(body.digest(), dtype=np.uint8), fx[k + "body_sha"]), f"{case}: .vcf data lines differ (pop {ip+1} chr index {ic})"
            if full:
                for ext in ("hap", "ped", "map", "int"):
                    assert np.array_equal(digest(open(base + "." + ext, "rb").read()), fx[f"vcfrun_{ext}file_pop{ip}_chr{ic}_sha"]), f"{case}: .{ext} file of the VCF-panel run differs (pop {ip+1} chr index {ic})"


@pytest.mark.parametrize("case", ["vcf1", "dense"])
def test_reference_cli_with_vcf_panels_on_the_c_abi_oracle_backend(case, tmp_path):
    exe = os.path.join(REFDIR, "GeneEvolve_glue_on_oracle")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/GeneEvolve_glue_on_oracle not built (needs the reference tree)")
    run_cli_vcf_panel(exe, case, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["vcf1", "dense"])
def test_reference_cli_with_vcf_panels_on_the_hip_library(case, tmp_path):
    exe = os.path.join(REFDIR, "GeneEvolve_gpu")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/GeneEvolve_gpu not built (needs the reference tree at build time)")
    run_cli_vcf_panel(exe, case, tmp_path)


@pytest.mark.parametrize("case", CASES)
def test_reference_cli_on_the_c_abi_oracle_backend(case, tmp_path):
    exe = os.path.join(REFDIR, "GeneEvolve_glue_on_oracle")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/GeneEvolve_glue_on_oracle not built (needs the reference tree: python -c 'import __graft_entry__ as g; g.build()')")
    run_cli(exe, case, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_reference_cli_on_the_hip_library(case, tmp_path):
    exe = os.path.join(REFDIR, "GeneEvolve_gpu")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/GeneEvolve_gpu not built (needs the reference tree at build time)")
    run_cli(exe, case, tmp_path)


def test_multithreaded_info_writer_is_byte_identical_at_a_size_that_uses_several_threads(tmp_path):
    """the glue's ras_save_human_info replacement formats slices of the population on several host threads (fixtures are too small
    to start more than one): 10 000 individuals, the unmodified reference vs the bound program (oracle backend), .info files
    of every generation byte for byte"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(helpers.GOLDEN), "..", "tools"))
    from cli_timing import write_inputs
    ref, glue = os.path.join(REFDIR, "GeneEvolve_ref"), os.path.join(REFDIR, "GeneEvolve_glue_on_oracle")
    if not (os.path.exists(ref) and os.path.exists(glue)):
        pytest.skip("oracle/_ref binaries not built (needs the reference tree at build time)")
    wd = str(tmp_path)
    args = write_inputs(wd, 10000, 40, 2, False)
    for exe, tag in ((ref, "ref"), (glue, "glue")):
        a = list(args); a[a.index("--prefix") + 1] = os.path.join(wd, tag)
        assert subprocess.run([exe] + a, stdout=subprocess.DEVNULL, timeout=900).returncode == 0
    for g in range(3):
        x = open(os.path.join(wd, f"ref.info.pop1.gen{g}.txt"), "rb").read(); y = open(os.path.join(wd, f"glue.info.pop1.gen{g}.txt"), "rb").read()
        assert len(x) > 500_000 and x == y, f".info file of generation {g} differs"
