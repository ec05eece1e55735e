#!/usr/bin/env python3
"""Build oracle/_ref/GeneEvolve_gpu: the reference's own command-line program with its reproduction hot path bound to
libgeneevolve_amd.so (the patch of INTEGRATION.md, applied mechanically).

Nothing of the reference is stored in this repository: an edited build copy of src/Simulation.{h,cpp} is generated under a
temporary directory from the sources where they lie, compiled together with integration/gev_glue.cpp, linked with the
reference's other (unmodified) objects that oracle/Makefile.ref already produced, and deleted.  (tests/build_glue_on_oracle.py
reuses build() to link the same objects against the CPU oracle: a checker for machines without a GPU, not a product.)  Edits (all located by the
function signatures / the one marker comment, not by line numbers):

  Simulation.h    class Simulation gets `friend struct GevGlue;`
  Simulation.cpp  ras_init_parameters       + hand the static inputs to the library before its final `return true`
                  ras_initial_human_gen0    + gev_init_gen0 with the function's own `seed` before its final `return true`
                  reproduce                 body -> gevglue_reproduce
                  ras_compute_AD            body -> gevglue_compute_AD
                  ras_do_migration          + gev_migrate before the "remove migrants from the origin population" block
                  ras_convert_interval_to_hap_matrix     body -> gevglue_hap_matrix      (--out_hap)
                  ras_convert_interval_to_format_plink   body -> gevglue_plink_matrix    (--out_plink, --out_plink01)
                  ras_write_hap_to_interval_format       body -> gevglue_write_interval  (--out_interval)
                  ras_convert_interval_from_vcf_to_hap_matrix      body -> gevglue_vcf_hap_matrix    (VCF panel: --out_hap)
                  ras_convert_interval_from_vcf_to_vcf_structure   body -> gevglue_vcf_structure     (VCF panel: --out_vcf)
                  ras_convert_interval_from_vcf_to_plink           body -> gevglue_vcf_plink_matrix  (VCF panel: --out_plink, --out_plink01)
                  assort_mate               the two CommFunc::ras_rank calls -> gevglue_rank (gev_rank_f64: stable sort on the device)
                  sim_next_generation       + gevglue_presample before the mating of each population (gev_presample; random mating)
                  ras_init_generation0 / sim_next_generation   population[ipop].ras_save_human_info(gen_num) -> gevglue_save_human_info
                                            (the same .info bytes, written once instead of flushed per individual)
                  ras_scale_AD_compute_GEF  + first statement: with GEV_GEF_DEVICE=1 return gevglue_scale_gef (gev_scale_ad_compute_gef)
                  random_mate               + first statement: return gevglue_random_mate (gev_random_mate) unless GEV_MATE_HOST=1
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("GEV_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "oracle", "_ref")


def body_span(src, signature_regex):
    """(start of '{', index after matching '}') of the function whose definition matches the regex"""
    m = re.search(signature_regex, src, flags=re.M)
    if not m:
        raise SystemExit(f"build_gpu_cli: cannot find {signature_regex!r} in the reference")
    i = src.index("{", m.end() - 1)
    depth, j = 0, i
    while True:
        c = src[j]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return i, j + 1
        j += 1


def replace_body(src, signature_regex, new_body):
    i, j = body_span(src, signature_regex)
    return src[:i] + "{\n" + new_body + "\n}" + src[j:]


def insert_before_last_return_true(src, signature_regex, stmt):
    i, j = body_span(src, signature_regex)
    body = src[i:j]
    k = body.rindex("return true;")
    return src[:i] + body[:k] + stmt + "\n    " + body[k:] + src[j:]


def build(out_name, backend_link, extra_sources=()):
    """edit + compile + link; backend_link = linker arguments of the C-ABI implementation; extra_sources = more C++ files"""
    if not os.path.isdir(REF):
        print("build_gpu_cli: no reference tree, skipped"); return None
    need = [os.path.join(OUT, "ge", f"{n}.o") for n in ("Main", "Population", "CommFunc", "RasRandomNumber", "RasMatrix", "format_hap", "format_plink", "format_vcf_return", "parameters")]   # format_vcf_return: format_vcf.cpp with
    # the `return` its read_vcf_header_sample lacks (oracle/ref_vcf_return.cpp); without it every --file_ref_vcf run crashes at start-up
    need.append(os.path.join(OUT, "libStatGen.a"))
    for f in need:
        if not os.path.exists(f):
            raise SystemExit(f"build_gpu_cli: {f} missing -- run make -C oracle -f Makefile.ref first")
    h = open(os.path.join(REF, "src", "Simulation.h")).read()
    cpp = open(os.path.join(REF, "src", "Simulation.cpp")).read()
    m = re.search(r"class\s+Simulation\s*\{", h)
    h = h[:m.end()] + "\n    friend struct GevGlue;\n" + h[m.end():]
    decl = ("\nclass Simulation;\nbool gevglue_init_static(Simulation&);\nbool gevglue_after_gen0(Simulation&, int, unsigned);\n"
            "std::vector<Human> gevglue_reproduce(Simulation&, int, int);\nbool gevglue_compute_AD(Simulation&, int, int);\n"
            "bool gevglue_migrate(Simulation&, const std::vector<std::vector<unsigned long int> >&, const std::vector<std::vector<unsigned long int> >&);\n"
            "bool gevglue_hap_matrix(Simulation&, int, std::vector<Legend>&, int, Hap_SNP&);\n"
            "bool gevglue_plink_matrix(Simulation&, int, std::vector<Legend>&, int, std::vector<std::vector<bool> >&, plink_PED_ids&, plink_MAP&);\n"
            "bool gevglue_write_interval(Simulation&, int);\n"
            "bool gevglue_vcf_hap_matrix(Simulation&, std::vector<vcf_structure>&, int, int, Hap_SNP&);\n"
            "bool gevglue_vcf_structure(Simulation&, vcf_structure&, int, int, int, std::vector<vcf_structure>&);\n"
            "bool gevglue_vcf_plink_matrix(Simulation&, std::vector<std::vector<bool> >&, plink_PED_ids&, plink_MAP&, int, int, std::vector<vcf_structure>&);\n"
            "bool gevglue_save_human_info(Simulation&, int, int);\n"
            "bool gevglue_presample(Simulation&, int, int);\nstd::vector<unsigned long int> gevglue_rank(std::vector<double>&);\n"
            "bool gevglue_use_device_gef();\nbool gevglue_scale_gef(Simulation&, int, int, int, double, double);\n"
            "bool gevglue_use_device_mate();\nbool gevglue_random_mate(Simulation&, int, int);\n")
    inc = re.search(r'#include\s+"Simulation.h"', cpp)
    cpp = cpp[:inc.end()] + decl + cpp[inc.end():]
    cpp = insert_before_last_return_true(cpp, r"^bool\s+Simulation::ras_init_parameters\s*\(", "if (!gevglue_init_static(*this)) return false;")
    cpp = insert_before_last_return_true(cpp, r"^bool\s+Simulation::ras_initial_human_gen0\s*\(\s*int\s+ipop\s*\)", "if (!gevglue_after_gen0(*this, ipop, seed)) return false;")
    cpp = replace_body(cpp, r"^std::vector<Human>\s+Simulation::reproduce\s*\(\s*int\s+ipop\s*,\s*int\s+gen_num\s*\)", "    return gevglue_reproduce(*this, ipop, gen_num);")
    cpp = replace_body(cpp, r"^bool\s+Simulation::ras_compute_AD\s*\(\s*int\s+ipop\s*,\s*int\s+gen_num\s*\)", "    return gevglue_compute_AD(*this, ipop, gen_num);")
    cpp = replace_body(cpp, r"^bool\s+Simulation::ras_convert_interval_to_hap_matrix\s*\(", "    return gevglue_hap_matrix(*this, ipop, pops_legend, ichr, hap_snp);")
    cpp = replace_body(cpp, r"^bool\s+Simulation::ras_convert_interval_to_format_plink\s*\(", "    return gevglue_plink_matrix(*this, ipop, pops_legend, ichr, matrix_plink_ped, plink_ped_ids, plink_map);")
    cpp = replace_body(cpp, r"^bool\s+Simulation::ras_write_hap_to_interval_format\s*\(\s*int\s+gen_num\s*\)", "    return gevglue_write_interval(*this, gen_num);")
    # the VCF-panel siblings (:1477, :1690, :1838)
    cpp = replace_body(cpp, r"^bool\s+Simulation::ras_convert_interval_from_vcf_to_hap_matrix\s*\(", "    return gevglue_vcf_hap_matrix(*this, vcf_structure_allpops_chr, ipop, ichr, hap_snp);")
    cpp = replace_body(cpp, r"^bool\s+Simulation::ras_convert_interval_from_vcf_to_vcf_structure\s*\(", "    return gevglue_vcf_structure(*this, vcf_out, gen_num, ipop, ichr, vcf_structure_allpops_chr);")
    cpp = replace_body(cpp, r"^bool\s+Simulation::ras_convert_interval_from_vcf_to_plink\s*$", "    return gevglue_vcf_plink_matrix(*this, matrix_plink_ped, plink_ped_ids, plink_map, ipop, ichr, vcf_structure_allpops_chr);")
    # assort_mate (:2278-2279): the two O(n^2) CommFunc::ras_rank calls on the bivariate-normal template -> device sort
    i, j = body_span(cpp, r"^bool\s+Simulation::assort_mate\s*\(")
    body = cpp[i:j]
    if body.count("CommFunc::ras_rank(") != 2:
        raise SystemExit("build_gpu_cli: expected two CommFunc::ras_rank calls in assort_mate")
    cpp = cpp[:i] + body.replace("CommFunc::ras_rank(", "gevglue_rank(") + cpp[j:]
    # sim_next_generation (:1907): head start for the sampling kernels before the host mates (random mating only)
    i, j = body_span(cpp, r"^bool\s+Simulation::sim_next_generation\s*\(")
    body = cpp[i:j]
    k = body.index("if (population[ipop]._RM)")
    cpp = cpp[:i] + body[:k] + "if (!gevglue_presample(*this, ipop, gen_num)) return false;\n        " + body[k:] + cpp[j:]
    # ras_scale_AD_compute_GEF (:3075): optional device version (GEV_GEF_DEVICE=1), else the function's own body
    i, j = body_span(cpp, r"^bool\s+Simulation::ras_scale_AD_compute_GEF\s*\(")
    cpp = cpp[:i + 1] + "\n    if (gevglue_use_device_gef()) return gevglue_scale_gef(*this, gen_num, ipop, iphen, s2_a_gen0, s2_d_gen0);\n" + cpp[i + 1:]
    # random_mate (:2090): the library forms the couples (gev_random_mate); the function's own body stays behind GEV_MATE_HOST=1
    i, j = body_span(cpp, r"^bool\s+Simulation::random_mate\s*\(\s*int\s+ipop\s*,\s*int\s+gen_ind\s*\)")
    cpp = cpp[:i + 1] + "\n    if (gevglue_use_device_mate()) return gevglue_random_mate(*this, ipop, gen_ind);\n" + cpp[i + 1:]
    # Population::ras_save_human_info call sites (:610, :2018): same bytes, one write instead of one flush per individual
    n_calls = cpp.count("population[ipop].ras_save_human_info(gen_num);")
    if n_calls != 2:
        raise SystemExit(f"build_gpu_cli: expected two ras_save_human_info call sites, found {n_calls}")
    cpp = cpp.replace("population[ipop].ras_save_human_info(gen_num);", "gevglue_save_human_info(*this, ipop, gen_num);")
    i, j = body_span(cpp, r"^bool\s+Simulation::ras_do_migration\s*\(")
    body = cpp[i:j]
    k = body.index("// remove migrants from the origin population")
    cpp = cpp[:i] + body[:k] + "if (!gevglue_migrate(*this, move_sample_pop, num_move)) return false;\n    " + body[k:] + cpp[j:]

    tmp = tempfile.mkdtemp(prefix="gev_gpu_cli_")
    try:
        open(os.path.join(tmp, "Simulation.h"), "w").write(h)
        open(os.path.join(tmp, "Simulation.cpp"), "w").write(cpp)
        sg = os.path.join(REF, "Library", "libStatGen")
        asan = ["-fsanitize=address"] if os.environ.get("GEV_GLUE_ASAN") else []      # CPU-side debugging of the glue only
        flags = (["-O1", "-g"] + asan if asan else ["-O3"]) + ["-std=c++11", "-pthread", "-w", "-D__ZLIB_AVAILABLE__", "-D_FILE_OFFSET_BITS=64", "-D__STDC_LIMIT_MACROS",
                 "-I" + tmp, "-I" + os.path.join(sg, "general"), "-I" + os.path.join(sg, "vcf"), "-I" + os.path.join(sg, "samtools"),
                 "-I" + os.path.join(REF, "Library", "eigen3"), "-I" + os.path.join(REF, "src"), "-I" + os.path.join(ROOT, "include"),
                 "-include", os.path.join(ROOT, "oracle", "ref_compat.h")]
        objs = []
        for n, srcf in enumerate([os.path.join(tmp, "Simulation.cpp"), os.path.join(ROOT, "integration", "gev_glue.cpp")] + list(extra_sources)):
            o = os.path.join(tmp, f"o{n}.o")
            subprocess.run(["g++"] + flags + ["-c", srcf, "-o", o], check=True)
            objs.append(o)
        exe = os.path.join(OUT, out_name)
        subprocess.run(["g++", "-o", exe] + asan + objs + [f for f in need if f.endswith(".o")] + [os.path.join(OUT, "libStatGen.a"), "-lz", "-pthread"] + list(backend_link), check=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    print("built", exe)
    return exe


def main():
    libdir = os.path.join(ROOT, "geneevolve_amd", "csrc")
    build("GeneEvolve_gpu", ["-L" + libdir, "-lgeneevolve_amd", "-Wl,-rpath,$ORIGIN/../../geneevolve_amd/csrc"])
    return 0


if __name__ == "__main__":
    sys.exit(main())
