// Test infrastructure: lets the reference's VCF-panel path run under a modern compiler so that its VCF OUTPUT can pin
// gev_format_vcf_gt.  format_vcf::read_vcf_header_sample (reference src/format_vcf.cpp:367-389) is declared `bool` but has no
// return statement; the caller branches on the result (src/Simulation.cpp:263).  Built with the reference's own -O3 by
// g++ 11 the function runs off its end and the program crashes on every --file_ref_vcf input.  This translation unit
// compiles the reference's file UNCHANGED (included from where it lies, nothing copied) with that one function renamed,
// and supplies the missing `return` around it.  It is linked into oracle/_ref/GeneEvolve_ref_vcf only, which
// tests/golden/make_golden.py uses for the VCF fixture and nothing else; every other fixture comes from the unmodified build.
#define read_vcf_header_sample read_vcf_header_sample_as_written
#include "format_vcf.cpp"
#undef read_vcf_header_sample

namespace format_vcf {
bool read_vcf_header_sample(std::string filename, std::vector<std::string>& sample)
{
    read_vcf_header_sample_as_written(filename, sample);     // fills `sample` (or prints its error and leaves it empty)
    return !sample.empty();
}
}
