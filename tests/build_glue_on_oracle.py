#!/usr/bin/env python3
"""Test infrastructure: oracle/_ref/GeneEvolve_glue_on_oracle = the program of integration/build_gpu_cli.py with its C-ABI calls
forwarded to the CPU oracle (tests/glue_on_oracle.cpp) instead of libgeneevolve_amd.so.  It lets tests/test_cli_dropin.py check
the reference-side glue and the edit script on a machine without a GPU.  Never a product path."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("build_gpu_cli", os.path.join(ROOT, "integration", "build_gpu_cli.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

if __name__ == "__main__":
    mod.build("GeneEvolve_glue_on_oracle", ["-L" + os.path.join(ROOT, "oracle"), "-lgev_oracle", "-Wl,-rpath,$ORIGIN/.."],
              extra_sources=[os.path.join(ROOT, "tests", "glue_on_oracle.cpp")])
    sys.exit(0)
