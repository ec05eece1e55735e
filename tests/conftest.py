import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build them the way __graft_entry__.build() does, once.
    (Building is not a fallback: the product path still is the HIP library and nothing else.)"""
    lib = os.path.join(ROOT, "geneevolve_amd", "csrc", "libgeneevolve_amd.so")
    demo = os.path.join(ROOT, "tools", "host_demo")
    if not (os.path.exists(lib) and os.path.exists(demo)):
        import shutil
        if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
            import __graft_entry__
            __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle_api
    return oracle_api.load()


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library (HIP).  Fails loudly when it is missing: there is no fallback."""
    from geneevolve_amd.capi import GevLibrary
    return GevLibrary()


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The reference-built binaries (oracle/_ref/, git-ignored, built where the reference tree is and travelling with the working
    tree) are what the drop-in CLI tests and bench.py's `cpu_baseline.kind = "reference"` depend on: say plainly whether they are
    here and how many tests were skipped because one was not."""
    ref = os.path.join(ROOT, "oracle", "_ref")
    names = ("ref_harness", "GeneEvolve_ref", "GeneEvolve_ref_vcf", "GeneEvolve_gpu", "GeneEvolve_glue_on_oracle")
    have = [n for n in names if os.path.exists(os.path.join(ref, n))]
    skipped = [r for r in terminalreporter.stats.get("skipped", []) if "oracle/_ref" in str(getattr(r, "longrepr", ""))]
    terminalreporter.write_line(f"oracle/_ref binaries present: {', '.join(have) if have else 'none'}"
                                + (f"; MISSING: {', '.join(n for n in names if n not in have)}" if len(have) < len(names) else "")
                                + f"; {len(skipped)} test(s) skipped for a missing oracle/_ref binary")
