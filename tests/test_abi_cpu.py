"""CPU tests of the product library's host side: it loads without a GPU, exports every symbol of
include/geneevolve_amd.h, refuses to compute without a device (no fallback), and its host-built
RNG tables / integer Bernoulli thresholds agree with the oracle's exact arithmetic."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from geneevolve_amd import capi
from oracle import oracle_api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_all_exported(gpu_lib):
    hdr = open(os.path.join(ROOT, "include", "geneevolve_amd.h")).read()
    declared = set(re.findall(r"\b(gev_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"gev_ctx"}
    assert len(declared) >= 30
    for name in sorted(declared):
        assert gpu_lib.exports(name[len("gev_"):]), f"{name} declared in the header but not exported"
    assert declared == {"gev_" + s for s in capi.ABI_SYMBOLS}, "capi.ABI_SYMBOLS out of sync with the header"


def test_no_cpu_fallback_without_device(gpu_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.GevError) as e:
        gpu_lib.create(1, 1, 1)
    assert e.value.code == -3 and "no CPU fallback" in str(e.value)


def test_threshold_tables_match_exact_canonical(gpu_lib, oracle_lib):
    thr = gpu_lib.lib.gev_dbg_threshold
    can = gpu_lib.lib.gev_dbg_canonical
    can.restype = C.c_double
    rs = np.random.RandomState(7)
    NMAX = 2147483646
    probs = [0.0, 1.0, 5e-4, 0.002, 1e-8, 0.5, 0.999999, 1e-300, 2.0, -1.0, 4.6566128752457969e-10, 1.0 / 3] + list(rs.uniform(0, 1, 40)) + list(10 ** rs.uniform(-9, -1, 40))
    for p in probs:
        out = (C.c_uint32 * 4)()
        assert thr(C.c_double(p), out) == 0
        a_lo, a_hi, b0, b1 = [int(x) for x in out]
        assert a_hi - a_lo <= 2

        def hit_thr(a, b):
            if a < a_lo:
                return True
            if a >= a_hi:
                return False
            return b < (b0 if a == a_lo else b1)
        # probe the whole neighbourhood of the window plus random digits
        cand_a = {0, 1, NMAX - 1, max(a_lo - 1, 0), a_lo, min(a_lo + 1, NMAX - 1), min(a_hi, NMAX - 1), min(a_hi + 1, NMAX - 1)} | {int(x) for x in rs.randint(0, NMAX, 6)}
        for a in cand_a:
            if a >= NMAX:
                continue
            cand_b = {0, 1, NMAX - 1, b0, max(b0 - 1, 0), min(b0 + 1, NMAX - 1), b1, max(b1 - 1, 0), min(b1 + 1, NMAX - 1)} | {int(x) for x in rs.randint(0, NMAX, 6)}
            for b in cand_b:
                if b >= NMAX:
                    continue
                r = oracle_api.kat_canonical(oracle_lib, b + 1, a + 1)      # oracle takes engine outputs x1 (low), x2 (high)
                assert can(C.c_uint32(a), C.c_uint32(b)) == r
                assert hit_thr(a, b) == (r < p), (p, a, b)


def test_glibc_linear_tables_reproduce_rand(gpu_lib, oracle_lib):
    """x_{344+k} = sum_i W[i][k] r_i (mod 2^32): emulate the device GlibcWave in numpy."""
    class T(C.Structure):
        _fields_ = [("pow_even", C.c_uint32 * 64), ("pow128", C.c_uint32), ("pow512", C.c_uint32), ("inv16807", C.c_uint32), ("pow_lcg", C.c_uint32 * 31),
                    ("w_init", C.c_uint32 * (31 * 64)), ("w_next", C.c_uint32 * (31 * 64)), ("pad", C.c_uint32 * 2)]
    t = T()
    assert gpu_lib.lib.gev_dbg_tables(C.byref(t), C.c_size_t(C.sizeof(t))) == 0
    M = 2147483647
    assert [int(x) for x in t.pow_even][:3] == [pow(16807, 2, M), pow(16807, 4, M), pow(16807, 6, M)]
    assert (int(t.inv16807) * 16807) % M == 1
    assert int(t.pow128) == pow(16807, 128, M) and int(t.pow512) == pow(16807, 512, M)
    w_init = np.array(t.w_init, dtype=np.uint64).reshape(31, 64)
    w_next = np.array(t.w_next, dtype=np.uint64).reshape(31, 64)
    for seed in [0, 1, 12345, 2147483646, 2147483647, 2147483648, 4294967295, 987654321]:
        s = seed if seed else 1
        w0 = s - (1 << 32) if s >= (1 << 31) else s
        hi, lo = int(w0 / 127773), int(np.fmod(w0, 127773))          # C truncating division
        w = 16807 * lo - 2836 * hi
        if w < 0:
            w += M
        r = np.zeros(31, dtype=np.uint64)
        r[0] = s
        for i in range(1, 31):
            r[i] = (pow(16807, i - 1, M) * w) % M
        assert int(t.pow_lcg[5]) == pow(16807, 4, M)
        x = (r[:, None] * w_init).sum(axis=0) & 0xFFFFFFFF                  # block 0
        out = list(x >> 1)
        for _ in range(2):                                                  # two more blocks
            x = (x[33:64, None] * w_next).sum(axis=0) & 0xFFFFFFFF
            out += list(x >> 1)
        want = oracle_api.kat_rand(oracle_lib, seed, 192)
        assert [int(v) for v in out] == [int(v) for v in want], f"seed {seed}"


def test_cpp_host_fails_loudly_without_gpu():
    """error convention of the reference's seam: message on stdout, failure status, no exception across the ABI"""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    exe = os.path.join(ROOT, "tools", "host_demo")
    if not os.path.exists(exe):
        pytest.skip("tools/host_demo not built (run __graft_entry__.build())")
    r = subprocess.run([exe, "10", "100", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stdout
