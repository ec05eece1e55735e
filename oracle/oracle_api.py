"""Python handle on the CPU oracle (oracle/libgev_oracle.so, symbols gevo_*).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from geneevolve_amd/.
It reuses the product's generic ctypes binding class (same ABI shapes, other prefix).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from geneevolve_amd.capi import GevLibrary

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "libgev_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "gev_oracle.cpp")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "libgev_oracle.so"], check=True, stdout=subprocess.DEVNULL)
    return LIB


def load():
    return GevLibrary(build(), prefix="gevo_")


# known-answer helpers -----------------------------------------------------------------------
def kat_rand(lib, seed, n):
    out = np.zeros(n, dtype=np.int32)
    lib.lib.gevo_kat_rand(C.c_uint(seed), C.c_int(n), out.ctypes.data_as(C.c_void_p))
    return out


def kat_minstd(lib, seed, n):
    out = np.zeros(n, dtype=np.uint64)
    lib.lib.gevo_kat_minstd(C.c_uint(seed), C.c_int(n), out.ctypes.data_as(C.c_void_p))
    return out


def kat_u01(lib, seed, n):
    out = np.zeros(n, dtype=np.float64)
    lib.lib.gevo_kat_u01(C.c_uint(seed), C.c_int(n), out.ctypes.data_as(C.c_void_p))
    return out


def kat_uint(lib, engine_seed, lo, hi, n):
    out = np.zeros(n, dtype=np.uint64)
    lib.lib.gevo_kat_uint(C.c_uint(engine_seed), C.c_uint64(lo), C.c_uint64(hi), C.c_int(n), out.ctypes.data_as(C.c_void_p))
    return out


def kat_normal(lib, seed, sd, n):
    out = np.zeros(n, dtype=np.float64)
    lib.lib.gevo_kat_normal(C.c_uint(seed), C.c_double(sd), C.c_int(n), out.ctypes.data_as(C.c_void_p))
    return out


def kat_canonical(lib, x1, x2):
    f = lib.lib.gevo_kat_canonical
    f.restype = C.c_double
    return f(C.c_uint64(x1), C.c_uint64(x2))


def kat_sim_loc_rec(lib, bp, prob, bp_dist, seed, cap=4096):
    bp = np.ascontiguousarray(bp, dtype=np.uint64); prob = np.ascontiguousarray(prob, dtype=np.float64)
    locs = np.zeros(cap, dtype=np.uint64); nx = np.zeros(2, dtype=np.int32)
    n = lib.lib.gevo_kat_sim_loc_rec(bp.ctypes.data_as(C.c_void_p), prob.ctypes.data_as(C.c_void_p), C.c_size_t(len(bp)),
                                     C.c_uint64(int(bp_dist)), C.c_uint(seed), locs.ctypes.data_as(C.c_void_p), C.c_size_t(cap),
                                     nx.ctypes.data_as(C.c_void_p))
    return locs[:n].copy(), nx


def kat_recombine(lib, parts, mut_counts, muts, hap_n, start, locs):
    from geneevolve_amd.capi import PART_DTYPE
    parts = np.ascontiguousarray(parts, dtype=PART_DTYPE)
    mc = np.ascontiguousarray(mut_counts, dtype=np.uint64); mu = np.ascontiguousarray(muts, dtype=np.uint64)
    hn = np.ascontiguousarray(hap_n, dtype=np.uint64); lc = np.ascontiguousarray(locs, dtype=np.uint64)
    cap_p, cap_m = 4 * (len(parts) + len(lc)) + 16, 4 * len(mu) + 16
    op = np.zeros(cap_p, dtype=PART_DTYPE); omc = np.zeros(cap_p, dtype=np.uint64); om = np.zeros(cap_m, dtype=np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    n = lib.lib.gevo_kat_recombine(p(parts), p(mc), p(mu), p(hn), C.c_int(start), p(lc), C.c_size_t(len(lc)),
                                   p(op), p(omc), p(om), C.c_size_t(cap_p), C.c_size_t(cap_m))
    return op[:n].copy(), omc[:n].copy(), om[:int(omc[:n].sum())].copy()
