"""Deterministic synthetic founder panels (pure integer arithmetic, identical on every host).

The device generator gev_synth_founders (geneevolve_amd/csrc) implements exactly this
function; tests compare the two.  SURVEY.md section 8(d): alleles ~ Bernoulli(f_i),
f_i ~ U(0.05, 0.5).

    T_i       = A + ((mix64(seed ^ C1 ^ (i * G)) >> 32) * B >> 32)            (32-bit threshold)
    bit(h, i) = (mix64(((h << 32) | i) + seed * C2) >> 32) < T_i
"""
import numpy as np

_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_C1 = np.uint64(0xA5A5A5A5A5A5A5A5)
_C2 = np.uint64(0xD1342543DE82EF95)
_A = np.uint64(214748365)     # ~0.05 * 2^32
_B = np.uint64(1932735283)    # ~0.45 * 2^32


def mix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + _G
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def synth_thresholds(seed, L):
    i = np.arange(L, dtype=np.uint64)
    with np.errstate(over="ignore"):
        u = mix64(np.uint64(seed) ^ _C1 ^ (i * _G)) >> np.uint64(32)
        return _A + ((u * _B) >> np.uint64(32))


def synth_bits(seed, nhap, L, row_begin=0):
    """-> uint8 [nhap][L] of 0/1"""
    T = synth_thresholds(seed, L)
    i = np.arange(L, dtype=np.uint64)
    out = np.empty((nhap, L), dtype=np.uint8)
    with np.errstate(over="ignore"):
        off = np.uint64(seed) * _C2
        for r in range(nhap):
            ctr = ((np.uint64(row_begin + r) << np.uint64(32)) | i) + off
            out[r] = (mix64(ctr) >> np.uint64(32)) < T
    return out


def synth_packed(seed, nhap, L, row_begin=0):
    """-> uint64 [nhap][ceil(L/64)] in the C-ABI bit packing"""
    b = synth_bits(seed, nhap, L, row_begin)
    pad = (-L) % 64
    if pad:
        b = np.concatenate([b, np.zeros((nhap, pad), dtype=np.uint8)], axis=1)
    return np.packbits(b, axis=1, bitorder="little").view(np.uint64)
