"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

numpy twin of the device-side synthetic data generator:
    value(col c, global row r) = splitmix64(seed ^ (c * GAMMA) ^ r) reduced to the column domain.
The product generates the same bytes directly in HBM (gen_kernel, viyadb_amd/csrc/vh_kernels.h);
this file exists so that the oracle can be fed identical inputs without copying 60 GB around.
"""
from __future__ import annotations

import numpy as np

GAMMA = np.uint64(0x9E3779B97F4A7C15)
GEN_UNIFORM, GEN_ROWID, GEN_CONST, GEN_ZIPF, GEN_SORTED, GEN_HOT = 0, 1, 2, 3, 4, 5


def splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + GAMMA
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def gen_column(col_index: int, dtype, spec, seed: int, row_base: int, nrows: int) -> np.ndarray:
    """spec = (mode, mod, add, scale[, param]) — same meaning as vh_gen_spec (include/viya_hip.h)."""
    mode, mod, add, scale = spec[:4]
    param = spec[4] if len(spec) > 4 else 0
    dtype = np.dtype(dtype)
    r = np.arange(row_base, row_base + nrows, dtype=np.uint64)
    if mode == GEN_ROWID:
        return r.astype(dtype)
    if mode == GEN_CONST:
        return np.full(nrows, add).astype(dtype)
    if mode == GEN_SORTED:
        return ((r // np.uint64(mod)).astype(np.int64) + np.int64(add)).astype(dtype)
    with np.errstate(over="ignore"):
        colseed = np.uint64(seed) ^ (np.uint64(col_index) * GAMMA)
    h = splitmix64(colseed ^ r)
    if mode == GEN_ZIPF:
        octaves = np.uint64(int(mod).bit_length())               # floor(log2 mod) + 1
        b = (h >> np.uint64(32)) % octaves
        one = np.uint64(1)
        v = ((one << b) - one + ((h & np.uint64(0xFFFFFFFF)) % (one << b))) % np.uint64(mod)
        iv = v.astype(np.int64) + np.int64(add)
    else:
        iv = (h % np.uint64(mod)).astype(np.int64) + np.int64(add)
        if mode == GEN_HOT:
            hot = splitmix64(np.uint64(seed) ^ r ^ np.uint64(0x407)) % np.uint64(1000) < np.uint64(param)
            iv = np.where(hot, np.int64(add) + np.int64(mod // 2), iv)
    if dtype.kind == "f":
        return (iv.astype(np.float64) * float(scale)).astype(dtype)
    return iv.astype(dtype)


def gen_bitset_column(col_index: int, spec, seed: int, row_base: int, nrows: int, wide: bool = False):
    """Twin of gen_csr_kernel: `add` ids per row from [0, mod). Returns a list of Python sets (one per row)."""
    mode, mod, add, scale = spec
    k = max(1, min(int(add), 8))
    r = np.arange(row_base, row_base + nrows, dtype=np.uint64)
    with np.errstate(over="ignore"):
        colseed = np.uint64(seed) ^ (np.uint64(col_index) * GAMMA)
        cols = [splitmix64(colseed ^ (r * np.uint64(8) + np.uint64(j))) % np.uint64(mod) for j in range(k)]
    vals = np.stack(cols, axis=1)
    if not wide:
        vals = vals.astype(np.uint32)
    return [set(int(v) for v in row) for row in vals]


def gen_bitset_csr(col_index: int, spec, seed: int, row_base: int, nrows: int, wide: bool = False):
    """The same ids as gen_bitset_column, as CSR arrays (offsets uint64[nrows + 1], ids) — what oracle.cpu_twin reads; exactly `add`
    ids per row, in draw order (a row may repeat an id: sets do not care)."""
    mode, mod, add, scale = spec[:4]
    k = max(1, min(int(add), 8))
    r = np.arange(row_base, row_base + nrows, dtype=np.uint64)
    with np.errstate(over="ignore"):
        colseed = np.uint64(seed) ^ (np.uint64(col_index) * GAMMA)
        cols = [splitmix64(colseed ^ (r * np.uint64(8) + np.uint64(j))) % np.uint64(mod) for j in range(k)]
    vals = np.stack(cols, axis=1).reshape(-1)
    return np.arange(nrows + 1, dtype=np.uint64) * np.uint64(k), (vals if wide else vals.astype(np.uint32))
