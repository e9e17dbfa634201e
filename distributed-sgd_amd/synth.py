"""Synthetic "RCV1-like" CSR data (csrc/synth.c): the input both the engine and the oracle get."""

from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SYNTH_LIB = os.path.join(HERE, "lib", "libdsgd_synth.so")
RCV1_DIM = 47236  # ref: utils/Dataset.scala:16

_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(SYNTH_LIB):
            raise OSError("%s not built: run __graft_entry__.build()" % SYNTH_LIB)
        lib = C.CDLL(SYNTH_LIB)
        lib.dsgd_synth_create.restype = C.c_void_p
        lib.dsgd_synth_create.argtypes = [C.c_uint64, C.c_int32]
        lib.dsgd_synth_create_shaped.restype = C.c_void_p
        lib.dsgd_synth_create_shaped.argtypes = [C.c_uint64, C.c_int32, C.c_double, C.c_double]
        lib.dsgd_synth_destroy.argtypes = [C.c_void_p]
        lib.dsgd_synth_row_ptr.restype = C.c_int64
        lib.dsgd_synth_row_ptr.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
        lib.dsgd_synth_fill.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.dsgd_synth_tau.restype = C.c_double
        lib.dsgd_synth_tau.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


@dataclass
class Csr:
    dim: int
    row_ptr: np.ndarray  # int64 [n_rows + 1]
    col: np.ndarray      # int32, 1-based, ascending per row
    val: np.ndarray      # float32
    label: np.ndarray    # int8 +1/-1

    @property
    def n_rows(self):
        return len(self.row_ptr) - 1

    @property
    def nnz(self):
        return int(self.row_ptr[-1])

    def algorithmic_bytes_per_row(self):
        """8 B per non-zero (fp32 value + int32 column) + 12 B per row."""
        return 8.0 * self.nnz / self.n_rows + 12.0

    def rows(self, begin, end):
        """Sub-matrix copy of rows [begin, end) (used for bounded CPU samples)."""
        b, e = int(self.row_ptr[begin]), int(self.row_ptr[end])
        return Csr(self.dim, (self.row_ptr[begin:end + 1] - b).astype(np.int64), self.col[b:e].copy(),
                   self.val[b:e].copy(), self.label[begin:end].copy())


def generate(n_rows, seed=0, dim=RCV1_DIM, row0=0, zipf=None, nnz_mean=None):
    """Rows [row0, row0 + n_rows) of the infinite synthetic stream for (seed, dim); zipf / nnz_mean: another column-frequency
    exponent / mean row length than the RCV1-like 1.1 / 75."""
    lib = _load()
    if zipf is None and nnz_mean is None:
        g = lib.dsgd_synth_create(C.c_uint64(seed), C.c_int32(dim))
    else:
        g = lib.dsgd_synth_create_shaped(C.c_uint64(seed), C.c_int32(dim), C.c_double(1.1 if zipf is None else zipf),
                                         C.c_double(75.0 if nnz_mean is None else nnz_mean))
    if not g:
        raise ValueError("bad generator arguments")
    try:
        row_ptr = np.zeros(n_rows + 1, dtype=np.int64)
        nnz = lib.dsgd_synth_row_ptr(g, C.c_int64(row0), C.c_int64(n_rows), row_ptr.ctypes.data_as(C.c_void_p))
        col = np.empty(nnz, dtype=np.int32)
        val = np.empty(nnz, dtype=np.float32)
        label = np.empty(n_rows, dtype=np.int8)
        lib.dsgd_synth_fill(g, C.c_int64(row0), C.c_int64(n_rows), row_ptr.ctypes.data_as(C.c_void_p),
                            col.ctypes.data_as(C.c_void_p), val.ctypes.data_as(C.c_void_p), label.ctypes.data_as(C.c_void_p))
    finally:
        lib.dsgd_synth_destroy(g)
    return Csr(dim, row_ptr, col, val, label)


def from_rows(dim, rows):
    """rows: list of ({key: value}, label) -- small hand-written cases (KATs)."""
    row_ptr, col, val, label = [0], [], [], []
    for m, y in rows:
        for k in sorted(m):
            col.append(k)
            val.append(m[k])
        row_ptr.append(len(col))
        label.append(y)
    return Csr(dim, np.asarray(row_ptr, np.int64), np.asarray(col, np.int32), np.asarray(val, np.float32),
               np.asarray(label, np.int8))
