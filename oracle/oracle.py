"""ORACLE (test infrastructure, NOT product code) -- ctypes face of oracle.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
See oracle.c for the restatement itself and its "parity unpinned" status.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(src) > os.path.getmtime(LIB_PATH):
        proc = subprocess.run(["make", "-C", HERE, "-B"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if proc.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + proc.stdout)
    return LIB_PATH


class _Csr(C.Structure):
    _fields_ = [
        ("n_rows", C.c_int64),
        ("dim", C.c_int32),
        ("row_ptr", C.c_void_p),
        ("col", C.c_void_p),
        ("val", C.c_void_p),
        ("label", C.c_void_p),
    ]


class GateStats(C.Structure):
    _fields_ = [("n_active", C.c_int64), ("n_exact_zero", C.c_int64), ("min_abs_margin", C.c_double)]

    def as_dict(self):
        return {"n_active": self.n_active, "n_exact_zero": self.n_exact_zero, "min_abs_margin": self.min_abs_margin}


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_row_dot.restype = C.c_double
        _lib.orc_dense_dot.restype = C.c_double
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """fp64 restatement bound to one CSR data set (the reference's Array[(Vec, Int)])."""

    def __init__(self, dim, row_ptr, col, val, label, lam, ds=None):
        self.dim = int(dim)
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        self.val = np.ascontiguousarray(val, dtype=np.float32)
        self.label = np.ascontiguousarray(label, dtype=np.int8)
        self.n_rows = len(self.row_ptr) - 1
        self.lam = float(lam)
        self._csr = _Csr(self.n_rows, self.dim, _p(self.row_ptr), _p(self.col), _p(self.val), _p(self.label))
        self.ds = None if ds is None else np.ascontiguousarray(ds, dtype=np.float64)
        self.last_stats = None

    # -- Main.scala:54-65 ---------------------------------------------------------------------
    def dim_sparsity(self, n_train):
        ds = np.zeros(self.dim + 1, dtype=np.float64)
        lib().orc_dim_sparsity(C.byref(self._csr), C.c_int64(n_train), _p(ds))
        return ds

    def set_dim_sparsity(self, ds):
        self.ds = np.ascontiguousarray(ds, dtype=np.float64)

    def row_dot(self, i, w):
        w = np.ascontiguousarray(w, dtype=np.float64)
        return lib().orc_row_dot(C.byref(self._csr), C.c_int64(i), _p(w))

    # -- core/Slave.scala:142-157 ---------------------------------------------------------------
    def gradient(self, w, idx, literal=False):
        w = np.ascontiguousarray(w, dtype=np.float64)
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        g = np.zeros(self.dim + 1, dtype=np.float64)
        if literal:
            rc = lib().orc_lit_gradient(C.byref(self._csr), _p(w), _p(self.ds), C.c_double(self.lam), _p(idx),
                                        C.c_int64(len(idx)), _p(g))
        else:
            st = GateStats()
            rc = lib().orc_gradient(C.byref(self._csr), _p(w), _p(self.ds), C.c_double(self.lam), _p(idx),
                                    C.c_int64(len(idx)), _p(g), C.byref(st))
            self.last_stats = st.as_dict()
        _check(rc)
        return g

    # -- core/Master.scala:186-197 --------------------------------------------------------------
    def sync_step(self, w, idx_per_worker, lr, literal=False):
        """In place on w (float64, D+1)."""
        assert w.dtype == np.float64 and w.flags.c_contiguous
        lists = [np.ascontiguousarray(i, dtype=np.int32) for i in idx_per_worker]
        k = len(lists)
        ptrs = (C.c_void_p * k)(*[_p(a) for a in lists])
        ns = (C.c_int64 * k)(*[len(a) for a in lists])
        if literal:
            rc = lib().orc_lit_sync_step(C.byref(self._csr), _p(w), _p(self.ds), C.c_double(self.lam), ptrs, ns,
                                         C.c_int32(k), C.c_double(lr))
        else:
            st = GateStats()
            rc = lib().orc_sync_step(C.byref(self._csr), _p(w), _p(self.ds), C.c_double(self.lam), ptrs, ns,
                                     C.c_int32(k), C.c_double(lr), C.byref(st))
            self.last_stats = st.as_dict()
        _check(rc)
        return w

    def sync_step_range_omp(self, w, row_begin, row_end, lr):
        assert w.dtype == np.float64 and w.flags.c_contiguous
        n_active = C.c_int64(0)
        rc = lib().orc_sync_step_range_omp(C.byref(self._csr), _p(w), _p(self.ds), C.c_double(self.lam),
                                           C.c_int64(row_begin), C.c_int64(row_end), C.c_double(lr), C.byref(n_active))
        _check(rc)
        return n_active.value

    def gate_profile(self, w, row_begin, row_end, eps=1e-5):
        """(rows with 0 < |x.w| < eps, per-coordinate sum of |x_j| over those rows): see orc_range_gate_profile."""
        w = np.ascontiguousarray(w, dtype=np.float64)
        near = np.zeros(self.dim + 1, dtype=np.float64)
        n = C.c_int64(0)
        _check(lib().orc_range_gate_profile(C.byref(self._csr), _p(w), C.c_int64(row_begin), C.c_int64(row_end),
                                            C.c_double(eps), C.byref(n), _p(near)))
        return n.value, near

    def gradient_range_omp(self, w, row_begin, row_end):
        w = np.ascontiguousarray(w, dtype=np.float64)
        g = np.zeros(self.dim + 1, dtype=np.float64)
        n_active = C.c_int64(0)
        rc = lib().orc_gradient_range_omp(C.byref(self._csr), _p(w), _p(self.ds), C.c_double(self.lam),
                                          C.c_int64(row_begin), C.c_int64(row_end), _p(g), C.byref(n_active))
        _check(rc)
        return g, n_active.value

    # -- core/Slave.scala:92-101 ----------------------------------------------------------------
    def async_step(self, w, idx, lr, want_delta=False):
        assert w.dtype == np.float64 and w.flags.c_contiguous
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        delta = np.zeros(self.dim + 1, dtype=np.float64) if want_delta else None
        st = GateStats()
        rc = lib().orc_async_step(C.byref(self._csr), _p(w), _p(self.ds), C.c_double(self.lam), _p(idx),
                                  C.c_int64(len(idx)), C.c_double(lr), _p(delta) if want_delta else None, C.byref(st))
        self.last_stats = st.as_dict()
        _check(rc)
        return delta

    # -- core/Slave.scala:129-140 ---------------------------------------------------------------
    def forward(self, w, idx):
        w = np.ascontiguousarray(w, dtype=np.float64)
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        pred = np.zeros(len(idx), dtype=np.float64)
        _check(lib().orc_forward(C.byref(self._csr), _p(w), _p(idx), C.c_int64(len(idx)), _p(pred)))
        return pred

    # -- core/Master.scala:100-107 --------------------------------------------------------------
    def loss_acc(self, w, row_begin, row_end):
        w = np.ascontiguousarray(w, dtype=np.float64)
        loss, acc, mam = C.c_double(0), C.c_double(0), C.c_double(0)
        counts = (C.c_int64 * 3)()
        _check(lib().orc_loss_acc(C.byref(self._csr), _p(w), C.c_double(self.lam), C.c_int64(row_begin),
                                  C.c_int64(row_end), C.byref(loss), C.byref(acc), counts, C.byref(mam)))
        return loss.value, acc.value, list(counts), mam.value


def num_threads():
    return lib().orc_num_threads()


def _check(rc):
    if rc == -1:
        raise ValueError("Cannot sum an empty list of vectors")  # math/Vec.scala:129
    if rc == -2:
        raise IndexError("sample index outside the data")
    if rc != 0:
        raise RuntimeError("oracle error %d" % rc)
