"""DenseLogistic: the dense mini-batch variant of BASELINE.json configs[4] (K8) as a Python object -- a thin face
over the dsgd_dense_* entry points of include/dsgd.h.  No reference counterpart (the reference's only model is the
sparse hinge SVM, core/ml/SparseSVM.scala:11); all arithmetic happens in the HIP library."""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, f32, ptr


class DenseLogistic:
    def __init__(self, n_features, device=0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self.dim = int(n_features)
        check(self._lib.dsgd_dense_create(C.c_int32(self.dim), C.c_int32(device), C.byref(self._h)))
        self.n_rows = 0

    def close(self):
        if self._h:
            self._lib.dsgd_dense_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def generate(self, n_rows, seed=0):
        check(self._lib.dsgd_dense_generate(self._h, C.c_int64(n_rows), C.c_uint64(seed)))
        self.n_rows = int(n_rows)

    def load(self, X, y):
        X = np.ascontiguousarray(X, dtype=np.float32)
        y = f32(y)
        if X.ndim != 2 or X.shape[1] != self.dim or len(y) != X.shape[0]:
            raise ValueError("X must be n_rows x %d and y n_rows" % self.dim)
        check(self._lib.dsgd_dense_load(self._h, C.c_int64(X.shape[0]), ptr(X), ptr(y)))
        self.n_rows = X.shape[0]

    def set_weights(self, w):
        check(self._lib.dsgd_dense_set_weights(self._h, ptr(f32(w, self.dim))))

    def get_weights(self):
        out = np.zeros(self.dim, dtype=np.float32)
        check(self._lib.dsgd_dense_get_weights(self._h, ptr(out)))
        return out

    def step(self, row_begin, row_end, lr):
        check(self._lib.dsgd_dense_step(self._h, C.c_int64(row_begin), C.c_int64(row_end), C.c_float(lr)))

    def synchronize(self):
        check(self._lib.dsgd_dense_synchronize(self._h))

    def loss(self, row_begin, row_end):
        l, a = C.c_double(0), C.c_double(0)
        check(self._lib.dsgd_dense_loss(self._h, C.c_int64(row_begin), C.c_int64(row_end), C.byref(l), C.byref(a)))
        return l.value, a.value

    def comm_init(self, unique_id, world_size, rank):
        check(self._lib.dsgd_dense_comm_init(self._h, C.c_char_p(unique_id), C.c_int32(world_size), C.c_int32(rank)))

    def prof(self, enable=True):
        ms, n = C.c_double(0), C.c_int64(0)
        check(self._lib.dsgd_dense_prof(self._h, C.c_int32(1 if enable else 0), C.byref(ms), C.byref(n)))
        return ms.value, n.value
