"""RCV1-v2 text files <-> CSR with the reference loader's semantics (utils/Dataset.scala:13-60).

    load(folder, full=True)            Dataset.rcv1: train file, then test_pt0..3; CCAT -> +1 else -1, LAST qrels line
                                       of a document wins; tokens 2.. of a vector line are the features
    export(folder, data, ...)          write a Csr in the same formats, so that a JVM owner can run the reference
                                       (`DSGD_NODE_COUNT=1 sbt run`) on exactly the data the engine was measured on

The parser itself is native (csrc/rcv1.c); this module is the ctypes wrapper and the writer.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _build
from .synth import Csr

RCV1_DIM = 47236  # utils/Dataset.scala:16
FILES = ["lyrl2004_vectors_train.dat"] + ["lyrl2004_vectors_test_pt%d.dat" % d for d in range(4)]
QRELS = "rcv1-v2.topics.qrels"
N_TRAIN_OFFICIAL = 23149  # documents in lyrl2004_vectors_train.dat

_lib = None


def _load():
    global _lib
    if _lib is None:
        lib = C.CDLL(_build.build_rcv1())
        lib.dsgd_rcv1_load.restype = C.c_void_p
        lib.dsgd_rcv1_load.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        lib.dsgd_rcv1_rows.restype = C.c_int64
        lib.dsgd_rcv1_rows.argtypes = [C.c_void_p]
        lib.dsgd_rcv1_nnz.restype = C.c_int64
        lib.dsgd_rcv1_nnz.argtypes = [C.c_void_p]
        lib.dsgd_rcv1_copy.restype = None
        lib.dsgd_rcv1_copy.argtypes = [C.c_void_p] * 6
        lib.dsgd_rcv1_free.restype = None
        lib.dsgd_rcv1_free.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def load(folder: str, full: bool = True, dim: int = RCV1_DIM, with_ids: bool = False):
    """Dataset.rcv1(folder, full).  Raises ValueError on what makes the reference throw (malformed line, missing label,
    missing file)."""
    lib = _load()
    err = C.create_string_buffer(512)
    h = lib.dsgd_rcv1_load(os.fsencode(folder), 1 if full else 0, err, len(err))
    if not h:
        raise ValueError(err.value.decode("utf-8", "replace") or "rcv1 load failed")
    try:
        n, nnz = lib.dsgd_rcv1_rows(h), lib.dsgd_rcv1_nnz(h)
        row_ptr = np.zeros(n + 1, dtype=np.int64)
        col = np.empty(nnz, dtype=np.int32)
        val = np.empty(nnz, dtype=np.float32)
        label = np.empty(n, dtype=np.int8)
        ids = np.empty(n, dtype=np.int32)
        lib.dsgd_rcv1_copy(h, row_ptr.ctypes.data, col.ctypes.data, val.ctypes.data, label.ctypes.data, ids.ctypes.data)
    finally:
        lib.dsgd_rcv1_free(h)
    data = Csr(dim, row_ptr, col, val, label)
    return (data, ids) if with_ids else data


def export(folder: str, data: Csr, n_train_file: int = N_TRAIN_OFFICIAL, first_id: int = 2286, other_topic_lines: bool = True):
    """Write `data` as RCV1-v2 text: rows [0, n_train_file) into the train file, the rest split evenly over the four
    test parts, labels into the qrels file.  A +1 row gets the line `CCAT <id> 1` LAST; a -1 row gets only other
    topics -- or, when `other_topic_lines`, `CCAT` followed by another topic, which the reference also reads as -1
    (last line wins) -- alternating, so both spellings of -1 occur.  Values are written with 9 significant digits
    (fp32 round trip)."""
    os.makedirs(folder, exist_ok=True)
    n = data.n_rows
    ids = np.arange(first_id, first_id + n, dtype=np.int64)
    bounds = [0, min(n, n_train_file)]
    rest = n - bounds[1]
    for d in range(4):
        bounds.append(bounds[-1] + (rest + 3 - d) // 4)
    rp, col, val = data.row_ptr, data.col, data.val
    for k, name in enumerate(FILES):
        with open(os.path.join(folder, name), "w") as f:
            for i in range(bounds[k], bounds[k + 1]):
                b, e = int(rp[i]), int(rp[i + 1])
                feats = " ".join("%d:%.9g" % (int(c), float(v)) for c, v in zip(col[b:e], val[b:e]))
                f.write("%d  %s\n" % (ids[i], feats))  # two spaces after the id, as in the official files
    with open(os.path.join(folder, QRELS), "w") as f:
        for i in range(n):
            if data.label[i] > 0:
                f.write("ECAT %d 1\nCCAT %d 1\n" % (ids[i], ids[i]) if i % 3 == 0 else "CCAT %d 1\n" % ids[i])
            elif other_topic_lines and i % 2 == 0:
                f.write("CCAT %d 1\nGCAT %d 1\n" % (ids[i], ids[i]))
            else:
                f.write("MCAT %d 1\n" % ids[i])
    return ids
