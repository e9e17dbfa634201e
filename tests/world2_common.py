"""Shared by tests/test_gpu_world2.py and its rank processes (tests/world2_worker.py): the configuration, the row
partition (SplitStrategy.vanilla per rank: core/ml/SplitStrategy.scala:13-14), the per-rank index lists."""

from collections import namedtuple

import numpy as np

CFG = {
    "n_rows": 120000, "n_train": 100000, "seed": 17, "lam": 1e-5,
    "lr_range": 0.5 * 100 / 100000,            # per hosted worker and rank: the per-sample step of application.conf:15,18
    "list_steps": [(1, 100), (3, 100), (1, 5000), (2, 700)],
    "exch_every": 5, "async_updates": 40, "async_batch": 100, "async_seed": 900, "async_range": (200, 40200),
}

Shard = namedtuple("Shard", "csr n_train train_lo train_hi test_lo test_hi")


def split_range(lo, hi, rank, world):
    n = hi - lo
    size = -(-n // world)
    a = min(hi, lo + rank * size)
    return a, min(hi, a + size)


def shard_of(data, n_train, rank, world):
    """Rank `rank` holds its contiguous part of the train rows followed by its part of the test rows."""
    import dsgd_amd

    tr = split_range(0, n_train, rank, world)
    te = split_range(n_train, data.n_rows, rank, world)
    rows = np.concatenate([np.arange(*tr), np.arange(*te)])
    starts, ends = data.row_ptr[rows], data.row_ptr[rows + 1]
    lens = ends - starts
    row_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    flat = np.arange(int(lens.sum()), dtype=np.int64) + np.repeat(starts - row_ptr[:-1], lens)
    csr = dsgd_amd.synth.Csr(data.dim, row_ptr, data.col[flat].copy(), data.val[flat].copy(), data.label[rows].copy())
    return Shard(csr, tr[1] - tr[0], tr[0], tr[1], te[0], te[1])


def local_lists(rank, step, k, b, n_train_local):
    rng = np.random.default_rng(1000 * rank + step)
    size = -(-n_train_local // k)
    return [(min(j * size, n_train_local - 1) + rng.permutation(min(size, n_train_local - j * size))[:b]).astype(np.int32) for j in range(k)]


def dense_problem():
    rng = np.random.default_rng(5)
    d, n_steps, bsz = 512, 4, 192
    n = 2 * n_steps * bsz
    X = (rng.normal(size=(n, d)) / np.sqrt(d)).astype(np.float32)
    w_star = rng.normal(size=d)
    y = (X @ w_star + 0.1 * rng.normal(size=n) > 0).astype(np.float32)
    return X, y, n_steps, bsz
