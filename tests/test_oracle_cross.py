"""Cross-check the two independent restatements (dict-based vs C, fast and literal flavours) on
seeded random inputs: same fp32 data, fp64 arithmetic, agreement to 1e-12."""

import numpy as np
import pytest

import dsgd_amd
from oracle import oracle as orc
from oracle import ref_dict as rd


def random_rows(rng, n_rows, dim, max_nnz):
    rows = []
    for _ in range(n_rows):
        n = int(rng.integers(1, max_nnz + 1))
        keys = rng.choice(np.arange(1, dim + 1), size=min(n, dim), replace=False)
        vals = np.abs(rng.normal(size=len(keys))).astype(np.float32) + np.float32(0.1)
        vals = (vals / np.float32(np.sqrt((vals.astype(np.float64) ** 2).sum()))).astype(np.float32)
        rows.append(({int(k): float(v) for k, v in zip(keys, vals)}, int(rng.choice([-1, 1]))))
    return rows


def to_dense(s, dim):
    return np.array([s.map.get(k, 0.0) for k in range(dim + 1)])


@pytest.mark.parametrize("seed,dim,n_rows,k_workers,batch", [(0, 12, 40, 3, 5), (1, 50, 120, 4, 10), (2, 200, 300, 1, 64)])
def test_sync_steps_dict_vs_c(seed, dim, n_rows, k_workers, batch):
    rng = np.random.default_rng(seed)
    rows = random_rows(rng, n_rows, dim, max_nnz=min(dim, 12))
    n_train = int(n_rows * 0.8)
    data_d = [(rd.Sparse(dict(m), dim), y) for m, y in rows]
    ds_d = rd.dim_sparsity(data_d[:n_train])
    lam, lr = 1e-2, 0.5
    model = rd.SparseSVM(lam, ds_d)
    csr = dsgd_amd.synth.from_rows(dim, rows)
    o = orc.Oracle(dim, csr.row_ptr, csr.col, csr.val, csr.label, lam)
    o.set_dim_sparsity(o.dim_sparsity(n_train))
    np.testing.assert_array_equal(o.ds, to_dense(ds_d, dim))
    split = rd.split_vanilla(n_train, k_workers)
    w_d = rd.Sparse({}, dim)
    w_c = np.zeros(dim + 1)
    w_l = np.zeros(dim + 1)
    for step in range(6):
        lists = [list(rng.permutation(np.asarray(r))[:batch]) for r in split]
        w_d = rd.master_sync_step(model, data_d, w_d, lists, lr)
        o.sync_step(w_c, lists, lr)
        o.sync_step(w_l, lists, lr, literal=True)
        np.testing.assert_allclose(w_c, to_dense(w_d, dim), rtol=0, atol=1e-12)
        np.testing.assert_allclose(w_l, w_c, rtol=0, atol=1e-12)
    # evaluation
    loss_c, acc_c, counts, _ = o.loss_acc(w_c, n_train, n_rows)
    assert abs(loss_c - rd.local_loss(model, w_d, data_d[n_train:])) < 1e-12
    assert acc_c == rd.local_accuracy(model, w_d, data_d[n_train:])
    assert sum(counts) == n_rows - n_train
    # forward
    idx = list(range(0, n_rows, 3))
    np.testing.assert_array_equal(o.forward(w_c, idx), rd.slave_forward(model, data_d, w_d, idx))


@pytest.mark.parametrize("seed", [3, 4])
def test_async_steps_dict_vs_c(seed):
    rng = np.random.default_rng(seed)
    dim, n_rows = 40, 100
    rows = random_rows(rng, n_rows, dim, max_nnz=8)
    data_d = [(rd.Sparse(dict(m), dim), y) for m, y in rows]
    lam, lr = 1e-2, 0.5
    model = rd.SparseSVM(lam, rd.dim_sparsity(data_d))
    csr = dsgd_amd.synth.from_rows(dim, rows)
    o = orc.Oracle(dim, csr.row_ptr, csr.col, csr.val, csr.label, lam)
    o.set_dim_sparsity(o.dim_sparsity(n_rows))
    w_d = rd.Sparse({}, dim)
    w_c = np.zeros(dim + 1)
    for step in range(10):
        idx = list(rng.permutation(n_rows)[: (1 if step % 3 == 0 else 7)])
        w_d, upd = rd.async_step(model, data_d, w_d, idx, lr)
        delta = o.async_step(w_c, idx, lr, want_delta=True)
        np.testing.assert_allclose(delta, to_dense(upd, dim), rtol=0, atol=1e-12)
        np.testing.assert_allclose(w_c, to_dense(w_d, dim), rtol=0, atol=1e-12)


def test_omp_gradient_matches_sequential():
    csr = dsgd_amd.synth.generate(3000, seed=11)
    o = orc.Oracle(csr.dim, csr.row_ptr, csr.col, csr.val, csr.label, 1e-5)
    o.set_dim_sparsity(o.dim_sparsity(2400))
    w = np.zeros(csr.dim + 1)
    o.sync_step(w, [np.arange(0, 500)], 0.5)
    g_seq = o.gradient(w, np.arange(2400))
    g_omp, n_active = o.gradient_range_omp(w, 0, 2400)
    assert n_active == o.last_stats["n_active"]
    np.testing.assert_allclose(g_omp, g_seq, rtol=0, atol=1e-10)


def test_split_vanilla_quirks():  # SplitStrategy.scala:13-14: may yield fewer groups than workers
    assert [list(r) for r in rd.split_vanilla(9, 4)] == [[0, 1, 2], [3, 4, 5], [6, 7, 8]]
    assert [len(r) for r in rd.split_vanilla(18519, 3)] == [6173, 6173, 6173]
    assert [len(r) for r in rd.split_vanilla(10, 3)] == [4, 4, 2]


def test_early_stopping_no_improvement():  # EarlyStopping.scala:13-46, newest-first list
    crit = rd.no_improvement(patience=2, min_delta=0.01)
    assert not crit([])
    assert not crit([0.5])
    assert not crit([0.4, 0.5])            # newest is the min
    assert not crit([0.45, 0.4, 0.5])      # min at index 1 < patience
    assert crit([0.46, 0.45, 0.4, 0.5])    # min at index 2 >= patience
    # within min_delta of the running min counts as a new min (<=) scanning oldest-last
    assert crit([0.5, 0.5, 0.405, 0.4])
