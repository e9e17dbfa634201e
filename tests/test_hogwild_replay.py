"""CPU: the ORACLE-side replay of a traced lock-free run (oracle/hogwild_replay.py) checked on its own.

A trace is synthesised from a schedule model written independently here (k workers of equal speed: every gradient is
computed on the weights as they were k - 1 updates before its commit, looked up in the full history of weight
vectors); replay_segment fed that trace must land on the same weights, and every negative control (a deliberately
wrong update rule) must leave the stated tolerances -- the check that the GPU test relies on can fail."""

import numpy as np
import pytest

import dsgd_amd
from dsgd_amd import host
from oracle import hogwild_replay as hr
from oracle import oracle as orc


def problem(n_rows=6000, lam=1e-5, seed=31):
    data = dsgd_amd.synth.generate(n_rows, seed=seed)
    n_train = int(n_rows * 0.8)
    o = orc.Oracle(data.dim, data.row_ptr, data.col, data.val, data.label, lam)
    o.set_dim_sparsity(o.dim_sparsity(n_train))
    return data, o, n_train


def model_run(o, split, batch, lr, seed, n_updates):
    """k equally fast workers, written without the replay's machinery: the whole history of weight vectors is kept and
    update c reads entry max(0, c - k).  (Reconstructing a stale vector by adding recent updates back would not do: a
    row whose x.w is EXACTLY zero -- most rows early on -- would come out at +-1e-18 and be gated at random.)
    Returns (final weights, trace)."""
    k = len(split)
    hist = [np.zeros(o.dim + 1)]
    worker, it, read_at = [], [], []
    for c in range(1, n_updates + 1):
        j, i, r = (c - 1) % k, (c - 1) // k, max(0, c - k)
        b, e = split[j]
        delta = o.async_step(hist[r].copy(), hr.hog_rows(seed, j, i, b, e - b, batch), lr, want_delta=True)
        w = hist[-1] - delta
        w[np.abs(w) <= 1e-20] = 0.0
        hist.append(w)
        worker.append(j)
        it.append(i)
        read_at.append(r)
    return hist[-1], (np.asarray(worker, np.int32), np.asarray(it, np.uint32), np.asarray(read_at, np.int64))


@pytest.mark.parametrize("k", [1, 4, 16])
def test_replay_reproduces_a_modelled_schedule(k):
    data, o, n_train = problem()
    split = [(r.start, r.stop) for r in host.split_vanilla(n_train, k)]
    w_model, trace = model_run(o, split, 50, 0.5, 99, 160)
    w = np.zeros(o.dim + 1)
    info = hr.replay_segment(o, w, split, 50, 0.5, 99, trace)
    assert info["updates"] == 160 and info["max_lag"] == k - 1
    assert np.abs(w - w_model).max() <= 1e-12 * max(1.0, np.abs(w_model).max())
    cmp = hr.compare(o, w_model, w, (n_train, data.n_rows))
    assert all(hr.within(cmp).values()) and cmp["rel_distance"] < 1e-12


def test_every_negative_control_leaves_the_tolerances():
    """What the traced check is FOR: a broken update rule must be rejected.  (Dropping the regulariser needs a lambda at
    which it matters: at the reference's 1e-5 the term is 1e-4 of the gradient -- that one is covered by the exact
    single-worker replay, tests/test_gpu_parity.py::test_hogwild_single_worker_replays_the_oracle.)"""
    data, o, n_train = problem(lam=3e-2)
    k = 8
    split = [(r.start, r.stop) for r in host.split_vanilla(n_train, k)]
    w_model, trace = model_run(o, split, 50, 0.5, 7, 400)
    rejected = {}
    for fault in hr.FAULTS:
        w = np.zeros(o.dim + 1)
        hr.replay_segment(o, w, split, 50, 0.5, 7, trace, fault=fault)
        cmp = hr.compare(o, w_model, w, (n_train, data.n_rows))
        rejected[fault] = (not all(hr.within(cmp).values()), round(cmp["rel_distance"], 3))
    # (with 8 workers the staleness hardly matters: `fresh_reads` may stay inside -- the 256-worker GPU test is where
    #  ignoring the staleness is far off)
    for fault in ("double_apply", "drop_third", "sum_not_mean", "half_step", "no_regulariser"):
        assert rejected[fault][0], rejected


def test_inconsistent_traces_are_refused():
    data, o, n_train = problem(n_rows=2000)
    split = [(0, n_train)]
    w = np.zeros(o.dim + 1)
    with pytest.raises(ValueError):   # an update cannot have read the weights its own commit produced
        hr.replay_segment(o, w, split, 10, 0.5, 1, (np.zeros(3, np.int32), np.arange(3, dtype=np.uint32), np.asarray([0, 2, 1])))
    assert hr.replay_segment(o, w, split, 10, 0.5, 1, (np.zeros(0, np.int32), np.zeros(0, np.uint32), np.zeros(0, np.int64)))["updates"] == 0
