"""-m gpu: many-worker parity of the persistent lock-free engine (BASELINE.json configs[3]) that CAN FAIL.

The engine records, for every mini-batch update in commit order, {worker, the worker's iteration (the sampler's key),
the update count its weights were read at} (dsgd_async_set_trace); the oracle replays the reference's asynchronous
iteration (core/Slave.scala:92-101) with exactly that schedule (oracle/hogwild_replay.py) and the two weight vectors are
compared at every checkpoint: test loss, test accuracy, |w|, relative distance -- under tolerances an order of magnitude
inside round 3's chance-to-perfect band.  The same replay with a deliberately wrong rule (updates applied twice, a
third of them lost, sum instead of mean, half the step, staleness ignored at 256 workers, no regulariser where lambda
makes it matter) must leave those tolerances: the statement has teeth.  The trace itself is held to exact invariants."""

import numpy as np
import pytest

import dsgd_amd
from dsgd_amd import host
from conftest import has_gpu
from oracle import hogwild_replay as hr
from oracle import oracle as orc

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="no gfx950 device")]

BATCH, LR = 100, 0.5   # application.conf:15,18


def check_trace(trace, k, updates, target):
    """Exact properties of a trace: one record per update; a worker's iterations count up from 0 in commit order; the
    weights of a worker's update were read at the commit of its previous one (for its first: somewhere before)."""
    worker, it, read_at = trace
    assert len(worker) == updates and target <= updates < target + k
    assert worker.min() >= 0 and worker.max() < k
    commit = np.arange(1, updates + 1)
    assert np.all(read_at >= 0) and np.all(read_at < commit)
    for j in range(k):
        m = np.flatnonzero(worker == j)
        assert np.array_equal(it[m], np.arange(len(m), dtype=it.dtype)), "worker %d: iterations out of order" % j
        assert np.array_equal(read_at[m][1:], commit[m][:-1]), "worker %d: read_at is not its previous commit" % j
    return int((commit - 1 - read_at).max()), float((commit - 1 - read_at).mean())


def traced_run(k, n_rows, checkpoints, lam, data_seed=13):
    data = dsgd_amd.synth.generate(n_rows, seed=data_seed)
    n_train = int(n_rows * 0.8)
    o = orc.Oracle(data.dim, data.row_ptr, data.col, data.val, data.label, lam)
    o.set_dim_sparsity(o.dim_sparsity(n_train))
    split = [(r.start, r.stop) for r in host.split_vanilla(n_train, k)]
    ev = (n_train, data.n_rows)
    segs, cmps = [], []
    w_rep = np.zeros(data.dim + 1)
    with dsgd_amd.Engine(data.dim, lam) as eng:
        eng.load_csr(data.row_ptr, data.col, data.val, data.label)
        eng.build_dim_sparsity(n_train)
        eng.set_weights(np.zeros(data.dim + 1, dtype=np.float32))
        eng.async_set_trace(max(np.diff([0] + list(checkpoints))) + k)
        prev = 0
        for c, target in enumerate(checkpoints):
            seed = 4242 + 7919 * c
            eng.async_start(split, batch=BATCH, lr=LR, max_updates=target - prev, seed=seed, positional_bug=False)
            eng.async_wait()
            u, running = eng.async_updates()
            assert not running
            trace = eng.async_read_trace()
            max_lag, mean_lag = check_trace(trace, k, u, target - prev)
            loss, acc, _ = eng.loss_acc(*ev)
            w_eng = eng.get_weights().astype(np.float64)
            info = hr.replay_segment(o, w_rep, split, BATCH, LR, seed, trace)
            assert info["max_lag"] == max_lag
            cmp = hr.compare(o, w_eng, w_rep, ev, engine_eval=(loss, acc))
            cmp.update(updates=u, max_lag=max_lag, mean_lag=mean_lag)
            cmps.append(cmp)
            segs.append((seed, trace))
            prev = target
        eng.async_set_trace(0)
    return o, split, ev, segs, cmps, w_eng


def replay_with(o, split, segs, fault):
    w = np.zeros(o.dim + 1)
    for seed, trace in segs:
        hr.replay_segment(o, w, split, BATCH, LR, seed, trace, fault=fault)
    return w


@pytest.mark.parametrize("k,n_rows,checkpoints", [
    (4, 40000, [400, 800, 1200, 1600]),             # the reference deploys 4 slaves (kube/dsgd.yaml:95)
    (64, 40000, [800, 1600, 2400, 3200]),           # the staleness of a wide machine
    (256, 100000, [2048, 4096, 6144, 8192]),        # the benchmarked shape (bench.py hogwild: 256 workers x batch 100)
])
def test_traced_run_is_the_replayed_schedule(k, n_rows, checkpoints):
    o, split, ev, segs, cmps, w_eng = traced_run(k, n_rows, checkpoints, 1e-5)
    for c in cmps:
        print("k=%d after %5d updates (lag max %d mean %.1f): loss %.4f / %.4f  acc %.4f / %.4f  |w| %.3f / %.3f  "
              "rel.dist %.4f  cos %.5f" % (k, c["updates"], c["max_lag"], c["mean_lag"], c["loss_engine"], c["loss_replay"],
                                           c["acc_engine"], c["acc_replay"], c["wnorm_engine"], c["wnorm_replay"],
                                           c["rel_distance"], c["cosine"]))
    for c in cmps:
        ok = hr.within(c)
        assert all(ok.values()), (ok, c)
    # negative controls: the same schedule under a broken rule must NOT pass
    faults = ["double_apply", "drop_third", "sum_not_mean", "half_step"] + (["fresh_reads"] if k >= 256 else [])
    for fault in faults:
        with np.errstate(all="ignore"):
            w_bad = replay_with(o, split, segs, fault)
            cmp = hr.compare(o, w_eng, w_bad, ev, engine_eval=(cmps[-1]["loss_engine"], cmps[-1]["acc_engine"]))
        ok = hr.within(cmp)
        print("k=%d control %-14s rel.dist %.3f |w| %.3f vs %.3f loss %.3f vs %.3f -> %s" % (
            k, fault, cmp["rel_distance"], cmp["wnorm_replay"], cmp["wnorm_engine"], cmp["loss_replay"], cmp["loss_engine"],
            "rejected" if not all(ok.values()) else "NOT rejected"))
        assert not all(ok.values()), (fault, cmp)


def test_traced_run_with_a_regulariser_that_matters():
    """lambda = 3e-2 instead of the reference's 1e-5: the support-only regulariser (core/ml/SparseSVM.scala:31) is then a
    tenth of the gradient, the engine's incrementally kept scalar s = 2 lambda (w . ds) is exercised for real, and a
    replay WITHOUT the regulariser is far off."""
    k, lam = 16, 3e-2
    o, split, ev, segs, cmps, w_eng = traced_run(k, 40000, [400, 800], lam, data_seed=17)
    for c in cmps:
        print("lambda %.0e after %d updates: loss %.4f / %.4f  |w| %.3f / %.3f  rel.dist %.4f" % (
            lam, c["updates"], c["loss_engine"], c["loss_replay"], c["wnorm_engine"], c["wnorm_replay"], c["rel_distance"]))
        assert all(hr.within(c).values()), c
    w_bad = replay_with(o, split, segs, "no_regulariser")
    cmp = hr.compare(o, w_eng, w_bad, ev)
    assert not all(hr.within(cmp).values()), cmp


def test_trace_api_states():
    data = dsgd_amd.synth.generate(4000, seed=3)
    with dsgd_amd.Engine(data.dim, 1e-5) as eng:
        eng.load_csr(data.row_ptr, data.col, data.val, data.label)
        eng.build_dim_sparsity(3200)
        with pytest.raises(dsgd_amd.DsgdError):      # no traced run yet
            eng.async_read_trace()
        with pytest.raises(dsgd_amd.DsgdInvalidArgument):
            eng.async_set_trace(-1)
        eng.async_set_trace(8)                       # a trace shorter than the run keeps its first records
        eng.async_start([(0, 1600), (1600, 3200)], batch=10, lr=0.5, max_updates=40, seed=1, positional_bug=False)
        with pytest.raises(dsgd_amd.DsgdError):      # engine running / not joined
            eng.async_read_trace()
        eng.async_wait()
        worker, it, read_at = eng.async_read_trace()
        assert len(worker) == 8 and np.all(read_at < np.arange(1, 9))
        eng.async_set_trace(0)
        eng.async_start([(0, 3200)], batch=10, lr=0.5, max_updates=5, seed=1, positional_bug=False)   # untraced runs go on as before
        eng.async_wait()
        assert eng.async_updates()[0] == 5
