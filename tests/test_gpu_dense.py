"""-m gpu: K8, the dense logistic mini-batch step (BASELINE.json configs[4]) through the C ABI against the fp64 oracle
(oracle/dense_ref.py).  No reference counterpart: parity unpinned by construction (see the oracle's header).

Stated tolerance: weights after T steps within 2e-6 * max(1, |w|_inf) of the oracle fed the same fp32 data (fp32
products and sums of <= 8192 terms per row and <= 4096 rows per column partial); loss within 1e-6 relative."""

import numpy as np
import pytest

import dsgd_amd
from conftest import has_gpu
from oracle import dense_ref

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="no gfx950 device")]


def make(n, d, seed):
    rng = np.random.default_rng(seed)
    X = (rng.normal(size=(n, d)) / np.sqrt(d)).astype(np.float32)
    y = (X.astype(np.float64) @ rng.normal(size=d) + 0.1 * rng.normal(size=n) > 0).astype(np.float32)
    return X, y


@pytest.mark.parametrize("n,d,batch,mfma", [(1000, 512, 100, 0), (4099, 1024, 1024, 0), (2048, 4096, 777, 0), (300, 8192, 300, 0),
                                            (1000, 512, 100, 1), (4099, 1024, 1024, 1), (2048, 4096, 777, 1)])
def test_steps_match_the_oracle(monkeypatch, n, d, batch, mfma):
    """mfma = 1: the forward product on the matrix cores (v_mfma_f32_16x16x4_f32, DSGD_DENSE_MFMA=1) -- the variant
    BASELINE.json configs[4] names; same oracle, same tolerance."""
    monkeypatch.setenv("DSGD_DENSE_MFMA", str(mfma))
    X, y = make(n, d, n + d)
    with dsgd_amd.DenseLogistic(d) as eng:
        eng.load(X, y)
        w_ref = np.zeros(d)
        lr = 4.0
        for b in range(0, n, batch):
            e = min(n, b + batch)   # ragged last batch
            eng.step(b, e, lr)
            w_ref, _, _ = dense_ref.step(X[b:e], y[b:e], w_ref, lr)
        eng.synchronize()
        w = eng.get_weights().astype(np.float64)
        assert np.abs(w - w_ref).max() <= 2e-6 * max(1.0, np.abs(w_ref).max()), np.abs(w - w_ref).max()
        assert np.abs(w_ref).max() > 1e-3
        loss, acc = eng.loss(0, n)
        loss_ref, _, acc_ref = dense_ref.loss_grad(X, y, w)   # same weights on both sides
        assert abs(loss - loss_ref) <= 1e-6 * max(1.0, loss_ref)
        assert abs(acc - acc_ref) <= 2.0 / n   # rows with z within round-off of 0


def test_generated_shard_trains_and_errors():
    with pytest.raises(ValueError):
        dsgd_amd.DenseLogistic(1000)        # D must be a multiple of 512
    with dsgd_amd.DenseLogistic(4096) as eng:
        with pytest.raises(dsgd_amd.DsgdError):
            eng.step(0, 10, 1.0)            # no data yet
        eng.generate(65536, seed=3)
        with pytest.raises(ValueError):
            eng.step(5, 5, 1.0)
        with pytest.raises(IndexError):
            eng.step(0, 65537, 1.0)
        l0, a0 = eng.loss(0, 65536)
        assert abs(l0 - np.log(2.0)) < 1e-6   # w = 0
        for ep in range(3):
            for b in range(0, 49152, 4096):
                eng.step(b, b + 4096, 8.0)
        eng.synchronize()
        l1, a1 = eng.loss(49152, 65536)     # held-out rows
        assert l1 < l0 - 0.005 and a1 > 0.7, (l1, a1)   # (|x . w| stays small: x ~ N(0,1)/sqrt(D), 36 steps)
        # same seed, same data: the generator is counter-based
        w1 = eng.get_weights()
    with dsgd_amd.DenseLogistic(4096) as eng2:
        eng2.generate(65536, seed=3)
        for ep in range(3):
            for b in range(0, 49152, 4096):
                eng2.step(b, b + 4096, 8.0)
        eng2.synchronize()
        np.testing.assert_array_equal(eng2.get_weights(), w1)   # fixed reduction order: bit-reproducible


def test_communicator_of_one_rank_changes_nothing():
    X, y = make(2000, 512, 5)
    out = []
    for with_comm in (False, True):
        with dsgd_amd.DenseLogistic(512) as eng:
            if with_comm:
                eng.comm_init(dsgd_amd.Engine.comm_unique_id(), 1, 0)
            eng.load(X, y)
            for b in range(0, 2000, 500):
                eng.step(b, b + 500, 2.0)
            eng.synchronize()
            out.append(eng.get_weights())
    np.testing.assert_array_equal(out[0], out[1])
