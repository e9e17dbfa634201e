"""Pin the oracle: (1) every golden vector the reference's own tests hold for this path
(src/test/scala/epfl/distributed/data/VecTests.scala:14-40), (2) the two hand-derived
known-answer tests of SURVEY.md 8(c), computed here by two independent restatements
(dict-based ref_dict.py and the C oracle) and compared with the tabulated values."""

import math

import numpy as np
import pytest

import dsgd_amd
from oracle import oracle as orc
from oracle import ref_dict as rd


# ---- VecTests.scala ---------------------------------------------------------------------------
def test_vectests_sparse_add():  # VecTests.scala:26-29
    v1 = rd.Sparse({0: 1, 1: 2, 2: 3}, 4)
    v2 = rd.Sparse({1: 1, 2: 2, 3: 3}, 4)
    assert v1 + v2 == rd.Sparse({0: 1, 1: 3, 2: 5, 3: 3}, 4)


def test_vectests_dense_values_on_sparse_type():  # VecTests.scala:14-21 (values; the training path uses Sparse)
    v = rd.Sparse({0: 1, 1: 2, 2: 3}, 3)
    assert v + v == rd.Sparse({0: 2, 1: 4, 2: 6}, 3)
    assert v.dot(v) == 1 + 4 + 9
    assert v * 2 == rd.Sparse({0: 2, 1: 4, 2: 6}, 3)
    assert 3 * v == rd.Sparse({0: 3, 1: 6, 2: 9}, 3)
    assert v.norm() == math.sqrt(1 + 4 + 9)


def test_vectests_division_by_zero():  # VecTests.scala:33-34 (IllegalArgumentException)
    with pytest.raises(ValueError):
        rd.Sparse({0: 1, 1: 2, 2: 3}, 4) / 0


def test_vectests_sparsity():  # VecTests.scala:38-40
    assert rd.Sparse({0: 1, 1: 2}, 10).sparsity() == 0.8


def test_c_oracle_dot_matches_vectests():
    # (1,2,3).(1,2,3) = 14 through the CSR row-dot of the C oracle
    data = dsgd_amd.synth.from_rows(3, [({1: 1.0, 2: 2.0, 3: 3.0}, 1)])
    o = orc.Oracle(3, data.row_ptr, data.col, data.val, data.label, 0.0)
    assert o.row_dot(0, np.array([0.0, 1.0, 2.0, 3.0])) == 14.0


# ---- KATs of SURVEY.md 8(c) -----------------------------------------------------------------------
KAT_ROWS = [
    ({1: .6, 3: .8}, +1), ({2: 1.0}, -1), ({3: .6, 4: .8}, -1),
    ({1: .8, 6: .6}, +1), ({1: .6, 3: .8}, -1), ({2: .6, 6: .8}, +1),
]


def _dict_data(rows, dim):
    return [(rd.Sparse(dict(m), dim), y) for m, y in rows]


def _close(a: rd.Sparse, expect: dict, tol=1e-9):
    assert set(a.map) == set(expect), (a, expect)
    for k, v in expect.items():
        assert abs(a.map[k] - v) < tol, (k, a.map[k], v)


def test_kat1_dict():
    dim = 6
    data = _dict_data(KAT_ROWS, dim)
    ds = rd.dim_sparsity(data)
    _close(ds, {0: .25, 1: 1 / 3, 2: .25, 3: .5, 5: 1 / 3})
    model = rd.SparseSVM(0.1, ds)
    w = rd.Sparse({}, dim)
    assert rd.local_loss(model, w, data) == 1.0 and rd.local_accuracy(model, w, data) == 0.0
    # step 0
    g0 = rd.slave_gradient(model, data, w, [0, 1, 2])
    g1 = rd.slave_gradient(model, data, w, [3, 4, 5])
    _close(g0, {1: .6, 2: -1, 3: .2, 4: -.8})
    _close(g1, {1: .2, 2: .6, 3: -.8, 6: 1.4})
    w = rd.master_sync_step(model, data, w, [[0, 1, 2], [3, 4, 5]], 0.25)
    _close(w, {1: -.1, 2: .05, 3: .075, 4: .1, 6: -.175})
    assert abs(rd.local_loss(model, w, data) - 0.3392083333333333) < 1e-12
    assert rd.local_accuracy(model, w, data) == 4 / 6
    # step 1 (s = 2*lambda*(w.ds) = 1/300: only keys 1,2,3 overlap because of the off-by-one)
    s = model.lam * 2.0 * w.dot(ds)
    assert abs(s - 1 / 300) < 1e-15
    g0 = rd.slave_gradient(model, data, w, [0, 1, 2])
    g1 = rd.slave_gradient(model, data, w, [3, 4, 5])
    _close(g0, {1: .6 + s, 3: .8 + s})
    _close(g1, {1: -.6 + s, 3: -.8 + s})
    w = rd.master_sync_step(model, data, w, [[0, 1, 2], [3, 4, 5]], 0.25)
    _close(w, {1: -.1 - .25 * s, 2: .05, 3: .075 - .25 * s, 4: .1, 6: -.175})
    assert abs(rd.local_loss(model, w, data) - 0.339212638) < 1e-8
    assert rd.local_accuracy(model, w, data) == 5 / 6
    # step 2
    s2 = model.lam * 2.0 * w.dot(ds)
    assert abs(s2 - 0.003194444) < 1e-8
    g0 = rd.slave_gradient(model, data, w, [0, 1, 2])
    g1 = rd.slave_gradient(model, data, w, [3, 4, 5])
    _close(g0, {})
    _close(g1, {1: -.6 + s2, 3: -.8 + s2})
    w = rd.master_sync_step(model, data, w, [[0, 1, 2], [3, 4, 5]], 0.25)
    _close(w, {1: -.026232638, 2: .05, 3: .173767361, 4: .1, 6: -.175}, tol=1e-8)
    assert abs(rd.local_loss(model, w, data) - 0.340734158) < 1e-8
    assert rd.local_accuracy(model, w, data) == 5 / 6


def test_kat2_dict():
    dim = 6
    data = _dict_data(KAT_ROWS[:4], dim)
    model = rd.SparseSVM(0.1, rd.dim_sparsity(data))
    w = rd.Sparse({}, dim)
    _close(rd.slave_gradient(model, data, w, [0, 1]), {1: .6, 2: -1, 3: .8})
    _close(rd.slave_gradient(model, data, w, [2, 3]), {1: .8, 3: -.6, 4: -.8, 6: .6})
    w = rd.master_sync_step(model, data, w, [[0, 1], [2, 3]], 0.5)
    _close(w, {1: -.35, 2: .25, 3: -.05, 4: .2, 6: -.15})
    assert abs(rd.local_loss(model, w, data) - 0.025) < 1e-12
    assert rd.local_accuracy(model, w, data) == 1.0
    for _ in range(2):  # every row inactive: empty-support path of valueLike (Vec.scala:66-67)
        _close(rd.slave_gradient(model, data, w, [0, 1]), {})
        _close(rd.slave_gradient(model, data, w, [2, 3]), {})
        w2 = rd.master_sync_step(model, data, w, [[0, 1], [2, 3]], 0.5)
        assert w2 == w


def _c_oracle(rows, dim, lam):
    data = dsgd_amd.synth.from_rows(dim, rows)
    o = orc.Oracle(dim, data.row_ptr, data.col, data.val.astype(np.float32), data.label, lam)
    o.set_dim_sparsity(o.dim_sparsity(len(rows)))
    return o


def test_kat1_c_oracle_matches_dict_with_same_fp32_inputs():
    # the C oracle takes fp32 values (the engine's input type); run the dict restatement on the
    # SAME fp32-rounded values and require 1e-12 agreement over three sync steps
    dim = 6
    rows32 = [({k: float(np.float32(v)) for k, v in m.items()}, y) for m, y in KAT_ROWS]
    data = _dict_data(rows32, dim)
    model = rd.SparseSVM(0.1, rd.dim_sparsity(data))
    o = _c_oracle(KAT_ROWS, dim, 0.1)
    np.testing.assert_allclose(o.ds, [model.dim_sparsity.map.get(k, 0.0) for k in range(dim + 1)], rtol=0, atol=0)
    w_d = rd.Sparse({}, dim)
    w_c = np.zeros(dim + 1)
    for step in range(3):
        for idx in ([0, 1, 2], [3, 4, 5]):
            g_c = o.gradient(w_c, idx)
            g_l = o.gradient(w_c, idx, literal=True)
            g_d = rd.slave_gradient(model, data, w_d, idx)
            np.testing.assert_allclose(g_c, [g_d.map.get(k, 0.0) for k in range(dim + 1)], rtol=0, atol=1e-15)
            np.testing.assert_allclose(g_l, g_c, rtol=0, atol=1e-15)
        w_d = rd.master_sync_step(model, data, w_d, [[0, 1, 2], [3, 4, 5]], 0.25)
        o.sync_step(w_c, [[0, 1, 2], [3, 4, 5]], 0.25)
        np.testing.assert_allclose(w_c, [w_d.map.get(k, 0.0) for k in range(dim + 1)], rtol=0, atol=1e-15)
        loss_c, acc_c, counts, _ = o.loss_acc(w_c, 0, 6)
        assert abs(loss_c - rd.local_loss(model, w_d, data)) < 1e-14
        assert acc_c == rd.local_accuracy(model, w_d, data)
    assert abs(loss_c - 0.340734158) < 1e-6 and acc_c == 5 / 6


def test_empty_batch_raises_like_vec_sum():  # math/Vec.scala:129
    o = _c_oracle(KAT_ROWS, 6, 0.1)
    with pytest.raises(ValueError):
        o.gradient(np.zeros(7), [])
    with pytest.raises(ValueError):
        rd.vec_sum([])


# ---- the committed fixtures (tests/golden/*.json) against BOTH restatements --------------------------------------
def _golden(name):
    import json
    import os

    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)))


def test_fixture_vectests_json():
    g = _golden("vectests.json")
    d = g["dense_14_21"]
    v = rd.Sparse({i: x for i, x in enumerate(d["v"])}, 3)
    assert (v + v) == rd.Sparse({i: x for i, x in enumerate(d["v_plus_v"])}, 3)
    assert v.dot(v) == d["v_dot_v"] and v.norm() == math.sqrt(d["norm_squared"])
    assert v * 2 == rd.Sparse({i: x for i, x in enumerate(d["v_times_2"])}, 3)
    assert 3 * v == rd.Sparse({i: x for i, x in enumerate(d["3_times_v"])}, 3)
    s = g["sparse_add_26_29"]
    a, b = (rd.Sparse({int(k): x for k, x in s[n].items()}, s["size"]) for n in ("a", "b"))
    assert a + b == rd.Sparse({int(k): x for k, x in s["sum"].items()}, s["size"])
    sp = g["sparsity_38_40"]
    assert rd.Sparse({int(k): x for k, x in sp["map"].items()}, sp["size"]).sparsity() == sp["sparsity"]


def test_fixture_kat_json_dict_and_c():
    import sys
    import os

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import check_fixtures

    kat = _golden("kat.json")
    assert check_fixtures.check(kat)            # dict-based restatement
    rows = [({int(k): v for k, v in m.items()}, y) for m, y in kat["rows"]]
    assert rows == KAT_ROWS                     # the inline copy used by the GPU tests is the same data
    # C oracle: the whole KAT-1 trajectory
    k1 = kat["kat1"]
    data = dsgd_amd.synth.from_rows(6, rows)
    o = orc.Oracle(6, data.row_ptr, data.col, data.val, data.label, k1["lambda"])
    ds = o.dim_sparsity(6)
    np.testing.assert_allclose(ds, [k1["ds"].get(str(j), 0.0) for j in range(7)], rtol=0, atol=1e-12)
    o.set_dim_sparsity(ds)
    w = np.zeros(7)
    for st in k1["steps"]:
        for idx, name in zip(k1["batches"], ("g0", "g1")):
            g = o.gradient(w, np.asarray(idx, dtype=np.int32))
            np.testing.assert_allclose(g, [st[name].get(str(j), 0.0) for j in range(7)], rtol=0, atol=2e-7)  # float32 inputs
        o.sync_step(w, [np.asarray(i, dtype=np.int32) for i in k1["batches"]], k1["lr"])
        np.testing.assert_allclose(w, [st["w"].get(str(j), 0.0) for j in range(7)], rtol=0, atol=2e-7)
        loss, acc, _, _ = o.loss_acc(w, 0, 6)
        assert abs(loss - st["loss"]) < 1e-6 and abs(acc - st["acc"]) < 1e-12


def test_committed_jvm_expectation_is_what_the_oracle_produces_today():
    """tests/golden/oracle_expected_for_jvm/: the Main.scala scenario on a small exported workload, as the JVM should log it
    (see tests/golden/make_oracle_expected_for_jvm.py).  Regenerated here and compared with the committed file, so that whoever
    diffs the JVM's log against it diffs against the oracle the parity tests actually use."""
    import json
    import os
    import sys

    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, golden)
    try:
        import make_oracle_expected_for_jvm as mk
    finally:
        sys.path.remove(golden)
    now = mk.compute()
    exp = json.load(open(os.path.join(golden, "oracle_expected_for_jvm", "expected.json")))
    assert now["config"] == exp["config"] and now["batches"] == exp["batches"] == 6
    for key in ("initial_loss", "initial_accuracy", "final_test_loss", "final_test_accuracy"):
        assert abs(now[key] - exp[key]) <= 1e-12 * max(1.0, abs(exp[key])), key
    assert now["final_weights"].keys() == exp["final_weights"].keys()
    assert max(abs(now["final_weights"][k] - exp["final_weights"][k]) for k in exp["final_weights"]) <= 1e-12
    assert exp["initial_loss"] == 1.0 and exp["initial_accuracy"] == 0.0   # w = 0: every prediction is 0 (SparseSVM.scala:14-18)
    # the exported text loads back to exactly the rows the expectation was computed on
    import dsgd_amd
    from dsgd_amd import rcv1

    back = rcv1.load(os.path.join(golden, "oracle_expected_for_jvm", "data"), full=False)
    data = dsgd_amd.synth.generate(mk.ROWS, seed=0)
    assert back.n_rows == mk.ROWS and (back.col == data.col).all() and (back.label == data.label).all()
    assert np.abs(back.val - data.val).max() == 0.0
