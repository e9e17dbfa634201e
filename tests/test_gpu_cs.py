"""-m gpu: the column-slice kernel (csrc/dsgd_cs.hpp: the reference's own batch sizes as a feature-parallel persistent
kernel -- G workgroups own the columns, one small exchange per step) against the fp64 oracle.

Held to the same statements as every other index-list kernel: ONE step from identical NON-ZERO weights under the derived
per-coordinate bound of oracle/bounds.py with the shift the step used (30 - ceil(log2 batch)) and the stated
1e-5 * max(1, |w|_inf) tolerance, active-row counts equal up to the rows within 1e-5 of the gate; a plan of several steps
equals its steps run one plan at a time BIT FOR BIT; a whole epoch of the reference's configuration (3 workers x batch
100) ends on the oracle's weights; ragged rows (empty, single-entry, long rows that need several slots per slice);
eligibility (beyond 8 hosted workers / 1,024 rows per step, with DSGD_CS=0, with a communicator the row-parallel
kernels run); what comes after a launch (evaluation, per-request steps, range steps) sees its weights."""

import os

import numpy as np
import pytest

import dsgd_amd
import waivers
from conftest import has_gpu
from oracle import bounds as orb
from oracle import oracle as orc
from oracle import ref_dict as rd

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="no gfx950 device")]

LAM = 1e-5
CS = "dsgd_cs_step_kernel"


@pytest.fixture(scope="module", autouse=True)
def _column_slices_on():
    old = os.environ.pop("DSGD_CS", None)
    yield
    if old is not None:
        os.environ["DSGD_CS"] = old


def make_pair(data, n_train):
    o = orc.Oracle(data.dim, data.row_ptr, data.col, data.val, data.label, LAM)
    o.set_dim_sparsity(o.dim_sparsity(n_train))
    eng = dsgd_amd.Engine(data.dim, LAM)
    eng.load_csr(data.row_ptr, data.col, data.val, data.label)
    eng.build_dim_sparsity(n_train)
    return o, eng


def batches(rng, n_train, k, b, steps):
    split = rd.split_vanilla(n_train, k)
    return [[rng.permutation(np.asarray(r))[:b].astype(np.int32) for r in split] for _ in range(steps)]   # Master.scala:184


def nonzero_weights(dim, rng, n=6000):
    w0 = np.zeros(dim + 1, dtype=np.float32)
    hot = rng.choice(np.arange(1, dim + 1), size=n, replace=False)
    w0[hot] = rng.normal(scale=0.05, size=n).astype(np.float32)
    return w0


def plan_step(o, eng, lists, lr, family, kernel=CS):
    w0 = eng.get_weights().astype(np.float64)
    w_ref = w0.copy()
    plan = eng.plan([lists])
    eng.synchronize()
    eng.plan_run(plan, 0, 1, lr)
    st = eng.synchronize()
    plan.destroy()
    assert eng.grad_kernel_name() == kernel
    shift = eng.tuning_info()["fix_shift"]
    o.sync_step(w_ref, lists, lr)
    tol_v, n_near, near_part = orb.list_bound(o, w0, w_ref, lists, lr, shift, parts=True)
    assert st["n_samples"] == sum(len(a) for a in lists)
    assert abs(st["n_active"] - o.last_stats["n_active"]) <= n_near, (st, o.last_stats, n_near)
    w = eng.get_weights()
    ratio, j = orb.worst_ratio(w, w_ref, tol_v)
    assert ratio <= 1.0, "coordinate %d: error %.3g x its derived bound (shift %d, %d rows near the gate)" % (j, ratio, shift, n_near)
    assert np.abs(w - w_ref).max() <= 1e-5 * max(1.0, np.abs(w_ref).max()) or n_near > 0
    tight = st["n_active"] == o.last_stats["n_active"] and orb.worst_ratio(w, w_ref, tol_v - near_part)[0] <= 1.0
    waivers.tight(family + ":gates_as_the_oracle", tight, n_near > 0, "%d rows near the gate" % n_near)
    return shift


@pytest.fixture(scope="module")
def mid():
    data = dsgd_amd.synth.generate(60000, seed=3)
    n_train = 48000
    o, eng = make_pair(data, n_train)
    yield data, n_train, o, eng
    eng.close()


@pytest.mark.parametrize("k,b", [(3, 100), (4, 200), (1, 100), (1, 1), (2, 7), (1, 800), (8, 100), (5, 64)])
def test_column_slices_match_oracle(mid, k, b):
    data, n_train, o, eng = mid
    rng = np.random.default_rng(11 * b + k)
    eng.set_weights(nonzero_weights(data.dim, rng))
    # (the reference's per-sample step, capped: at batch 1 .. 7 the scaled step is 7 .. 50 and two workers' gradients that
    #  nearly cancel then leave fp32 roundings of EACH worker's sum that the derived bound -- written for the net update --
    #  does not price; every index-list kernel shares that arithmetic)
    lr = min(0.5 * 100 / b, 1.0)
    steps = batches(rng, n_train, k, b, 3)
    for lists in steps:
        shift = plan_step(o, eng, lists, lr, "column_slices")
        assert shift == 30 - int(np.ceil(np.log2(b))) if b > 1 else shift == 30
    # a plan of several steps = its steps, one plan each, BIT FOR BIT (the same slices, sums and order either way)
    w_a = eng.get_weights()
    eng.set_weights(nonzero_weights(data.dim, np.random.default_rng(5)))
    w_start = eng.get_weights()
    plan = eng.plan(steps)
    eng.plan_run(plan, 0, 2, lr)
    eng.plan_run(plan, 2, 3, lr)
    st = eng.synchronize()
    plan.destroy()
    assert eng.grad_kernel_name() == CS and st["n_samples"] == 3 * k * b
    w_multi = eng.get_weights()
    eng.set_weights(w_start)
    for lists in steps:
        p1 = eng.plan([lists])
        eng.plan_run(p1, 0, 1, lr)
        eng.synchronize()
        p1.destroy()
    assert np.array_equal(eng.get_weights(), w_multi)
    assert not np.array_equal(w_multi, w_a)


def test_an_epoch_of_the_reference_configuration(mid):
    """application.conf: 3 workers x batch 100, lr 0.5 -- one epoch over 18,519 train rows (62 steps) in ONE launch from
    w = 0 against the oracle stepping over the same lists; evaluation and a per-request step afterwards see the weights."""
    data, _, _, _ = mid
    n_train = 18519
    sub = data.rows(0, 23149)
    o, eng = make_pair(sub, n_train)
    with eng:
        rng = np.random.default_rng(2)
        split = rd.split_vanilla(n_train, 3)
        size = len(split[0])
        steps = []
        for s in range(0, size, 100):
            ls = [rng.permutation(np.asarray(r))[:100].astype(np.int32) for r in split]
            steps.append(ls)
        w_ref = np.zeros(sub.dim + 1)
        exposed = 0
        for ls in steps:
            o.sync_step(w_ref, ls, 0.5)
            exposed += o.last_stats["min_abs_margin"] < 1e-5
        plan = eng.plan(steps)
        eng.plan_run(plan, 0, len(steps), 0.5)
        st = eng.synchronize()
        assert eng.grad_kernel_name() == CS and st["n_samples"] == 300 * len(steps)
        w = eng.get_weights().astype(np.float64)
        err = np.abs(w - w_ref).max()
        waivers.tight("column_slices:epoch", err <= 1e-5 * max(1.0, np.abs(w_ref).max()), exposed > 0,
                      "%d steps with a row within 1e-5 of the gate, err %.3g" % (exposed, err))
        loss, acc, counts = eng.loss_acc(n_train, sub.n_rows)
        l_ref, a_ref, c_ref, _ = o.loss_acc(w, n_train, sub.n_rows)
        assert counts == c_ref and abs(loss - l_ref) < 1e-6
        # the same plan again through the row-parallel kernels ends within the tolerance of the same oracle weights
        plan.destroy()
        st1 = eng.sync_step(steps[0], 0.5)   # a per-request step after the launch starts from its weights
        w1 = w.copy()
        o.sync_step(w1, steps[0], 0.5)
        assert np.abs(eng.get_weights() - w1).max() <= 1e-5 * max(1.0, np.abs(w1).max()) and st1["n_samples"] == 300


def test_ragged_rows_and_long_rows():
    """Empty rows, single-entry rows, a value the Sparse constructor drops, and rows whose entries inside ONE slice need
    several slots (a 1,200-entry row: 150 entries per slice of eight)."""
    base = dsgd_amd.synth.generate(6000, seed=23)
    rng = np.random.default_rng(23)
    row_ptr, col, val = [0], [], []
    for i in range(base.n_rows):
        b, e = int(base.row_ptr[i]), int(base.row_ptr[i + 1])
        kind = rng.integers(0, 10)
        if kind == 0:
            pass
        elif kind == 1:
            col.append(base.col[b]); val.append(np.float32(1.0))
        elif kind == 2:
            keys = np.sort(rng.choice(np.arange(1, base.dim + 1), size=1200, replace=False))
            v = np.abs(rng.normal(size=1200)).astype(np.float32) + 0.1
            v /= np.sqrt((v * v).sum())
            col.extend(keys.tolist()); val.extend(v.tolist())
        else:
            c, v = base.col[b:e], base.val[b:e].copy()
            if kind == 3:
                v[0] = np.float32(1e-25)
            col.extend(c.tolist()); val.extend(v.tolist())
        row_ptr.append(len(col))
    data = dsgd_amd.synth.Csr(base.dim, np.asarray(row_ptr, np.int64), np.asarray(col, np.int32), np.asarray(val, np.float32),
                              base.label.copy())
    n_train = 5000
    o, eng = make_pair(data, n_train)
    with eng:
        eng.set_weights(nonzero_weights(data.dim, rng, 20000))
        for k, b in ((3, 100), (2, 150), (1, 64)):   # (a tenth of the rows needs ten slots per slice)
            for lists in batches(rng, n_train, k, b, 2):
                plan_step(o, eng, lists, 0.5 * 100 / b, "column_slices_ragged")
        # eighty 1,200-entry rows in one step touch more than the 4,096 listed columns of a slice (of its 5,905): such a
        # step is not a small one -- the plan falls back to the row-parallel kernels, same answer
        for lists in batches(rng, n_train, 2, 400, 1):
            plan_step(o, eng, lists, 0.125, "column_slices_ragged", kernel="dsgd_vt_grad_kernel")


def test_slice_major_weights_between_launches_are_seen_by_everything_else(mid):
    """Consecutive column-slice launches keep the weights slice-major on the device (a launch then loads ONE contiguous
    piece per workgroup); the rank-ordered vector is stale meanwhile.  Whatever else reads or writes the weights -- range
    steps, a plan the column slices cannot take, per-request steps, evaluation, get / set -- must see them: every stage of
    an interleaved sequence against the oracle (a stale vector would be off by ~1e-2, the tolerance is 1e-4)."""
    data, n_train, o, eng = mid
    rng = np.random.default_rng(77)
    eng.set_weights(nonzero_weights(data.dim, rng))
    w_ref = eng.get_weights().astype(np.float64)
    steps = batches(rng, n_train, 3, 100, 7)
    big = batches(rng, n_train, 1, 3000, 1)          # 3,000 rows in a step: the row-parallel kernels
    plan, plan_big = eng.plan(steps), eng.plan(big)

    def check(stage):
        w = eng.get_weights().astype(np.float64)
        assert np.abs(w - w_ref).max() <= 1e-4 * max(1.0, np.abs(w_ref).max()), stage

    eng.plan_run(plan, 0, 1, 0.5)
    eng.plan_run(plan, 1, 2, 0.5)                    # stays slice-major: no conversion in between
    eng.synchronize()                                # ... nor here
    eng.plan_run(plan, 2, 3, 0.5)
    assert eng.grad_kernel_name() == CS
    for ls in steps[:3]:
        o.sync_step(w_ref, ls, 0.5)
    check("three one-step launches")
    eng.sync_step_ranges([(0, 5000)], 0.01)          # a range step straight after a launch
    o.sync_step(w_ref, [np.arange(5000, dtype=np.int32)], 0.01)
    check("range step")
    eng.plan_run(plan, 3, 4, 0.5)
    eng.plan_run(plan_big, 0, 1, 0.02)               # another plan, row-parallel, straight after
    assert eng.grad_kernel_name() != CS
    o.sync_step(w_ref, steps[3], 0.5)
    o.sync_step(w_ref, big[0], 0.02)
    check("row-parallel plan behind a column-slice launch")
    eng.plan_run(plan, 4, 5, 0.5)
    o.sync_step(w_ref, steps[4], 0.5)
    loss, acc, counts = eng.loss_acc(n_train, data.n_rows)   # evaluation straight after a launch
    l_ref, a_ref, c_ref, _ = o.loss_acc(w_ref, n_train, data.n_rows)
    assert abs(loss - l_ref) < 1e-4 and abs(acc - a_ref) < 2e-3
    eng.sync_step(steps[5], 0.5)                     # a per-request step
    o.sync_step(w_ref, steps[5], 0.5)
    eng.plan_run(plan, 6, 7, 0.5)
    o.sync_step(w_ref, steps[6], 0.5)
    check("per-request step between launches")
    w_new = nonzero_weights(data.dim, np.random.default_rng(78))
    eng.set_weights(w_new)                           # set_weights while the slice-major copy is live
    eng.plan_run(plan, 0, 1, 0.5)
    w_ref = w_new.astype(np.float64)
    o.sync_step(w_ref, steps[0], 0.5)
    check("set_weights behind a launch")
    plan.destroy()
    plan_big.destroy()


def test_eligibility_and_fallbacks(monkeypatch):
    data = dsgd_amd.synth.generate(30000, seed=9)
    n_train = 24000
    o, eng = make_pair(data, n_train)
    rng = np.random.default_rng(9)
    with eng:
        eng.set_weights(nonzero_weights(data.dim, rng))
        # beyond CS_MAX_K workers / 1,024 rows per step: the row-parallel kernels, same statements
        plan_step(o, eng, batches(rng, n_train, 9, 50, 1)[0], 1.0, "column_slices_fallback", kernel="dsgd_vt_grad_kernel")
        plan_step(o, eng, batches(rng, n_train, 2, 600, 1)[0], 0.1, "column_slices_fallback", kernel="dsgd_vt_grad_kernel")
        plan_step(o, eng, batches(rng, n_train, 3, 100, 1)[0], 0.5, "column_slices_fallback")
        # a communicator (even of one rank) puts a collective inside the step: not this kernel's business
        eng.comm_init(dsgd_amd.Engine.comm_unique_id(), 1, 0)
        plan_step(o, eng, batches(rng, n_train, 3, 100, 1)[0], 0.5, "column_slices_fallback", kernel="dsgd_vt_grad_kernel")
    monkeypatch.setenv("DSGD_CS", "0")
    o, eng = make_pair(data, n_train)
    with eng:
        eng.set_weights(nonzero_weights(data.dim, rng))
        plan_step(o, eng, batches(rng, n_train, 3, 100, 1)[0], 0.5, "column_slices_fallback", kernel="dsgd_vt_grad_kernel")
    monkeypatch.setenv("DSGD_CS", "1")
    monkeypatch.setenv("DSGD_CS_G", "16")   # sixteen slices for few workers: the same statements
    o, eng = make_pair(data, n_train)
    with eng:
        eng.set_weights(nonzero_weights(data.dim, rng))
        plan_step(o, eng, batches(rng, n_train, 3, 100, 1)[0], 0.5, "column_slices_g16")
        plan_step(o, eng, batches(rng, n_train, 1, 100, 1)[0], 0.5, "column_slices_g16")


def test_wide_and_narrow_models(monkeypatch):
    """Models that are not RCV1-shaped: 200,000 features (the slices still fit LDS for few workers; beyond that the kernel
    declines) and a 300-feature toy (fewer columns per slice than lanes)."""
    for dim, k in ((200000, 1), (300, 3)):
        data = dsgd_amd.synth.generate(8000, seed=31, dim=dim)
        n_train = 6400
        o, eng = make_pair(data, n_train)
        rng = np.random.default_rng(dim)
        with eng:
            eng.set_weights(nonzero_weights(data.dim, rng, min(4000, dim // 2)))
            for lists in batches(rng, n_train, k, 100, 2):
                plan_step(o, eng, lists, 0.5, "column_slices_dims")
