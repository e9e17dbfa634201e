"""Column lists (-m gpu): whole-split steps of 10^3 .. 10^5 rows as dot + column-wise gradient + reduce
(csrc/dsgd_tcol.hpp: dsgd_tc_dot_kernel, dsgd_tc_grad_kernel) against the fp64 CPU oracle, through the C ABI
(dsgd_sync_step_ranges) -- the step `Master.fit`'s batch closure takes when a batch is a worker's whole split
(core/Master.scala:179-199 with batch-size >= the split) at the reference's small data set (N = 23,149: application.conf:24)
and at what one GPU of eight holds of RCV1 (core/ml/SplitStrategy.scala:13-14).

Held to the DERIVED per-coordinate bound of oracle/bounds.py (tests/test_gpu_parity.py `ranged_step`).  The column lists
accumulate the same fixed-point integers as every row-parallel kernel, so -- at the SAME shift and with the same gate
decisions -- a step through them and one through the chunked launch end on bit-identical weights; asserted too.
"""

import numpy as np
import pytest

import dsgd_amd
import waivers
from conftest import has_gpu
from test_gpu_fstep import some_weights, with_long_rows
from test_gpu_parity import GATE_EPS, make_pair, ragged_data, ranged_step

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="no gfx950 device")]

TCOL = "dsgd_tc_grad_kernel"
KNOBS = ("DSGD_TCOL", "DSGD_TCOL_MIN", "DSGD_TCOL_MAX", "DSGD_TCOL_MAX_NNZ", "DSGD_TCOL_SHARE", "DSGD_FSTEP", "DSGD_FSTEP_MIN", "DSGD_FSTEP_MAX",
         "DSGD_FSTEP_ROWS", "DSGD_STREAM_MIN", "DSGD_FIX_SHIFT")


def clean(monkeypatch):
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)


@pytest.mark.parametrize("n_rows", [23149, 100552])
def test_column_lists_match_oracle(monkeypatch, n_rows):
    """N = 23,149 (the product's choice) and 80,441 train rows (one GPU of eight's share of RCV1: until round 6 the product's
    choice too -- now row chunks from 4.5 M non-zeros on, tests/test_gpu_dispatch.py; the column lists are asked for by
    DSGD_TCOL_MAX_NNZ): whole-split steps from non-zero weights, one / two / three workers (SplitStrategy.vanilla's contiguous
    ranges, and uneven ones), under the derived bound; tallies of the test rows exact; the same step twice = the same bits."""
    clean(monkeypatch)
    monkeypatch.setenv("DSGD_TCOL_MAX_NNZ", "1000000000")
    data = dsgd_amd.synth.generate(n_rows, seed=53)
    n_train = int(n_rows * 0.8)
    o, eng = make_pair(data, 1e-5, n_train)
    with eng:
        eng.set_weights(some_weights(data.dim, 53))
        third = n_train // 3
        for ranges in ([(0, n_train)], [(0, n_train // 2), (n_train // 2, n_train)], [(0, third), (third, 2 * third + 7), (2 * third + 7, n_train)],
                       [(0, n_train)]):
            ranged_step(o, eng, ranges, 0.5 * 100 / n_train * len(ranges))
            assert eng.grad_kernel_name() == TCOL
            assert eng.tuning_info()["fix_shift"] == 21
        loss, acc, counts = eng.loss_acc(n_train, n_rows)
        l_ref, a_ref, c_ref, mam = o.loss_acc(eng.get_weights().astype(np.float64), n_train, n_rows)
        assert abs(loss - l_ref) <= 1e-6
        waivers.tight("column_lists:tallies", counts == c_ref, mam < GATE_EPS, "margin %.2g" % mam)
        w0 = eng.get_weights()
        eng.sync_step_ranges([(0, n_train)], 0.5 * 100 / n_train)
        w1 = eng.get_weights()
        eng.set_weights(w0)
        eng.sync_step_ranges([(0, n_train)], 0.5 * 100 / n_train)
        np.testing.assert_array_equal(eng.get_weights(), w1)


def test_column_lists_from_zero_weights_and_a_trajectory(monkeypatch):
    """From w = 0 every row is ON the gate (y (x . w) = 0 >= 0: all active, core/ml/SparseSVM.scala:27-28); then ten
    whole-split steps in a row with the engine's weights carried over, each under the derived bound."""
    clean(monkeypatch)
    n_rows = 23149
    data = dsgd_amd.synth.generate(n_rows, seed=59)
    n_train = int(n_rows * 0.8)
    o, eng = make_pair(data, 1e-5, n_train)
    with eng:
        st = eng.sync_step_ranges([(0, n_train)], 0.0)
        assert st["n_active"] == n_train and eng.grad_kernel_name() == TCOL
        for i in range(10):
            ranges = [(0, n_train)] if i % 2 == 0 else [(0, 6173), (6173, 12346), (12346, n_train)]
            ranged_step(o, eng, ranges, 0.5 * 100 / n_train * len(ranges))
            assert eng.grad_kernel_name() == TCOL


@pytest.mark.parametrize("share", ["0", "64", "1000", "8192"])
def test_column_lists_equal_the_chunked_launch_bit_for_bit(monkeypatch, share):
    """Same fixed-point grid (shift 21 on both sides, hot and cold columns alike: chunks of 512 rows leave the chunked
    launch at the cap too) => the integer sums do not care whether they were formed row by row or column by column, nor
    where a share ends (shares of 64 entries: most columns are cut; 8,192: the LDS table at its largest): the same bits as
    the chunked launch, provided both gate the same rows (the two kernels add a row's products in different orders: a row
    within rounding of the gate may differ -- counted, not hidden)."""
    clean(monkeypatch)
    monkeypatch.setenv("DSGD_TCOL_SHARE", share)
    n_rows, n_train = 60000, 50000
    data = dsgd_amd.synth.generate(n_rows, seed=61)
    w0 = some_weights(data.dim, 61)
    res = {}
    for tcol in ("1", "0"):
        monkeypatch.setenv("DSGD_TCOL", tcol)
        monkeypatch.setenv("DSGD_FSTEP_MIN", "8192")
        with dsgd_amd.Engine(data.dim, 1e-5) as eng:
            eng.load_csr(data.row_ptr, data.col, data.val, data.label)
            eng.build_dim_sparsity(n_train)
            eng.set_weights(w0)
            acts = []
            for ranges in ([(0, n_train)], [(0, 20001), (20001, n_train)], [(5, 17000), (17000, 33000), (33000, 49999)]):
                st = eng.sync_step_ranges(ranges, 0.5 * 100 / n_train * len(ranges))
                acts.append(st["n_active"])
                assert eng.grad_kernel_name() == (TCOL if tcol == "1" else "dsgd_fstep_kernel")
                assert eng.tuning_info()["fix_shift"] == 21
            res[tcol] = (eng.get_weights(), acts)
    same_gates = res["1"][1] == res["0"][1]
    same_bits = bool(np.array_equal(res["1"][0], res["0"][0]))
    waivers.tight("column_lists:bits_of_the_chunked_launch", same_gates and same_bits, not same_gates,
                  "active rows %s vs %s" % (res["1"][1], res["0"][1]))


def test_column_lists_on_ragged_rows(monkeypatch):
    """Empty rows, rows of one entry, rows of 3,000 entries (many rounds of a 16-lane group), 1e-25 entries (zero on the
    fixed-point grid: outside the support, math/Sparse.scala:108-118); sub-ranges that do not start at row 0; overlapping
    ranges of two workers (the same row in two lists: two entries, two sums)."""
    clean(monkeypatch)
    data = with_long_rows(ragged_data(29, n_rows=24000), 29)
    n_train = 20000
    o, eng = make_pair(data, 1e-5, n_train)
    with eng:
        lr = 0.5 * 100 / 10000
        for ranges in ([(0, n_train)], [(0, 10000), (10000, n_train)], [(100, 2500), (2500, 4900)], [(0, 12000), (8000, n_train)], [(0, n_train)]):
            ranged_step(o, eng, ranges, lr)
            assert eng.grad_kernel_name() == TCOL
        eng.set_weights(np.zeros(data.dim + 1, dtype=np.float32))
        st = eng.sync_step_ranges([(0, n_train)], 0.0)
        assert st["n_active"] == n_train and eng.grad_kernel_name() == TCOL


def test_ranges_the_column_lists_decline(monkeypatch):
    """Below DSGD_TCOL_MIN and above DSGD_TCOL_MAX the other kernels take the range; DSGD_TCOL=0 switches the path off."""
    clean(monkeypatch)
    data = dsgd_amd.synth.generate(30000, seed=5)
    o, eng = make_pair(data, 1e-5, 25000)
    with eng:
        for ranges, tc in (([(0, 25000)], True), ([(0, 300)], False), ([(0, 300), (300, 600)], True), ([(0, 512)], True)):
            ranged_step(o, eng, ranges, 0.5 * 100 / 25000)
            assert (eng.grad_kernel_name() == TCOL) == tc, ranges
    monkeypatch.setenv("DSGD_TCOL_MAX", "20000")
    o, eng = make_pair(data, 1e-5, 25000)
    with eng:
        ranged_step(o, eng, [(0, 25000)], 0.5 * 100 / 25000)
        assert eng.grad_kernel_name() == "dsgd_mb_grad_kernel"
        ranged_step(o, eng, [(0, 15000)], 0.5 * 100 / 25000)
        assert eng.grad_kernel_name() == TCOL
    monkeypatch.setenv("DSGD_TCOL", "0")
    o, eng = make_pair(data, 1e-5, 25000)
    with eng:
        ranged_step(o, eng, [(0, 15000)], 0.5 * 100 / 25000)
        assert eng.grad_kernel_name() == "dsgd_mb_grad_kernel"


def test_more_configurations_than_the_cache_holds(monkeypatch):
    """Ten different (ranges) configurations alternate over a cache of eight: evicted layouts are rebuilt, results stay
    those of a fresh layout; a plan, per-request steps and an evaluation pass in between see the same weights."""
    clean(monkeypatch)
    data = dsgd_amd.synth.generate(30000, seed=9)
    n_train = 25000
    w0 = some_weights(data.dim, 9)
    with dsgd_amd.Engine(data.dim, 1e-5) as eng:
        eng.load_csr(data.row_ptr, data.col, data.val, data.label)
        eng.build_dim_sparsity(n_train)
        cfgs = [[(0, 12000 + 1000 * i)] for i in range(10)]
        first = []
        for rg in cfgs:
            eng.set_weights(w0)
            eng.sync_step_ranges(rg, 1e-3)
            assert eng.grad_kernel_name() == TCOL
            first.append(eng.get_weights())
        rng = np.random.default_rng(9)
        lists = [np.sort(rng.choice(n_train, size=100, replace=False)).astype(np.int32) for _ in range(3)]
        eng.sync_step(lists, 0.1)
        eng.loss_acc(n_train, 30000)
        for rg, w1 in list(zip(cfgs, first))[::-1]:
            eng.set_weights(w0)
            eng.sync_step_ranges(rg, 1e-3)
            np.testing.assert_array_equal(eng.get_weights(), w1)


def test_asynchronous_range_steps_and_a_communicator_of_one(monkeypatch):
    """dsgd_sync_step_ranges_async + dsgd_synchronize (the bench's loop) report the same active rows and end on the same
    bits as the blocking calls; with a communicator of one attached (the all-reduce between reduce and update: two launches
    around it) the weights are those of the engine without one."""
    clean(monkeypatch)
    data = dsgd_amd.synth.generate(23149, seed=67)
    n_train = 18519
    w0 = some_weights(data.dim, 67)
    lr = 0.5 * 100 / n_train
    out = []
    for mode in ("blocking", "async", "comm"):
        with dsgd_amd.Engine(data.dim, 1e-5) as eng:
            eng.load_csr(data.row_ptr, data.col, data.val, data.label)
            if mode == "comm":
                eng.comm_init(dsgd_amd.Engine.comm_unique_id(), 1, 0)
            eng.build_dim_sparsity(n_train)
            eng.set_weights(w0)
            act = 0
            if mode == "async":
                for _ in range(5):
                    eng.sync_step_ranges([(0, n_train)], lr, asynchronous=True)
                act = eng.synchronize()["n_active"]
            else:
                for _ in range(5):
                    act += eng.sync_step_ranges([(0, n_train)], lr)["n_active"]
            assert eng.grad_kernel_name() == TCOL
            out.append((eng.get_weights(), act))
    assert out[0][1] == out[1][1] == out[2][1]
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][0], out[2][0])


def test_a_caller_that_never_repeats_its_ranges_is_left_alone(monkeypatch):
    """A layout costs milliseconds, a step 20 us: thirteen configurations in a row without one being used again (more than
    the cache of eight could ever serve) and the context's ranges go back to the row-wise kernel -- every step, before and
    after, under the derived bound."""
    clean(monkeypatch)
    data = dsgd_amd.synth.generate(30000, seed=11)
    o, eng = make_pair(data, 1e-5, 25000)
    with eng:
        eng.set_weights(some_weights(data.dim, 11))
        names = []
        for i in range(16):
            ranged_step(o, eng, [(100 * i, 100 * i + 4000)], 0.5 * 100 / 4000)
            names.append(eng.grad_kernel_name())
        assert names[:12] == [TCOL] * 12 and names[-1] == "dsgd_mb_grad_kernel", names
        # a configuration that IS repeated keeps its column lists in a fresh context
    o, eng = make_pair(data, 1e-5, 25000)
    with eng:
        for _ in range(20):
            eng.sync_step_ranges([(0, 4000)], 1e-3, asynchronous=True)
        eng.synchronize()
        assert eng.grad_kernel_name() == TCOL
