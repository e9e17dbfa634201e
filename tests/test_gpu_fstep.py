"""Row chunks (-m gpu): the whole gradient of a row range in ONE launch (csrc/dsgd_fstep.hpp, dsgd_fstep_kernel) against
the fp64 CPU oracle, through the C ABI (dsgd_sync_step_ranges) -- the step `Master.fit`'s batch closure takes when a
batch is a worker's whole split (core/Master.scala:179-199 with batch-size >= the split), at the reference's own data-set
sizes (N = 804,414: DatasetTests.scala:18; N = 23,149: application.conf:24) and what one GPU of eight holds of the first.

Held to the DERIVED per-coordinate bound of oracle/bounds.py (tests/test_gpu_parity.py `ranged_step`): the chunked
launch accumulates the same fixed-point integers as the three streaming launches, so -- at the SAME shift -- a step
through either path ends on bit-identical weights; that is asserted too (the shift capped to 15 on both sides).
"""

import numpy as np
import pytest

import dsgd_amd
import waivers
from conftest import has_gpu
from test_gpu_parity import GATE_EPS, make_pair, ragged_data, ranged_step

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="no gfx950 device")]

FSTEP = "dsgd_fstep_kernel"


@pytest.fixture(autouse=True)
def _no_column_lists(monkeypatch):
    """Ranges below 65,536 rows are the column lists' in the product (tests/test_gpu_tcol.py); this module forces the
    chunked launch onto them."""
    monkeypatch.setenv("DSGD_TCOL", "0")


def some_weights(dim, seed, n=8000, scale=0.05):
    rng = np.random.default_rng(seed)
    w0 = np.zeros(dim + 1, dtype=np.float32)
    hot = rng.choice(np.arange(1, dim + 1), size=n, replace=False)
    w0[hot] = rng.normal(scale=scale, size=n).astype(np.float32)
    return w0


def with_long_rows(base, seed, every=97, n_entries=3000):
    """every 97th row replaced by one of 3,000 entries (its hot part alone exceeds a wave tile's 504 slots: the long-row list)"""
    rng = np.random.default_rng(seed)
    row_ptr, col, val = [0], [], []
    for i in range(base.n_rows):
        b, e = int(base.row_ptr[i]), int(base.row_ptr[i + 1])
        if i % every == 5:
            keys = np.sort(rng.choice(np.arange(1, base.dim + 1), size=n_entries, replace=False))
            v = np.abs(rng.normal(size=n_entries)).astype(np.float32) + 0.1
            v /= np.sqrt((v * v).sum())
            col.append(keys.astype(np.int32)); val.append(v)
        else:
            col.append(base.col[b:e]); val.append(base.val[b:e])
        row_ptr.append(row_ptr[-1] + len(col[-1]))
    return dsgd_amd.synth.Csr(base.dim, np.asarray(row_ptr, np.int64), np.concatenate(col).astype(np.int32),
                              np.concatenate(val).astype(np.float32), base.label.copy())


@pytest.mark.parametrize("n_rows", [23149, 125000, 804414])
def test_row_chunks_match_oracle(monkeypatch, n_rows):
    """The reference's sizes (N = 23,149 forced onto the chunked launch: the product keeps the row-wise kernel below
    65,536 rows; 100,000 train rows = one GPU of eight's share of RCV1; N = 804,414 = RCV1): whole-split steps from non-zero
    weights, one / two / three workers (SplitStrategy.vanilla's contiguous ranges), under the derived bound; tallies of
    the test rows exact."""
    for k in ("DSGD_FSTEP", "DSGD_FSTEP_MIN", "DSGD_FSTEP_MAX", "DSGD_FSTEP_ROWS", "DSGD_STREAM_MIN"):
        monkeypatch.delenv(k, raising=False)
    if n_rows < 100000:
        monkeypatch.setenv("DSGD_FSTEP_MIN", "4096")
    data = dsgd_amd.synth.generate(n_rows, seed=43)
    n_train = int(n_rows * 0.8)
    o, eng = make_pair(data, 1e-5, n_train)
    with eng:
        eng.set_weights(some_weights(data.dim, 43))
        third = n_train // 3
        for ranges in ([(0, n_train)], [(0, n_train // 2), (n_train // 2, n_train)], [(0, third), (third, 2 * third + 7), (2 * third + 7, n_train)],
                       [(0, n_train)]):
            ranged_step(o, eng, ranges, 0.5 * 100 / n_train * len(ranges))
            assert eng.grad_kernel_name() == FSTEP
        loss, acc, counts = eng.loss_acc(n_train, n_rows)
        l_ref, a_ref, c_ref, mam = o.loss_acc(eng.get_weights().astype(np.float64), n_train, n_rows)
        assert abs(loss - l_ref) <= 1e-6
        waivers.tight("row_chunks:tallies", counts == c_ref, mam < GATE_EPS, "margin %.2g" % mam)
        # bit-reproducible, whatever the scheduling of the workgroups: the same step from the same weights twice
        w0 = eng.get_weights()
        eng.sync_step_ranges([(0, n_train)], 0.5 * 100 / n_train)
        w1 = eng.get_weights()
        eng.set_weights(w0)
        eng.sync_step_ranges([(0, n_train)], 0.5 * 100 / n_train)
        np.testing.assert_array_equal(eng.get_weights(), w1)


@pytest.mark.parametrize("rows_per_chunk", ["64", "300", "4096"])
def test_row_chunks_equal_the_three_launches_bit_for_bit(monkeypatch, rows_per_chunk):
    """Same fixed-point grid (shift capped to 15 on both sides) => the integer sums do not care which workgroup's partial
    a contribution lands in: the chunked launch and the three streaming launches end on the SAME bits, for every chunk
    size (chunks of 64 rows: most tiles are cut short; 4,096: few workgroups)."""
    monkeypatch.setenv("DSGD_FIX_SHIFT", "15")
    monkeypatch.setenv("DSGD_STREAM_MIN", "8192")
    monkeypatch.setenv("DSGD_FSTEP_ROWS", rows_per_chunk)
    monkeypatch.setenv("DSGD_FSTEP_MIN", "8192")
    n_rows, n_train = 60000, 50000
    data = dsgd_amd.synth.generate(n_rows, seed=47)
    w0 = some_weights(data.dim, 47)
    res = {}
    for fstep in ("1", "0"):
        monkeypatch.setenv("DSGD_FSTEP", fstep)
        with dsgd_amd.Engine(data.dim, 1e-5) as eng:
            eng.load_csr(data.row_ptr, data.col, data.val, data.label)
            eng.build_dim_sparsity(n_train)
            eng.set_weights(w0)
            acts = []
            for ranges in ([(0, n_train)], [(0, 20001), (20001, n_train)], [(5, 17000), (17000, 33000), (33000, 49999)]):
                st = eng.sync_step_ranges(ranges, 0.5 * 100 / n_train * len(ranges))
                acts.append(st["n_active"])
                assert eng.grad_kernel_name() == (FSTEP if fstep == "1" else "dsgd_wseg_kernel<true>")
                assert eng.tuning_info()["fix_shift"] == 15
            res[fstep] = (eng.get_weights(), acts)
    assert res["1"][1] == res["0"][1]
    np.testing.assert_array_equal(res["1"][0], res["0"][0])


def test_row_chunks_on_ragged_rows(monkeypatch):
    """Empty rows, rows of one entry, rows longer than a wave tile (the long-row list), 1e-25 entries: chunk boundaries
    fall between all of them (chunks of 50 rows)."""
    monkeypatch.setenv("DSGD_FSTEP_MIN", "1000")
    monkeypatch.setenv("DSGD_FSTEP_ROWS", "50")
    data = with_long_rows(ragged_data(23, n_rows=24000), 23)
    n_train = 20000
    o, eng = make_pair(data, 1e-5, n_train)
    with eng:
        lr = 0.5 * 100 / 10000
        for ranges in ([(0, n_train)], [(0, 10000), (10000, n_train)], [(100, 2500), (2500, 4900)], [(0, n_train)]):
            ranged_step(o, eng, ranges, lr)
            assert eng.grad_kernel_name() == FSTEP
        eng.set_weights(np.zeros(data.dim + 1, dtype=np.float32))
        st = eng.sync_step_ranges([(0, n_train)], 0.0)
        assert st["n_active"] == n_train and eng.grad_kernel_name() == FSTEP


def test_layouts_the_chunked_launch_declines(monkeypatch):
    """More cold columns than the LDS tile holds (D = 70,000), or no cold stream at all (D = 3,000): the three streaming
    launches take the range, as before -- and ranges above DSGD_FSTEP_MAX or below DSGD_FSTEP_MIN."""
    monkeypatch.setenv("DSGD_STREAM_MIN", "8192")
    monkeypatch.setenv("DSGD_FSTEP_MIN", "8192")
    for dim in (70000, 3000):
        data = dsgd_amd.synth.generate(30000, seed=5, dim=dim)
        o, eng = make_pair(data, 1e-5, 25000)
        with eng:
            ranged_step(o, eng, [(0, 25000)], 0.5 * 100 / 25000)
            assert eng.grad_kernel_name() != FSTEP
    monkeypatch.setenv("DSGD_FSTEP_MAX", "20000")
    monkeypatch.setenv("DSGD_FSTEP_MIN", "10000")
    data = dsgd_amd.synth.generate(30000, seed=5)
    o, eng = make_pair(data, 1e-5, 25000)
    with eng:
        for ranges, chunked in (([(0, 25000)], False), ([(0, 15000)], True), ([(0, 9000)], False), ([(0, 7000), (7000, 15000)], True)):
            ranged_step(o, eng, ranges, 0.5 * 100 / 25000)
            assert (eng.grad_kernel_name() == FSTEP) == chunked, ranges


def test_more_configurations_than_the_cache_holds(monkeypatch):
    """Ten different (ranges) configurations alternate over a cache of eight: evicted layouts are rebuilt, results stay
    those of a fresh layout (each configuration's step repeated from the same weights gives the same bits)."""
    monkeypatch.setenv("DSGD_FSTEP_MIN", "1000")
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
            assert eng.grad_kernel_name() == FSTEP
            first.append(eng.get_weights())
        for rg, w1 in list(zip(cfgs, first))[::-1]:
            eng.set_weights(w0)
            eng.sync_step_ranges(rg, 1e-3)
            np.testing.assert_array_equal(eng.get_weights(), w1)


def test_chunks_recut_by_measured_time_change_no_bit(monkeypatch):
    """Round 6: after 24 launches of a configuration the workgroups' own durations re-cut its chunks (twice at most;
    dsgd_tuning_info slot 6).  Rows move between workgroups -- integer partial sums do not care: a step from the same
    weights gives the same bits before the first re-cut, after it, and with the re-cut switched off."""
    monkeypatch.setenv("DSGD_TCOL", "0")
    monkeypatch.setenv("DSGD_FSTEP_MIN", "8192")
    data = dsgd_amd.synth.generate(150000, seed=15)
    n_train = 120000
    w0 = some_weights(data.dim, 15)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("DSGD_FSTEP_REBALANCE", mode)
        with dsgd_amd.Engine(data.dim, 1e-5) as eng:
            eng.load_csr(data.row_ptr, data.col, data.val, data.label)
            eng.build_dim_sparsity(n_train)
            seen = []
            for rep in range(5):
                eng.set_weights(w0)
                eng.sync_step_ranges([(0, n_train)], 1e-3)
                assert eng.grad_kernel_name() == FSTEP
                seen.append((eng.tuning_info()["fstep_rebalances"], eng.get_weights()))
                for _ in range(30):
                    eng.sync_step_ranges([(0, n_train)], 1e-6, asynchronous=True)
                eng.synchronize()
            outs[mode] = seen
    assert [r for r, _ in outs["0"]] == [0] * 5
    assert outs["1"][0][0] == 0 and outs["1"][-1][0] >= 1, [r for r, _ in outs["1"]]
    for _, w in outs["1"] + outs["0"]:
        assert np.array_equal(w, outs["1"][0][1])
