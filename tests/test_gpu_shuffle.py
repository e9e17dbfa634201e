"""-m gpu: an epoch's index lists DRAWN BY THE DEVICE (dsgd_plan_create_from_seed, csrc/dsgd_shuffle.hpp) are the
reference's stream draw for draw.

core/Master.scala:184 reshuffles every worker's whole split for EVERY batch (scala.util.Random.shuffle over
java.util.Random, seeded 0 at Main.scala:32) and slices it.  csrc/jrand.c reproduces that stream on the host
(tests/test_host_mirror.py pins it against the pure-Python restatement of the JVM's generator); the device form -- raw
stream scanned for rejection candidates, the candidates walked on the host, every list traced backwards through its
Fisher-Yates by one workgroup -- must give the same lists entry for entry and leave the generator in the same state,
rejections included (at N = 804,414 every shuffle has ~10 of them)."""

import time

import numpy as np
import pytest

import dsgd_amd
from dsgd_amd import host
from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="no gfx950 device")]


def engine(n_rows, seed=5, ds_rows=None):
    data = dsgd_amd.synth.generate(n_rows, seed=seed)
    eng = dsgd_amd.Engine(data.dim, 1e-5)
    eng.load_csr(data.row_ptr, data.col, data.val, data.label)
    if ds_rows:
        eng.build_dim_sparsity(ds_rows)
    return data, eng


def both(eng, rnd_state, split, max_samples, batch):
    rnd = host.JavaRandom(0)
    rnd.seed = rnd_state
    idx_h, offs_h, n_h = host.epoch_lists(rnd, split, max_samples, batch, native=True)
    plan, n_d, state_d, draws = eng.plan_from_seed(rnd_state, split, max_samples, batch)
    return (idx_h, offs_h, n_h, rnd.seed), (plan, n_d, state_d, draws)


@pytest.mark.parametrize("n_train,k,batch", [(18519, 3, 100), (18519, 1, 100), (18519, 4, 200), (5000, 3, 64), (4097, 2, 1024)])
def test_device_drawn_lists_are_the_hosts_over_three_epochs(n_train, k, batch):
    data, eng = engine(n_train + 100)
    with eng:
        split = host.split_vanilla(n_train, k)
        max_samples = max(len(r) for r in split)
        state = host.JavaRandom(0).seed                  # Main.scala:32
        for epoch in range(3):
            (idx_h, offs_h, n_h, state_h), (plan, n_d, state_d, draws) = both(eng, state, split, max_samples, batch)
            assert n_d == n_h >= len(range(0, min(len(r) for r in split), batch)) and plan is not None
            idx_d, offs_d = eng.plan_lists(plan)
            assert np.array_equal(offs_d, offs_h), "epoch %d: list lengths differ" % epoch
            assert np.array_equal(idx_d, idx_h), "epoch %d: first differing entry %d" % (epoch, int(np.flatnonzero(idx_d != idx_h)[0]))
            assert state_d == state_h and draws >= n_h * sum(len(r) - 1 for r in split)
            plan.destroy()
            state = state_d


def test_short_splits_single_rows_and_limits():
    data, eng = engine(3000)
    with eng:
        # the epoch ends in front of the first batch that hands a worker an empty slice (math/Vec.scala:129 throws there)
        split = [range(0, 1), range(1, 6), range(6, 400)]
        (idx_h, offs_h, n_h, state_h), (plan, n_d, state_d, _) = both(eng, host.JavaRandom(7).seed, split, 394, 2)
        assert n_h == n_d == 1 and state_d == state_h
        idx_d, offs_d = eng.plan_lists(plan)
        assert np.array_equal(idx_d, idx_h) and np.array_equal(offs_d, offs_h) and idx_d[0] == 0
        plan.destroy()
        # the last batch of a split that is not a multiple of the batch size is shorter
        split = [range(0, 250), range(250, 500)]
        (idx_h, offs_h, n_h, state_h), (plan, n_d, state_d, _) = both(eng, 12345, split, 250, 100)
        assert n_h == n_d == 3 and state_d == state_h
        idx_d, offs_d = eng.plan_lists(plan)
        assert np.array_equal(idx_d, idx_h) and np.array_equal(offs_d, offs_h) and offs_d[-1] == 500
        assert sorted(idx_d[offs_d[0]:offs_d[1]].tolist() + idx_d[offs_d[2]:offs_d[3]].tolist()) != list(range(200))   # (each batch its own shuffle)
        plan.destroy()
        # outside the device form: the caller draws on the host
        with pytest.raises(dsgd_amd.DsgdError) as ei:
            eng.plan_from_seed(1, [range(0, 3000)], 3000, 2000)
        assert ei.value.code == -7
        with pytest.raises(dsgd_amd.DsgdIndexError):
            eng.plan_from_seed(1, [range(0, 4000)], 4000, 100)
        with pytest.raises(dsgd_amd.DsgdInvalidArgument):
            eng.plan_from_seed(1, [range(5, 5)], 10, 100)


def test_an_epoch_of_rcv1_full_is_drawn_on_the_device_draw_for_draw():
    """N = 804,414 (DatasetTests.scala:18): 643,531 training rows, 3 workers x batch 100 (application.conf:15,27): 2,146
    batches x 3 shuffles of 214,510 rows = 1.38 G draws, ~70 K of them rejections that shift every later draw.  The lists, the
    generator's state -- and what they are for: the plan runs to the same weights as one made from the host's lists."""
    n_train = 643531
    data, eng = engine(n_train + 64, seed=0, ds_rows=n_train)
    with eng:
        split = host.split_vanilla(n_train, 3)
        max_samples = max(len(r) for r in split)
        state = host.JavaRandom(0).seed
        t0 = time.perf_counter()
        rnd = host.JavaRandom(0)
        idx_h, offs_h, n_h = host.epoch_lists(rnd, split, max_samples, 100, native=True)
        t_host = time.perf_counter() - t0
        t0 = time.perf_counter()
        plan, n_d, state_d, draws = eng.plan_from_seed(state, split, max_samples, 100)
        t_dev = time.perf_counter() - t0
        t0 = time.perf_counter()
        plan2, n_d2, state_d2, draws2 = eng.plan_from_seed(state_d, split, max_samples, 100)   # (second epoch: blocks from the cache)
        t_dev2 = time.perf_counter() - t0
        idx_d, offs_d = eng.plan_lists(plan)
        print("one epoch of Master.fit at N = 804,414, 3 x 100: %d steps, %d draws (%d rejections); lists on the host (csrc/jrand.c, %d "
              "threads) %.1f ms, on the device incl. the plan's layout %.1f ms (next epoch %.1f ms)" % (
                  n_d, draws, draws - n_d * sum(len(r) - 1 for r in split), host.host_threads() if hasattr(host, "host_threads") else -1,
                  1e3 * t_host, 1e3 * t_dev, 1e3 * t_dev2))
        assert n_d == n_h == 2146 and state_d == rnd.seed
        assert np.array_equal(offs_d, offs_h) and np.array_equal(idx_d, idx_h)
        assert draws - n_d * sum(len(r) - 1 for r in split) > 10000          # (the rejections were there to be handled)
        idx_h2, offs_h2, _ = host.epoch_lists(rnd, split, max_samples, 100, native=True)
        idx_d2, _ = eng.plan_lists(plan2)
        assert np.array_equal(idx_d2, idx_h2) and state_d2 == rnd.seed
        plan2.destroy()
        # the same plan from the host's lists: the same kernel, the same bits
        assert plan.info()["kind"] == "column_slices"
        eng.set_weights(np.zeros(data.dim + 1, dtype=np.float32))
        eng.plan_run(plan, 0, 200, 0.5)
        eng.synchronize()
        w_dev = eng.get_weights()
        plan.destroy()
        plan_h = eng.plan_flat(idx_h, offs_h, n_h, 3)
        eng.set_weights(np.zeros(data.dim + 1, dtype=np.float32))
        eng.plan_run(plan_h, 0, 200, 0.5)
        eng.synchronize()
        assert np.array_equal(eng.get_weights(), w_dev)
        plan_h.destroy()


def test_device_drawn_plans_beyond_the_column_slices_run_like_host_drawn_ones():
    """Steps of more than 1,024 rows do not run on column slices: their plan is laid out on the HOST at its first run, which
    fetches the device-drawn lists back (plan_host_idx).  2 workers x 700 rows per step: the same weights as the plan made
    from the host's lists, bit for bit."""
    n_train = 20000
    data, eng = engine(n_train + 500, seed=8, ds_rows=n_train)
    with eng:
        split = host.split_vanilla(n_train, 2)
        rnd = host.JavaRandom(0)
        state = rnd.seed
        idx_h, offs_h, n_h = host.epoch_lists(rnd, split, 10000, 700, native=True)
        plan, n_d, state_d, _ = eng.plan_from_seed(state, split, 10000, 700)
        assert n_d == n_h == 15 and state_d == rnd.seed
        out = []
        for p in (plan, eng.plan_flat(idx_h, offs_h, n_h, 2)):
            eng.set_weights(np.zeros(data.dim + 1, dtype=np.float32))
            eng.plan_run(p, 0, n_h, 0.05)
            eng.synchronize()
            out.append((eng.get_weights(), p.info()["kind"], eng.grad_kernel_name()))
            p.destroy()
        assert out[0][1] == out[1][1] != "column_slices" and out[0][2] == out[1][2], out
        assert np.array_equal(out[0][0], out[1][0]) and np.abs(out[0][0]).max() > 0


def test_random_small_epochs_device_equals_host():
    """Forty random configurations (1-8 workers, splits of 1-6,000 rows incl. powers of two -- java.util.Random.nextInt takes
    another branch there --, batches of 1-700, arbitrary generator states): lists, step counts and final states equal."""
    data, eng = engine(50000, seed=2)
    rng = np.random.default_rng(20260930)
    with eng:
        for case in range(40):
            k = int(rng.integers(1, 9))
            lens = [int(rng.choice([1, 2, 3, 64, 100, 1024, 4096, int(rng.integers(1, 6001))])) for _ in range(k)]
            split, at = [], 0
            for ln in lens:
                split.append(range(at, at + ln))
                at += ln
            batch = int(rng.choice([1, 7, 100, 256, int(rng.integers(1, 701))]))
            state = int(rng.integers(0, 1 << 48))
            max_samples = max(lens) if case % 3 else int(rng.integers(1, max(lens) + 60))   # (Master.fit passes the longest split; any value works)
            (idx_h, offs_h, n_h, state_h), (plan, n_d, state_d, draws) = both(eng, state, split, max_samples, batch)
            assert n_d == n_h and state_d == state_h, (case, lens, batch)
            if n_h == 0:
                assert plan is None
                continue
            idx_d, offs_d = eng.plan_lists(plan)
            assert np.array_equal(offs_d, offs_h) and np.array_equal(idx_d, idx_h), (case, lens, batch, int(np.flatnonzero(idx_d != idx_h)[0]) if len(idx_d) == len(idx_h) else -1)
            plan.destroy()
