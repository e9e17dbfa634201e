"""CPU tests of the host-side mirror (distributed-sgd_amd/host.py) and of the N > 1 orchestration with
world_size 2 over gloo."""

import os

import numpy as np
import pytest

import dsgd_amd
from dsgd_amd import host
from oracle import oracle as orc
from oracle_backend import OracleBackend

HERE = os.path.dirname(os.path.abspath(__file__))


def test_java_random_known_answers():
    r = host.JavaRandom(0)
    assert [r.next_int() for _ in range(3)] == [-1155484576, -723955400, 1033096058]  # new java.util.Random(0)
    r = host.JavaRandom(0)
    assert [r.next_int(10) for _ in range(10)] == [0, 8, 9, 7, 5, 3, 1, 1, 9, 4]
    r = host.JavaRandom(42)
    assert r.next_int() == -1170105035


def test_scala_shuffle_is_a_seeded_permutation():
    a = host.scala_shuffle(list(range(20)), host.JavaRandom(0))
    b = host.scala_shuffle(list(range(20)), host.JavaRandom(0))
    assert a == b and sorted(a) == list(range(20)) and a != list(range(20))
    # the last element is decided first: swap(n - 1, nextInt(n)) with nextInt(20) of seed 0
    assert a[19] == host.JavaRandom(0).next_int(20)


def test_config_defaults_env_and_roles():
    # the HOCON text a deployment would hold, synthesised from the key table of tests/golden/config_keys.json
    import json

    keys = json.load(open(os.path.join(HERE, "golden", "config_keys.json")))["keys"]
    lines = ["dsgd {"]
    for k, (default, env_var) in keys.items():
        if default is not None:
            lines.append("  %s = %s" % (k, default))
        lines.append("  %s = ${?%s}   # environment override" % (k, env_var))
    text = "\n".join(lines + ["}", "kamon { metric { tick-interval = 1 seconds } }"])
    assert set(keys) == set(host.Config._KEYS) and all(host.Config._KEYS[k][1] == v[1] for k, v in keys.items())
    c = host.Config.load(text, env={})
    assert (c.batch_size, c.learning_rate, c.lambda_, c.node_count, c.max_epochs) == (100, 0.5, 1e-5, 3, 10)
    assert (c.check_every, c.leaky_loss, c.patience, c.conv_delta, c.full, c.async_) == (100, 0.9, 5, 0.01, False, False)
    assert c.role() == "dev" and c.port == 4000 and c.host == "127.0.0.1"
    # kube/config-sync.yaml:8-20 overrides
    env = {"DSGD_BATCH_SIZE": "200", "DSGD_NODE_COUNT": "4", "DSGD_CONV_DELTA": "0.001", "DSGD_ASYNC": "true",
           "DSGD_MASTER_HOST": "10.0.0.1", "DSGD_MASTER_PORT": "46746", "DSGD_NODE_PORT": "46746"}
    c = host.Config.load(text, env=env)
    assert (c.batch_size, c.node_count, c.conv_delta, c.async_) == (200, 4, 0.001, True)
    assert c.role() == "slave"
    env["DSGD_NODE_HOST"] = "10.0.0.1"
    assert host.Config.load(text, env=env).role() == "master"


def test_early_stopping_and_split():
    crit = host.EarlyStopping.no_improvement(patience=2, min_delta=0.01)
    assert not crit([]) and not crit([0.4, 0.5]) and not crit([0.45, 0.4, 0.5]) and crit([0.46, 0.45, 0.4, 0.5])
    assert host.EarlyStopping.target(0.3)([0.29, 0.5]) and not host.EarlyStopping.target(0.3)([0.31])
    assert [len(r) for r in host.split_vanilla(18519, 3)] == [6173, 6173, 6173]
    assert len(host.split_vanilla(9, 4)) == 3  # SplitStrategy.scala:14 can yield fewer groups than workers


def small_problem(n_rows=3000, seed=31):
    data = dsgd_amd.synth.generate(n_rows, seed=seed)
    n_train = int(n_rows * 0.8)
    o = orc.Oracle(data.dim, data.row_ptr, data.col, data.val, data.label, 1e-5)
    o.set_dim_sparsity(o.dim_sparsity(n_train))
    return data, n_train, o


def test_master_sync_fit_loop():
    data, n_train, o = small_problem()
    be = OracleBackend(o)
    m = host.MasterSync(be, n_train, data.n_rows, node_count=3, rnd=host.JavaRandom(0))
    state = m.fit(np.zeros(data.dim + 1), max_epochs=2, batch_size=100, learning_rate=0.5,
                  stopping_criterion=host.EarlyStopping.no_improvement(5, 0.01))
    per_epoch = -(-len(host.split_vanilla(n_train, 3)[0]) // 100)  # ceil(800 / 100) batches per epoch
    assert len(be.steps) == 2 * per_epoch and all(s == [100, 100, 100] for s in be.steps)
    assert len(m.losses) == 2 and len(m.test_losses) == 2 and state.updates == 2 and state.end is not None
    assert m.accs[0] > 0.5
    # replaying the same random stream gives the same weights (Main.scala:32 seeds the global Random with 0)
    be2 = OracleBackend(o)
    m2 = host.MasterSync(be2, n_train, data.n_rows, node_count=3, rnd=host.JavaRandom(0))
    state2 = m2.fit(np.zeros(data.dim + 1), 2, 100, 0.5, host.EarlyStopping.no_improvement(5, 0.01))
    np.testing.assert_array_equal(state.grad, state2.grad)
    # a criterion that fires immediately stops after the first epoch's evaluation
    be3 = OracleBackend(o)
    m3 = host.MasterSync(be3, n_train, data.n_rows, node_count=3)
    m3.fit(np.zeros(data.dim + 1), 10, 100, 0.5, lambda losses: len(losses) >= 1)
    assert len(be3.steps) == per_epoch


# ---- world_size 2 over gloo: host-owned all-reduce == one process hosting all workers ---------------------------
def _rank_main(rank, world, port, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data, n_train, o = small_problem()
    local = OracleBackend(o)
    be = host.HostAllReduceBackend(local, dist, n_train)
    # four workers in total, two per rank; rank r hosts workers 2r and 2r+1 (contiguous ranges of the train rows)
    split = host.split_vanilla(n_train, 2 * world)
    rng = np.random.default_rng(123)
    for step in range(5):
        lists = [rng.permutation(np.asarray(r))[:100].astype(np.int32) for r in split]  # same stream on every rank
        be.sync_step(lists[2 * rank:2 * rank + 2], 0.5)
    loss, acc, counts = be.loss_acc(n_train + rank * 100, n_train + rank * 100 + 100)   # disjoint eval shards
    np.save(os.path.join(out_dir, "w%d.npy" % rank), local.w)
    np.save(os.path.join(out_dir, "e%d.npy" % rank), np.asarray([loss, acc] + counts))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo_matches_single_process(tmp_path):
    import torch.multiprocessing as mp

    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    w0, w1 = np.load(tmp_path / "w0.npy"), np.load(tmp_path / "w1.npy")
    np.testing.assert_array_equal(w0, w1)  # replicas apply the identical update
    data, n_train, o = small_problem()
    ref = OracleBackend(o)
    split = host.split_vanilla(n_train, 4)
    rng = np.random.default_rng(123)
    for step in range(5):
        lists = [rng.permutation(np.asarray(r))[:100].astype(np.int32) for r in split]
        ref.sync_step(lists, 0.5)  # mean over 4 workers in one process (Master.scala:194)
    np.testing.assert_allclose(w0, ref.w, rtol=0, atol=1e-6)  # the host path rounds g_mean to fp32 at the ABI
    e0, e1 = np.load(tmp_path / "e0.npy"), np.load(tmp_path / "e1.npy")
    np.testing.assert_array_equal(e0, e1)  # tallies are summed over ranks
    ref.w = w0.copy()
    c = np.zeros(3)
    for r in range(2):
        c += np.asarray(ref.loss_acc(n_train + r * 100, n_train + r * 100 + 100)[2])
    assert list(e0[2:]) == list(c) and abs(e0[1] - c[0] / 200) < 1e-12


# ---- asynchronous mode across ranks: replicas + periodic exchange of the summed updates ---------------------------
def _async_schedule(n_train, world, rounds, every):
    """The sample lists of every rank, round and local update (the same on whoever computes them)."""
    rng = np.random.default_rng(321)
    split = host.split_vanilla(n_train, world)
    return [[[rng.permutation(np.asarray(split[r]))[:50].astype(np.int32) for _ in range(every)] for _ in range(rounds)]
            for r in range(world)]


def _async_rank_main(rank, world, port, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data, n_train, o = small_problem()
    local = OracleBackend(o)
    ex = host.HostAsyncExchange(local, dist)
    for lists in _async_schedule(n_train, world, 4, 3)[rank]:
        ex.run_round(lists, 0.5)
    np.save(os.path.join(out_dir, "aw%d.npy" % rank), local.w)
    dist.barrier()
    dist.destroy_process_group()


class _OneRank:
    """torch.distributed's surface for a single rank: the peers' part of an exchange is exactly zero."""

    @staticmethod
    def get_world_size():
        return 1

    @staticmethod
    def all_reduce(t):
        return None


def test_async_exchange_world_2_gloo_matches_the_in_process_simulation(tmp_path):
    """core/Slave.scala:99-105,177-185 batched per round (the scheme of dsgd_async_set_exchange): two gloo ranks end
    with the same replica, equal to a single-process simulation of the two replicas exchanging their summed updates."""
    import torch.multiprocessing as mp

    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_async_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    w0, w1 = np.load(tmp_path / "aw0.npy"), np.load(tmp_path / "aw1.npy")
    # (w - A) - B on one rank, (w - B) - A on the other: equal up to the order of two fp64 subtractions
    np.testing.assert_allclose(w0, w1, rtol=0, atol=1e-14)
    data, n_train, o = small_problem()
    reps = [OracleBackend(o), OracleBackend(o)]
    sched = _async_schedule(n_train, 2, 4, 3)
    w_prev = [r.w.copy() for r in reps]
    for rnd in range(4):
        for r in range(2):
            for idx in sched[r][rnd]:
                reps[r].async_step(idx, 0.5)
        d = [w_prev[r] - reps[r].w for r in range(2)]
        for r in range(2):
            reps[r].w = reps[r].w - d[1 - r]   # the peer's part
            w_prev[r] = reps[r].w.copy()
    np.testing.assert_allclose(w0, reps[0].w, rtol=0, atol=1e-14)
    assert np.abs(w0).max() > 0.01   # something was learnt: the comparison is not 0 == 0


def test_async_exchange_with_one_rank_changes_nothing():
    data, n_train, o = small_problem()
    plain, local = OracleBackend(o), OracleBackend(o)
    ex = host.HostAsyncExchange(local, _OneRank)
    for lists in _async_schedule(n_train, 1, 3, 4)[0]:
        for idx in lists:
            plain.async_step(idx, 0.5)
        ex.run_round(lists, 0.5)
    np.testing.assert_array_equal(local.w, plain.w)   # bit for bit: the peers' part is exactly zero
    assert ex.rounds == 3


# ---- the reference's random stream, natively (csrc/jrand.c) ---------------------------------------------------------
def test_native_epoch_lists_are_the_reference_stream_draw_for_draw():
    """host.epoch_lists: Master.scala:184's per-batch reshuffle of every split for one epoch.  The native form (the
    epoch's shuffles drawn in parallel from jump-ahead states, rejections resolved exactly) against the pure-Python
    restatement of java.util.Random / scala.util.Random.shuffle, lists and generator state alike; uneven splits, a batch
    size that does not divide them, one worker, batch 1."""
    assert host._host_lib() is not None, "libdsgd_host.so not built (__graft_entry__.build())"
    for n, k, b, seed in ((1000, 3, 100, 0), (18519, 3, 100, 0), (997, 4, 64, 7), (130, 1, 1, 3), (5000, 8, 37, 11), (12, 5, 5, 1)):
        split = host.split_vanilla(n, k)
        mx = max(len(r) for r in split)
        ra, rb = host.JavaRandom(seed), host.JavaRandom(seed)
        ia, oa, na = host.epoch_lists(ra, split, mx, b, native=True)
        ib, ob, nb = host.epoch_lists(rb, split, mx, b, native=False)
        assert na == nb and np.array_equal(oa, ob) and np.array_equal(ia, ib), (n, k, b)
        assert ra.seed == rb.seed                       # the generator stands where the JVM's would
        assert ra.next_int(1000) == rb.next_int(1000)
        if n == 18519:                                  # the reference's own shape: 62 steps of 3 x 100, every row id in its split
            assert na == 62 and len(ia) == 62 * 300 - (62 * 100 - 6173) * 3
            first = ia[:100]
            assert first.min() >= 0 and first.max() < 6173 and len(set(first.tolist())) == 100
    # a short last split: the epoch stops where the reference's slave would be handed an empty slice
    split = host.split_vanilla(9, 4)                    # three groups of 3 (SplitStrategy.scala:14)
    ia, oa, na = host.epoch_lists(host.JavaRandom(0), split, 3, 2, native=True)
    ib, ob, nb = host.epoch_lists(host.JavaRandom(0), split, 3, 2, native=False)
    assert na == nb == 2 and np.array_equal(ia, ib) and np.array_equal(oa, ob)


def test_native_stream_with_rejections():
    """nextInt(bound) rejects a raw value with probability ~bound / 2^31: visible only for large splits.  One shuffle of
    600,000 elements consumes ~80 more raw values than it has draws; the parallel epoch form must land on the same lists
    and the same generator state as shuffles drawn one after the other."""
    import ctypes as C

    lib = host._host_lib()
    n = 600000
    # sequential native shuffle against the pure-Python one (which follows java.util.Random's loop literally)
    st = C.c_uint64(host.JavaRandom(5).seed)
    buf = np.arange(n, dtype=np.int32)
    used = lib.dsgd_jrand_shuffle(C.byref(st), buf.ctypes.data_as(C.c_void_p), C.c_int64(n))
    rp = host.JavaRandom(5)
    ref = host.scala_shuffle(range(n), rp)
    assert np.array_equal(buf, np.asarray(ref, dtype=np.int32)) and st.value == rp.seed
    assert used > n - 1                                 # rejections happened: the stream is longer than the draws
    # the parallel epoch against shuffles drawn one after the other
    split = host.split_vanilla(2 * n, 2)
    r1 = host.JavaRandom(9)
    idx, offs, ns = host.epoch_lists(r1, split, n, 250000, native=True)   # 3 steps x 2 workers = 6 shuffles of 600,000
    st = C.c_uint64(host.JavaRandom(9).seed)
    want = []
    for step in range(3):
        for r in split:
            buf = np.arange(r.start, r.stop, dtype=np.int32)
            lib.dsgd_jrand_shuffle(C.byref(st), buf.ctypes.data_as(C.c_void_p), C.c_int64(len(buf)))
            want.append(buf[step * 250000:(step + 1) * 250000].copy())
    assert ns == 3 and np.array_equal(idx, np.concatenate(want)) and r1.seed == st.value
    # short epochs are drawn in ONE speculative pass (no rejection expected); forced onto this stream -- which has hundreds --
    # the speculation must notice and give way to the exact two-pass form
    os.environ["DSGD_HOST_SPECULATE"] = "1"
    try:
        r2 = host.JavaRandom(9)
        idx2, offs2, ns2 = host.epoch_lists(r2, split, n, 250000, native=True)
    finally:
        del os.environ["DSGD_HOST_SPECULATE"]
    assert ns2 == 3 and np.array_equal(idx2, idx) and r2.seed == r1.seed


def test_fit_through_plans_is_fit_step_by_step():
    """MasterSync.fit with an epoch as ONE plan (the path that reaches the 5 us kernel) against the per-batch form: the
    same lists from the same stream, the same weights bit for bit, the same curves, counters and number of timer entries;
    the prefetched plan of an epoch that never runs is dropped and the stream put back."""
    data, n_train, o = small_problem()
    crit = host.EarlyStopping.no_improvement(5, 0.01)
    be_a, be_b = OracleBackend(o), OracleBackend(o)
    logs_a, logs_b = [], []
    ma = host.MasterSync(be_a, n_train, data.n_rows, 3, rnd=host.JavaRandom(0), plans=False, log=logs_a.append)
    mb = host.MasterSync(be_b, n_train, data.n_rows, 3, rnd=host.JavaRandom(0), plans=True, log=logs_b.append)
    sa = ma.fit(np.zeros(data.dim + 1), 3, 100, 0.5, crit)
    sb = mb.fit(np.zeros(data.dim + 1), 3, 100, 0.5, crit)
    np.testing.assert_array_equal(sa.grad, sb.grad)
    assert be_a.steps == be_b.steps and ma.test_losses == mb.test_losses and ma.accs == mb.accs
    assert ma.rnd.seed == mb.rnd.seed and be_b.plans_made == 3 and mb.steps_run == ma.steps_run == 24
    ca, cb = ma.metrics.snapshot(), mb.metrics.snapshot()
    assert ca["counters"] == cb["counters"] and len(ca["histograms"]["master.sync.batch.duration"]) == len(cb["histograms"]["master.sync.batch.duration"]) == 24
    assert logs_a == logs_b
    # stopped by the criterion after the first epoch: the prefetched second epoch is dropped, the stream stands where the
    # per-batch form leaves it
    be_c, be_d = OracleBackend(o), OracleBackend(o)
    mc = host.MasterSync(be_c, n_train, data.n_rows, 3, rnd=host.JavaRandom(3), plans=False)
    md = host.MasterSync(be_d, n_train, data.n_rows, 3, rnd=host.JavaRandom(3), plans=True, prefetch=True)
    mc.fit(np.zeros(data.dim + 1), 10, 100, 0.5, lambda losses: len(losses) >= 1)
    md.fit(np.zeros(data.dim + 1), 10, 100, 0.5, lambda losses: len(losses) >= 1)
    assert mc.rnd.seed == md.rnd.seed and be_c.steps == be_d.steps and be_d.plans_made == 2
    np.testing.assert_array_equal(be_c.w, be_d.w)
    # a short last split: the reference's slave throws on the empty slice -- after the batches before it ran
    be_e = OracleBackend(o)
    me = host.MasterSync(be_e, 10, 12, 4, rnd=host.JavaRandom(0), plans=True)   # splits of 3, 3, 3, 1 rows (SplitStrategy.scala:14)
    with pytest.raises(ValueError, match="empty list"):
        me.fit(np.zeros(data.dim + 1), 1, 2, 0.5, crit)
    assert be_e.steps == [[2, 2, 2, 1]]


def test_epoch_lists_edge_cases_native_equals_python():
    """a batch size beyond int32 selects whole splits natively as in the Python restatement (same lists, same
    generator state); an empty split ends the epoch before its first batch on both paths (the reference's Vec.sum throws
    there, math/Vec.scala:129 -- MasterSync raises it)."""
    split = [range(0, 50), range(50, 90)]
    ra, rb = host.JavaRandom(0), host.JavaRandom(0)
    a = host.epoch_lists(ra, split, 50, 2 ** 40)
    b = host.epoch_lists(rb, split, 50, 2 ** 40, native=False)
    assert a[2] == b[2] == 1 and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and ra.seed == rb.seed
    for native in (None, False):
        idx, offs, n = host.epoch_lists(host.JavaRandom(0), [range(0, 50), range(50, 50)], 50, 10, native=native)
        assert n == 0 and len(idx) == 0
