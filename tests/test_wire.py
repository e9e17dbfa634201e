"""The wire-level worker (distributed-sgd_amd/wire.py) against the protocol of src/main/protobuf/proto.proto and the
handler semantics of core/Slave.scala -- CPU only: the backend is the oracle behind the Engine surface."""

import time

import numpy as np
import pytest

import dsgd_amd
from dsgd_amd import host, wire
from oracle import oracle as orc
from oracle_backend import OracleBackend

grpc = pytest.importorskip("grpc")


def hexs(m):
    return m.SerializeToString().hex()


def test_messages_encode_as_the_proto_file_says():
    """Known-answer encodings derived by hand from proto.proto:20-70 (field numbers and wire types)."""
    M = wire.messages()
    assert hexs(M["Node"](host="a", port=1)) == "0a0161" "1001"                  # string host = 1; int32 port = 2
    assert hexs(M["Ack"]()) == ""
    s = M["Sparse"](size=5)
    s.map[3] = 0.5                                                                # map<int32,double> map = 1; int32 size = 2
    assert hexs(s) == "0a0b" "0803" "11000000000000e03f" "1005"
    assert hexs(M["GradUpdate"](gradUpdate=s)) == "0a0f" + hexs(s)               # Sparse gradUpdate = 1
    fr = M["ForwardRequest"](samples=[1, 300], weights=s)                         # repeated int32 samples = 1 [packed]; weights = 2
    assert hexs(fr) == "0a03" "01ac02" "120f" + hexs(s)
    assert hexs(M["ForwardReply"](predictions=[1.0, -1.0])) == "0a10" "000000000000f03f" "000000000000f0bf"
    gr = M["GradientRequest"](weights=s, samples=[7])                             # weights = 1; samples = 2 [packed]
    assert hexs(gr) == "0a0f" + hexs(s) + "1201" "07"
    sa = M["StartAsyncRequest"](weights=s, samples=[2], batchSize=100, learningRate=0.5)
    assert hexs(sa) == "0a0f" + hexs(s) + "120102" "1864" "21000000000000e03f"   # batchSize = 3; double learningRate = 4
    # an unpacked encoding of `samples` (what a proto2-style writer emits) parses to the same message
    assert M["GradientRequest"].FromString(bytes.fromhex("1007" "1008")).samples == [7, 8]
    assert sorted(wire.SLAVE_METHODS) == ["Forward", "Gradient", "RegisterSlave", "StartAsync", "StopAsync", "UnregisterSlave", "UpdateGrad"]


def test_vec_sparse_mapping():
    w = np.zeros(11, dtype=np.float32)
    w[[1, 4, 10]] = [0.5, -2.0, 1e-25]          # below Sparse.epsilon: not stored (math/Sparse.scala:108-118)
    s = wire.to_sparse(w, 10)
    assert dict(s.map) == {1: 0.5, 4: -2.0} and s.size == 10
    np.testing.assert_array_equal(wire.from_sparse(s, 11), np.where(np.abs(w) > 1e-20, w, 0))
    s.map[11] = 1.0
    with pytest.raises(IndexError):
        wire.from_sparse(s, 11)


def make_backend(seed, n_rows=400):
    data = dsgd_amd.synth.generate(n_rows, seed=seed)
    o = orc.Oracle(data.dim, data.row_ptr, data.col, data.val, data.label, 1e-5)
    o.set_dim_sparsity(o.dim_sparsity(n_rows))
    return data, o, OracleBackend(o)


def test_sync_handlers_over_the_wire():
    data, o, backend = make_backend(71)
    master = wire.MasterService(expected_nodes=2).start()
    a = wire.SlaveWorker(backend, data.dim, master=("127.0.0.1", master.port)).start()
    b = wire.SlaveWorker(make_backend(72)[2], data.dim, master=("127.0.0.1", master.port)).start()
    try:
        assert master.ready.wait(5)
        # the master cross-registered the colleagues (core/Master.scala:229-233)
        assert list(a.others) == [("127.0.0.1", b.port)] and list(b.others) == [("127.0.0.1", a.port)]
        stub = wire.Stub(wire.new_channel("127.0.0.1", a.port), "Slave")
        M = wire.messages()
        rng = np.random.default_rng(1)
        w = np.zeros(data.dim + 1)
        w[rng.choice(np.arange(1, data.dim + 1), 3000, replace=False)] = rng.normal(scale=0.1, size=3000)
        idx = rng.permutation(400)[:64].astype(np.int32)
        reply = stub.Gradient(M["GradientRequest"](weights=wire.to_sparse(w, data.dim), samples=idx.tolist()))
        g_ref = o.gradient(w.astype(np.float32).astype(np.float64), idx)   # the worker receives fp64, computes from fp32
        got = np.zeros(data.dim + 1)
        for k, v in reply.gradUpdate.map.items():
            got[k] = v
        assert reply.gradUpdate.size == data.dim
        np.testing.assert_allclose(got, g_ref, rtol=0, atol=1e-12)
        assert set(reply.gradUpdate.map) == set(np.flatnonzero(g_ref).tolist())   # zeros are not sent
        pred = stub.Forward(M["ForwardRequest"](samples=idx.tolist(), weights=wire.to_sparse(w, data.dim)))
        np.testing.assert_array_equal(np.asarray(pred.predictions), o.forward(w.astype(np.float32).astype(np.float64), idx))
        assert a.metrics.snapshot()["counters"] == {"slave.sync.backward": 64, "slave.sync.forward": 64}
        # Vec.sum of an empty batch throws in the handler's Future: the RPC fails (math/Vec.scala:129)
        with pytest.raises(grpc.RpcError):
            stub.Gradient(M["GradientRequest"](weights=wire.to_sparse(w, data.dim), samples=[]))
        # a synchronous-mode slave refuses the async calls (core/Slave.scala:160,178,188)
        for call, req in ((stub.StartAsync, M["StartAsyncRequest"](batchSize=1)), (stub.UpdateGrad, M["GradUpdate"]()),
                          (stub.StopAsync, M["Empty"]())):
            with pytest.raises(grpc.RpcError) as ei:
                call(req)
            assert "synchronous mode" in ei.value.details()
        b.stop()
        time.sleep(0.2)
        assert a.others == {}                               # unregisterSlave reached the colleague through the master
    finally:
        a.stop()
        master.stop()


def test_async_task_gossips_updates():
    data, o, backend_a = make_backend(81)
    _, _, backend_b = make_backend(81)
    seen = []
    master = wire.MasterService(expected_nodes=2, on_update=seen.append).start()
    a = wire.SlaveWorker(backend_a, data.dim, master=("127.0.0.1", master.port), asynchronous=True, rnd=host.JavaRandom(0)).start()
    b = wire.SlaveWorker(backend_b, data.dim, master=("127.0.0.1", master.port), asynchronous=True).start()
    try:
        assert master.ready.wait(5)
        M = wire.messages()
        stub = wire.Stub(wire.new_channel("127.0.0.1", a.port), "Slave")
        w0 = np.zeros(data.dim + 1, dtype=np.float32)
        stub.StartAsync(M["StartAsyncRequest"](weights=wire.to_sparse(w0, data.dim), samples=list(range(100, 300)), batchSize=10,
                                               learningRate=0.5))
        with pytest.raises(grpc.RpcError):                  # already running (core/Slave.scala:161)
            stub.StartAsync(M["StartAsyncRequest"](batchSize=1))
        deadline = time.time() + 10
        while a.metrics.snapshot()["counters"].get("slave.async.batch", 0) < 5 and time.time() < deadline:
            time.sleep(0.01)
        stub.StopAsync(M["Empty"]())
        a._thread.join(5)
        assert a._async_error is None
        n_batches = a.metrics.snapshot()["counters"]["slave.async.batch"]
        assert n_batches >= 5 and a.metrics.snapshot()["counters"]["slave.async.backward"] == 10 * n_batches
        time.sleep(0.5)                                      # fire-and-forget updates drain
        # the first batch is the reference's: Random(0).shuffle(0 until 200) take 10, used as DATA indices (:87)
        first = host.scala_shuffle(list(range(200)), host.JavaRandom(0))[:10]
        w_ref = np.zeros(data.dim + 1)
        d_ref = o.async_step(w_ref, np.asarray(first, dtype=np.int32), 0.5, want_delta=True)
        support = set(np.flatnonzero(d_ref).tolist())
        match = [u for u in seen if set(u) == support]       # (fire-and-forget calls may overtake each other)
        assert len(match) == 1
        np.testing.assert_allclose([match[0][k] for k in sorted(support)], d_ref[sorted(support)], rtol=1e-6, atol=1e-9)
        # colleague b applied the same updates: w_b = -sum of deltas = w_a (b computes nothing itself)
        assert b.metrics.snapshot()["counters"]["slave.async.grad.update"] == len(seen) == n_batches
        np.testing.assert_allclose(backend_b.get_weights(), backend_a.get_weights(), rtol=0, atol=1e-5)
    finally:
        a.stop()
        b.stop()
        master.stop()


def test_metrics_and_final_weights_line():
    m = host.Metrics()
    m.counter("slave.sync.backward", 300)
    m.counter("slave.sync.backward", 300)
    m.histogram("master.sync.loss", 0.93)   # Master.scala:150: losses.head.toLong
    with m.timer("master.sync.batch.duration"):
        pass
    snap = m.snapshot()
    assert snap["counters"] == {"slave.sync.backward": 600}
    assert snap["histograms"]["master.sync.loss"] == [0] and len(snap["histograms"]["master.sync.batch.duration"]) == 1
    assert any(l.startswith("slave.sync.backward count=600i") for l in m.influx_lines())
    w = np.zeros(6, dtype=np.float32)
    w[[1, 4]] = [0.1, -2.5]
    assert host.format_final_weights(w) == "1:0.10000000149011612 4:-2.5"   # Main.scala:114 (fp32 values printed as doubles)


def test_master_fit_over_the_wire_equals_the_in_process_fit():
    """core/Master.scala:120-218 with the slaves behind gRPC: same java.util.Random stream, same batches, same result as
    the mirror driving one backend directly (up to the fp32 weights the workers compute from)."""
    data, o, direct = make_backend(91, n_rows=900)
    n_train, n_rows, k = 720, 900, 3
    master = wire.MasterService(expected_nodes=k).start()
    workers = [wire.SlaveWorker(make_backend(91, n_rows=900)[2], data.dim, master=("127.0.0.1", master.port)).start() for _ in range(k)]
    try:
        assert master.ready.wait(5)
        stubs = [wire.Stub(wire.new_channel("127.0.0.1", w.port), "Slave") for w in workers]
        over_wire = host.MasterSync(wire.WireBackend(stubs, data.dim, 1e-5, data.label), n_train, n_rows, node_count=k, rnd=host.JavaRandom(0))
        s_wire = over_wire.fit(np.zeros(data.dim + 1), 2, 100, 0.5, host.EarlyStopping.no_improvement(5, 0.01))
        in_proc = host.MasterSync(direct, n_train, n_rows, node_count=k, rnd=host.JavaRandom(0))
        s_ref = in_proc.fit(np.zeros(data.dim + 1), 2, 100, 0.5, host.EarlyStopping.no_improvement(5, 0.01))
        assert s_wire.updates == s_ref.updates == 2
        np.testing.assert_allclose(s_wire.grad, s_ref.grad, rtol=0, atol=2e-5 * max(1.0, np.abs(s_ref.grad).max()))
        np.testing.assert_allclose(over_wire.test_losses, in_proc.test_losses, rtol=0, atol=5e-3)
        np.testing.assert_allclose(over_wire.test_accs, in_proc.test_accs, rtol=0, atol=5e-3)
        # 2 epochs x 3 batches (240 rows per worker, batch 100) x 3 workers: one Gradient RPC each, per sample counted
        assert sum(w.metrics.snapshot()["counters"]["slave.sync.backward"] for w in workers) == 2 * 720
    finally:
        for w in workers:
            w.stop()
        master.stop()
