"""Wire-level drop-in worker: the reference's `Slave` gRPC service hosted in front of an engine backend, so that an
unmodified Scala `Master` can drive a GPU worker without JNI (SURVEY.md 8(b) seam S3, 8(f) item 3).

What is restated here (nothing is generated: the image has no protoc / grpc_tools, the descriptors are built
programmatically and the field numbers, types and method paths below ARE the protocol):

    src/main/protobuf/proto.proto:13-70      messages Node, Ack, Sparse{map<int32,double> map = 1; int32 size = 2},
                                             GradUpdate, ForwardRequest, ForwardReply, GradientRequest,
                                             StartAsyncRequest; services Master and Slave (plaintext HTTP/2,
                                             core/package.scala:20-21); the ScalaPB options are codegen-only
    core/Slave.scala:113-198                 SlaveImpl: registerSlave / unregisterSlave keep the colleague stubs,
                                             forward, gradient, startAsync, updateGrad, stopAsync
    core/Slave.scala:79-111                  asyncTask: sample, mean of gated sub-gradients, regularise, scale by the
                                             learning rate, subtract locally, gossip the update to colleagues + master
    core/Slave.scala:45-60,67-76             register with / unregister from the master
    core/package.scala:11-13                 Vec <-> Sparse: the map holds the non-zero entries, `size` the dimension

The backend is anything with the Engine surface used below (`gradient(idx, w)`, `forward(idx, w)`,
`async_step(idx, lr, want_delta)`, `update_grad(keys, values)`, `set_weights`, `get_weights`, `dp`): the HIP engine
on a GPU box, the oracle-backed stand-in in the CPU tests.  Vectors cross the wire as the reference sends them
(sparse fp64 maps keyed by feature id); the engine side is dense fp32 indexed by key.
"""

from __future__ import annotations

import threading
from concurrent import futures
from typing import Callable, Dict, Optional, Sequence, Tuple

import numpy as np

from . import host

PACKAGE = "epfl.distributed"


# ---- descriptors (proto.proto:13-70) ---------------------------------------------------------------------------------
def _build_messages():
    from google.protobuf import descriptor_pb2, descriptor_pool, empty_pb2, message_factory  # noqa: F401

    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "epfl/distributed/proto.proto"
    fd.package = PACKAGE
    fd.syntax = "proto3"
    fd.dependency.append("google/protobuf/empty.proto")

    def msg(name, *fields):
        m = fd.message_type.add()
        m.name = name
        for fname, number, ftype, label, type_name in fields:
            f = m.field.add()
            f.name, f.number, f.type, f.label = fname, number, ftype, label
            if type_name:
                f.type_name = type_name
            f.json_name = fname
        return m

    OPT, REP = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    msg("Node", ("host", 1, F.TYPE_STRING, OPT, None), ("port", 2, F.TYPE_INT32, OPT, None))
    msg("Ack")
    sparse = msg("Sparse", ("map", 1, F.TYPE_MESSAGE, REP, ".%s.Sparse.MapEntry" % PACKAGE),
                 ("size", 2, F.TYPE_INT32, OPT, None))
    entry = sparse.nested_type.add()  # map<int32, double> == repeated MapEntry{key = 1; value = 2} with map_entry
    entry.name = "MapEntry"
    entry.options.map_entry = True
    for fname, number, ftype in (("key", 1, F.TYPE_INT32), ("value", 2, F.TYPE_DOUBLE)):
        f = entry.field.add()
        f.name, f.number, f.type, f.label, f.json_name = fname, number, ftype, OPT, fname
    S = ".%s.Sparse" % PACKAGE
    msg("GradUpdate", ("gradUpdate", 1, F.TYPE_MESSAGE, OPT, S))
    msg("ForwardRequest", ("samples", 1, F.TYPE_INT32, REP, None), ("weights", 2, F.TYPE_MESSAGE, OPT, S))
    msg("ForwardReply", ("predictions", 1, F.TYPE_DOUBLE, REP, None))
    msg("GradientRequest", ("weights", 1, F.TYPE_MESSAGE, OPT, S), ("samples", 2, F.TYPE_INT32, REP, None))
    msg("StartAsyncRequest", ("weights", 1, F.TYPE_MESSAGE, OPT, S), ("samples", 2, F.TYPE_INT32, REP, None),
        ("batchSize", 3, F.TYPE_INT32, OPT, None), ("learningRate", 4, F.TYPE_DOUBLE, OPT, None))
    pool = descriptor_pool.Default()
    try:
        pool.FindFileByName(fd.name)
    except KeyError:
        pool.Add(fd)
    get = message_factory.GetMessageClass
    names = ["Node", "Ack", "Sparse", "GradUpdate", "ForwardRequest", "ForwardReply", "GradientRequest", "StartAsyncRequest"]
    classes = {n: get(pool.FindMessageTypeByName("%s.%s" % (PACKAGE, n))) for n in names}
    classes["Empty"] = empty_pb2.Empty
    return classes


_MSG = None


def messages():
    global _MSG
    if _MSG is None:
        _MSG = _build_messages()
    return _MSG


# method -> (request message, reply message); paths are /epfl.distributed.<Service>/<Method>
SLAVE_METHODS = {
    "RegisterSlave": ("Node", "Ack"), "UnregisterSlave": ("Node", "Ack"),
    "Forward": ("ForwardRequest", "ForwardReply"), "Gradient": ("GradientRequest", "GradUpdate"),
    "StartAsync": ("StartAsyncRequest", "Ack"), "StopAsync": ("Empty", "Ack"), "UpdateGrad": ("GradUpdate", "Ack"),
}
MASTER_METHODS = {"RegisterSlave": ("Node", "Ack"), "UnregisterSlave": ("Node", "Ack"), "UpdateGrad": ("GradUpdate", "Ack")}


# ---- Vec <-> Sparse (core/package.scala:11-13) ------------------------------------------------------------------------
def to_sparse(dense, size: int):
    """Dense array indexed by key -> Sparse{map, size}: zeros are never stored (math/Sparse.scala:108-118)."""
    M = messages()
    out = M["Sparse"]()
    out.size = size
    arr = np.asarray(dense)
    nz = np.flatnonzero(np.abs(arr) > host.SPARSE_EPSILON)
    for k in nz:
        out.map[int(k)] = float(arr[k])
    return out


def from_sparse(sparse, dp: int) -> np.ndarray:
    """Sparse -> dense float32[dp] indexed by key; a key outside [0, dp) is what Vec.apply would reject later."""
    w = np.zeros(dp, dtype=np.float32)
    for k, v in sparse.map.items():
        if k < 0 or k >= dp:
            raise IndexError("key %d outside [0, %d)" % (k, dp))
        w[k] = v
    return w


# ---- stubs ---------------------------------------------------------------------------------------------------------
class Stub:
    """Client for one of the two services: `Stub(channel, "Slave").Gradient(request)`."""

    def __init__(self, channel, service: str):
        M = messages()
        table = SLAVE_METHODS if service == "Slave" else MASTER_METHODS
        for method, (req, rep) in table.items():
            call = channel.unary_unary("/%s.%s/%s" % (PACKAGE, service, method),
                                       request_serializer=M[req].SerializeToString,
                                       response_deserializer=M[rep].FromString)
            setattr(self, method, call)


def new_channel(host_: str, port: int):
    import grpc

    return grpc.insecure_channel("%s:%d" % (host_, port))  # core/package.scala:20-21: usePlaintext


# ---- the worker -------------------------------------------------------------------------------------------------------
class SlaveWorker:
    """core/Slave.scala: one worker process.  `backend` holds the resident data (rows are indexed as in the
    reference: positions in the train array every node loads, Main.scala:138,149) and the model."""

    def __init__(self, backend, n_features: int, host_: str = "127.0.0.1", port: int = 0, master: Optional[Tuple[str, int]] = None,
                 asynchronous: bool = False, rnd: Optional[host.JavaRandom] = None, metrics: Optional[host.Metrics] = None,
                 max_workers: int = 8):
        import grpc

        self.backend, self.size, self.dp = backend, n_features, n_features + 1
        self.asynchronous = asynchronous
        self.rnd = rnd or host.JavaRandom(0)
        self.metrics = metrics or host.Metrics()
        self.others: Dict[Tuple[str, int], Stub] = {}
        self.lock = threading.Lock()
        self.running_async = False
        self._thread: Optional[threading.Thread] = None
        self._async_error: Optional[BaseException] = None
        self.master_addr = master
        self.master_stub = Stub(new_channel(*master), "Master") if master else None
        M = messages()
        handlers = {}
        for method, (req, rep) in SLAVE_METHODS.items():
            fn = getattr(self, "_rpc_" + method)
            handlers[method] = grpc.unary_unary_rpc_method_handler(self._guard(fn), request_deserializer=M[req].FromString,
                                                                   response_serializer=M[rep].SerializeToString)
        # utils/Pool.scala:13: the reference serves from a fixed pool of 8 threads
        self.server = grpc.server(futures.ThreadPoolExecutor(max_workers=max_workers))
        self.server.add_generic_rpc_handlers((grpc.method_handlers_generic_handler("%s.Slave" % PACKAGE, handlers),))
        self.host = host_
        self.port = self.server.add_insecure_port("%s:%d" % (host_, port))

    # -- lifecycle (core/Slave.scala:36-76) ------------------------------------------------------------------------
    def start(self):
        self.server.start()
        if self.master_stub is not None:
            self.master_stub.RegisterSlave(messages()["Node"](host=self.host, port=self.port))
        return self

    def stop(self):
        self.running_async = False
        if self._thread is not None:
            self._thread.join(timeout=10)
        if self.master_stub is not None:
            try:
                self.master_stub.UnregisterSlave(messages()["Node"](host=self.host, port=self.port))
            except Exception:  # the master may be gone already
                pass
        self.server.stop(grace=None)

    @staticmethod
    def _guard(fn: Callable):
        """`require` failures and model exceptions fail the RPC, as an exception inside the handler's Future does."""
        import grpc

        def wrapped(request, context):
            try:
                return fn(request)
            except (ValueError, IndexError, RuntimeError) as e:
                context.abort(grpc.StatusCode.UNKNOWN, "%s: %s" % (type(e).__name__, e))

        return wrapped

    # -- SlaveImpl (core/Slave.scala:113-198) ----------------------------------------------------------------------
    def _rpc_RegisterSlave(self, node):
        with self.lock:
            self.others[(node.host, node.port)] = Stub(new_channel(node.host, node.port), "Slave")
        return messages()["Ack"]()

    def _rpc_UnregisterSlave(self, node):
        with self.lock:
            self.others.pop((node.host, node.port), None)
        return messages()["Ack"]()

    def _rpc_Forward(self, request):  # :129-140
        w = from_sparse(request.weights, self.dp)
        idx = np.asarray(request.samples, dtype=np.int32)
        self.metrics.counter("slave.sync.forward", len(idx))
        pred = self.backend.forward(idx, w)
        return messages()["ForwardReply"](predictions=[float(p) for p in pred])

    def _rpc_Gradient(self, request):  # :142-157
        w = from_sparse(request.weights, self.dp)
        idx = np.asarray(request.samples, dtype=np.int32)
        if len(idx) == 0:
            raise ValueError("requirement failed: Vec.sum of an empty batch (math/Vec.scala:129)")
        self.metrics.counter("slave.sync.backward", len(idx))
        g, _ = self.backend.gradient(idx, w)
        return messages()["GradUpdate"](gradUpdate=to_sparse(g, self.size))

    def _rpc_StartAsync(self, request):  # :159-175
        if not self.asynchronous:
            raise ValueError("requirement failed: Cannot initialize async computation: slave is in synchronous mode.")
        if self.running_async:
            raise ValueError("requirement failed: Async computation already running, can't be initialized unless stopped first")
        self.backend.set_weights(from_sparse(request.weights, self.dp))
        self.assigned = np.asarray(request.samples, dtype=np.int32)
        self.batch_size, self.learning_rate = int(request.batchSize), float(request.learningRate)
        self.running_async = True
        self._thread = threading.Thread(target=self._async_task, name="slave-async", daemon=True)
        self._thread.start()
        return messages()["Ack"]()

    def _rpc_UpdateGrad(self, request):  # :177-185
        if not self.asynchronous:
            raise ValueError("requirement failed: Cannot update gradient: slave is in synchronous mode.")
        keys = np.fromiter(request.gradUpdate.map.keys(), dtype=np.int32, count=len(request.gradUpdate.map))
        vals = np.fromiter(request.gradUpdate.map.values(), dtype=np.float32, count=len(request.gradUpdate.map))
        self.backend.update_grad(keys, vals)  # weights - gradUpdate
        self.metrics.counter("slave.async.grad.update")
        return messages()["Ack"]()

    def _rpc_StopAsync(self, request):  # :187-196
        if not self.asynchronous:
            raise ValueError("requirement failed: Cannot stop async computation: slave is in synchronous mode.")
        self.running_async = False
        return messages()["Ack"]()

    # -- asyncTask (core/Slave.scala:79-111) -----------------------------------------------------------------------
    def _sample(self) -> np.ndarray:
        n = len(self.assigned)
        if self.batch_size == 1:
            return np.asarray([self.assigned[self.rnd.next_int(n)]], dtype=np.int32)
        # `Random.shuffle(assignedSamples.indices) take batchSize map data`: the POSITIONS are used as data indices
        # (the indexing bug noted in SURVEY.md 3.4) -- reproduced, not fixed
        order = host.scala_shuffle(list(range(n)), self.rnd)
        return np.asarray(order[:self.batch_size], dtype=np.int32)

    def _async_task(self):
        GradUpdate = messages()["GradUpdate"]
        pending: list = []
        try:
            while self.running_async:
                idx = self._sample()
                self.metrics.counter("slave.async.backward", len(idx))
                delta, _ = self.backend.async_step(idx, self.learning_rate, want_delta=True)
                update = GradUpdate(gradUpdate=to_sparse(delta, self.size))
                with self.lock:
                    others = list(self.others.values())
                # otherSlaves.values.foreach(_.updateGrad(...)); masterStub.updateGrad(...): fire and forget.  (A grpc
                # future that is garbage-collected before it completes is CANCELLED: keep it until done.)
                pending = [f for f in pending if not f.done()]
                for stub in others:
                    pending.append(stub.UpdateGrad.future(update))
                if self.master_stub is not None:
                    pending.append(self.master_stub.UpdateGrad.future(update))
                self.metrics.counter("slave.async.batch")
            for f in pending:       # stopAsync: let the updates already computed reach their destinations
                try:
                    f.result(timeout=5)
                except Exception:
                    pass
        except BaseException as e:  # surfaced by tests / callers; the reference logs and dies
            self._async_error = e
            self.running_async = False


# ---- a minimal master for tests and Python-only deployments (core/Master.scala:218-262) -----------------------------
class MasterService:
    """Registration bookkeeping of AbstractMasterGrpc: keeps a stub per slave and cross-registers the colleagues;
    `UpdateGrad` hands async updates to `on_update` (MasterAsync.updateGrad, core/MasterAsync.scala:164-177)."""

    def __init__(self, expected_nodes: int, host_: str = "127.0.0.1", port: int = 0,
                 on_update: Optional[Callable[[Dict[int, float]], None]] = None, max_workers: int = 8):
        import grpc

        self.expected, self.on_update = expected_nodes, on_update
        self.slaves: Dict[Tuple[str, int], Stub] = {}
        self.lock = threading.Lock()
        self.ready = threading.Event()
        M = messages()
        handlers = {}
        for method, (req, rep) in MASTER_METHODS.items():
            handlers[method] = grpc.unary_unary_rpc_method_handler(getattr(self, "_rpc_" + method), request_deserializer=M[req].FromString,
                                                                   response_serializer=M[rep].SerializeToString)
        self.server = grpc.server(futures.ThreadPoolExecutor(max_workers=max_workers))
        self.server.add_generic_rpc_handlers((grpc.method_handlers_generic_handler("%s.Master" % PACKAGE, handlers),))
        self.host = host_
        self.port = self.server.add_insecure_port("%s:%d" % (host_, port))

    def start(self):
        self.server.start()
        return self

    def stop(self):
        self.server.stop(grace=None)

    def _rpc_RegisterSlave(self, node, context):
        M = messages()
        with self.lock:
            if len(self.slaves) > self.expected:
                import grpc
                context.abort(grpc.StatusCode.UNKNOWN, "requirement failed: too many nodes have joined")
            snap = dict(self.slaves)
            stub = Stub(new_channel(node.host, node.port), "Slave")
            self.slaves[(node.host, node.port)] = stub
            n = len(self.slaves)
        for (h, p), other in snap.items():   # every pair of slaves learns about each other (:229-233)
            other.RegisterSlave(node)
            stub.RegisterSlave(M["Node"](host=h, port=p))
        if n >= self.expected:
            self.ready.set()
        return M["Ack"]()

    def _rpc_UnregisterSlave(self, node, context):
        M = messages()
        with self.lock:
            self.slaves.pop((node.host, node.port), None)
            rest = list(self.slaves.values())
        for other in rest:
            other.UnregisterSlave(node)
        return M["Ack"]()

    def _rpc_UpdateGrad(self, request, context):
        if self.on_update is not None:
            self.on_update(dict(request.gradUpdate.map))
        return messages()["Ack"]()


# ---- the master's side of the synchronous protocol, for JVM-free deployments ------------------------------------------
class WireBackend:
    """`host.MasterSync(WireBackend(...), ...)` is the reference's synchronous master over the wire: one `Gradient` RPC
    per worker and batch (`core/Master.scala:184-190`), `Vec.mean`, `w - lr * mean` (:192-197) on the master's own
    copy of the weights; loss and accuracy through `Forward` RPCs fanned out over the slaves
    (`predict` / `distributedLoss` / `distributedAccuracy`, `core/Master.scala:60-98`) -- the master needs the labels
    only.  Worker k of a batch is served by slave k (zip of splits and slaves, :184)."""

    def __init__(self, slaves: Sequence[Stub], n_features: int, lam: float, labels):
        self.slaves, self.size, self.dp, self.lam = list(slaves), n_features, n_features + 1, lam
        self.labels = np.asarray(labels, dtype=np.float64)
        self.w = np.zeros(self.dp, dtype=np.float64)

    def set_weights(self, w):
        self.w = np.asarray(w, dtype=np.float64).copy()

    def get_weights(self):
        return self.w.copy()

    def _dense(self, sparse) -> np.ndarray:
        g = np.zeros(self.dp, dtype=np.float64)
        for k, v in sparse.map.items():
            g[k] = v
        return g

    def sync_step(self, idx_lists, lr):
        M = messages()
        if len(idx_lists) > len(self.slaves):
            raise ValueError("%d workers in the batch but %d slaves registered" % (len(idx_lists), len(self.slaves)))
        weights = to_sparse(self.w, self.size)
        calls = [stub.Gradient.future(M["GradientRequest"](weights=weights, samples=[int(i) for i in idx]))
                 for stub, idx in zip(self.slaves, idx_lists)]
        total = np.zeros(self.dp, dtype=np.float64)
        for c in calls:                      # Future.sequence: a failed RPC fails the batch
            total += self._dense(c.result().gradUpdate)
        self.w = self.w - lr * (total / len(idx_lists))   # Vec.mean, then weights - learningRate * grad
        return {"n_samples": sum(len(i) for i in idx_lists), "n_active": None}

    def loss_acc(self, lo, hi):
        M = messages()
        rows = np.arange(lo, hi)
        if len(rows) == 0:
            raise ValueError("requirement failed: empty row range")
        parts = [p for p in np.array_split(rows, len(self.slaves)) if len(p)]
        weights = to_sparse(self.w, self.size)
        calls = [stub.Forward.future(M["ForwardRequest"](samples=[int(i) for i in p], weights=weights))
                 for stub, p in zip(self.slaves, parts)]
        pred = np.concatenate([np.asarray(c.result().predictions, dtype=np.float64) for c in calls])
        y = self.labels[lo:hi]
        hinge = np.maximum(0.0, 1.0 - y * pred)                     # core/ml/SparseSVM.scala:16
        counts = [int((pred == y).sum()), int((pred == 0).sum()), int((pred == -y).sum())]
        return self.lam * float(self.w @ self.w) + float(hinge.mean()), counts[0] / len(rows), counts
