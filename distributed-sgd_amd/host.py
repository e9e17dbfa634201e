"""Host-side mirror of the reference's coordination layer for the hot path.

Same names, argument meaning and stopping behaviour as the Scala classes, so that parity tests read like
the reference.  Citations are relative to /root/reference/src/main/scala/epfl/distributed/.

    Config            utils/Config.scala:3-21 + resources/application.conf:1-52 (DSGD_* overrides)
    SplitStrategy     core/ml/SplitStrategy.scala:13-14
    EarlyStopping     core/ml/EarlyStopping.scala:11-46
    GradState         core/ml/GradState.scala:6-24  (`grad` holds the WEIGHTS)
    JavaRandom        java.util.Random (the generator behind scala.util.Random, seeded 0 at Main.scala:32)
    scala_shuffle     scala.util.Random.shuffle (2.12): the per-batch reshuffle of Master.scala:184
    MasterSync.fit    core/Master.scala:120-218
    MasterAsync.fit   core/MasterAsync.scala:32-62, 96-177
    Metrics           the Kamon instruments of the path, same names (core/Slave.scala:90-181, core/Master.scala:150-183)
    format_final_weights   the `final weights: idx:val ...` log line (Main.scala:114)

All arithmetic happens behind a *backend* (the HIP `Engine`); this module only orchestrates.  A backend
offers: sync_step(idx_lists, lr), gradient(idx) -> (g, stats), apply(g_mean, lr), loss_acc(lo, hi),
get_weights(), set_weights(w), optionally resident plans (plan_flat(idx, offsets, n_steps, n_workers) -> plan with
.destroy(), plan_run(plan, step_begin, step_end, lr), synchronize()): an epoch of Master.fit then runs as ONE plan; and for
the asynchronous mode async_start/async_updates/async_stop.
"""

from __future__ import annotations

import math
import os
import time
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np


SPARSE_EPSILON = 1e-20  # math/Sparse.scala:104: entries with abs(v) <= epsilon are not stored


# ---- instruments (Kamon names, so dashboards such as kube/monitor.yaml keep working) ----------------------
class Metrics:
    """Counters / histograms / timers under the reference's instrument names:
    counters `slave.sync.forward`, `slave.sync.backward` (one increment per SAMPLE, core/Slave.scala:131-150),
    `slave.async.backward` (per sample, :90-95), `slave.async.batch` (:107), `slave.async.grad.update` (:181),
    `master.async.loss` (MasterAsync.scala:126); histograms `master.sync.loss`, `master.sync.acc`
    (Master.scala:150-151 record `losses.head.toLong` and `100 * accs.head.toLong`: the values are TRUNCATED to
    integers before they are recorded -- kept); timer `master.sync.batch.duration` (:183)."""

    def __init__(self):
        import threading

        self._lock = threading.Lock()
        self.counters: dict = {}
        self.histograms: dict = {}

    def counter(self, name: str, times: int = 1):
        with self._lock:
            self.counters[name] = self.counters.get(name, 0) + int(times)

    def histogram(self, name: str, value):
        with self._lock:
            self.histograms.setdefault(name, []).append(int(value))  # .toLong

    def timer(self, name: str):
        metrics = self

        class _T:
            def __enter__(self):
                self.t0 = time.perf_counter_ns()
                return self

            def __exit__(self, *exc):
                with metrics._lock:
                    metrics.histograms.setdefault(name, []).append(time.perf_counter_ns() - self.t0)

        return _T()

    def snapshot(self) -> dict:
        with self._lock:
            return {"counters": dict(self.counters), "histograms": {k: list(v) for k, v in self.histograms.items()}}

    def influx_lines(self, tags: str = "") -> List[str]:
        """InfluxDB line protocol (the reference reports through kamon-influxdb, Main.scala:42)."""
        snap, t = self.snapshot(), time.time_ns()
        tag = ("," + tags) if tags else ""
        out = ["%s%s count=%di %d" % (k, tag, v, t) for k, v in sorted(snap["counters"].items())]
        for k, v in sorted(snap["histograms"].items()):
            if v:
                out.append("%s%s count=%di,sum=%di,min=%di,max=%di %d" % (k, tag, len(v), sum(v), min(v), max(v), t))
        return out


def format_final_weights(w, eps: float = SPARSE_EPSILON) -> str:
    """Main.scala:114: `w1.map.map { case (idx, n) => s"$idx:$n" }.mkString(" ")` -- the stored (non-zero) entries as
    `idx:value`.  The reference iterates an immutable HashMap (unspecified order) and prints fp64; here ascending keys
    and the shortest decimal that round-trips the engine's fp32 value."""
    arr = np.asarray(w)
    return " ".join("%d:%s" % (int(k), repr(float(np.float32(arr[k])))) for k in np.flatnonzero(np.abs(arr) > eps))


# ---- configuration --------------------------------------------------------------------------------------
@dataclass
class Config:
    """utils/Config.scala:3-21; defaults are application.conf:1-52."""

    data_path: str = "data"
    host: str = "127.0.0.1"
    port: int = 4000
    master_host: Optional[str] = None
    master_port: Optional[int] = None
    batch_size: int = 100
    learning_rate: float = 0.5
    lambda_: float = 1e-5
    full: bool = False
    node_count: int = 3
    async_: bool = False
    record: bool = False
    max_epochs: int = 10
    check_every: int = 100
    leaky_loss: float = 0.9
    patience: int = 5
    conv_delta: float = 0.01

    # application.conf key -> (field, env override, parser)
    _KEYS = {
        "data-path": ("data_path", "DSGD_DATA_PATH", str),
        "host": ("host", "DSGD_NODE_HOST", str),
        "port": ("port", "DSGD_NODE_PORT", int),
        "master-host": ("master_host", "DSGD_MASTER_HOST", str),
        "master-port": ("master_port", "DSGD_MASTER_PORT", int),
        "batch-size": ("batch_size", "DSGD_BATCH_SIZE", int),
        "learning-rate": ("learning_rate", "DSGD_LEARNING_RATE", float),
        "lambda": ("lambda_", "DSGD_LAMBDA", float),
        "full": ("full", "DSGD_FULL", lambda s: str(s).lower() == "true"),
        "node-count": ("node_count", "DSGD_NODE_COUNT", int),
        "async": ("async_", "DSGD_ASYNC", lambda s: str(s).lower() == "true"),
        "record": ("record", "DSGD_RECORD", lambda s: str(s).lower() == "true"),
        "max-epochs": ("max_epochs", "DSGD_MAX_EPOCHS", int),
        "check-every": ("check_every", "DSGD_CHECK_EVERY", int),
        "leaky-loss": ("leaky_loss", "DSGD_LEAKY_LOSS", float),
        "patience": ("patience", "DSGD_PATIENCE", int),
        "conv-delta": ("conv_delta", "DSGD_CONV_DELTA", float),
    }

    @classmethod
    def load(cls, conf_text: Optional[str] = None, env=os.environ) -> "Config":
        """`key = value` lines of the dsgd { ... } block, then the DSGD_* environment overrides
        (`key = ${?DSGD_X}` lines of application.conf)."""
        cfg = cls()
        if conf_text:
            depth, in_dsgd = 0, False
            for raw in conf_text.splitlines():
                line = raw.split("#", 1)[0].strip()
                if not line:
                    continue
                if line.startswith("dsgd") and line.endswith("{"):
                    in_dsgd, depth = True, 1
                    continue
                if in_dsgd:
                    depth += line.count("{") - line.count("}")
                    if depth <= 0:
                        in_dsgd = False
                        continue
                    if "=" in line:
                        k, v = (t.strip() for t in line.split("=", 1))
                        if k in cls._KEYS and not v.startswith("${"):
                            name, _, parse = cls._KEYS[k]
                            setattr(cfg, name, parse(v.strip('"')))
        for k, (name, var, parse) in cls._KEYS.items():
            if var in env and env[var] != "":
                setattr(cfg, name, parse(env[var]))
        return cfg

    def role(self) -> str:
        """Main.scala:122-159: master-host/port equal to own -> master; set but different -> slave; unset -> dev."""
        if self.master_host is not None and self.master_port is not None:
            return "master" if (self.master_host == self.host and self.master_port == self.port) else "slave"
        return "dev"


# ---- split / stopping / state -----------------------------------------------------------------------------
def split_vanilla(n: int, n_slaves: int) -> List[range]:
    """SplitStrategy.vanilla: indices.grouped(ceil(n / K)); may yield fewer than K groups."""
    size = int(math.ceil(n / float(n_slaves)))
    return [range(b, min(n, b + size)) for b in range(0, n, size)]


class EarlyStopping:
    @staticmethod
    def target(target: float) -> Callable[[Sequence[float]], bool]:
        return lambda losses: bool(losses) and losses[0] <= target  # EarlyStopping.scala:11

    @staticmethod
    def no_improvement(patience: int = 5, min_delta: float = 1e-3, min_steps: Optional[int] = None):
        """EarlyStopping.scala:13-46 over a NEWEST-FIRST list."""

        def crit(losses: Sequence[float]) -> bool:
            abs_min_delta = abs(min_delta)

            def check() -> bool:
                mn, idx_min = 1.7976931348623157e308, -1
                for index, num in enumerate(losses):
                    if (num - mn) <= abs_min_delta:
                        mn, idx_min = num, index
                return False if idx_min == 0 else idx_min >= patience

            if not losses:
                return False
            if min_steps is None:
                return check()
            return False if min_steps < len(losses) else check()

        return crit


@dataclass
class GradState:
    """GradState.scala:6-24 -- `grad` is the weight vector."""

    grad: np.ndarray
    loss: Optional[float] = None
    start: float = field(default_factory=time.time)
    updates: int = 0
    end: Optional[float] = None

    def replace_grad(self, new_grad) -> "GradState":
        return GradState(new_grad, self.loss, self.start, self.updates + 1, self.end)

    def finish(self, final_loss) -> "GradState":
        return GradState(self.grad, final_loss, self.start, self.updates, time.time())


# ---- the reference's random stream ---------------------------------------------------------------------------
class JavaRandom:
    """java.util.Random: 48-bit LCG; scala.util.Random delegates to it (seed 0 at Main.scala:32)."""

    def __init__(self, seed: int = 0):
        self.seed = (seed ^ 0x5DEECE66D) & ((1 << 48) - 1)

    def _next(self, bits: int) -> int:
        self.seed = (self.seed * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
        v = self.seed >> (48 - bits)
        return v - (1 << 32) if (bits == 32 and v >= (1 << 31)) else v  # (int) cast of the Java original

    def next_int(self, bound: Optional[int] = None) -> int:
        if bound is None:
            return self._next(32)
        if bound <= 0:
            raise ValueError("bound must be positive")
        r = self._next(31)
        m = bound - 1
        if (bound & m) == 0:
            return (bound * r) >> 31
        u = r
        while True:
            r = u % bound
            if u - r + m < (1 << 31):
                return r
            u = self._next(31)


def scala_shuffle(xs: Sequence[int], rnd: JavaRandom) -> List[int]:
    """scala.util.Random.shuffle (2.12): for n <- len to 2 by -1: swap(n - 1, nextInt(n))."""
    buf = list(xs)
    for n in range(len(buf), 1, -1):
        k = rnd.next_int(n)
        buf[n - 1], buf[k] = buf[k], buf[n - 1]
    return buf


# ---- the same stream, natively (csrc/jrand.c -> lib/libdsgd_host.so) -----------------------------------------------
_HOST_LIB = None


def _host_lib():
    """libdsgd_host.so or None (not built: the pure-Python restatement above is the fallback -- HOST logic, not the
    compute path; the product's kernels have no fallback)."""
    global _HOST_LIB
    if _HOST_LIB is None:
        import ctypes as C

        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libdsgd_host.so")
        try:
            lib = C.CDLL(path)
            lib.dsgd_jrand_seed.restype = C.c_uint64
            lib.dsgd_jrand_seed.argtypes = [C.c_int64]
            lib.dsgd_jrand_shuffle.restype = C.c_int64
            lib.dsgd_jrand_epoch_lists.restype = C.c_int
            _HOST_LIB = lib
        except OSError:
            _HOST_LIB = False
    return _HOST_LIB or None


def epoch_lists(rnd: JavaRandom, split: Sequence[range], max_samples: int, batch_size: int, native: Optional[bool] = None):
    """The index lists of ONE epoch of Master.fit (core/Master.scala:179-199), drawn from `rnd` exactly as the reference
    draws them -- for every batch b in (0 until maxSamples by batchSize), for every worker k in order:
    Random.shuffle(split_k).slice(b, b + batchSize) (:184: a fresh shuffle of the WHOLE split per worker and batch).

    Returns (idx int32 [total], offsets int64 [n_steps * K + 1], n_steps): the flat form dsgd_plan_create takes.  Steps are
    emitted up to the first one in which some worker's slice is empty (Vec.sum would throw in that slave, math/Vec.scala:129);
    `rnd` advances over the steps emitted.  Natively (csrc/jrand.c: the epoch's shuffles in parallel, draw for draw the
    same stream) when lib/libdsgd_host.so is there; `native=False` forces the pure-Python restatement."""
    K = len(split)
    lib = _host_lib() if native in (None, True) else None
    if native is True and lib is None:
        raise RuntimeError("libdsgd_host.so is not built")
    # (an empty split takes the Python path, which stops where the reference's Vec.sum throws; a batch size beyond the
    #  longest split selects whole splits either way -- clamped: the native side takes it as an int32)
    if lib is not None and all(r.step == 1 and len(r) > 0 for r in split):
        import ctypes as C

        batch_size = max(1, min(int(batch_size), max(max_samples, max(len(r) for r in split)), 2 ** 31 - 1))
        n_steps_max = len(range(0, max_samples, batch_size))
        sb = np.asarray([r.start for r in split], dtype=np.int64)
        se = np.asarray([r.stop for r in split], dtype=np.int64)
        cap = n_steps_max * sum(min(batch_size, len(r)) for r in split)
        idx = np.empty(max(cap, 1), dtype=np.int32)
        offsets = np.zeros(n_steps_max * K + 1, dtype=np.int64)
        state = C.c_uint64(rnd.seed)
        n_steps, draws = C.c_int64(0), C.c_int64(0)
        rc = lib.dsgd_jrand_epoch_lists(C.byref(state), sb.ctypes.data_as(C.c_void_p), se.ctypes.data_as(C.c_void_p), C.c_int32(K),
                                        C.c_int64(max_samples), C.c_int32(batch_size), idx.ctypes.data_as(C.c_void_p),
                                        offsets.ctypes.data_as(C.c_void_p), C.byref(n_steps), C.byref(draws))
        if rc != 0:
            raise RuntimeError("dsgd_jrand_epoch_lists failed (%d)" % rc)
        rnd.seed = int(state.value)
        ns = int(n_steps.value)
        return idx[:int(offsets[ns * K])], offsets[:ns * K + 1], ns
    flat, offs, n_steps = [], [0], 0
    for batch in range(0, max_samples, batch_size):
        if any(batch >= len(r) for r in split):
            break
        for r in split:
            shuffled = scala_shuffle(list(r), rnd)
            flat.append(np.asarray(shuffled[batch:batch + batch_size], dtype=np.int32))
            offs.append(offs[-1] + len(flat[-1]))
        n_steps += 1
    idx = np.concatenate(flat) if flat else np.zeros(0, dtype=np.int32)
    return idx, np.asarray(offs, dtype=np.int64), n_steps


# ---- distributed aggregate owned by the host (alternative to the in-library RCCL all-reduce) -----------------
class HostAllReduceBackend:
    """Mean over world_size x hosted workers with the collective owned by the host (`torch.distributed`,
    gloo on CPU / nccl on GPU).  core/Master.scala:190-197: every rank computes the regularised sums of its
    own workers, the sums are all-reduced, every rank applies the identical update."""

    def __init__(self, local, dist_module, n_rows_local_train: int):
        self.local, self.dist = local, dist_module
        self.world = dist_module.get_world_size()
        self.n_train = n_rows_local_train

    def sync_step(self, idx_lists, lr):
        import torch

        total = None
        stats = {"n_samples": 0, "n_active": 0}
        for idx in idx_lists:
            g, st = self.local.gradient(idx)
            total = g.astype(np.float64) if total is None else total + g
            stats["n_samples"] += st["n_samples"]
            stats["n_active"] += st["n_active"]
        t = torch.from_numpy(np.ascontiguousarray(total))
        self.dist.all_reduce(t)  # sum over ranks
        k_total = len(idx_lists) * self.world
        self.local.apply((t.numpy() / k_total).astype(np.float32), lr)  # Vec.mean then w - lr * grad
        return stats

    def loss_acc(self, lo, hi):
        import torch

        _, _, counts = self.local.loss_acc(lo, hi)
        t = torch.tensor(list(counts) + [hi - lo], dtype=torch.int64)
        self.dist.all_reduce(t)
        c0, c1, c2, n = (int(x) for x in t)
        w = self.local.get_weights().astype(np.float64)
        lam = getattr(self.local, "lam")
        return lam * float((w * w).sum()) + (c1 + 2.0 * c2) / n, c0 / n, [c0, c1, c2]

    def get_weights(self):
        return self.local.get_weights()

    def set_weights(self, w):
        self.local.set_weights(w)


class HostAsyncExchange:
    """The asynchronous mode across ranks with the collective owned by the host: the host-side twin of
    `dsgd_async_set_exchange` (csrc/dsgd_hip.hip, RCCL inside the library), usable with gloo on CPU.

    core/Slave.scala:99-105 applies every update locally and gossips it to every peer, who subtracts it when it
    arrives (:177-185).  One replica per rank runs `every` local updates per round; between rounds the replicas
    all-reduce what each subtracted since the last exchange (d_local = w_prev - w) and subtract their PEERS' part
    (d_sum - d_local) on top of their own.  With one rank the peers' part is exactly zero."""

    def __init__(self, local, dist_module):
        self.local, self.dist = local, dist_module
        self.world = dist_module.get_world_size()
        self.w_prev = np.asarray(local.get_weights(), dtype=np.float64).copy()
        self.rounds = 0

    def run_round(self, idx_lists, lr):
        """`idx_lists`: the sample lists of this rank's local updates of the round, in order (Slave.asyncTask draws
        them itself; they are an input here so that a run can be replayed)."""
        stats = {"n_samples": 0, "n_active": 0}
        for idx in idx_lists:
            _, st = self.local.async_step(idx, lr)
            stats["n_samples"] += st["n_samples"]
            stats["n_active"] += st["n_active"]
        self.exchange()
        return stats

    def exchange(self):
        import torch

        w = np.asarray(self.local.get_weights(), dtype=np.float64)
        d_local = self.w_prev - w
        t = torch.from_numpy(np.ascontiguousarray(d_local.copy()))
        self.dist.all_reduce(t)  # sum over ranks
        others = t.numpy() - d_local   # exactly 0 with a single rank
        if self.world > 1 or np.any(others != 0.0):
            w = w - others
            self.local.set_weights(w)
            # what the backend really holds now: an fp32 engine rounds on store, and a w_prev that kept the fp64 value
            # would gossip that rounding residue to the peers as if it were an update (the device kernel sets its
            # w_prev to the rounded value too)
            w = np.asarray(self.local.get_weights(), dtype=np.float64)
        self.w_prev = w.copy()
        self.rounds += 1


# ---- Master.fit (synchronous) ---------------------------------------------------------------------------------
class MasterSync:
    """core/Master.scala:120-218 for the workers hosted behind one backend.

    data layout: rows [0, n_train) are the train set, [n_train, n_rows) the test set (Main.scala:52)."""

    def __init__(self, backend, n_train: int, n_rows: int, node_count: int, rnd: Optional[JavaRandom] = None, log=None,
                 metrics: Optional[Metrics] = None, plans: Optional[bool] = None, prefetch: bool = True):
        """plans: run an epoch's batches as ONE resident plan of the backend (Engine.plan_flat / plan_run: the column-slice
        kernel runs all of the epoch's steps in one launch, 5 us per 3 x 100 step) instead of one backend.sync_step call per
        batch (32 us each); None = whenever the backend offers plans.  The random stream, the lists, the order of the
        steps and the arithmetic are the same either way; what changes is when the host sees a batch: the per-batch log
        lines and the `master.sync.batch.duration` timer are written after the epoch's launch (one entry per batch, the
        epoch's time shared evenly).  prefetch: while an epoch's steps run, the next epoch's plan is laid out and the lists of
        the epoch after it are drawn on a helper thread (if fit stops early the stream is put back to where the reference's
        would be: nothing of an epoch that never ran is kept)."""
        self.backend, self.n_train, self.n_rows, self.node_count = backend, n_train, n_rows, node_count
        self.metrics = metrics or Metrics()
        self.rnd = rnd or JavaRandom(0)
        self._logging = log is not None
        self.log = log or (lambda *a: None)
        self.plans = hasattr(backend, "plan_flat") if plans is None else bool(plans)
        # the epoch's lists drawn BY THE DEVICE, draw for draw the same stream (Engine.plan_from_seed, csrc/dsgd_shuffle.hpp):
        # 14 ms per epoch of RCV1 at full = true against 195 ms on 32 host threads; off by itself where the device form does
        # not apply (batches beyond 1,024 rows, splits beyond 2^20 rows) or DSGD_DEVICE_LISTS=0
        self.device_lists = self.plans and hasattr(backend, "plan_from_seed") and os.environ.get("DSGD_DEVICE_LISTS", "1") != "0"
        # ... and only for epochs of at least this many draws: the device form has ~1.5 ms of fixed cost per epoch (two
        # synchronisations, the host's walk over the candidates), the host draws 7 G values per second -- the full = false
        # configuration (1.15 M draws per epoch) is cheaper on the host (profiles/r06_fit_loop.txt)
        self.device_lists_min_draws = int(os.environ.get("DSGD_DEVICE_LISTS_MIN_DRAWS", 8 << 20))
        self.prefetch = prefetch
        self.losses: List[float] = []
        self.accs: List[float] = []
        self.test_losses: List[float] = []
        self.test_accs: List[float] = []
        self._pending = None
        self._lists_future = None
        self._executor = None
        self.batch_loop_s = 0.0      # wall time of the batch loops (shuffles, plan set-up, steps): Master.scala:179-199
        self.shuffle_s = 0.0         # ... of which drawing the lists (not overlapped by prefetch: see fit)
        self.steps_run = 0

    def local_loss(self, test: bool = False):  # Master.scala:104-106
        lo, hi = (self.n_train, self.n_rows) if test else (0, self.n_train)
        return self.backend.loss_acc(lo, hi)[0]

    def local_accuracy(self, test: bool = False):  # Master.scala:100-102
        lo, hi = (self.n_train, self.n_rows) if test else (0, self.n_train)
        return self.backend.loss_acc(lo, hi)[1]

    def fit(self, initial_weights, max_epochs: int, batch_size: int, learning_rate: float,
            stopping_criterion: Callable[[Sequence[float]], bool]) -> GradState:
        split = split_vanilla(self.n_train, self.node_count)  # Master.scala:136
        max_samples = max(len(r) for r in split)              # :138
        self.backend.set_weights(np.asarray(initial_weights, dtype=np.float32))
        state = GradState(np.asarray(initial_weights, dtype=np.float32))
        self._pending = None   # (plans + prefetch) the next epoch's plan, laid out while the current one runs
        try:
            return self._fit_loop(state, 0, split, max_samples, max_epochs, batch_size, learning_rate, stopping_criterion)
        finally:
            self._drop_pending()

    def _drop_pending(self):
        """fit is over: whatever was drawn or laid out for epochs that never ran is dropped, and the stream goes back to
        where the reference's generator stands (the draws are undone oldest first)."""
        fut, self._lists_future = getattr(self, "_lists_future", None), None
        ahead = fut.result() if fut is not None else None
        p, self._pending = getattr(self, "_pending", None), None
        if p is not None:
            self.rnd.seed = p["seed_before"]
            if p["plan"] is not None:
                p["plan"].destroy()
        elif ahead is not None:
            self.rnd.seed = ahead["seed_before"]
        ex, self._executor = getattr(self, "_executor", None), None
        if ex is not None:
            ex.shutdown(wait=True)

    def _draw_lists(self, split, max_samples, batch_size):
        seed_before = self.rnd.seed
        t0 = time.perf_counter()
        idx, offsets, n_steps = epoch_lists(self.rnd, split, max_samples, batch_size)
        return {"idx": idx, "offsets": offsets, "n_steps": n_steps, "seed_before": seed_before, "shuffle_s": time.perf_counter() - t0}

    def _take_lists(self, split, max_samples, batch_size):
        """The next epoch's lists: drawn ahead by the helper thread if one is at it (the wait, if any, counts as shuffle time
        that was not hidden), otherwise drawn now."""
        fut, self._lists_future = getattr(self, "_lists_future", None), None
        t0 = time.perf_counter()
        lists = fut.result() if fut is not None else self._draw_lists(split, max_samples, batch_size)
        self.shuffle_s += time.perf_counter() - t0
        return lists

    def _draw_ahead(self, split, max_samples, batch_size):
        """Start drawing the lists of the epoch after the next on a helper thread (the native generator releases the GIL):
        the main thread lays the next epoch's plan out meanwhile.  One drawer at a time: the stream is sequential."""
        if getattr(self, "_executor", None) is None:
            from concurrent.futures import ThreadPoolExecutor

            self._executor = ThreadPoolExecutor(max_workers=1, thread_name_prefix="dsgd-lists")
        self._lists_future = self._executor.submit(self._draw_lists, split, max_samples, batch_size)

    def _make_plan(self, lists, n_workers):
        plan = self.backend.plan_flat(lists["idx"], lists["offsets"], lists["n_steps"], n_workers) if lists["n_steps"] else None
        return {"plan": plan, "n_steps": lists["n_steps"], "n_samples": int(lists["offsets"][lists["n_steps"] * n_workers]),
                "seed_before": lists["seed_before"]}

    def _next_plan(self, split, max_samples, batch_size, n_workers, ahead_ok=False):
        """The next epoch as a plan: its lists drawn by the device where that applies, else by the host (csrc/jrand.c)."""
        draws = len(range(0, max_samples, batch_size)) * sum(len(r) - 1 for r in split)
        if self.device_lists and draws >= self.device_lists_min_draws and all(r.step == 1 for r in split):
            seed_before = self.rnd.seed
            t0 = time.perf_counter()
            try:
                plan, n_steps, state, _ = self.backend.plan_from_seed(seed_before, split, max_samples, batch_size)
            except Exception as e:   # (DSGD_EUNSUPPORTED: outside the device form -- the host draws from here on)
                if getattr(e, "code", None) != -7:
                    raise
                self.device_lists = False
            else:
                self.rnd.seed = state
                self.shuffle_s += time.perf_counter() - t0
                return {"plan": plan, "n_steps": n_steps, "n_samples": plan.n_samples if plan is not None else 0, "seed_before": seed_before}
        lists = self._take_lists(split, max_samples, batch_size)
        if ahead_ok and lists["n_steps"] == len(range(0, max_samples, batch_size)):
            self._draw_ahead(split, max_samples, batch_size)
        return self._make_plan(lists, n_workers)

    def _epoch_through_a_plan(self, split, max_samples, batch_size, learning_rate, epochs_left):
        """One epoch's batch loop (core/Master.scala:179-199) as one resident plan."""
        K = len(split)
        n_expected = len(range(0, max_samples, batch_size))
        if batch_size >= max_samples and all(r.step == 1 for r in split) and hasattr(self.backend, "sync_step_ranges"):
            # batch-size >= split size: slice(0, batchSize) of a shuffled split is the WHOLE split, and a sum does not
            # depend on the order -- the step is a sum over contiguous row ranges (dsgd_sync_step_ranges: the streaming
            # kernels).  The shuffles are still drawn: the generator must stand where the reference's stands.
            t_sh = time.perf_counter()
            epoch_lists(self.rnd, split, max_samples, batch_size)
            self.shuffle_s += time.perf_counter() - t_sh
            t0 = time.perf_counter_ns()
            self.backend.sync_step_ranges([(r.start, r.stop) for r in split], learning_rate)
            self.log("samples %d - %d / %d" % (1, max_samples, max_samples))
            with self.metrics._lock:
                self.metrics.histograms.setdefault("master.sync.batch.duration", []).append(time.perf_counter_ns() - t0)
            self.metrics.counter("slave.sync.backward", sum(len(r) for r in split))
            self.steps_run += 1
            return
        cur = getattr(self, "_pending", None)
        self._pending = None
        if cur is None:
            cur = self._next_plan(split, max_samples, batch_size, K)
        t0 = time.perf_counter_ns()
        try:
            if cur["n_steps"]:
                self.backend.plan_run(cur["plan"], 0, cur["n_steps"], learning_rate)   # enqueued: ALL the epoch's steps, one launch
            if self.prefetch and epochs_left > 1 and cur["n_steps"] == n_expected:
                # ... and while they run: the next epoch's lists (drawn ahead already, from the second epoch on), the draw of the
                # epoch after it started on the helper thread, the next epoch's plan laid out (the device's build stream)
                self._pending = self._next_plan(split, max_samples, batch_size, K, ahead_ok=epochs_left > 2)
        finally:
            if cur["plan"] is not None:        # (also when the run or the next plan's set-up failed: the device blocks go back)
                cur["plan"].destroy()                                                # (behind the run; no synchronisation)
                cur["plan"] = None
        self.backend.synchronize()
        dt = time.perf_counter_ns() - t0
        # what the per-batch closure would have logged and recorded (:181-183, Slave.scala:145-150), written now
        if self._logging:
            for s_ in range(cur["n_steps"]):
                batch = s_ * batch_size
                self.log("samples %d - %d / %d" % (batch + 1, min(batch + batch_size, max_samples), max_samples))
        if cur["n_steps"]:
            with self.metrics._lock:   # one entry per batch, as the per-batch closure records them
                self.metrics.histograms.setdefault("master.sync.batch.duration", []).extend([dt // cur["n_steps"]] * cur["n_steps"])
            self.metrics.counter("slave.sync.backward", cur["n_samples"])
        self.steps_run += cur["n_steps"]
        if cur["n_steps"] < n_expected:
            # the reference's next batch hands some slave an empty slice: Vec.sum throws there (math/Vec.scala:129)
            raise ValueError("Cannot sum an empty list of vectors (batch %d of the epoch: a worker's slice is empty)" % cur["n_steps"])

    def _finished(self, state):
        if self.plans and state.updates:
            state = GradState(self.backend.get_weights(), state.loss, state.start, state.updates, state.end)
        return state.finish(self.losses[0] if self.losses else None)

    def _fit_loop(self, state, epoch, split, max_samples, max_epochs, batch_size, learning_rate, stopping_criterion):
        while True:
            if self.losses:
                self.log("loss after epoch %d: %s" % (epoch, self.losses[0]))
                self.log("acc after epoch %d: %s" % (epoch, self.accs[0]))
                self.metrics.histogram("master.sync.loss", self.losses[0])       # Master.scala:150 (.toLong)
                self.metrics.histogram("master.sync.acc", 100 * int(self.accs[0]))  # :151: 100 * accs.head.toLong
            if epoch >= max_epochs:                             # :154
                self.log("Reached max number of epochs: stopping computation")
                return self._finished(state)
            if stopping_criterion(self.test_losses):            # :166
                self.log("Converged to target: stopping computation")
                return self._finished(state)
            t_loop = time.perf_counter()
            if self.plans:
                self._epoch_through_a_plan(split, max_samples, batch_size, learning_rate, epochs_left=max_epochs - epoch)
            else:
                # :184 -- every worker's split is reshuffled for EVERY batch, then sliced.  Nothing else draws from the
                # generator inside the loop: the epoch's lists are drawn up front (the same draws in the same order;
                # natively when lib/libdsgd_host.so is there), then one request per batch
                t_sh = time.perf_counter()
                idx, offs, n_steps = epoch_lists(self.rnd, split, max_samples, batch_size)
                self.shuffle_s += time.perf_counter() - t_sh
                K = len(split)
                for s_, batch in enumerate(range(0, max_samples, batch_size)):    # :179
                    self.log("samples %d - %d / %d" % (batch + 1, min(batch + batch_size, max_samples), max_samples))   # :181
                    if s_ >= n_steps:
                        # a slice past the end of a short last split is empty: Vec.sum throws in that slave (math/Vec.scala:129)
                        raise ValueError("Cannot sum an empty list of vectors (batch %d of the epoch: a worker's slice is empty)" % s_)
                    lists = [idx[offs[s_ * K + j]:offs[s_ * K + j + 1]] for j in range(K)]
                    with self.metrics.timer("master.sync.batch.duration"):   # :183
                        st = self.backend.sync_step(lists, learning_rate)    # :186-197
                    self.steps_run += 1
                    if st:
                        self.metrics.counter("slave.sync.backward", st.get("n_samples", 0))  # Slave.scala:145-150
            if self.plans:
                # (the weights stay on the device between the epochs: GradState.grad is filled when fit returns -- the
                #  reference's master needs the vector every batch only because it ships it to the slaves)
                self.batch_loop_s += time.perf_counter() - t_loop
                state = state.replace_grad(state.grad)
            else:
                w = self.backend.get_weights()   # (synchronises: the epoch's steps are done)
                self.batch_loop_s += time.perf_counter() - t_loop
                state = state.replace_grad(w)
            # :206-209 -- four full passes per epoch; newest first
            l, a, _ = self.backend.loss_acc(0, self.n_train)
            tl, ta, _ = self.backend.loss_acc(self.n_train, self.n_rows)
            self.losses.insert(0, l)
            self.accs.insert(0, a)
            self.test_losses.insert(0, tl)
            self.test_accs.insert(0, ta)
            epoch += 1


# ---- MasterAsync.fit ------------------------------------------------------------------------------------------------
class MasterAsync:
    """core/MasterAsync.scala:32-177 on top of the lock-free engine: start the workers, check the test loss
    every `check_every` updates with a leaky average, keep the best weights, stop on the criterion or at
    maxSteps = n_train * max_epochs updates (MasterAsync.scala:83)."""

    def __init__(self, backend, n_train: int, n_rows: int, node_count: int, log=None, poll_s: float = 0.0):
        self.backend, self.n_train, self.n_rows, self.node_count = backend, n_train, n_rows, node_count
        self.log = log or (lambda *a: None)
        self.poll_s = poll_s
        self.test_losses: List[float] = []
        self.test_accs: List[float] = []

    def fit(self, initial_weights, max_epoch: int, batch_size: int, learning_rate: float,
            stopping_criterion: Callable[[Sequence[float]], bool], check_every: int, leak_loss_coef: float,
            seed: int = 0, positional_bug: bool = True, max_steps: Optional[int] = None) -> GradState:
        if not (0 <= leak_loss_coef <= 1):
            raise ValueError("leaking coefficient must be between 0 and 1")  # MasterAsync.scala:97
        split = [(r.start, r.stop) for r in split_vanilla(self.n_train, self.node_count)]
        steps = self.n_train * max_epoch if max_steps is None else max_steps
        self.backend.set_weights(np.asarray(initial_weights, dtype=np.float32))
        self.backend.async_start(split, batch=batch_size, lr=learning_rate, max_updates=steps, seed=seed,
                                 positional_bug=positional_bug)
        best_w, best_loss = np.asarray(initial_weights, dtype=np.float32), float("inf")
        last_step = -check_every
        state = GradState(best_w)
        try:
            while True:
                updates, running = self.backend.async_updates()
                if updates - last_step >= check_every or not running:
                    computed_loss, computed_acc, _ = self.backend.loss_acc(self.n_train, self.n_rows)
                    prev_l = self.test_losses[0] if self.test_losses else computed_loss
                    prev_a = self.test_accs[0] if self.test_accs else computed_acc
                    loss = leak_loss_coef * computed_loss + (1 - leak_loss_coef) * prev_l   # :122-125
                    acc = leak_loss_coef * computed_acc + (1 - leak_loss_coef) * prev_a
                    if best_loss > loss:                                                     # :132-138
                        best_loss, best_w = loss, self.backend.get_weights()
                    self.test_losses.insert(0, loss)
                    self.test_accs.insert(0, acc)
                    last_step = updates
                    if stopping_criterion(self.test_losses):                                # :146
                        self.log("converged to target: stopping computation")
                        break
                if not running:
                    break
                if self.poll_s:
                    time.sleep(self.poll_s)
        finally:
            self.backend.async_stop()                                                        # endComputation :87-94
        updates, _ = self.backend.async_updates()
        state = GradState(best_w, updates=int(updates)).finish(best_loss)
        return state
