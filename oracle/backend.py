"""ORACLE (test infrastructure, NOT product code): the CPU oracle behind the backend interface of
distributed-sgd_amd/host.py, so that the host-side orchestration (Master.fit mirror, host-owned all-reduce) runs over the
fp64 restatement -- the reference side of every "same mirror, same random stream" comparison.

Only tests/ and bench.py's checker / target legs may import this module."""

import threading

import numpy as np


class OracleBackend:
    def __init__(self, oracle):
        self.o = oracle
        self.lam = oracle.lam
        self.w = np.zeros(oracle.dim + 1)
        self.steps = []
        self.min_margins = []
        self.actives = []
        self.mu = threading.Lock()  # the engine serialises calls on a context; handlers arrive from 8 pool threads

    def gradient(self, idx, w=None):
        wv = self.w if w is None else np.asarray(w, dtype=np.float64)
        g = self.o.gradient(wv, idx)
        return g, {"n_samples": len(idx), "n_active": self.o.last_stats["n_active"]}

    def forward(self, idx, w=None):
        return self.o.forward(self.w if w is None else np.asarray(w, dtype=np.float64), idx)

    def async_step(self, idx, lr, want_delta=False):
        with self.mu:
            w = np.ascontiguousarray(self.w, dtype=np.float64).copy()
            delta = self.o.async_step(w, idx, lr, want_delta=want_delta)
            self.w = w
            return delta, {"n_samples": len(idx), "n_active": self.o.last_stats["n_active"]}

    def update_grad(self, keys, values):
        with self.mu:
            w = self.w.copy()
            np.subtract.at(w, np.asarray(keys, dtype=np.int64), np.asarray(values, dtype=np.float64))
            self.w = w

    @property
    def dp(self):
        return self.o.dim + 1

    def apply(self, g_mean, lr):
        self.w = self.w - lr * np.asarray(g_mean, dtype=np.float64)

    def sync_step(self, lists, lr):
        self.steps.append([len(a) for a in lists])
        self.o.sync_step(self.w, lists, lr)
        # flip accounting for engine-vs-oracle runs: a step is "exposed" when some row's fp64 margin is within fp32
        # round-off of the gate (core/ml/SparseSVM.scala:27-28) -- only then may an fp32 engine gate a row differently
        self.min_margins.append(self.o.last_stats["min_abs_margin"])
        self.actives.append(self.o.last_stats["n_active"])
        return {"n_samples": sum(len(a) for a in lists), "n_active": self.o.last_stats["n_active"]}

    # resident plans (the surface of dsgd_amd.Engine that host.MasterSync.fit uses): here simply the steps in order
    class _Plan:
        def __init__(self, idx, offsets, n_steps, k):
            self.idx, self.offsets, self.n_steps, self.k, self.destroyed = np.asarray(idx), np.asarray(offsets), n_steps, k, False

        def destroy(self):
            self.destroyed = True

    def plan_flat(self, idx, offsets, n_steps, n_workers):
        self.plans_made = getattr(self, "plans_made", 0) + 1
        return OracleBackend._Plan(idx, offsets, n_steps, n_workers)

    def plan_run(self, plan, step_begin, step_end, lr):
        assert not plan.destroyed
        for s in range(step_begin, step_end):
            o = plan.offsets[s * plan.k:(s + 1) * plan.k + 1]
            self.sync_step([plan.idx[o[j]:o[j + 1]].astype(np.int32) for j in range(plan.k)], lr)

    def synchronize(self):
        return {"n_samples": 0, "n_active": 0}

    def sync_step_ranges(self, ranges, lr):
        return self.sync_step([np.arange(a, b, dtype=np.int32) for a, b in ranges], lr)

    def loss_acc(self, lo, hi):
        loss, acc, counts, _ = self.o.loss_acc(self.w, lo, hi)
        return loss, acc, counts

    def get_weights(self):
        return self.w.copy()

    def set_weights(self, w):
        self.w = np.asarray(w, dtype=np.float64).copy()
