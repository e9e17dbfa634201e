"""Test helper: the CPU oracle behind the backend interface of distributed-sgd_amd/host.py, so that the
host-side orchestration (Master.fit mirror, host-owned all-reduce) can be exercised without a GPU."""

import numpy as np


class OracleBackend:
    def __init__(self, oracle):
        self.o = oracle
        self.lam = oracle.lam
        self.w = np.zeros(oracle.dim + 1)
        self.steps = []

    def gradient(self, idx):
        g = self.o.gradient(self.w, idx)
        return g, {"n_samples": len(idx), "n_active": self.o.last_stats["n_active"]}

    def apply(self, g_mean, lr):
        self.w = self.w - lr * np.asarray(g_mean, dtype=np.float64)

    def sync_step(self, lists, lr):
        self.steps.append([len(a) for a in lists])
        self.o.sync_step(self.w, lists, lr)
        return {"n_samples": sum(len(a) for a in lists), "n_active": self.o.last_stats["n_active"]}

    def loss_acc(self, lo, hi):
        loss, acc, counts, _ = self.o.loss_acc(self.w, lo, hi)
        return loss, acc, counts

    def get_weights(self):
        return self.w.copy()

    def set_weights(self, w):
        self.w = np.asarray(w, dtype=np.float64).copy()
