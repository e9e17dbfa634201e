"""ORACLE (test infrastructure, NOT product code) for K8, the dense logistic mini-batch step of BASELINE.json
configs[4].  PARITY UNPINNED BY CONSTRUCTION: the reference has no dense / logistic model at all
(core/ml/SparseSVM.scala:11 is its only model; Main.scala:67 "could use another model"), so there is nothing of the
reference to restate or to take golden vectors from.  What this file pins instead is the mathematics:

    z = X w,  loss(w) = mean_i softplus(z_i) - y_i z_i,  grad = X^T (sigmoid(z) - y) / B,  w' = w - lr * grad

in fp64 numpy, and tests/test_dense_oracle.py checks `grad` against central finite differences of `loss`.
"""

from __future__ import annotations

import numpy as np


def loss_grad(X, y, w):
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    z = X @ w
    loss = float(np.mean(np.maximum(z, 0.0) + np.log1p(np.exp(-np.abs(z))) - y * z))
    p = 1.0 / (1.0 + np.exp(-z))
    grad = X.T @ (p - y) / X.shape[0]
    acc = float(np.mean((z > 0) == (y > 0.5)))
    return loss, grad, acc


def step(X, y, w, lr):
    loss, grad, acc = loss_grad(X, y, w)
    return np.asarray(w, dtype=np.float64) - lr * grad, loss, acc
