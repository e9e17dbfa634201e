"""CPU: the K8 oracle (oracle/dense_ref.py) pinned by finite differences -- the reference has nothing to pin it to."""

import numpy as np

from oracle import dense_ref


def test_gradient_matches_central_finite_differences():
    rng = np.random.default_rng(0)
    X = rng.normal(size=(200, 24)) / np.sqrt(24)
    w_true = rng.normal(size=24)
    y = (X @ w_true + 0.1 * rng.normal(size=200) > 0).astype(np.float64)
    w = 0.3 * rng.normal(size=24)
    loss, grad, acc = dense_ref.loss_grad(X, y, w)
    num = np.zeros_like(w)
    for j in range(24):
        e = np.zeros(24)
        e[j] = 1e-6
        num[j] = (dense_ref.loss_grad(X, y, w + e)[0] - dense_ref.loss_grad(X, y, w - e)[0]) / 2e-6
    assert np.abs(num - grad).max() < 1e-8
    assert 0.0 <= acc <= 1.0 and loss > 0


def test_known_values():
    # one row x = (1, 0), y = 1, w = 0: z = 0, loss = log 2, grad = (sigmoid(0) - 1) * x = (-0.5, 0)
    loss, grad, acc = dense_ref.loss_grad([[1.0, 0.0]], [1.0], [0.0, 0.0])
    assert abs(loss - np.log(2.0)) < 1e-15 and np.allclose(grad, [-0.5, 0.0]) and acc == 0.0
    # extreme margins do not overflow
    loss, grad, _ = dense_ref.loss_grad([[1000.0], [-1000.0]], [1.0, 0.0], [1.0])
    assert loss == 0.0 and np.allclose(grad, 0.0)


def test_training_decreases_the_loss():
    rng = np.random.default_rng(1)
    X = rng.normal(size=(500, 16)) / 4.0
    y = (X @ rng.normal(size=16) > 0).astype(np.float64)
    w = np.zeros(16)
    losses = []
    for _ in range(50):
        w, loss, acc = dense_ref.step(X, y, w, 2.0)
        losses.append(loss)
    assert losses[-1] < 0.6 * losses[0] and all(b <= a + 1e-12 for a, b in zip(losses, losses[1:]))
