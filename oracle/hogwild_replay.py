"""ORACLE side (test infrastructure, NOT product code): replay of a TRACED lock-free ("Hogwild") run.

A many-worker lock-free run is not reproducible, so round 3 held it to a band between the orderings the oracle can
invent (oracle/hogwild_band.py) -- a band from chance to near-perfect that a broken engine would also sit in.  This
module replaces invention by measurement: a traced run (dsgd_async_set_trace, include/dsgd.h) records, for every
mini-batch update in COMMIT order, which worker made it, that worker's iteration number (the key of the engine's
replayable sampler) and the update count its weights were read at.  The oracle then replays the reference's
asynchronous iteration (core/Slave.scala:92-101 = oracle.c orc_async_step) with exactly that schedule:

    update c (record c - 1):  rows   = the engine's sample for (seed, worker, iteration)        -- hog_rows below
                              W_snap = the weights after update number read_at[c]                -- a ring of snapshots
                              delta  = lr * regularize(mean_i backward(W_snap, x_i, y_i), W_snap)   (Slave.scala:93-99)
                              W_c    = W_{c-1} - delta                                              (Slave.scala:101,
                                                                                                    GradState.scala:8)

What the replay cannot know is the handful of updates in flight while a worker's reads were being served (a worker's
LDS copy of the hot weights is requested right next to the atomic whose return value is `read_at`), and the engine
computes in fp32: the two trajectories separate slowly, and a constant-step run amplifies differences through gate
flips.  They stay close enough for tolerances an order of magnitude tighter than the band (tests/
test_gpu_hogwild_trace.py states them), and -- the point -- replays with a deliberately WRONG rule (every update applied
twice, a third of the updates lost, the batch summed instead of averaged, half the step length, the staleness ignored)
land far outside those tolerances: the check can fail.
"""

from __future__ import annotations

import math

import numpy as np

M64 = (1 << 64) - 1


def hog_mix(z):  # csrc/dsgd_batch.hpp: hog_mix (splitmix64 finaliser)
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def hog_rows(seed, worker, it, begin, n_k, batch, positional_bug=False):
    """The rows iteration `it` of worker `worker` samples (mirror of dsgd_hogwild_kernel's hog_sampler / row_at)."""
    key = hog_mix(seed ^ hog_mix((worker * 0x100000001B3 + it) & M64))
    mul = 1 + hog_mix(key) % n_k
    while math.gcd(mul, n_k) != 1:
        mul = mul % n_k + 1
    off = hog_mix(key ^ 0xABCDEF12345) % n_k
    base = 0 if positional_bug else begin
    # (mul, off < n_k < 2^31 and batch <= 4096: the products stay below 2^43)
    return np.asarray(base + (mul * np.arange(batch, dtype=np.int64) + off) % n_k, dtype=np.int32)


FAULTS = ("double_apply", "drop_third", "sum_not_mean", "half_step", "no_regulariser", "fresh_reads")


def replay_segment(o, w, split, batch, lr, seed, trace, fault=None, positional_bug=False):
    """Replay one traced engine run (one dsgd_async_start ... dsgd_async_wait) in place on `w` (float64, D + 1).
    trace = (worker, iteration, read_at) as Engine.async_read_trace returns them; record i is update number i + 1 and
    read_at counts the updates of THIS run (0 = the weights the run started from).  `fault` (one of FAULTS) breaks the
    rule on purpose: the negative controls of the tests.  Returns {updates, max_lag, mean_lag}."""
    worker, it, read_at = (np.asarray(a) for a in trace)
    n = len(worker)
    if n == 0:
        return {"updates": 0, "max_lag": 0, "mean_lag": 0.0}
    commit = np.arange(1, n + 1, dtype=np.int64)
    if np.any(read_at < 0) or np.any(read_at >= commit):
        raise ValueError("trace inconsistent: an update read weights from its own future")
    lag = commit - 1 - read_at                      # updates applied between the read and the commit
    if fault == "fresh_reads":
        read_at = commit - 1
    ring_n = 1
    while ring_n < int(lag.max()) + 2:
        ring_n *= 2
    if ring_n * (o.dim + 1) * 8 > 3 << 30:
        raise MemoryError("staleness of %d updates needs a %d-entry snapshot ring" % (int(lag.max()), ring_n))
    ring = np.empty((ring_n, o.dim + 1))
    ring[0] = w
    lam = o.lam
    for c in range(1, n + 1):
        k = int(worker[c - 1])
        b, e = split[k]
        rows = hog_rows(seed, k, int(it[c - 1]), b, e - b, batch, positional_bug)
        snap = ring[int(read_at[c - 1]) % ring_n].copy()
        if fault == "no_regulariser":
            o.lam = 0.0
        try:
            delta = o.async_step(snap, rows, lr, want_delta=True)
        finally:
            o.lam = lam
        if fault == "double_apply":
            delta = 2.0 * delta
        elif fault == "sum_not_mean":
            delta = float(batch) * delta
        elif fault == "half_step":
            delta = 0.5 * delta
        if not (fault == "drop_third" and c % 3 == 0):
            w -= delta
            w[np.abs(w) <= 1e-20] = 0.0             # math/Sparse.scala:108-118
        ring[c % ring_n] = w
    return {"updates": n, "max_lag": int(lag.max()), "mean_lag": float(lag.mean())}


def compare(o, w_engine, w_replay, eval_range, engine_eval=None):
    """The statistics a traced run is held to: test loss / accuracy of both weight vectors over `eval_range`
    (core/Master.scala:100-107), |w|_2, the sum of the weights (a linear functional: a lost or doubled update moves it),
    the relative distance and the cosine between the two vectors.  engine_eval: (loss, acc) as the ENGINE evaluated
    its own weights, when the caller has them."""
    w_engine = np.asarray(w_engine, dtype=np.float64)
    le, ae, _, _ = o.loss_acc(w_engine, eval_range[0], eval_range[1])
    lr_, ar, _, _ = o.loss_acc(w_replay, eval_range[0], eval_range[1])
    if engine_eval is not None:
        le, ae = engine_eval
    with np.errstate(all="ignore"):   # (a negative control may have diverged to inf / nan: it then fails every comparison)
        ne, nr = float(np.sqrt(w_engine @ w_engine)), float(np.sqrt(w_replay @ w_replay))
        dist = float(np.sqrt(((w_engine - w_replay) ** 2).sum()) / max(nr, 1e-300))
        cos = float((w_engine @ w_replay) / max(ne * nr, 1e-300))
    return {
        "loss_engine": float(le), "loss_replay": float(lr_), "acc_engine": float(ae), "acc_replay": float(ar),
        "wnorm_engine": ne, "wnorm_replay": nr,
        "wsum_engine": float(w_engine.sum()), "wsum_replay": float(w_replay.sum()),
        "rel_distance": dist, "cosine": cos,
    }


# The stated tolerances of the traced parity check (tests/test_gpu_hogwild_trace.py, bench.py hogwild.traced_replay):
# |loss_engine - loss_replay| <= LOSS, |acc_engine - acc_replay| <= ACC, |w| within WNORM_REL, relative distance of
# the weight vectors <= REL_DISTANCE.  (Round 3's band, for comparison: loss +- 0.43, accuracy +- 0.21, |w| x 55.)
TOL = {"loss": 0.02, "acc": 0.02, "wnorm_rel": 0.03, "rel_distance": 0.25}


def within(cmp, tol=None):
    """{quantity: bool} -- all True = the engine's run is the traced schedule's run within the stated tolerances."""
    t = dict(TOL if tol is None else tol)
    return {
        "loss": abs(cmp["loss_engine"] - cmp["loss_replay"]) <= t["loss"],
        "acc": abs(cmp["acc_engine"] - cmp["acc_replay"]) <= t["acc"],
        "wnorm": abs(cmp["wnorm_engine"] - cmp["wnorm_replay"]) <= t["wnorm_rel"] * cmp["wnorm_replay"],
        "rel_distance": cmp["rel_distance"] <= t["rel_distance"],
    }
