"""ORACLE side (test infrastructure, NOT product code): replay of a TRACED lock-free ("Hogwild") run.

A many-worker lock-free run is not reproducible, so round 3 held it to a band between the orderings the oracle can
invent (oracle/hogwild_band.py) -- a band from chance to near-perfect that a broken engine would also sit in.

Why a plain re-simulation cannot do better: a constant-step run from w = 0 is CHAOTIC.  Every margin starts exactly
at the gate (x . 0 = 0 is active, core/ml/SparseSVM.scala:27-28), so a perturbation of 1e-7 in the initial weights
moves the test loss by 0.1 and the weights by a third of their norm within 400 updates (measured on this oracle;
tests/test_hogwild_replay.py keeps the experiment).  Whatever re-decides the gates -- in fp64 instead of fp32, a few
updates earlier or later -- leaves the engine's trajectory at once.

So the engine RECORDS its decisions (dsgd_async_set_trace, include/dsgd.h): for every mini-batch update, in commit
order, {worker, the worker's iteration = the key of the engine's replayable sampler, the update count its weights were
read at, the regulariser scalar s it used, the gate decision of every sampled row}.  With those on record the
reference's asynchronous iteration (core/Slave.scala:92-101) is a LINEAR recurrence the oracle evaluates exactly:

    update c:  rows   = the engine's sample for (seed, worker, iteration)                      -- hog_rows
               g      = (sum over the rows the ENGINE found active of y_i x_i) / batch         (Slave.scala:93-98, Vec.mean)
               g_j   += s_c on supp(g)                                                         (SparseSVM.scala:31; s_c as recorded)
               W_c    = W_{c-1} - lr * g                                                       (Slave.scala:99-101, GradState.scala:8)

Three statements follow, each of which fails for a broken engine:
  (A) ACCOUNTING, to rounding: the engine's final weights equal W_n coordinate by coordinate within ACCOUNT_TOL *
      max(1, |w|_inf) -- every update applied exactly once, the batch averaged, the step length, the sign, the
      support-only regulariser.  (An update lost in 8,000, a doubled one, a sum instead of a mean, a missing
      regulariser at lambda = 1e-5: all far outside.)
  (B) GATES: the recorded decision of row i of update c must be what the reference's gate gives on the replayed weights
      the update read, W_{read_at(c)}: y_i (x_i . W) >= 0.  What the replay cannot know is the handful of updates in
      flight while a worker's reads were served, so rows whose margin is within MARGIN_CLEAR of zero are reported but
      excused; on the CLEAR rows the disagreement must stay below GATE_TOL -- and the same check against the WRONG
      snapshot (the weights at the commit instead of at the read: staleness ignored) must be visibly worse with many
      workers, which is what shows that `read_at` means something.
  (C) SCALAR: the recorded s_c equals 2 lambda (W_{read_at(c)} . ds) within S_TOL (relative to the run's largest |s|).
"""

from __future__ import annotations

import math

import numpy as np

M64 = (1 << 64) - 1

ACCOUNT_TOL = 2e-4     # (A): max_j |w_engine - W_n|_j <= ACCOUNT_TOL * max(1, |W_n|_inf)
MARGIN_CLEAR = 0.02    # (B): rows with |x . W| above this are "clear"
GATE_TOL = 0.01        # (B): fraction of the clear rows whose recorded decision may differ
S_TOL = 0.05           # (C): |s_engine - s_replay| <= S_TOL * max_c |s_replay| for all but S_OUTLIERS of the updates
S_OUTLIERS = 0.01


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


def _entries(o, rows):
    """(flat positions, row number of every entry) of the CSR rows `rows`."""
    rows = np.asarray(rows, dtype=np.int64)
    starts = o.row_ptr[rows]
    lens = o.row_ptr[rows + 1] - starts
    total = int(lens.sum())
    first = np.cumsum(lens) - lens
    flat = np.arange(total, dtype=np.int64) + np.repeat(starts - first, lens)
    return flat, np.repeat(np.arange(len(rows), dtype=np.int64), lens)


def margins(o, w, rows):
    """x_i . w of the rows (products filtered as math/Sparse.scala:46 -> :112-114 does)."""
    flat, rid = _entries(o, rows)
    prod = o.val[flat].astype(np.float64) * w[o.col[flat]]
    prod[np.abs(prod) <= 1e-20] = 0.0
    return np.bincount(rid, weights=prod, minlength=len(rows))


def forced_delta(o, rows, active, s, batch, lr, fault=None):
    """lr * regularize(mean_i backward_i) with the gate decisions GIVEN (core/Slave.scala:93-99): the mean divides by
    the batch size -- an inactive row contributes a zero vector, it is still one of the vectors (math/Vec.scala:139)."""
    g = np.zeros(o.dim + 1)
    act = np.asarray(rows)[np.asarray(active, dtype=bool)]
    if len(act):
        flat, rid = _entries(o, act)
        yx = o.val[flat].astype(np.float64) * o.label[act].astype(np.float64)[rid]
        yx[np.abs(yx) <= 1e-20] = 0.0               # math/Sparse.scala:108-118
        g = np.bincount(o.col[flat], weights=yx, minlength=o.dim + 1)
    if fault != "sum_not_mean":
        g = g / float(batch)
    g[np.abs(g) <= 1e-20] = 0.0
    if fault != "no_regulariser" and s != 0.0 and abs(s) > 1e-20:   # core/ml/SparseSVM.scala:31, math/Vec.scala:65-75
        g[g != 0.0] += s
    step = lr * g
    if fault == "double_apply":
        step = 2.0 * step
    elif fault == "half_step":
        step = 0.5 * step
    return step


FAULTS = ("double_apply", "drop_one", "sum_not_mean", "half_step", "no_regulariser", "wrong_rows")


def replay_forced(o, w, split, batch, lr, seed, trace, fault=None, positional_bug=False, check=True):
    """Replay one traced engine run (one dsgd_async_start ... dsgd_async_wait) in place on `w` (float64, D + 1) with the
    engine's recorded gate decisions and scalars.  trace: the dict Engine.async_read_trace returns; read_at counts the
    updates of THIS run (0 = the weights the run started from).  `fault` (one of FAULTS) breaks the rule on purpose: the
    negative controls.  Returns the statistics of checks (B) and (C) (module docstring) over this run."""
    worker, it, read_at = np.asarray(trace["worker"]), np.asarray(trace["it"]), np.asarray(trace["read_at"])
    s_rec, mask, n_act = np.asarray(trace["s"], dtype=np.float64), np.asarray(trace["mask"]), np.asarray(trace["n_active"])
    n = len(worker)
    st = {"updates": n, "max_lag": 0, "mean_lag": 0.0, "rows": 0, "rows_clear": 0, "gate_differs": 0, "gate_differs_clear": 0,
          "gate_differs_clear_if_fresh": 0, "rows_clear_if_fresh": 0, "s_max_abs": 0.0, "s_err": []}
    if n == 0:
        return st
    commit = np.arange(1, n + 1, dtype=np.int64)
    if np.any(read_at < 0) or np.any(read_at >= commit):
        raise ValueError("trace inconsistent: an update read weights from its own future")
    if np.any(mask[:, batch:]) or np.any(mask[:, :batch].sum(axis=1) != n_act):
        raise ValueError("trace inconsistent: gate masks and active counts disagree")
    lag = commit - 1 - read_at                      # updates applied between the read and the commit
    st["max_lag"], st["mean_lag"] = int(lag.max()), float(lag.mean())
    ring_n = 1
    while ring_n < int(lag.max()) + 2:
        ring_n *= 2
    if ring_n * (o.dim + 1) * 8 > 3 << 30:
        raise MemoryError("staleness of %d updates needs a %d-entry snapshot ring" % (int(lag.max()), ring_n))
    ring = np.empty((ring_n, o.dim + 1))
    ring[0] = w
    for c in range(1, n + 1):
        k = int(worker[c - 1])
        b, e = split[k]
        rows = hog_rows(seed, k, int(it[c - 1]), b, e - b, batch, positional_bug)
        active = mask[c - 1, :batch]
        if check:
            snap = ring[int(read_at[c - 1]) % ring_n]
            y = o.label[rows].astype(np.float64)
            d = margins(o, snap, rows)
            want = ~(y * d < 0.0)                   # core/ml/SparseSVM.scala:27-28
            clear = np.abs(d) > MARGIN_CLEAR
            st["rows"] += batch
            st["rows_clear"] += int(clear.sum())
            st["gate_differs"] += int((want != active).sum())
            st["gate_differs_clear"] += int(((want != active) & clear).sum())
            if lag[c - 1] > 0:                      # the same against the WRONG snapshot: staleness ignored
                d2 = margins(o, w, rows)
                clear2 = np.abs(d2) > MARGIN_CLEAR
                st["rows_clear_if_fresh"] += int(clear2.sum())
                st["gate_differs_clear_if_fresh"] += int(((~(y * d2 < 0.0) != active) & clear2).sum())
            s_ref = 2.0 * o.lam * float(snap @ o.ds)
            st["s_max_abs"] = max(st["s_max_abs"], abs(s_ref))
            st["s_err"].append(abs(float(s_rec[c - 1]) - s_ref))
        if fault == "wrong_rows":
            rows = hog_rows(seed + 1, k, int(it[c - 1]), b, e - b, batch, positional_bug)
        if not (fault == "drop_one" and c == n // 2):
            w -= forced_delta(o, rows, active, float(s_rec[c - 1]), batch, lr, fault)
            w[np.abs(w) <= 1e-20] = 0.0             # math/Sparse.scala:108-118
        ring[c % ring_n] = w
    return st


def merge(stats):
    """The statistics of several consecutive runs (segments between checkpoints) as one."""
    out = {"updates": sum(s["updates"] for s in stats), "max_lag": max(s["max_lag"] for s in stats),
           "mean_lag": float(np.average([s["mean_lag"] for s in stats], weights=[max(1, s["updates"]) for s in stats])),
           "s_max_abs": max(s["s_max_abs"] for s in stats), "s_err": [e for s in stats for e in s["s_err"]]}
    for q in ("rows", "rows_clear", "gate_differs", "gate_differs_clear", "gate_differs_clear_if_fresh", "rows_clear_if_fresh"):
        out[q] = sum(s[q] for s in stats)
    return out


def verdict(o, w_engine, w_replay, stats, eval_range=None):
    """The three statements (module docstring) as numbers and booleans."""
    w_engine = np.asarray(w_engine, dtype=np.float64)
    with np.errstate(all="ignore"):   # (a negative control may have diverged to inf / nan: it then fails every comparison)
        err = float(np.abs(w_engine - w_replay).max())
        scale = max(1.0, float(np.abs(w_replay).max()))
        nr = float(np.sqrt(w_replay @ w_replay))
        dist = float(np.sqrt(((w_engine - w_replay) ** 2).sum()) / max(nr, 1e-300))
    s_err = np.asarray(stats["s_err"]) if len(stats["s_err"]) else np.zeros(1)
    s_bad = float((s_err > S_TOL * max(stats["s_max_abs"], 1e-300)).mean())
    gate_clear = stats["gate_differs_clear"] / max(1, stats["rows_clear"])
    out = {
        "updates": stats["updates"], "max_lag": stats["max_lag"], "mean_lag": stats["mean_lag"],
        "account_max_abs_err": err, "account_err_over_tol": err / (ACCOUNT_TOL * scale), "account_tolerance": ACCOUNT_TOL,
        "rel_distance": dist, "wnorm_engine": float(np.sqrt(w_engine @ w_engine)), "wnorm_replay": nr,
        "rows": stats["rows"], "rows_clear": stats["rows_clear"], "margin_clear": MARGIN_CLEAR,
        "gate_differs_all_rows": stats["gate_differs"] / max(1, stats["rows"]),
        "gate_differs_clear_rows": gate_clear, "gate_tolerance": GATE_TOL,
        "gate_differs_clear_rows_if_staleness_ignored": stats["gate_differs_clear_if_fresh"] / max(1, stats["rows_clear_if_fresh"]),
        "s_max_abs": stats["s_max_abs"], "s_err_median": float(np.median(s_err)), "s_err_max": float(s_err.max()),
        "s_fraction_outside": s_bad, "s_tolerance": S_TOL,
    }
    if eval_range is not None:
        out["loss_engine"], out["acc_engine"] = o.loss_acc(w_engine, eval_range[0], eval_range[1])[:2]
        out["loss_replay"], out["acc_replay"] = o.loss_acc(w_replay, eval_range[0], eval_range[1])[:2]
    out["ok"] = {"accounting": bool(err <= ACCOUNT_TOL * scale), "gates": bool(gate_clear <= GATE_TOL),
                 "scalar": bool(s_bad <= S_OUTLIERS)}
    return out
