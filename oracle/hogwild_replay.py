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
  (B) STALENESS, as far as a chaotic system lets it be seen: the margins of such a run are small (median |x . w| = 0.05
      while one update moves them by 0.03: replaying the gates of a MODELLED schedule against a snapshot that is off by ONE
      update already disagrees on 12 % of the rows, by two on 24 % -- tests/test_hogwild_replay.py), and what a worker
      really read is W at `read_at` plus whatever part of the updates in flight had landed.  So the recorded decisions
      cannot be reproduced row by row; what can be shown is WHERE along the interval [read_at, commit) the weights they
      were taken on lie: the fraction of rows whose recorded decision differs from the reference's gate y_i (x_i . W) >= 0
      is evaluated on W_{read_at + f * lag} for f in FRACTIONS.  Measured (256 workers x batch 100): 0.35 at f = 0, 0.10 at
      f = 0.5, 0.42-0.50 at f = 1 -- the decisions were taken on weights from the MIDDLE of the interval: the LDS copy of the
      hot weights is requested next to the atomic that returns `read_at`, but updates LAND before they COMMIT (an update's
      atomics go out during its sweep, its commit number is drawn behind it) and the engine is bound by those atomics, so the
      sweep is most of an iteration and about half of the updates in flight are already visible.  Asserted with many
      workers: the best-fitting point lies clearly below the commit end (staleness ignored), and so does the read end.
  (C) SCALAR: the recorded s_c against 2 lambda (W_{read_at(c)} . ds): median within S_TOL_MEDIAN, 90th percentile
      within S_TOL_P90 of the run's largest |s| (the scalar is kept incrementally with one atomic per update, the same
      in-flight fuzz applies).
"""

from __future__ import annotations

import math

import numpy as np

M64 = (1 << 64) - 1

ACCOUNT_TOL = 2e-4     # (A): max_j |w_engine - W_n|_j <= ACCOUNT_TOL * max(1, |W_n|_inf)
FRACTIONS = (0.0, 0.25, 0.5, 0.75, 1.0)   # (B): where in [read_at, commit) the gates are compared
STALE_LAG = 32.0       # (B): asserted for runs whose mean lag is at least this many updates ...
STALE_RATIO = 0.6      # ...: min_f differs(f) <= STALE_RATIO * differs(f = 1), and differs(f = 0) < differs(f = 1)
S_TOL_MEDIAN = 0.02    # (C): relative to max_c |s_replay|
S_TOL_P90 = 0.25


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


def replay_forced(o, w, split, batch, lr, seed, trace, fault=None, positional_bug=False, check=True, fractions=FRACTIONS):
    """Replay one traced engine run (one dsgd_async_start ... dsgd_async_wait) in place on `w` (float64, D + 1) with the
    engine's recorded gate decisions and scalars.  trace: the dict Engine.async_read_trace returns; read_at counts the
    updates of THIS run (0 = the weights the run started from).  `fault` (one of FAULTS) breaks the rule on purpose: the
    negative controls.  Returns the statistics of checks (B) and (C) (module docstring) over this run."""
    worker, it, read_at = np.asarray(trace["worker"]), np.asarray(trace["it"]), np.asarray(trace["read_at"])
    s_rec, mask, n_act = np.asarray(trace["s"], dtype=np.float64), np.asarray(trace["mask"]), np.asarray(trace["n_active"])
    n = len(worker)
    st = {"updates": n, "max_lag": 0, "mean_lag": 0.0, "rows": 0, "fractions": list(fractions), "gate_differs": [0] * len(fractions),
          "s_max_abs": 0.0, "s_err": []}
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
            y = o.label[rows].astype(np.float64)
            st["rows"] += batch
            for fi, f in enumerate(fractions):      # the reference's gate on W at read_at + f * lag
                at = int(read_at[c - 1]) + int(round(f * float(lag[c - 1])))
                d = margins(o, ring[at % ring_n], rows)
                st["gate_differs"][fi] += int((~(y * d < 0.0) != active).sum())   # core/ml/SparseSVM.scala:27-28
            s_ref = 2.0 * o.lam * float(ring[int(read_at[c - 1]) % ring_n] @ o.ds)
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
           "s_max_abs": max(s["s_max_abs"] for s in stats), "s_err": [e for s in stats for e in s["s_err"]],
           "rows": sum(s["rows"] for s in stats), "fractions": list(stats[0]["fractions"]),
           "gate_differs": [sum(s["gate_differs"][i] for s in stats) for i in range(len(stats[0]["fractions"]))]}
    return out


EMPTY = {"updates": 0, "max_lag": 0, "mean_lag": 0.0, "rows": 0, "fractions": list(FRACTIONS), "gate_differs": [0] * len(FRACTIONS),
         "s_max_abs": 0.0, "s_err": []}


def verdict(o, w_engine, w_replay, stats, eval_range=None):
    """The three statements (module docstring) as numbers and booleans."""
    w_engine = np.asarray(w_engine, dtype=np.float64)
    with np.errstate(all="ignore"):   # (a negative control may have diverged to inf / nan: it then fails every comparison)
        err = float(np.abs(w_engine - w_replay).max())
        scale = max(1.0, float(np.abs(w_replay).max()))
        nr = float(np.sqrt(w_replay @ w_replay))
        dist = float(np.sqrt(((w_engine - w_replay) ** 2).sum()) / max(nr, 1e-300))
    s_err = np.asarray(stats["s_err"]) if len(stats["s_err"]) else np.zeros(1)
    s_scale = max(stats["s_max_abs"], 1e-300)
    s_p50, s_p90 = float(np.median(s_err)) / s_scale, float(np.quantile(s_err, 0.9)) / s_scale
    prof = [g / max(1, stats["rows"]) for g in stats["gate_differs"]]
    stale_seen = stats["mean_lag"] < STALE_LAG or (min(prof) <= STALE_RATIO * prof[-1] and prof[0] < prof[-1])
    out = {
        "updates": stats["updates"], "max_lag": stats["max_lag"], "mean_lag": stats["mean_lag"],
        "account_max_abs_err": err, "account_err_over_tol": err / (ACCOUNT_TOL * scale), "account_tolerance": ACCOUNT_TOL,
        "rel_distance": dist, "wnorm_engine": float(np.sqrt(w_engine @ w_engine)), "wnorm_replay": nr,
        "rows": stats["rows"], "gate_fractions": list(stats["fractions"]), "gate_differs_at_fraction": prof,
        "gate_rule": "min over f <= %.2f x differs(commit end) and differs(read end) < differs(commit end) once the mean lag reaches "
                     "%d updates" % (STALE_RATIO, STALE_LAG),
        "s_max_abs": stats["s_max_abs"], "s_rel_err_median": s_p50, "s_rel_err_p90": s_p90, "s_rel_err_max": float(s_err.max()) / s_scale,
        "s_tolerances": [S_TOL_MEDIAN, S_TOL_P90],
    }
    if eval_range is not None:
        out["loss_engine"], out["acc_engine"] = o.loss_acc(w_engine, eval_range[0], eval_range[1])[:2]
        out["loss_replay"], out["acc_replay"] = o.loss_acc(w_replay, eval_range[0], eval_range[1])[:2]
    out["ok"] = {"accounting": bool(err <= ACCOUNT_TOL * scale), "staleness": bool(stale_seen),
                 "scalar": bool(s_p50 <= S_TOL_MEDIAN and s_p90 <= S_TOL_P90)}
    return out


# ---- a gate check WITH TEETH for few workers ---------------------------------------------------------------------------
# Statement (B) above only locates the decisions inside [read_at, commit): at 256 workers a tenth of the rows still
# disagrees at the best-fitting point, so a gate bug that mis-decides a few per cent of the rows only under concurrency
# would pass it.  With FEW workers most updates have nothing, or one update, in flight between their read and their commit:
# then the weights a worker read are known up to that one update, and a recorded decision can be held to the reference's
# gate row by row.
GATE_SLACK = 8.0


def gate_check_small_lag(o, w0, split, batch, lr, seed, trace, max_lag=1, positional_bug=False):
    """Replay the traced run with its recorded decisions (exact weights W_c after every commit), then hold every decision
    of every update with lag <= max_lag to the reference's gate y (x . W) >= 0 at BOTH ends of [read_at, commit): a
    decision that differs at both ends is legitimate only for a row whose margin, at one of the ends, is no larger than
    what the updates in flight can have moved it by while they were landing (sum_j |x_j| |delta_j| over them) plus the
    fp32 resolution of its dot product and of the replayed weights themselves.  Returns counts; `outside` lists the violations (empty for a correct engine)."""
    worker, it, read_at = np.asarray(trace["worker"]), np.asarray(trace["it"]), np.asarray(trace["read_at"])
    s_rec, mask = np.asarray(trace["s"], dtype=np.float64), np.asarray(trace["mask"])
    n = len(worker)
    out = {"updates": n, "updates_checked": 0, "rows_checked": 0, "differ_at_both_ends": 0, "explained_by_in_flight_or_resolution": 0, "outside": []}
    w = np.asarray(w0, dtype=np.float64).copy()
    window = {0: w.copy()}
    dwin = {}
    eps32 = 2.0 ** -24
    for c in range(1, n + 1):
        k = int(worker[c - 1])
        b, e = split[k]
        rows = hog_rows(seed, k, int(it[c - 1]), b, e - b, batch, positional_bug)
        ra = int(read_at[c - 1])
        lag = c - 1 - ra
        if lag <= max_lag:
            y = o.label[rows].astype(np.float64)
            w_read, w_commit = window[ra], window[c - 1]
            flat, rid = _entries(o, rows)
            v = o.val[flat].astype(np.float64)
            cols = o.col[flat]
            m_read = y * np.bincount(rid, weights=v * w_read[cols], minlength=batch)
            m_commit = y * np.bincount(rid, weights=v * w_commit[cols], minlength=batch)
            # what the updates in flight can have done to the margin while landing coordinate by coordinate: any SUBSET of
            # their coordinates may have been visible, so the margin the worker saw lies in [m_read - down, m_read + up]
            down, up = np.zeros(batch), np.zeros(batch)
            for cc in range(ra + 1, c):
                t = y[rid] * v * dwin[cc][cols]            # W_commit = W_read - delta: the margin moves by -t per coordinate
                down += np.bincount(rid, weights=np.maximum(t, 0.0), minlength=batch)
                up += np.bincount(rid, weights=np.maximum(-t, 0.0), minlength=batch)
            nnz = np.bincount(rid, minlength=batch)
            res = (nnz + 32) * eps32 * np.bincount(rid, weights=np.abs(v * w_read[cols]), minlength=batch)
            # ... and the weights the worker read are the ENGINE's fp32 weights, which the replay (fp64, recorded decisions)
            # follows to its accounting error only -- a few 1e-7 .. 1e-5 of |w|inf per coordinate after hundreds of atomic
            # updates (replay_forced's statement; 2e-4 |w|inf is its tolerance): a margin inside sum_j |x_j| times a
            # conservative 1e-6 |w|inf of the gate cannot be told apart (found on a 2-worker run: lag 0, margin -1.6e-5)
            res = res + np.bincount(rid, weights=np.abs(v), minlength=batch) * 1e-6 * max(1.0, float(np.abs(w_read).max())) / GATE_SLACK
            rec = mask[c - 1, :batch].astype(bool)
            d_read, d_commit = ~(m_read < 0.0), ~(m_commit < 0.0)
            both = (rec != d_read) & (rec != d_commit)
            out["updates_checked"] += 1
            out["rows_checked"] += batch
            out["differ_at_both_ends"] += int(both.sum())
            for r in np.flatnonzero(both):
                slack = GATE_SLACK * res[r] + 1e-30
                # recorded "inactive" needs a visible margin < 0, recorded "active" one >= 0, somewhere in the reachable range
                reachable = (m_read[r] - down[r] - slack < 0.0) if not rec[r] else (m_read[r] + up[r] + slack >= 0.0)
                if reachable:
                    out["explained_by_in_flight_or_resolution"] += 1
                else:
                    out["outside"].append({"update": c, "row": int(rows[r]), "lag": int(lag), "recorded_active": bool(rec[r]),
                                           "margin_read": float(m_read[r]), "margin_commit": float(m_commit[r]),
                                           "reachable": [float(m_read[r] - down[r]), float(m_read[r] + up[r])], "resolution": float(slack)})
        d = forced_delta(o, rows, mask[c - 1, :batch], float(s_rec[c - 1]), batch, lr)
        w -= d
        w[np.abs(w) <= 1e-20] = 0.0
        window[c] = w.copy()
        dwin[c] = d
        old = c - max_lag - 3
        window.pop(old, None)
        dwin.pop(old, None)
    out["ok"] = not out["outside"]
    out["outside"] = out["outside"][:5]
    return out
