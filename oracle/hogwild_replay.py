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
    of every update with at most max_lag OTHER updates between what it is known to have read and its commit to the
    reference's gate y (x . W) >= 0 at BOTH ends of that stretch: a decision that differs at both ends is legitimate only
    for a row whose margin, at one of the ends, is no larger than what the updates in flight can have moved it by while
    they were landing (sum_j |x_j| |delta_j| over them) plus the fp32 resolution of its dot product and of the replayed
    weights themselves.  Returns counts; `outside` lists the violations (empty for a correct engine).
    What is known to have been read: traces of the engine carry `seen_from` (an update count read BEFORE any weight of
    the iteration was requested: updates 1..seen_from had landed) and the worker's own previous update (commit number
    read_at: acknowledged before the copy was requested) -- the stretch starts at seen_from, NOT at read_at: the copy of
    the weights goes out before the worker's own commit number comes back, and an update of another worker committed in
    between carries a number <= read_at without being in the copy (found in round 6, when the engine's timing changed:
    58 of 143,000 rows).  Such traces also get, as in flight, the FIRST update of every other worker beyond the commit
    (updates land before they commit).  A trace without `seen_from` (hand-made schedules) is read as before:
    everything up to read_at in, the updates in (read_at, commit) in flight."""
    worker, it, read_at = np.asarray(trace["worker"]), np.asarray(trace["it"]), np.asarray(trace["read_at"])
    s_rec, mask = np.asarray(trace["s"], dtype=np.float64), np.asarray(trace["mask"])
    seen = np.asarray(trace["seen_from"], dtype=np.int64) if "seen_from" in trace else None
    n = len(worker)
    out = {"updates": n, "updates_checked": 0, "rows_checked": 0, "differ_at_both_ends": 0, "explained_by_in_flight_or_resolution": 0, "outside": []}
    # pass 1: every update's delta with the recorded decisions (sparse), the rows it sampled
    rows_of, dcols, dvals = [], [], []
    for c in range(1, n + 1):
        k = int(worker[c - 1])
        b, e = split[k]
        rows = hog_rows(seed, k, int(it[c - 1]), b, e - b, batch, positional_bug)
        cols, vals = _sparse_delta(o, rows, mask[c - 1, :batch], float(s_rec[c - 1]), batch, lr)
        rows_of.append(rows)
        dcols.append(cols)
        dvals.append(vals)
    by_worker = [np.flatnonzero(worker == k) + 1 for k in range(len(split))]
    w = np.asarray(w0, dtype=np.float64).copy()
    window = {0: w.copy()}
    eps32 = 2.0 ** -24
    keep = max_lag + 4
    for c in range(1, n + 1):
        k = int(worker[c - 1])
        rows = rows_of[c - 1]
        ra = int(read_at[c - 1])
        lo = int(seen[c - 1]) if seen is not None else ra
        own = ra if (seen is not None and ra > lo) else 0            # the worker's own previous update: in for certain
        flying = [cc for cc in range(lo + 1, c) if cc != own]
        lag = len(flying)
        if lag <= max_lag and lo in window:
            if seen is not None:                                     # ... and what can have landed ahead of its commit
                for kk in range(len(split)):
                    if kk != k:
                        pos = int(np.searchsorted(by_worker[kk], c, side="right"))
                        if pos < len(by_worker[kk]):
                            flying.append(int(by_worker[kk][pos]))
            y = o.label[rows].astype(np.float64)
            w_read, w_commit = window[lo], window[c - 1]
            flat, rid = _entries(o, rows)
            v = o.val[flat].astype(np.float64)
            cols = o.col[flat]
            wr = w_read[cols] - (_lookup(dcols[own - 1], dvals[own - 1], cols) if own else 0.0)
            m_read = y * np.bincount(rid, weights=v * wr, minlength=batch)
            m_commit = y * np.bincount(rid, weights=v * w_commit[cols], minlength=batch)
            # what the updates in flight can have done to the margin while landing coordinate by coordinate: any SUBSET of
            # their coordinates may have been visible, so the margin the worker saw lies in [m_read - down, m_read + up]
            down, up = np.zeros(batch), np.zeros(batch)
            for cc in flying:
                t = y[rid] * v * _lookup(dcols[cc - 1], dvals[cc - 1], cols)   # W_after = W_before - delta: the margin moves by -t per coordinate
                down += np.bincount(rid, weights=np.maximum(t, 0.0), minlength=batch)
                up += np.bincount(rid, weights=np.maximum(-t, 0.0), minlength=batch)
            nnz = np.bincount(rid, minlength=batch)
            res = (nnz + 32) * eps32 * np.bincount(rid, weights=np.abs(v * wr), minlength=batch)
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
        w[dcols[c - 1]] -= dvals[c - 1]
        w[np.abs(w) <= 1e-20] = 0.0
        window[c] = w.copy()
        window.pop(c - keep, None)
    out["ok"] = not out["outside"]
    out["outside"] = out["outside"][:5]
    return out


# ---- EVERY decision of a run of ANY worker count, held to what it was taken on (round 6) --------------------------------
# The engine also records the x . w every sampled row was gated on and `seen_from`, an update count it read before it
# requested any weight of the iteration (dsgd_async_read_trace_dots, include/dsgd.h).  Two statements per ROW, for 100 % of
# the rows of 100 % of the updates:
#   (a) RULE: recorded decision == !(y d < 0) for the recorded d                      (core/ml/SparseSVM.scala:27-28)
#   (b) RANGE: d is an x . w of weights the iteration CAN have read (core/Slave.scala:92): coordinate by coordinate between
#       the smallest and the largest value the replayed history allows.  What an iteration with commit number c can have
#       seen of coordinate j is W_lo[j], lo = seen_from (all of the updates 1..lo had landed), minus ANY SUBSET of the
#       deltas on j of the updates that may have been landing while it read:
#           S(c) = {lo < u < c}  +  {the first update of every OTHER worker with a commit number > c}
#       (updates land before they commit, in any order across workers; a worker's second update after c starts its sweep
#       after its first one committed, i.e. after c; the iteration's own update c goes out behind its reads; the worker's
#       OWN previous update, commit number read_at, is not in S either: its atomics were drained before the worker
#       requested any weight of this iteration, so it is part of what was seen for certain).  Hence
#           w_j in [W_lo[j] - sum_{u in S} max(delta_u[j], 0),  W_lo[j] - sum_{u in S} min(delta_u[j], 0)]
#       and d in the interval sum_j x_j * [.,.], widened by the fp32 resolution of the dot product and by the accounting
#       error of the replay (statement (A): the replayed weights follow the engine's fp32 weights to ~1e-6 |w|inf).
#       This is RIGOROUS: a correct engine has no row outside, whatever the interleaving.  Also reported, not rigorous:
#       the narrower interval of the coordinates' values over the replayed STATES W_t, lo <= t < c (plus the in-flight
#       part beyond c) -- the states differ from what can be read by the order in which concurrent updates land.
# How sharp the statement is is reported with it: the share of rows whose interval excludes zero (their decision is then
# pinned by the replayed weights alone, whatever d says) and the interval's width against |d|.
DOT_ACC = 1e-6          # accounting error of the replayed weights per coordinate, relative to max(1, |W|inf)  (5e-7 measured)
DOT_RES = 8.0           # fp32 resolution of a dot product of nnz terms: DOT_RES * (nnz + 32) * 2^-24 * sum |x_j| |w_j|


def _lookup(cols_sorted, vals, query):
    """values of a sparse vector (ascending columns) at the columns `query` (0 where absent)"""
    if len(cols_sorted) == 0:
        return np.zeros(len(query))
    pos = np.minimum(np.searchsorted(cols_sorted, query), len(cols_sorted) - 1)
    return np.where(cols_sorted[pos] == query, vals[pos], 0.0)


def _sparse_delta(o, rows, active, s, batch, lr):
    d = forced_delta(o, rows, active, s, batch, lr)
    nz = np.flatnonzero(d)
    return nz.astype(np.int32), d[nz]


def gate_check_recorded_dots(o, w0, split, batch, lr, seed, trace, positional_bug=False, prefix_every=1, collect=False):
    """Statements (a) and (b) above for every row of every update of one traced run started from `w0`.  Returns counts,
    the first few violations (`outside_rule`, `outside_range`: empty for a correct engine) and the sharpness figures.
    prefix_every: the non-rigorous states-only interval is evaluated for every prefix_every-th update (0: never).
    collect: also return the per-row intervals ("lo", "hi": [n, batch]) for the negative controls."""
    worker, it, read_at = np.asarray(trace["worker"]), np.asarray(trace["it"]), np.asarray(trace["read_at"])
    s_rec, mask = np.asarray(trace["s"], dtype=np.float64), np.asarray(trace["mask"])
    seen, dots = np.asarray(trace["seen_from"], dtype=np.int64), np.asarray(trace["dot"], dtype=np.float64)
    n, k_workers, dim1 = len(worker), len(split), o.dim + 1
    out = {"updates": n, "rows": 0, "gate_rows_checked": 0, "empty_rows": 0, "rule_violations": 0, "range_violations": 0,
           "outside_rule": [], "outside_range": [], "rows_pinned_by_the_replay": 0, "states_rows": 0, "states_inside": 0,
           "max_window": 0, "mean_window": 0.0}
    if n == 0:
        out["ok"] = True
        return out
    commit = np.arange(1, n + 1, dtype=np.int64)
    if np.any(seen < 0) or np.any(seen > read_at) or np.any(read_at >= commit):
        raise ValueError("trace inconsistent: seen_from <= read_at < commit violated")
    # pass 1: every update's delta with the recorded decisions (sparse), the rows it sampled
    w = np.asarray(w0, dtype=np.float64).copy()
    rows_of, dcols, dvals = [], [], []
    winf = 1.0
    for c in range(1, n + 1):
        k = int(worker[c - 1])
        b, e = split[k]
        rows = hog_rows(seed, k, int(it[c - 1]), b, e - b, batch, positional_bug)
        cols, vals = _sparse_delta(o, rows, mask[c - 1, :batch], float(s_rec[c - 1]), batch, lr)
        w[cols] -= vals
        winf = max(winf, float(np.abs(w[cols]).max()) if len(cols) else 0.0)
        rows_of.append(rows)
        dcols.append(cols)
        dvals.append(vals)
    w_final = w
    # a worker's updates in commit order, to find "the first update of worker k beyond c"
    by_worker = [np.flatnonzero(worker == k) + 1 for k in range(k_workers)]
    nxt = [0] * k_workers                      # position in by_worker[k] of its first update with a commit number >= c
    window = commit - 1 - seen                 # updates between what is known to be in and the commit
    out["max_window"], out["mean_window"] = int(window.max()), float(window.mean())
    ring_n = 1
    while ring_n < int(window.max()) + 2:
        ring_n *= 2
    if 2 * ring_n * dim1 * 8 > 6 << 30:
        raise MemoryError("a window of %d updates needs a %d-entry ring" % (int(window.max()), ring_n))
    cp_ring, cn_ring = np.zeros((ring_n, dim1)), np.zeros((ring_n, dim1))   # cumulative positive / negative parts after t updates
    cp, cn = np.zeros(dim1), np.zeros(dim1)
    pa, na = np.zeros(dim1), np.zeros(dim1)    # the same over every worker's first update with a commit number >= c
    for k in range(k_workers):
        if len(by_worker[k]):
            u = int(by_worker[k][0]) - 1
            np.add.at(pa, dcols[u], np.maximum(dvals[u], 0.0))
            np.add.at(na, dcols[u], np.minimum(dvals[u], 0.0))
    w0 = np.asarray(w0, dtype=np.float64)
    eps32 = 2.0 ** -24
    acc = DOT_ACC * winf
    lo_all = np.zeros((n, batch)) if collect else None
    hi_all = np.zeros((n, batch)) if collect else None
    width_over_d = []
    for c in range(1, n + 1):
        k = int(worker[c - 1])
        cols_c, vals_c = dcols[c - 1], dvals[c - 1]
        # the iteration's own update is not among what it can have read: out of the in-flight sums ...
        pa[cols_c] -= np.maximum(vals_c, 0.0)
        na[cols_c] -= np.minimum(vals_c, 0.0)
        rows = rows_of[c - 1]
        y = o.label[rows].astype(np.float64)
        flat, rid = _entries(o, rows)
        v = o.val[flat].astype(np.float64)
        cols = o.col[flat]
        lo_t = int(seen[c - 1])
        base = w0[cols] - cp_ring[lo_t % ring_n, cols] - cn_ring[lo_t % ring_n, cols]                  # W_lo
        p = cp[cols] - cp_ring[lo_t % ring_n, cols] + pa[cols]                                        # subsets can take this much off ...
        q = cn[cols] - cn_ring[lo_t % ring_n, cols] + na[cols]                                        # ... or add this much (q <= 0)
        ra = int(read_at[c - 1])
        if ra > lo_t:                           # the worker's own previous update: seen for certain, not a maybe
            own = _lookup(dcols[ra - 1], dvals[ra - 1], cols)
            base = base - own
            p = p - np.maximum(own, 0.0)
            q = q - np.minimum(own, 0.0)
        w_min, w_max = base - np.maximum(p, 0.0), base - np.minimum(q, 0.0)
        t_lo = np.where(v >= 0.0, v * w_min, v * w_max)
        t_hi = np.where(v >= 0.0, v * w_max, v * w_min)
        d_lo = np.bincount(rid, weights=t_lo, minlength=batch)
        d_hi = np.bincount(rid, weights=t_hi, minlength=batch)
        nnz = np.bincount(rid, minlength=batch)
        mag = np.bincount(rid, weights=np.abs(v) * np.maximum(np.abs(w_min), np.abs(w_max)), minlength=batch)
        slack = DOT_RES * (nnz + 32) * eps32 * mag + np.bincount(rid, weights=np.abs(v), minlength=batch) * acc + 1e-30
        d = dots[c - 1, :batch]
        rec = mask[c - 1, :batch].astype(bool)
        live = nnz > 0                          # (a row without non-zeros contributes nothing whatever its decision)
        out["rows"] += batch
        out["empty_rows"] += int((~live).sum())
        out["gate_rows_checked"] += int(live.sum())
        rule_bad = live & (rec != ~(y * d < 0.0))                                                     # (a)
        range_bad = live & ((d < d_lo - slack) | (d > d_hi + slack))                                  # (b)
        out["rule_violations"] += int(rule_bad.sum())
        out["range_violations"] += int(range_bad.sum())
        for r in np.flatnonzero(rule_bad)[:2]:
            if len(out["outside_rule"]) < 5:
                out["outside_rule"].append({"update": c, "row": int(rows[r]), "t": int(r), "d": float(d[r]), "y": float(y[r]), "recorded_active": bool(rec[r])})
        for r in np.flatnonzero(range_bad)[:2]:
            if len(out["outside_range"]) < 5:
                out["outside_range"].append({"update": c, "row": int(rows[r]), "t": int(r), "d": float(d[r]), "range": [float(d_lo[r]), float(d_hi[r])],
                                             "slack": float(slack[r]), "window": int(window[c - 1])})
        pinned = live & ((y * (d_lo - slack) >= 0.0) & (y * (d_hi + slack) >= 0.0) | (y * (d_lo - slack) < 0.0) & (y * (d_hi + slack) < 0.0))
        out["rows_pinned_by_the_replay"] += int(pinned.sum())
        width_over_d.append((d_hi - d_lo)[live] / np.maximum(np.abs(d[live]), 1e-12))
        if collect:
            lo_all[c - 1], hi_all[c - 1] = d_lo - slack, d_hi + slack
        if prefix_every and c % prefix_every == 0:
            # the narrower, NOT rigorous interval: each coordinate somewhere among its values in the states W_t, lo <= t < c
            ucols, inv = np.unique(cols, return_inverse=True)
            ts = np.arange(lo_t, c, dtype=np.int64) % ring_n
            hist = w0[ucols][None, :] - cp_ring[np.ix_(ts, ucols)] - cn_ring[np.ix_(ts, ucols)]
            if ra > lo_t:                       # (every state before the own previous update lacks it: shifted as above)
                pre = (np.arange(lo_t, c) < ra)[:, None] * _lookup(dcols[ra - 1], dvals[ra - 1], ucols)[None, :]
                hist = hist - pre
            s_min, s_max = hist.min(axis=0)[inv] - np.maximum(pa[cols], 0.0), hist.max(axis=0)[inv] - np.minimum(na[cols], 0.0)
            j_lo = np.bincount(rid, weights=np.where(v >= 0.0, v * s_min, v * s_max), minlength=batch)
            j_hi = np.bincount(rid, weights=np.where(v >= 0.0, v * s_max, v * s_min), minlength=batch)
            out["states_rows"] += int(live.sum())
            out["states_inside"] += int((live & (d >= j_lo - slack) & (d <= j_hi + slack)).sum())
        # ... and on to state c: the update joins the history, its worker's next one the in-flight set
        cp[cols_c] += np.maximum(vals_c, 0.0)
        cn[cols_c] += np.minimum(vals_c, 0.0)
        cp_ring[c % ring_n] = cp
        cn_ring[c % ring_n] = cn
        nxt[k] += 1
        if nxt[k] < len(by_worker[k]):
            u = int(by_worker[k][nxt[k]]) - 1
            pa[dcols[u]] += np.maximum(dvals[u], 0.0)
            na[dcols[u]] += np.minimum(dvals[u], 0.0)
    wod = np.concatenate(width_over_d) if width_over_d else np.zeros(1)
    out["width_over_abs_d_median"] = float(np.median(wod))
    out["width_over_abs_d_p10"] = float(np.quantile(wod, 0.1))
    out["share_pinned_by_the_replay"] = out["rows_pinned_by_the_replay"] / max(1, out["gate_rows_checked"])
    out["states_share_inside"] = out["states_inside"] / max(1, out["states_rows"]) if out["states_rows"] else None
    out["ok"] = out["rule_violations"] == 0 and out["range_violations"] == 0
    out["w_replayed"] = w_final
    if collect:
        out["lo"], out["hi"] = lo_all, hi_all
    return out
