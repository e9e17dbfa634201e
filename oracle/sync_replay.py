"""ORACLE (test infrastructure, NOT product code) -- replay of a LONG synchronous run with the engine's own gate decisions.

Only tests/ and bench.py's parity legs may import this module.

Why.  The synchronous step (core/Master.scala:184-197 over core/Slave.scala:147-155) decides every sampled row on a gate,
y (x . w) >= 0 (core/ml/SparseSVM.scala:27-28).  The engine computes x . w in fp32, the reference (and oracle.c) in fp64:
once in a few thousand steps a row sits closer to the gate than fp32 resolves, the two sides decide it differently, the
updates differ by that row's y x lr / K -- and a constant-step run never forgets it (round 4: 3 workers x batch 100 on
804,414 rows, max |w - w_oracle| = 0.44 after ONE epoch, "explained" by two steps with a row near the gate).  A free-running
comparison cannot tell such a flip from a bug.

What.  The engine records, per step of a plan, the gate decision of every row and the regulariser scalar it used
(include/dsgd.h: dsgd_plan_record).  With those decisions FORCED, this module recomputes every step in fp64 exactly as
the reference would (the per-worker sums of y x over the rows the engine let through, the support-only regulariser
s = 2 lambda (w . ds) of core/ml/SparseSVM.scala:31 from the REPLAYED weights, the mean over the workers, the update) and
states three things a test can assert:

  (a) accounting: the engine's final weights are the replayed ones to rounding -- every row the engine let through was
      added once, to its worker's sum, regularised, averaged, scaled as the reference does; nothing else moved w;
  (b) every decision that DIFFERS from the gate of the replayed weights (the oracle's own decision at that point of the
      trajectory) belongs to a row whose margin |y (x . w)| lies inside the fp32 resolution of that row's dot product;
  (c) the first step at which the two sides decide differently -- where a free-running oracle leaves the trajectory.

A gate bug (a row let through at a clearly negative margin, or stopped at a clearly positive one) breaks (b); a lost,
doubled or mis-scaled contribution breaks (a); the recorded scalar is held to the replayed one as well.
"""

from __future__ import annotations

import numpy as np

EPS32 = 2.0 ** -24          # unit round-off of fp32
SPARSE_EPS = 1e-20          # math/Sparse.scala:104
ACCOUNT_TOL = 1e-5          # (a): max |w_engine - w_replay| <= ACCOUNT_TOL * max(1, |w_replay|_inf)  (BASELINE.md's stated tolerance)


def _filt(v):
    return np.where(np.abs(v) > SPARSE_EPS, v, 0.0)


def replay(o, w, steps, lr, masks, s_used=None, fault=None):
    """Replay `steps` (list of steps, each a list of per-worker index arrays) on `w` (float64, D + 1, IN PLACE) with the
    decisions `masks` (bool [n_steps, >= rows per step]: bit r = row r of the step, workers in order, lists in order).

    Returns a dict of per-run statistics; per differing decision the row's margin and its fp32 resolution are kept.
    `fault` (negative controls): "drop_row" ignores one active row of step 0, "double_step" applies step 1's update twice."""
    row_ptr, col, val, label = o.row_ptr, o.col, o.val.astype(np.float64), o.label.astype(np.float64)
    ds, lam = o.ds, o.lam
    n_dec = n_diff = 0
    first_div = None
    diffs = []            # (step, row in step, global row, margin, resolution)
    s_err = 0.0
    w_inf = 0.0
    for t, lists in enumerate(steps):
        K = len(lists)
        rows = np.concatenate([np.asarray(a, dtype=np.int64) for a in lists])
        worker = np.repeat(np.arange(K), [len(a) for a in lists])
        R = len(rows)
        forced = np.asarray(masks[t][:R], dtype=bool).copy()
        if fault == "drop_row" and t == 0 and forced.any():
            forced[np.flatnonzero(forced)[0]] = False
        st, en = row_ptr[rows], row_ptr[rows + 1]
        lens = en - st
        seg = np.repeat(np.arange(R), lens)
        flat = np.arange(int(lens.sum()), dtype=np.int64) + np.repeat(st - np.concatenate([[0], np.cumsum(lens)[:-1]]), lens)
        c, v = col[flat], val[flat]
        prod = _filt(v * w[c])                                           # math/Sparse.scala:46 + the constructor's filter
        dot = np.bincount(seg, weights=prod, minlength=R)
        absdot = np.bincount(seg, weights=np.abs(prod), minlength=R)
        margin = label[rows] * dot
        own = ~(margin < 0.0)                                            # core/ml/SparseSVM.scala:27-28
        n_dec += R
        differ = np.flatnonzero(own != forced)
        if len(differ):
            n_diff += len(differ)
            if first_div is None:
                first_div = t
            absx = np.bincount(seg, weights=np.abs(v), minlength=R)
            for r in differ:
                diffs.append((t, int(r), int(rows[r]), float(margin[r]), float(absdot[r]), int(lens[r]), float(absx[r])))
        # the step with the FORCED decisions
        s = 2.0 * lam * float(_filt(w * ds).sum())                       # SparseSVM.scala:31 (regularize's scalar)
        if s_used is not None:
            s_err = max(s_err, abs(s - float(s_used[t])))
        add = (s != 0.0) and (abs(s) > SPARSE_EPS)
        gsum = np.zeros_like(w)
        for k in range(K):
            sel = forced[seg] & (worker[seg] == k)
            g = np.bincount(c[sel], weights=(label[rows][seg] * v)[sel], minlength=len(w))   # Vec.sum of the active y x
            g = _filt(g)
            if add:
                g = np.where(g != 0.0, _filt(g + s), g)                  # support only (math/Vec.scala:65-75)
            gsum = _filt(gsum + g)
        upd = _filt(_filt(gsum / K) * lr)                                # Vec.mean, learningRate * grad (Master.scala:194-197)
        reps = 2 if (fault == "double_step" and t == 1) else 1
        for _ in range(reps):
            w[:] = np.where(gsum != 0.0, _filt(w - upd), w)
        w_inf = max(w_inf, float(np.abs(w).max()))
    return {"steps": len(steps), "decisions": n_dec, "differing": n_diff, "first_divergent_step": first_div, "diffs": diffs,
            "s_max_abs_err": s_err, "w_inf_max": w_inf}


def verdict(stats, w_engine, w_replay, slack=8.0):
    """The three statements.  The resolution of a row's fp32 dot product: (nnz + 32) * 2^-24 * sum_j |x_j w_j| (sequential
    and pairwise sums alike stay below nnz roundings of the partial sums) plus what the weights themselves may differ by
    between the two sides at that point, sum_j |x_j| * delta_w, with delta_w the measured accounting error of the whole
    run (the engine's intermediate weights are not on record; `slack` covers that it is taken at the end)."""
    w_engine = np.asarray(w_engine, dtype=np.float64)
    err = float(np.abs(w_engine - w_replay).max())
    scale = max(1.0, float(np.abs(w_replay).max()))
    delta_w = max(err, EPS32 * scale)
    worst = 0.0
    outside = []
    for (t, r, row, m, absdot, nnz, absx) in stats["diffs"]:
        res = slack * ((nnz + 32) * EPS32 * absdot + absx * delta_w) + 1e-30
        worst = max(worst, abs(m) / res)
        if abs(m) > res:
            outside.append({"step": t, "row_of_step": r, "row": row, "margin": m, "resolution": res})
    return {"account_max_abs_err": err, "account_tolerance": ACCOUNT_TOL * scale, "account_err_over_tol": err / (ACCOUNT_TOL * scale),
            "accounting_agrees": bool(err <= ACCOUNT_TOL * scale),
            "decisions": stats["decisions"], "differing_decisions": stats["differing"],
            "first_divergent_step": stats["first_divergent_step"],
            "divergent_rows_all_near_gate": bool(not outside), "worst_margin_over_resolution": worst, "outside": outside[:5],
            "s_max_abs_err": stats["s_max_abs_err"],
            "s_agrees": bool(stats["s_max_abs_err"] <= 1e-5 * max(1.0, 2.0 * stats["w_inf_max"]))}
