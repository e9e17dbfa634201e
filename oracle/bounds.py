"""ORACLE side (test infrastructure): the DERIVED error bound of a whole-range synchronous step.

The engine's whole-range gradient is a sum of fixed-point contributions round(y*x * 2^shift / vmax2) accumulated
EXACTLY (integers) and rounded to fp32 once.  Against the fp64 oracle fed the same weights the update of coordinate
j can therefore differ by at most

    lr/K * ( cnt_j * 2^-(shift+1) * vmax2        every contribution is off by at most half a grid unit
           + near_j )                            rows whose margin is within 1e-5 of zero may be gated differently
    + 8 * 2^-24 * (|w_j| + |w_j - w_j_before|)   fp32 roundings of the sum, the mean, the product and the subtraction
    + 2 * 2^-24 * lr/K * sum_k |g_k,j|           (index lists) EACH worker's exact sum is rounded to fp32 once and regularised
                                                 before the workers are folded: when two workers' gradients nearly cancel
                                                 in a coordinate, those roundings are large against the NET update the
                                                 line above prices (found at 2 workers x batch 7, lr 1)
    + 1e-9                                       the regulariser scalar s = 2*lambda*(w.ds) in fp32 vs fp64

cnt_j = non-zeros of column j in the rows of the step, near_j = sum of |x_j| over the near-zero-margin rows
(oracle.c orc_range_gate_profile).  No blanket tolerance: tests and bench.py's parity gate assert this bound per
coordinate and report the worst ratio error / bound.
"""

from __future__ import annotations

import numpy as np

GATE_EPS = 1e-5


def vmax2_of(val):
    m = float(np.abs(val).max()) if len(val) else 1.0
    if m <= 0.0:
        return 1.0
    return float(2.0 ** np.ceil(np.log2(m)))


def column_counts(o, lo, hi):
    b, e = int(o.row_ptr[lo]), int(o.row_ptr[hi])
    keep = np.abs(o.val[b:e]) > 1e-20
    return np.bincount(o.col[b:e][keep], minlength=o.dim + 1).astype(np.float64)


def step_bound(o, w_before, w_after_ref, ranges, lr, shift, vmax2=None, parts=False):
    """Per-coordinate bound on |w_engine - w_after_ref| after ONE synchronous step over `ranges` (one range per worker,
    mean over the workers) starting from w_before on both sides.  Returns (tol vector, rows near the gate)."""
    if vmax2 is None:
        vmax2 = vmax2_of(o.val)
    k = len(ranges)
    cnt = np.zeros(o.dim + 1)
    near = np.zeros(o.dim + 1)
    n_near = 0
    for lo, hi in ranges:
        cnt += column_counts(o, lo, hi)
        n, l1 = o.gate_profile(np.ascontiguousarray(w_before, dtype=np.float64), lo, hi, GATE_EPS)
        n_near += n
        near += l1
    quantum = vmax2 * 2.0 ** (-(shift + 1))
    tol = (lr / k) * (cnt * quantum + near)
    tol += 8.0 * 2.0 ** -24 * (np.abs(w_after_ref) + np.abs(w_after_ref - w_before)) + 1e-9
    if parts:
        return tol, n_near, (lr / k) * near   # (tol - this = the bound with every near-gate row gated as the oracle does)
    return tol, n_near


def worst_ratio(w_engine, w_ref, tol):
    r = np.abs(np.asarray(w_engine, dtype=np.float64) - w_ref) / tol
    j = int(np.argmax(r))
    return float(r[j]), j


def _list_profile(o, w, rows, eps):
    """(cnt_j, near_j, rows near the gate) over the rows of an index list (numpy restatement of column_counts +
    orc_range_gate_profile for rows that are not a contiguous range)."""
    rows = np.asarray(rows, dtype=np.int64)
    starts = o.row_ptr[rows]
    lens = o.row_ptr[rows + 1] - starts
    total = int(lens.sum())
    if total == 0:
        z = np.zeros(o.dim + 1)
        return z, z.copy(), 0
    first = np.cumsum(lens) - lens
    flat = np.arange(total, dtype=np.int64) + np.repeat(starts - first, lens)
    row_id = np.repeat(np.arange(len(rows), dtype=np.int64), lens)
    cols = o.col[flat]
    vals = o.val[flat].astype(np.float64)
    keep = np.abs(vals) > 1e-20                       # math/Sparse.scala:108-118
    cnt = np.bincount(cols[keep], minlength=o.dim + 1).astype(np.float64)
    prod = vals * np.asarray(w, dtype=np.float64)[cols]
    prod[np.abs(prod) <= 1e-20] = 0.0                 # math/Sparse.scala:46 -> :112-114
    d = np.bincount(row_id, weights=prod, minlength=len(rows))
    near_rows = (np.abs(d) > 0.0) & (np.abs(d) < eps)
    mask = near_rows[row_id] & keep
    near = np.bincount(cols[mask], weights=np.abs(vals[mask]), minlength=o.dim + 1)
    return cnt, near, int(near_rows.sum())


def list_bound(o, w_before, w_after_ref, lists, lr, shift, vmax2=None, parts=False):
    """step_bound for index lists (one list per worker, mean over the workers): the index-list kernels accumulate the
    same fixed-point contributions exactly, at the shift the launch reports."""
    if vmax2 is None:
        vmax2 = vmax2_of(o.val)
    k = len(lists)
    cnt = np.zeros(o.dim + 1)
    near = np.zeros(o.dim + 1)
    n_near = 0
    for rows in lists:
        c, nr, n = _list_profile(o, w_before, rows, GATE_EPS)
        cnt += c
        near += nr
        n_near += n
    quantum = vmax2 * 2.0 ** (-(shift + 1))
    tol = (lr / k) * (cnt * quantum + near)
    tol += 8.0 * 2.0 ** -24 * (np.abs(w_after_ref) + np.abs(w_after_ref - w_before)) + 1e-9
    # every worker's own regularised sum, rounded once before the fold over the workers (Vec.sum, math/Vec.scala:128-131)
    keep = o.last_stats
    gabs = np.zeros(o.dim + 1)
    for rows in lists:
        gabs += np.abs(o.gradient(np.ascontiguousarray(w_before, dtype=np.float64), rows))
    o.last_stats = keep
    tol += 2.0 * 2.0 ** -24 * (lr / k) * gabs
    if parts:
        return tol, n_near, (lr / k) * near
    return tol, n_near
