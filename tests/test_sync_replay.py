"""CPU: oracle/sync_replay.py (the forced-decision replay of a long synchronous run) pinned against oracle.c itself.

With the decisions oracle.c takes on its own trajectory on record, the replay must land on oracle.c's weights (two
independent restatements of core/Master.scala:184-197 + core/Slave.scala:147-155 + core/ml/SparseSVM.scala:26-31: C with
per-row loops, numpy with segment sums), find no differing decision, and its statements must be able to FAIL: a dropped
row and a doubled step break the accounting; a decision flipped on a row far from the gate breaks "all differing rows
are near the gate"; a wrong scalar breaks "the recorded s is the replayed one"."""

import numpy as np

import dsgd_amd
from oracle import oracle as orc
from oracle import ref_dict as rd
from oracle import sync_replay as sr

LAM = 1e-5


def own_decisions(o, w, lists, width):
    rows = np.concatenate(lists)
    m = np.zeros(width, dtype=bool)
    for r, row in enumerate(rows):
        m[r] = not (o.label[row] * o.row_dot(int(row), w) < 0.0)
    return m


def run_oracle(o, steps, lr, dim, width):
    w = np.zeros(dim + 1)
    masks, s_used = [], []
    for lists in steps:
        masks.append(own_decisions(o, w, lists, width))
        prod = w * o.ds
        s_used.append(2.0 * LAM * float(prod[np.abs(prod) > 1e-20].sum()))
        o.sync_step(w, lists, lr)
    return w, np.asarray(masks), np.asarray(s_used)


def test_replay_reproduces_the_oracle_and_its_statements_can_fail():
    data = dsgd_amd.synth.generate(4000, seed=41)
    n_train = 3200
    o = orc.Oracle(data.dim, data.row_ptr, data.col, data.val, data.label, LAM)
    o.set_dim_sparsity(o.dim_sparsity(n_train))
    rng = np.random.default_rng(4)
    split = rd.split_vanilla(n_train, 3)
    steps = [[rng.permutation(np.asarray(r))[:100].astype(np.int32) for r in split] for _ in range(12)]
    w_o, masks, s_used = run_oracle(o, steps, 0.5, data.dim, 320)
    w = np.zeros(data.dim + 1)
    st = sr.replay(o, w, steps, 0.5, masks, s_used)
    v = sr.verdict(st, w_o, w)
    assert st["differing"] == 0 and v["first_divergent_step"] is None and v["divergent_rows_all_near_gate"]
    assert v["account_max_abs_err"] < 1e-12 and v["accounting_agrees"] and v["s_agrees"] and v["s_max_abs_err"] < 1e-15
    assert st["decisions"] == 12 * 300
    # -- negative controls: each statement can fail --
    for fault in ("drop_row", "double_step"):
        wb = np.zeros(data.dim + 1)
        vb = sr.verdict(sr.replay(o, wb, steps, 0.5, masks, s_used, fault=fault), w_o, wb)
        assert not vb["accounting_agrees"], fault
    flipped = masks.copy()
    # a row decided against a CLEAR margin (the largest |margin| of step 3 on the oracle's trajectory)
    w3 = np.zeros(data.dim + 1)
    for lists in steps[:3]:
        o.sync_step(w3, lists, 0.5)
    rows = np.concatenate(steps[3])
    margins = np.array([abs(o.row_dot(int(r), w3)) for r in rows])
    flipped[3, int(margins.argmax())] ^= True
    wf = np.zeros(data.dim + 1)
    stf = sr.replay(o, wf, steps, 0.5, flipped, s_used)
    vf = sr.verdict(stf, wf.copy(), wf)       # (accounting against itself: only the gate statement is under test)
    assert stf["differing"] >= 1 and vf["first_divergent_step"] == 3 and not vf["divergent_rows_all_near_gate"]
    assert vf["outside"][0]["step"] == 3 and vf["worst_margin_over_resolution"] > 100
    wb = np.zeros(data.dim + 1)
    vs = sr.verdict(sr.replay(o, wb, steps, 0.5, masks, s_used + 1e-3), w_o, wb)
    assert not vs["s_agrees"] and vs["accounting_agrees"]


def test_a_near_gate_flip_is_accepted_and_named():
    """A row decided differently INSIDE the fp32 resolution of its dot product is what fp32 against fp64 does: the replay
    follows the forced decision, names the step, and the gate statement holds."""
    data = dsgd_amd.synth.generate(3000, seed=43)
    n_train = 2400
    o = orc.Oracle(data.dim, data.row_ptr, data.col, data.val, data.label, LAM)
    o.set_dim_sparsity(o.dim_sparsity(n_train))
    rng = np.random.default_rng(6)
    steps = [[rng.permutation(n_train)[:100].astype(np.int32)]]
    # the step runs from w = 0: every margin is exactly 0, every row active (>= 0); a side that saw -0-ish noise would stop some
    w_o, own, s_used = run_oracle(o, steps, 0.5, data.dim, 128)
    forced = own.copy()
    forced[0, 5] = False                     # margin exactly 0: inside any resolution
    w = np.zeros(data.dim + 1)
    st = sr.replay(o, w, steps, 0.5, forced, None)
    v = sr.verdict(st, w.copy(), w)
    assert v["first_divergent_step"] == 0 and st["differing"] >= 1 and v["divergent_rows_all_near_gate"]
