"""CPU: the ORACLE-side replay of a traced lock-free run (oracle/hogwild_replay.py) checked on its own.

A trace is synthesised from a schedule model written independently here (k workers of equal speed: every gradient is
computed on the weights as they were k - 1 updates before its commit, looked up in the full history of weight vectors,
gates and scalar recorded as the engine records them); replay_forced fed that trace must land on the same weights with
no gate or scalar disagreement, every negative control (a deliberately wrong update rule) must break the accounting
statement, and the experiment that motivates recording the gate decisions -- a 1e-7 perturbation of the initial weights
moves a re-simulated run macroscopically -- is kept as a test."""

import numpy as np
import pytest

import dsgd_amd
from dsgd_amd import host
from oracle import hogwild_replay as hr
from oracle import oracle as orc


def problem(n_rows=6000, lam=1e-5, seed=31):
    data = dsgd_amd.synth.generate(n_rows, seed=seed)
    n_train = int(n_rows * 0.8)
    o = orc.Oracle(data.dim, data.row_ptr, data.col, data.val, data.label, lam)
    o.set_dim_sparsity(o.dim_sparsity(n_train))
    return data, o, n_train


def model_run(o, split, batch, lr, seed, n_updates, w0=None):
    """k equally fast workers, written without the replay's machinery: the whole history of weight vectors is kept and
    update c reads entry max(0, c - k); the update itself is the oracle's orc_async_step (core/Slave.scala:92-101), the
    gates and the scalar are recorded from the weights it read.  Returns (final weights, trace)."""
    k = len(split)
    hist = [np.zeros(o.dim + 1) if w0 is None else w0.copy()]
    tr = {q: [] for q in ("worker", "it", "read_at", "s", "n_active", "mask")}
    for c in range(1, n_updates + 1):
        j, i, r = (c - 1) % k, (c - 1) // k, max(0, c - k)
        b, e = split[j]
        rows = hr.hog_rows(seed, j, i, b, e - b, batch)
        snap = hist[r]
        y = o.label[rows].astype(np.float64)
        active = ~(y * np.asarray([o.row_dot(int(t), snap) for t in rows]) < 0.0)
        delta = o.async_step(snap.copy(), rows, lr, want_delta=True)
        w = hist[-1] - delta
        w[np.abs(w) <= 1e-20] = 0.0
        hist.append(w)
        m = np.zeros(32 * ((batch + 31) // 32), dtype=bool)
        m[:batch] = active
        for q, v in (("worker", j), ("it", i), ("read_at", r), ("s", np.float32(2.0 * o.lam * (snap @ o.ds))),
                     ("n_active", int(active.sum())), ("mask", m)):
            tr[q].append(v)
    return hist[-1], {"worker": np.asarray(tr["worker"], np.int32), "it": np.asarray(tr["it"], np.uint32),
                      "read_at": np.asarray(tr["read_at"], np.int64), "s": np.asarray(tr["s"], np.float32),
                      "n_active": np.asarray(tr["n_active"], np.int32), "mask": np.stack(tr["mask"])}


@pytest.mark.parametrize("k", [1, 4, 16])
def test_forced_replay_reproduces_a_modelled_schedule(k):
    data, o, n_train = problem()
    split = [(r.start, r.stop) for r in host.split_vanilla(n_train, k)]
    w_model, trace = model_run(o, split, 50, 0.5, 99, 160)
    w = np.zeros(o.dim + 1)
    st = hr.replay_forced(o, w, split, 50, 0.5, 99, trace)
    v = hr.verdict(o, w_model, w, st, (n_train, data.n_rows))
    assert st["updates"] == 160 and st["max_lag"] == k - 1
    # the recorded scalar is fp32: |s| ~ 1e-6, so the weights agree to ~1e-13 rather than bit for bit
    assert v["account_max_abs_err"] <= 1e-9 and all(v["ok"].values()), v
    assert v["gate_differs_at_fraction"][0] == 0.0 and v["s_rel_err_max"] <= 1e-6


def test_every_negative_control_breaks_the_accounting():
    """What the traced check is FOR: a broken update rule must be rejected -- including ONE lost update and, now that the
    gates are forced, a missing regulariser at the reference's lambda = 1e-5 (a re-simulation could not see either)."""
    data, o, n_train = problem()
    k = 8
    split = [(r.start, r.stop) for r in host.split_vanilla(n_train, k)]
    w_model, trace = model_run(o, split, 100, 0.5, 7, 400)
    out = {}
    for fault in hr.FAULTS:
        w = np.zeros(o.dim + 1)
        st = hr.replay_forced(o, w, split, 100, 0.5, 7, trace, fault=fault, check=False)
        out[fault] = hr.verdict(o, w_model, w, st)["account_err_over_tol"]
    for fault in ("double_apply", "drop_one", "sum_not_mean", "half_step", "wrong_rows"):
        assert out[fault] > 10.0, out
    # 400 updates at lambda = 1e-5 leave the regulariser's total at ~1e-4 of a weight: visible, though not yet beyond the
    # tolerance of a run this short (the 8,000-update GPU run is where it must be -- tests/test_gpu_hogwild_trace.py)
    assert out["no_regulariser"] > 1e-3, out


def test_a_resimulation_cannot_follow_a_perturbed_run():
    """Why the gates are recorded: the same schedule re-simulated (gates re-decided) from initial weights that differ by
    1e-7 on 100 coordinates ends a third of the norm away after 400 updates -- every margin starts AT the gate."""
    data, o, n_train = problem(n_rows=20000)
    k = 4
    split = [(r.start, r.stop) for r in host.split_vanilla(n_train, k)]
    w_a, _ = model_run(o, split, 100, 0.5, 4242, 400)
    w0 = np.zeros(o.dim + 1)
    w0[np.random.default_rng(0).integers(1, 47000, 100)] = 1e-7
    w_b, _ = model_run(o, split, 100, 0.5, 4242, 400, w0=w0)
    assert np.sqrt(((w_a - w_b) ** 2).sum()) > 0.05 * np.sqrt((w_a ** 2).sum())


def test_gates_are_hypersensitive_to_the_snapshot():
    """Why statement (B) is a profile and not a row-by-row check: the recorded gates of a MODELLED 4-worker schedule held
    against a snapshot that is off by ONE update disagree on ~12 % of the rows (two: ~24 %) -- the margins of a
    constant-step run are as small as what one update moves them by."""
    data, o, n_train = problem(n_rows=20000)
    k = 4
    split = [(r.start, r.stop) for r in host.split_vanilla(n_train, k)]
    _, tr = model_run(o, split, 100, 0.5, 4242, 300)
    commit = np.arange(1, 301)
    out = {}
    for shift in (0, 1, 2):
        t2 = dict(tr, read_at=np.clip(tr["read_at"] + shift, 0, commit - 1))
        st = hr.replay_forced(o, np.zeros(o.dim + 1), split, 100, 0.5, 4242, t2, fractions=(0.0,))
        out[shift] = st["gate_differs"][0] / st["rows"]
    assert out[0] == 0.0 and out[1] > 0.05 and out[2] > out[1], out


def test_inconsistent_traces_are_refused():
    data, o, n_train = problem(n_rows=2000)
    split = [(0, n_train)]
    w = np.zeros(o.dim + 1)
    _, good = model_run(o, split, 10, 0.5, 1, 3)
    bad = dict(good, read_at=np.asarray([0, 2, 1]))   # an update cannot have read the weights its own commit produced
    with pytest.raises(ValueError):
        hr.replay_forced(o, w, split, 10, 0.5, 1, bad)
    bad = dict(good, n_active=good["n_active"] + 1)
    with pytest.raises(ValueError):
        hr.replay_forced(o, w, split, 10, 0.5, 1, bad)
    empty = {q: v[:0] for q, v in good.items()}
    assert hr.replay_forced(o, w, split, 10, 0.5, 1, empty)["updates"] == 0


def test_small_lag_gate_check_on_a_modelled_schedule():
    """hogwild_replay.gate_check_small_lag (the few-worker gate statement with teeth) on schedules whose staleness is known:
    one worker (nothing in flight: every decision IS the gate at both ends) and two workers (exactly one update in flight,
    the snapshot is the read end): no decision differs at both ends; flipped on a clear margin, one does and is outside."""
    data, o, n_train = problem()
    for k in (1, 2):
        split = [(r.start, r.stop) for r in host.split_vanilla(n_train, k)]
        _, trace = model_run(o, split, 50, 0.5, 7, 120)
        g = hr.gate_check_small_lag(o, np.zeros(o.dim + 1), split, 50, 0.5, 7, trace, max_lag=1)
        assert g["ok"] and g["updates_checked"] == 120 and g["rows_checked"] == 120 * 50 and g["differ_at_both_ends"] == 0, g
        bad = {q: np.array(v, copy=True) for q, v in trace.items()}
        c = 60                                              # flip the clearest row of update 60
        w = np.zeros(o.dim + 1)
        for cc in range(1, c):
            j = int(trace["worker"][cc - 1])
            rows = hr.hog_rows(7, j, int(trace["it"][cc - 1]), split[j][0], split[j][1] - split[j][0], 50)
            w -= hr.forced_delta(o, rows, trace["mask"][cc - 1, :50], float(trace["s"][cc - 1]), 50, 0.5)
        j = int(trace["worker"][c - 1])
        rows = hr.hog_rows(7, j, int(trace["it"][c - 1]), split[j][0], split[j][1] - split[j][0], 50)
        bad["mask"][c - 1, int(np.abs(hr.margins(o, w, rows)).argmax())] ^= True
        gb = hr.gate_check_small_lag(o, np.zeros(o.dim + 1), split, 50, 0.5, 7, bad, max_lag=1)
        assert not gb["ok"] and gb["outside"][0]["update"] == c, gb
    # beyond the lag the check is asked for, updates are left out (16 equally fast workers: every update has 15 in flight)
    split = [(r.start, r.stop) for r in host.split_vanilla(n_train, 16)]
    _, trace = model_run(o, split, 50, 0.5, 7, 64)
    g = hr.gate_check_small_lag(o, np.zeros(o.dim + 1), split, 50, 0.5, 7, trace, max_lag=1)
    assert g["updates_checked"] == 2 and g["ok"]            # (only the first two updates of the run have lag <= 1)


# ---- every decision held to the x . w it was taken on (hogwild_replay.gate_check_recorded_dots, round 6) ----------------
def landing_model_run(o, split, batch, lr, seed, n_updates, rng, stale_bug=0):
    """A schedule model with what makes the real engine hard: workers advance in RANDOM interleaving, an update LANDS
    coordinate by coordinate (three chunks at three different moments) BEFORE it draws its commit number, and an
    iteration READS its weights in three chunks at three different moments -- so what it sees is a mixture no single
    replayed state equals.  `seen_from` is the commit count a worker notes before it reads anything of its next iteration
    (and before its own commit), as the engine does.  stale_bug > 0: the worker gates on weights it cached that many
    iterations ago (the kind of defect the range statement exists to catch).  Returns the trace with dots / seen_from."""
    k = len(split)
    w_live = np.zeros(o.dim + 1)
    w_committed = np.zeros(o.dim + 1)     # (w_live also holds what uncommitted updates have landed when the run ends)
    commits = 0
    tr = {q: [] for q in ("worker", "it", "read_at", "s", "n_active", "mask", "seen_from", "dot")}
    st = [{"phase": 0, "it": 0, "read_at": 0, "seen": 0, "seen_next": 0} for _ in range(k)]
    while commits < n_updates:
        j = int(rng.integers(k))
        s = st[j]
        ph = s["phase"]
        if ph == 0:      # the iteration's sample; first third of the reads
            b, e = split[j]
            s["rows"] = hr.hog_rows(seed, j, s["it"], b, e - b, batch)
            flat, rid = hr._entries(o, s["rows"])
            s["flat"], s["rid"] = flat, rid
            s["cols"] = o.col[flat]
            s["part"] = rng.integers(3, size=len(flat))
            s["wv"] = np.zeros(len(flat))
            s["s"] = np.float32(2.0 * o.lam * (w_live @ o.ds))
        if ph in (0, 1, 2):
            m = s["part"] == ph
            s["wv"][m] = w_live[s["cols"][m]]
        if ph == 2:      # x . w in fp32, the gate, the update to land
            wv = s["wv"]
            prod = (o.val[s["flat"]].astype(np.float32) * wv.astype(np.float32)).astype(np.float32)
            d = np.zeros(batch, dtype=np.float32)
            np.add.at(d, s["rid"], prod)
            if stale_bug and s["it"] >= stale_bug:
                d = s["dhist"][-stale_bug] * np.float32(1.0)   # the x . w of ANOTHER sample, many updates ago: weights AND rows stale
            s.setdefault("dhist", []).append(d.copy())
            y = o.label[s["rows"]].astype(np.float64)
            s["active"] = ~(y * d.astype(np.float64) < 0.0)
            s["d"] = d
            delta = hr.forced_delta(o, s["rows"], s["active"], float(s["s"]), batch, lr)
            nz = np.flatnonzero(delta)
            s["dcols"], s["dvals"] = nz, delta[nz]
            s["dpart"] = rng.integers(3, size=len(nz))
        if ph in (3, 4, 5):   # landing, a third of the coordinates at a time
            m = s["dpart"] == ph - 3
            w_live[s["dcols"][m]] -= s["dvals"][m]
        if ph == 6:      # noted before the next iteration requests any weight, and before the own commit
            s["seen_next"] = commits
        if ph == 7:      # commit
            commits += 1
            w_committed[s["dcols"]] -= s["dvals"]
            mk = np.zeros(32 * ((batch + 31) // 32), dtype=bool)
            mk[:batch] = s["active"]
            for q, v in (("worker", j), ("it", s["it"]), ("read_at", s["read_at"]), ("s", s["s"]), ("n_active", int(s["active"].sum())),
                         ("mask", mk), ("seen_from", s["seen"]), ("dot", s["d"])):
                tr[q].append(v)
            s["read_at"], s["seen"], s["it"] = commits, s["seen_next"], s["it"] + 1
        s["phase"] = (ph + 1) % 8
    return w_committed, {"worker": np.asarray(tr["worker"], np.int32), "it": np.asarray(tr["it"], np.uint32),
                    "read_at": np.asarray(tr["read_at"], np.int64), "s": np.asarray(tr["s"], np.float32),
                    "n_active": np.asarray(tr["n_active"], np.int32), "mask": np.stack(tr["mask"]),
                    "seen_from": np.asarray(tr["seen_from"], np.int64), "dot": np.stack(tr["dot"])}


@pytest.mark.parametrize("k,seed", [(1, 0), (3, 1), (8, 2), (24, 3)])
def test_recorded_dots_of_a_landing_model_are_inside_their_ranges(k, seed):
    """No false alarm, whatever the interleaving: rule and range hold for every row of every update of the landing model;
    the weights the check replays are the model's to rounding."""
    data, o, n_train = problem()
    split = [(r.start, r.stop) for r in host.split_vanilla(n_train, k)]
    n_upd = 60 + 12 * k
    w_model, trace = landing_model_run(o, split, 40, 0.5, 5, n_upd, np.random.default_rng(seed))
    g = hr.gate_check_recorded_dots(o, np.zeros(o.dim + 1), split, 40, 0.5, 5, trace)
    assert g["ok"] and g["gate_rows_checked"] + g["empty_rows"] == n_upd * 40 == g["rows"], {q: v for q, v in g.items() if q != "w_replayed"}
    assert np.abs(g["w_replayed"] - w_model).max() <= 1e-9
    if k == 1:   # nothing in flight: the interval is a point (to resolution), every clear row is pinned by the replay alone
        assert g["max_window"] <= 1 and g["share_pinned_by_the_replay"] > 0.9 and g["states_share_inside"] == 1.0


def test_recorded_dots_controls_are_caught():
    """Teeth: (1) a decision flipped against its own recorded d breaks the RULE; (2) a d moved outside its range breaks the
    RANGE; (3) an engine that gates on STALE products (the x . w it computed three iterations earlier) is caught on a
    large share of its rows although every decision still obeys the rule on the recorded d."""
    data, o, n_train = problem()
    k = 8
    split = [(r.start, r.stop) for r in host.split_vanilla(n_train, k)]
    _, trace = landing_model_run(o, split, 40, 0.5, 5, 200, np.random.default_rng(11))
    good = hr.gate_check_recorded_dots(o, np.zeros(o.dim + 1), split, 40, 0.5, 5, trace, collect=True)
    assert good["ok"]
    bad = {q: np.array(v, copy=True) for q, v in trace.items()}
    bad["mask"][120, 7] ^= True
    bad["n_active"][120] += 1 if bad["mask"][120, 7] else -1
    g1 = hr.gate_check_recorded_dots(o, np.zeros(o.dim + 1), split, 40, 0.5, 5, bad)
    assert not g1["ok"] and g1["rule_violations"] >= 1 and g1["outside_rule"][0]["update"] == 121
    bad = {q: np.array(v, copy=True) for q, v in trace.items()}
    width = good["hi"][150, 3] - good["lo"][150, 3]
    bad["dot"][150, 3] = np.float32(good["hi"][150, 3] + 0.05 * width + 1e-4)
    y = float(o.label[hr.hog_rows(5, int(trace["worker"][150]), int(trace["it"][150]), *_span(split, trace, 150), 40)[3]])
    bad["mask"][150, 3] = not (y * float(bad["dot"][150, 3]) < 0.0)          # (keeps the rule: only the range can object)
    bad["n_active"][150] = int(bad["mask"][150, :40].sum())
    g2 = hr.gate_check_recorded_dots(o, np.zeros(o.dim + 1), split, 40, 0.5, 5, bad)
    assert not g2["ok"] and g2["rule_violations"] == 0 and any(v["update"] == 151 and v["t"] == 3 for v in g2["outside_range"]), g2["outside_range"]
    _, stale = landing_model_run(o, split, 40, 0.5, 5, 200, np.random.default_rng(11), stale_bug=3)
    g3 = hr.gate_check_recorded_dots(o, np.zeros(o.dim + 1), split, 40, 0.5, 5, stale)
    assert g3["rule_violations"] == 0 and g3["range_violations"] > 0.2 * g3["gate_rows_checked"], {q: v for q, v in g3.items() if q != "w_replayed"}


def _span(split, trace, i):
    b, e = split[int(trace["worker"][i])]
    return b, e - b
