"""CPU: bench.py's host logic exercised without a GPU -- the partition of a strong-scaling run, and a dry run of the
legs that wrap the engine (sweep with its parity-through-the-timed-plan rule, the reference shapes, time to target)
over a stand-in engine that answers from the fp64 oracle.  What this pins is the PLUMBING (keys, the rule that the
checked kernel is the timed kernel, the stated 1e-5 tolerance beside the derived bound, the epoch lists); numbers come
from the GPU."""

import argparse
import json
import os
import sys

import numpy as np
import pytest

import dsgd_amd
from conftest import ROOT
from oracle import oracle as orc

sys.path.insert(0, ROOT)
import bench  # noqa: E402


class FakePlan:
    def __init__(self, steps):
        self.steps, self.n_steps = steps, len(steps)
        self.rec, self.masks, self.s = False, {}, {}

    def destroy(self):
        pass

    def record(self, on=True):
        self.rec = bool(on)

    def read_record(self, a=0, b=None):
        b = self.n_steps if b is None else b
        width = 32 * -(-max(sum(len(x) for x in st) for st in self.steps) // 32)
        m = np.zeros((b - a, width), dtype=bool)
        for i in range(a, b):
            m[i - a, :len(self.masks[i])] = self.masks[i]
        return m, np.asarray([self.s[i] for i in range(a, b)], dtype=np.float32)


class FakeEngine:
    """The surface of dsgd_amd.Engine that bench.py touches, answered by the oracle (fp64; weights cross as fp32)."""

    kernel_for_plans = "dsgd_vt_grad_kernel"

    def __init__(self, dim, lam, device=0):
        self.dim, self.dp, self.lam = dim, dim + 1, lam
        self.w = np.zeros(self.dp)
        self.act, self.kern, self.n_rows = 0, "", 0

    def __enter__(self):
        return self

    def __exit__(self, *a):
        pass

    def close(self):
        pass

    def load_csr(self, row_ptr, col, val, label):
        self.o = orc.Oracle(self.dim, row_ptr, col, val, label, self.lam)
        self.n_rows = len(row_ptr) - 1
        self.row_ptr = row_ptr

    def build_dim_sparsity(self, n_train):
        self.o.set_dim_sparsity(self.o.dim_sparsity(n_train))

    def set_weights(self, w):
        self.w = np.asarray(w, dtype=np.float64).copy()

    def get_weights(self):
        return self.w.astype(np.float32)

    def sync_step_ranges(self, ranges, lr, asynchronous=False):
        self.o.sync_step(self.w, [np.arange(a, b, dtype=np.int32) for a, b in ranges], lr)
        self.kern = "dsgd_wseg_kernel<true>"
        n = self.o.last_stats["n_active"]
        self.act += n
        return None if asynchronous else {"n_samples": sum(b - a for a, b in ranges), "n_active": n}

    def plan(self, steps):
        return FakePlan(steps)

    def sync_step(self, lists, lr):
        self.o.sync_step(self.w, lists, lr)
        self.kern = "dsgd_mb_grad_kernel"
        return {"n_samples": sum(len(a) for a in lists), "n_active": self.o.last_stats["n_active"]}

    def plan_flat(self, idx, offsets, n_steps, k):
        return FakePlan([[np.asarray(idx[offsets[s * k + j]:offsets[s * k + j + 1]], dtype=np.int32) for j in range(k)] for s in range(n_steps)])

    def plan_run(self, plan, a, b, lr):
        for i, s in enumerate(plan.steps[a:b], start=a):
            if plan.rec:   # the decisions the oracle takes on these weights, the scalar it uses
                rows = np.concatenate(s)
                plan.masks[i] = np.array([not (self.o.label[r] * self.o.row_dot(int(r), self.w) < 0.0) for r in rows])
                prod = self.w * self.o.ds
                plan.s[i] = 2.0 * self.lam * float(prod[np.abs(prod) > 1e-20].sum())
            self.o.sync_step(self.w, s, lr)
            self.act += self.o.last_stats["n_active"]
            self.kern = "dsgd_plan_kernel" if len(s) == 1 and len(s[0]) <= 192 else self.kernel_for_plans

    def synchronize(self):
        n, self.act = self.act, 0
        return {"n_samples": 0, "n_active": n}

    def grad_kernel_name(self):
        return self.kern

    def tuning_info(self):
        return {"fix_shift": 21, "hsplit": 18396}

    def prof_enable(self, on=True):
        pass

    def prof_read(self, reset=True):
        return 0.5, 100

    def range_nnz(self, a, b):
        return int(self.row_ptr[b] - self.row_ptr[a]), 0

    def loss_acc(self, lo, hi):
        loss, acc, counts, _ = self.o.loss_acc(self.w, lo, hi)
        return loss, acc, counts


@pytest.fixture
def fake(monkeypatch):
    monkeypatch.setattr(dsgd_amd, "Engine", FakeEngine)
    return FakeEngine


def test_strong_scaling_shards_are_the_reference_split():
    """--scaling strong: ONE data set, 80/20 (Main.scala:52), its train rows split over the ranks by
    SplitStrategy.vanilla (core/ml/SplitStrategy.scala:13-14): the ranks' train shards concatenated are exactly the train
    rows of the data set, in order; likewise the test rows."""
    n_tot = 10007
    whole = dsgd_amd.synth.generate(n_tot, seed=3)
    n_train_tot = int(n_tot * 0.8)
    for world in (1, 2, 3, 8):
        args = argparse.Namespace(scaling="strong", rows_total=n_tot, rows=0, seed=3)
        tr_rows, te_rows, lo_expect = [], [], 0
        for r in range(world):
            data, n_train, row0, n_job = bench.load_shard(dsgd_amd, args, r, world)
            assert n_job == n_train_tot and row0 == lo_expect
            lo_expect += n_train
            tr_rows.append(data.rows(0, n_train))
            te_rows.append(data.rows(n_train, data.n_rows))
        tr, te = bench.concat_csr(tr_rows), bench.concat_csr(te_rows)
        ref_tr, ref_te = whole.rows(0, n_train_tot), whole.rows(n_train_tot, n_tot)
        for a, b in ((tr, ref_tr), (te, ref_te)):
            assert np.array_equal(a.row_ptr, b.row_ptr) and np.array_equal(a.col, b.col)
            assert np.array_equal(a.val, b.val) and np.array_equal(a.label, b.label)
    # fewer groups than ranks (SplitStrategy.vanilla can yield < K groups): refused, not silently dropped
    with pytest.raises(SystemExit):
        bench.load_shard(dsgd_amd, argparse.Namespace(scaling="strong", rows_total=11, rows=0, seed=0), 0, 8)
    # weak: every rank its own rows of the stream
    a = bench.load_shard(dsgd_amd, argparse.Namespace(scaling="weak", rows=1000, rows_total=0, seed=3), 2, 4)
    assert a[1] == 800 and a[2] == 2000 and a[3] == 3200
    assert np.array_equal(a[0].col, dsgd_amd.synth.generate(1000, seed=3, row0=2000).col)


def test_split_vanilla_matches_the_host_mirror():
    from dsgd_amd import host

    for n, k in ((9, 4), (18519, 3), (100, 1), (7, 8), (643531, 256)):
        assert bench.split_vanilla(n, k) == [(r.start, r.stop) for r in host.split_vanilla(n, k)]


def test_sweep_checks_the_kernel_it_times(fake):
    data = dsgd_amd.synth.generate(6000, seed=5)
    n_train = 4800
    eng = FakeEngine(data.dim, bench.LAMBDA)
    eng.load_csr(data.row_ptr, data.col, data.val, data.label)
    eng.build_dim_sparsity(n_train)
    rows = bench.sweep(eng, data, n_train, with_parity=True, configs=((1, 100, 4), (3, 100, 4), (1, 1000, 3)))
    assert [(r["workers"], r["batch"]) for r in rows] == [(1, 100), (3, 100), (1, 1000)]
    for r in rows:
        p = r["parity"]
        assert p["kernel"] == r["kernel"] and p["checked"] == "step 0 of the timed plan"
        assert p["worst_err_over_bound"] <= 1.0 and p["max_rel_err"] <= bench.STATED_TOL == p["stated_tolerance"]
        assert p["n_active_engine"] == p["n_active_oracle"]
    assert rows[0]["kernel"] == "dsgd_plan_kernel" and rows[1]["kernel"] == "dsgd_vt_grad_kernel"

    # an engine that times another kernel than the one its parity step ran is refused
    class Shifty(FakeEngine):
        calls = 0

        def plan_run(self, plan, a, b, lr):
            super().plan_run(plan, a, b, lr)
            Shifty.calls += 1
            if Shifty.calls > 1:
                self.kern = "dsgd_mb_grad_kernel"

    eng2 = Shifty(data.dim, bench.LAMBDA)
    eng2.load_csr(data.row_ptr, data.col, data.val, data.label)
    eng2.build_dim_sparsity(n_train)
    with pytest.raises(SystemExit, match="parity was checked on"):
        bench.sweep(eng2, data, n_train, with_parity=True, configs=((3, 100, 3),))


def test_stated_tolerance_is_asserted_beside_the_derived_bound():
    from oracle import bounds as orb

    w_ref = np.zeros(10)
    w_ref[3] = 2.0
    tol = np.full(10, 1.0)                     # a derived bound so loose that it alone would pass anything
    ok = bench.check_step(None, orb, w_ref + 1e-6, w_ref, w_ref, tol, 0, 5, 5, "t")
    assert ok["max_rel_err"] <= bench.STATED_TOL
    with pytest.raises(SystemExit, match="stated tolerance"):
        bench.check_step(None, orb, w_ref + 1e-4, w_ref, w_ref, tol, 0, 5, 5, "t")
    with pytest.raises(SystemExit, match="derived bound"):
        bench.check_step(None, orb, w_ref + 1e-6, w_ref, w_ref, np.full(10, 1e-9), 0, 5, 5, "t")
    with pytest.raises(SystemExit, match="active rows"):
        bench.check_step(None, orb, w_ref, w_ref, w_ref, tol, 1, 5, 9, "t")


def test_reference_shape_and_time_to_target_dry_run(fake, monkeypatch):
    monkeypatch.setattr(bench, "SWEEP", ((1, 100, 3), (3, 100, 3)))
    r = bench.reference_shape(dsgd_amd, 0, 3000, with_parity=True, repeats=2, steps=2)
    assert r["rows"] == 3000 and r["train_rows"] == 2400 and r["whole_shard"]["repeats"] == 2
    assert r["parity_gate"]["max_rel_err"] <= bench.STATED_TOL and r["roofline"]["bound"] == "hbm"
    assert {(s["workers"], s["batch"]) for s in r["sweep"]} == {(1, 100), (3, 100)}
    t = bench.time_to_target(dsgd_amd, 0, n_rows=3000, oracle_budget_s=3.0, max_epochs_engine=12)
    assert len(t["configs"]) == 5 and [(c["workers"], c["batch"]) for c in t["configs"]][:2] == [(3, 100), (4, 200)]
    assert "MasterSync.fit" in t["through"]
    ref = t["configs"][0]
    # the stand-in engine IS the oracle: at the reference's configuration both reach the target in the same epoch
    assert ref["engine_epochs"] == ref["oracle_epochs"] and ref["engine_epochs"] is not None
    assert t["fastest"] is not None and t["target_test_loss"] == float(np.median(t["oracle_target_curve"]))
    assert all(c["epoch1_max_abs_diff"] is not None and c["epoch1_max_abs_diff"] < 1e-6 for c in t["configs"])   # the stand-in IS the oracle
    # the reference's configuration carries the forced replay of its first epoch: no differing decision (the stand-in IS the
    # oracle), the accounting agrees, and the fields the compact line quotes are there
    fr = ref["forced_replay_epoch1"]
    assert fr["accounting_agrees"] and fr["differing_decisions"] == 0 and ref["first_divergent_step"] is None
    assert ref["divergent_rows_all_near_gate"] is True and ref["forced_replay_account_err_over_tol"] < 0.1   # (the stand-in hands its weights over as fp32)
    assert ref["steps"] >= 8 and ref["batch_loop_us_per_step"] > 0 and all("fit_s" in c for c in t["configs"])
    json.dumps(t), json.dumps(r)   # everything in the line is JSON-serialisable


def test_epochs_to_target_runs_through_the_mirror(fake):
    """bench.epochs_to_target: both sides through host.MasterSync.fit with java.util.Random(0); the `fit` leg carries the
    boundary's cost per batch (plans vs one request per batch) and the forced replay of the whole 10-epoch trajectory."""
    e = bench.epochs_to_target(dsgd_amd, 0)
    assert e["engine_epochs"] == e["oracle_epochs"] and e["max_curve_difference"] < 1e-6 and e["steps_per_epoch"] == 62
    f = e["fit"]
    assert f["steps"] == 620 and f["summary"]["forced_replay_agrees"] and f["forced_replay_10_epochs"]["steps"] == 620
    assert f["batch_loop_us_per_step"] > 0 and f["per_request_us_per_step"] > 0
    json.dumps(e)


def test_final_line_is_compact_and_parses(tmp_path):
    """BENCH_r04 went unrecorded because the one JSON line had grown to 28 KB and the driver keeps an 8 KB tail.  The
    LAST stdout line is now a compact summary (contract keys, roofline, cpu_baseline, one row per leg); the full object
    goes to a file.  Fed with the fattest object a visit ever produced (profiles/r04_bench.json), the line must parse,
    stay far below 8 KB, and keep every key of the driver's contract with the detail's values."""
    import io

    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
    buf = io.StringIO()
    detail = tmp_path / "sub" / "bench_detail.json"
    line = bench.emit(full, str(detail), stream=buf)
    printed = buf.getvalue()
    assert printed.endswith(line + "\n") and printed.count("\n") == 1          # ONE line, the last one
    assert len(line.encode()) < bench.LINE_LIMIT == 8192 and len(line.encode()) < 6144
    c = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in c, key
    assert c["metric"] == full["metric"] and c["n_gpus"] == 1 and c["dtype"] == "f32" and "workload" in c["config"]
    assert abs(c["value"] - full["value"]) <= 1e-5 * full["value"] and abs(c["ms_per_step"] - full["ms_per_step"]) <= 1e-5 * full["ms_per_step"]
    r = c["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5 and "traffic" in r
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms_avg"] * 1e-3) / 1e9) <= 1e-4 * r["achieved"]
    assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["cores"] >= 1 and c["cpu_baseline"]["sample"]
    assert "legs_dropped_from_line" not in c and {"sweep", "reference_shapes", "hogwild", "time_to_target"} <= set(c["legs"])
    assert json.load(open(detail)) == full                                       # nothing is lost: the file has it all
    # a pathological object (a leg blown up a hundredfold) still yields a line under the limit: legs go, contract keys stay
    fat = dict(full, sweep=full["sweep"] * 100)
    for i, s in enumerate(fat["sweep"]):
        fat["sweep"][i] = dict(s, batch=s["batch"] + i)
    buf2 = io.StringIO()
    line2 = bench.emit(fat, None, stream=buf2)
    c2 = json.loads(line2)
    assert len(line2) < bench.LINE_LIMIT and "sweep" in c2["legs_dropped_from_line"] and c2["value"] == c["value"] and "roofline" in c2
