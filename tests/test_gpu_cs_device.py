"""-m gpu: what round 5 added to the column-slice path (csrc/dsgd_cs.hpp).

  * the layout of a plan's slices is built ON THE DEVICE (dsgd_cs_layout_kernel): the same plan laid out by the host
    builder of rounds 1-4 (DSGD_CS_HOST_LAYOUT=1) ends on the SAME BITS -- both builders emit the same slots in the same
    order, so the partial dots, the gates and the exact sums agree bit for bit; ragged and long rows included;
  * per-request steps (dsgd_sync_step with host index lists -- what core/Slave.scala:142-157 + core/Master.scala:184-197
    reach) run as ONE launch of dsgd_cs_request_kernel: held to the oracle like every index-list kernel, bit-identical
    to the same lists as a one-step plan, with a loud, correct fallback when a step does not fit the one-step layout;
  * the record of a plan's run (dsgd_plan_record): the engine's gate decisions are the oracle's on a short run, and the
    forced-decision replay (oracle/sync_replay.py) lands on the engine's weights;
  * plans come and go without device synchronisation: create / run / destroy cycles reuse cached blocks and stay correct
    when the next plan is created WHILE the previous one runs."""

import os
import subprocess
import sys

import numpy as np
import pytest

import dsgd_amd
import waivers
from conftest import has_gpu
from oracle import bounds as orb
from oracle import oracle as orc
from oracle import ref_dict as rd
from oracle import sync_replay as sr

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="no gfx950 device")]

LAM = 1e-5
CS, REQ = "dsgd_cs_step_kernel", "dsgd_cs_request_kernel"
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module", autouse=True)
def _request_kernel_on():
    """Per-request steps through the column-slice kernel are OPT-IN (DSGD_CS_REQ=1: one workgroup per slice lays the step
    out before it can run it, and that set-up is latency-bound -- 126 us per 3 x 100 request against 40 us through the
    row-parallel kernels, profiles/r05_probe.json); this module tests the path, so it switches it on."""
    old = os.environ.get("DSGD_CS_REQ")
    os.environ["DSGD_CS_REQ"] = "1"
    yield
    if old is None:
        os.environ.pop("DSGD_CS_REQ", None)
    else:
        os.environ["DSGD_CS_REQ"] = old


def make_pair(data, n_train, with_oracle=True):
    o = None
    if with_oracle:
        o = orc.Oracle(data.dim, data.row_ptr, data.col, data.val, data.label, LAM)
        o.set_dim_sparsity(o.dim_sparsity(n_train))
    eng = dsgd_amd.Engine(data.dim, LAM)
    eng.load_csr(data.row_ptr, data.col, data.val, data.label)
    eng.build_dim_sparsity(n_train)
    return o, eng


def batches(rng, n_train, k, b, steps):
    split = rd.split_vanilla(n_train, k)
    return [[rng.permutation(np.asarray(r))[:b].astype(np.int32) for r in split] for _ in range(steps)]


def nonzero_weights(dim, rng, n=6000):
    w0 = np.zeros(dim + 1, dtype=np.float32)
    hot = rng.choice(np.arange(1, dim + 1), size=n, replace=False)
    w0[hot] = rng.normal(scale=0.05, size=n).astype(np.float32)
    return w0


def ragged(seed=23, n=6000):
    base = dsgd_amd.synth.generate(n, seed=seed)
    rng = np.random.default_rng(seed)
    row_ptr, col, val = [0], [], []
    for i in range(base.n_rows):
        b, e = int(base.row_ptr[i]), int(base.row_ptr[i + 1])
        kind = rng.integers(0, 10)
        if kind == 0:
            pass
        elif kind == 1:
            col.append(base.col[b]); val.append(np.float32(1.0))
        elif kind == 2:
            keys = np.sort(rng.choice(np.arange(1, base.dim + 1), size=1200, replace=False))
            v = np.abs(rng.normal(size=1200)).astype(np.float32) + 0.1
            v /= np.sqrt((v * v).sum())
            col.extend(keys.tolist()); val.extend(v.tolist())
        else:
            col.extend(base.col[b:e].tolist()); val.extend(base.val[b:e].tolist())
        row_ptr.append(len(col))
    return dsgd_amd.synth.Csr(base.dim, np.asarray(row_ptr, np.int64), np.asarray(col, np.int32), np.asarray(val, np.float32),
                              base.label.copy())


@pytest.mark.parametrize("shape", ["rcv1", "ragged"])
def test_device_layout_is_the_host_layout_bit_for_bit(monkeypatch, shape):
    data = dsgd_amd.synth.generate(40000, seed=5) if shape == "rcv1" else ragged()
    n_train = int(data.n_rows * 0.8)
    rng = np.random.default_rng(8)
    w0 = nonzero_weights(data.dim, rng, 12000)
    cfgs = [(3, 100, 9), (4, 200, 4), (1, 1, 3), (8, 100, 3), (2, 37, 5)] if shape == "rcv1" else [(3, 100, 4), (1, 64, 3)]
    plans = [(k, b, batches(rng, n_train, k, b, n)) for k, b, n in cfgs]
    out = {}
    for mode in ("device", "host"):
        monkeypatch.setenv("DSGD_CS_HOST_LAYOUT", "1" if mode == "host" else "0")
        _, eng = make_pair(data, n_train, with_oracle=False)
        res = []
        with eng:
            for k, b, steps in plans:
                eng.set_weights(w0)
                plan = eng.plan(steps)
                info = plan.info()
                assert info["kind"] == "column_slices" and info["device_built"] == (mode == "device"), info   # laid out at creation
                eng.plan_run(plan, 0, len(steps), min(0.5 * 100 / b, 1.0))
                st = eng.synchronize()
                assert eng.grad_kernel_name() == CS and st["n_samples"] == sum(len(a) for s_ in steps for a in s_)
                res.append((eng.get_weights(), st["n_active"], {kk: info[kk] for kk in ("slices", "slot_stride", "row_stride", "col_list_stride", "slots_per_lane")}))
                plan.destroy()
        out[mode] = res
    for (wd, ad, infd), (wh, ah, infh), (k, b, _) in zip(out["device"], out["host"], plans):
        assert infd == infh, (k, b, infd, infh)           # the same strides: the same maxima were found
        assert ad == ah and np.array_equal(wd, wh), "%d x %d: the device-built layout ends on other bits than the host-built one" % (k, b)


def request_step(o, eng, lists, lr, family, kernel=REQ):
    w0 = eng.get_weights().astype(np.float64)
    w_ref = w0.copy()
    st = eng.sync_step(lists, lr)
    assert eng.grad_kernel_name() == kernel, eng.grad_kernel_name()
    shift = eng.tuning_info()["fix_shift"]
    o.sync_step(w_ref, lists, lr)
    tol_v, n_near, near_part = orb.list_bound(o, w0, w_ref, lists, lr, shift, parts=True)
    assert st["n_samples"] == sum(len(a) for a in lists)
    assert abs(st["n_active"] - o.last_stats["n_active"]) <= n_near, (st, o.last_stats, n_near)
    w = eng.get_weights()
    ratio, j = orb.worst_ratio(w, w_ref, tol_v)
    assert ratio <= 1.0, "coordinate %d: error %.3g x its derived bound (shift %d, %d rows near the gate)" % (j, ratio, shift, n_near)
    assert np.abs(w - w_ref).max() <= 1e-5 * max(1.0, np.abs(w_ref).max()) or n_near > 0
    tight = st["n_active"] == o.last_stats["n_active"] and orb.worst_ratio(w, w_ref, tol_v - near_part)[0] <= 1.0
    waivers.tight(family + ":gates_as_the_oracle", tight, n_near > 0, "%d rows near the gate" % n_near)
    return st


@pytest.fixture(scope="module")
def mid():
    data = dsgd_amd.synth.generate(60000, seed=3)
    n_train = 48000
    o, eng = make_pair(data, n_train)
    yield data, n_train, o, eng
    eng.close()


@pytest.mark.parametrize("k,b", [(3, 100), (4, 200), (1, 100), (1, 1), (2, 7), (8, 100), (1, 800), (5, 64)])
def test_per_request_steps_run_on_column_slices(mid, k, b):
    data, n_train, o, eng = mid
    rng = np.random.default_rng(13 * b + k)
    eng.set_weights(nonzero_weights(data.dim, rng))
    lr = min(0.5 * 100 / b, 1.0)
    steps = batches(rng, n_train, k, b, 4)
    for lists in steps:
        request_step(o, eng, lists, lr, "cs_request")
    # the same lists as one-step plans: the same slices, slots, sums and order -- the same bits
    w_req = eng.get_weights()
    eng.set_weights(nonzero_weights(data.dim, np.random.default_rng(5)))
    w_start = eng.get_weights()
    acts = []
    for lists in steps:
        acts.append(eng.sync_step(lists, lr)["n_active"])
    w_a = eng.get_weights()
    eng.set_weights(w_start)
    plan = eng.plan(steps)
    eng.synchronize()                                 # (the counters of the requests above are collected and cleared)
    eng.plan_run(plan, 0, len(steps), lr)
    st = eng.synchronize()
    plan.destroy()
    assert eng.grad_kernel_name() == CS and st["n_active"] == sum(acts)
    assert np.array_equal(eng.get_weights(), w_a) and not np.array_equal(w_a, w_req)


def test_a_request_that_does_not_fit_falls_back_and_the_next_one_runs():
    """Eighty 1,200-entry rows in one step touch more than the 4,096 columns a slice of the one-step layout lists: the
    request kernel gives the step up before anything is published (nothing applied), the row-parallel kernels take it, the
    abort word is cleared -- the next request runs on column slices again.  Requests beyond 1,024 rows or 8 workers never
    try."""
    data = ragged()
    n_train = 5000
    o, eng = make_pair(data, n_train)
    rng = np.random.default_rng(23)
    with eng:
        eng.set_weights(nonzero_weights(data.dim, rng, 20000))
        request_step(o, eng, batches(rng, n_train, 3, 100, 1)[0], 0.5, "cs_request_ragged")
        big = batches(rng, n_train, 2, 400, 1)[0]
        request_step(o, eng, big, 0.125, "cs_request_ragged", kernel="dsgd_mb_grad_kernel")
        request_step(o, eng, batches(rng, n_train, 3, 100, 1)[0], 0.5, "cs_request_ragged")
        request_step(o, eng, batches(rng, n_train, 1, 64, 1)[0], 0.5, "cs_request_ragged")
        request_step(o, eng, batches(rng, n_train, 2, 600, 1)[0], 0.1, "cs_request_ragged", kernel="dsgd_mb_grad_kernel")   # 1,200 rows
        request_step(o, eng, batches(rng, n_train, 9, 50, 1)[0], 1.0, "cs_request_ragged", kernel="dsgd_mb_grad_kernel")   # nine workers
        with pytest.raises(IndexError):
            eng.sync_step([np.array([1, 2, n_train + 10 ** 6], dtype=np.int32)], 0.5)
        with pytest.raises(ValueError):
            eng.sync_step([np.array([1, 2], dtype=np.int32), np.zeros(0, dtype=np.int32)], 0.5)
        request_step(o, eng, batches(rng, n_train, 3, 100, 1)[0], 0.5, "cs_request_ragged")


def test_request_path_can_be_switched_off(monkeypatch):
    data = dsgd_amd.synth.generate(20000, seed=9)
    n_train = 16000
    monkeypatch.setenv("DSGD_CS_REQ", "0")
    o, eng = make_pair(data, n_train)
    rng = np.random.default_rng(9)
    with eng:
        eng.set_weights(nonzero_weights(data.dim, rng))
        request_step(o, eng, batches(rng, n_train, 3, 100, 1)[0], 0.5, "cs_request_off", kernel="dsgd_mb_grad_kernel")


def test_the_record_of_a_run_and_its_forced_replay(mid):
    """dsgd_plan_record: 40 steps of 3 x 100 from non-zero weights.  The recorded decisions are the oracle's own wherever
    the oracle's margin is clear; the forced replay lands on the engine's weights (accounting), its scalar on the
    recorded one; a second run of the same plan rewrites the record; a plan that does not run on column slices refuses."""
    data, n_train, o, eng = mid
    rng = np.random.default_rng(99)
    w0 = nonzero_weights(data.dim, rng)
    steps = batches(rng, n_train, 3, 100, 40)
    plan = eng.plan(steps)
    plan.record(True)
    assert plan.info()["record_words"] == 10            # 300 rows per step
    eng.set_weights(w0)
    eng.plan_run(plan, 0, 25, 0.5)
    eng.plan_run(plan, 25, 40, 0.5)                       # two launches: the record is indexed by the plan's steps
    eng.synchronize()
    masks, s_used = plan.read_record()
    assert masks.shape == (40, 320) and not masks[:, 300:].any()
    w_eng = eng.get_weights()
    w = w0.astype(np.float64)
    st = sr.replay(o, w, steps, 0.5, masks, s_used)
    v = sr.verdict(st, w_eng, w)
    assert v["accounting_agrees"] and v["account_err_over_tol"] < 0.2, v
    assert v["s_agrees"] and v["divergent_rows_all_near_gate"], v
    assert st["decisions"] == 40 * 300 and v["differing_decisions"] <= 3, v
    # negative control through the same record: one decision flipped on a clear margin is caught
    w1 = w0.astype(np.float64)
    for lists in steps[:7]:
        o.sync_step(w1, lists, 0.5)
    rows = np.concatenate(steps[7])
    marg = np.array([abs(o.row_dot(int(r), w1)) for r in rows])
    bad = masks.copy()
    bad[7, int(marg.argmax())] ^= True
    wb = w0.astype(np.float64)
    vb = sr.verdict(sr.replay(o, wb, steps, 0.5, bad, s_used), w_eng, wb)
    assert not vb["divergent_rows_all_near_gate"] and not vb["accounting_agrees"]
    # partial read, and the record follows a second run
    m2, s2 = plan.read_record(10, 12)
    assert np.array_equal(m2, masks[10:12]) and np.array_equal(s2, s_used[10:12])
    eng.set_weights(np.zeros(data.dim + 1, dtype=np.float32))
    eng.plan_run(plan, 0, 1, 0.5)
    eng.synchronize()
    m3, s3 = plan.read_record(0, 1)
    assert m3[0, :300].all() and s3[0] == 0.0            # from w = 0 every margin is 0: every row is active, s = 0
    plan.record(False)
    with pytest.raises(dsgd_amd.DsgdError):
        plan.read_record(0, 1)
    plan.destroy()
    big = eng.plan(batches(rng, n_train, 1, 3000, 1))
    big.record(True)
    eng.plan_run(big, 0, 1, 0.01)
    eng.synchronize()
    with pytest.raises(dsgd_amd.DsgdError):
        big.read_record(0, 1)
    big.destroy()


def test_plans_come_and_go_without_synchronising(mid):
    """One epoch of Master.fit = one plan.  Six 'epochs': the next plan is created while the previous one still runs, the
    previous one destroyed right behind its run (blocks handed back, reused); the final weights are the oracle's."""
    data, n_train, o, eng = mid
    rng = np.random.default_rng(321)
    eng.set_weights(np.zeros(data.dim + 1, dtype=np.float32))
    w_ref = np.zeros(data.dim + 1)
    epochs = [batches(rng, n_train, 3, 100, 30) for _ in range(6)]
    exposed = 0
    nxt = eng.plan(epochs[0])
    for e, steps in enumerate(epochs):
        cur = nxt
        eng.plan_run(cur, 0, len(steps), 0.5)             # enqueued; nothing waits
        if e + 1 < len(epochs):
            nxt = eng.plan(epochs[e + 1])                 # laid out on the build stream beside the run
            assert nxt.info()["kind"] == "column_slices"
        cur.destroy()                                      # behind the run, without a device synchronisation
        for lists in steps:
            o.sync_step(w_ref, lists, 0.5)
            exposed += o.last_stats["min_abs_margin"] < 1e-5
    st = eng.synchronize()
    assert st["n_samples"] == 6 * 30 * 300
    err = np.abs(eng.get_weights().astype(np.float64) - w_ref).max()
    waivers.tight("column_slices:epoch", err <= 1e-5 * max(1.0, np.abs(w_ref).max()), exposed > 0,
                  "%d steps with a row within 1e-5 of the gate, err %.3g" % (exposed, err))


def test_the_exchange_gives_up_loudly_and_leaves_the_weights():
    """A slice that stops publishing (test build of the library: dsgd_test_cs_skip_publish) must end the launch with
    DSGD_ESTATE -- not hang -- with the weights bit-identical to before, the abort word cleared, and the next launch
    succeeding; per-request steps likewise."""
    from test_rccl_stub import seam_env

    proc = subprocess.run([sys.executable, os.path.join(HERE, "cs_abort_worker.py")], env=seam_env(), stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True, timeout=600)
    assert proc.returncode == 0 and "CS_ABORT_OK" in proc.stdout, proc.stdout[-3000:]
