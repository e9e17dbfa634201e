"""-m gpu: the in-library collective path with world = 2 EXECUTED on real kernels -- two processes, ONE device, each
with its half of the rows, the collectives served by the test-only shim tests/rccl_stub (RCCL refuses two ranks on one
device and the build environment reaches exactly one GPU; selected by DSGD_RCCL_LIB in the rank processes only).

What replaces `Future.sequence` + `Vec.mean` over the workers (core/Master.scala:190-197; the reference deploys 4
slaves, kube/dsgd.yaml:95) is checked at world = 2 where none of it is an identity any more:
  * replicas bit-identical after every whole-range step and every index-list step;
  * the weights equal the ORACLE's step with K = hosted workers x world (mean over all workers, Master.scala:194) under
    the derived per-coordinate bound, from identical weights, and so does a single process hosting all the workers;
  * one column ranking and one dimSparsity on all ranks, equal to the single process' (counts all-reduced);
  * evaluation tallies = the sum over the shards;
  * the asynchronous mode's exchange (dsgd_async_set_exchange) with two replicas equals the in-process simulation of
    two replicas exchanging their summed updates (host.HostAsyncExchange's scheme) over the oracle;
  * K8 (dsgd_dense_comm_init): two ranks = one process stepping over the union of their mini-batches.
No scaling curve is measured here: both ranks share one device."""

import os
import subprocess
import sys

import numpy as np
import pytest

import dsgd_amd
import waivers
from conftest import has_gpu
from oracle import bounds as orb
from oracle import dense_ref
from oracle import oracle as orc
from test_rccl_stub import seam_env
from world2_common import CFG, dense_problem, local_lists, shard_of

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="no gfx950 device")]

HERE = os.path.dirname(os.path.abspath(__file__))
WORLD = 2


@pytest.fixture(scope="module", params=["streaming_kernels", "column_lists"])
def ranks(request, tmp_path_factory):
    """Two rank processes, twice: their row ranges (24,000 .. 50,000 train rows each) on the streaming kernels -- the path
    bench.py --gpus N times in weak mode -- and on the column lists (csrc/dsgd_tcol.hpp) -- the product's choice for ranges
    of that size, i.e. `bench.py --gpus 8 --scaling strong --rows-total 804414`.  The collective sits between the exact
    column sums and the update in both."""
    wd = str(tmp_path_factory.mktemp("world2"))
    env = seam_env()   # the seam build of the library (DSGD_LIB_PATH) over the shim (DSGD_RCCL_LIB)
    env["DSGD_TCOL"] = "1" if request.param == "column_lists" else "0"
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "world2_worker.py"), str(r), str(WORLD), wd], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(WORLD)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=900)[0])
    finally:
        for p in procs:   # (exactly the processes started here)
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-4000:])
    outs = [dict(np.load(os.path.join(wd, "out_%d.npz" % r))) for r in range(WORLD)]
    for o_ in outs:   # the mode took effect in the rank processes
        assert str(o_["range_kernel"]) == ("dsgd_tc_grad_kernel" if request.param == "column_lists" else "dsgd_wseg_kernel<true>"), o_["range_kernel"]
        o_["_tcol"] = env["DSGD_TCOL"]
    return outs


@pytest.fixture(scope="module")
def problem():
    data = dsgd_amd.synth.generate(CFG["n_rows"], seed=CFG["seed"])
    o = orc.Oracle(data.dim, data.row_ptr, data.col, data.val, data.label, CFG["lam"])
    o.set_dim_sparsity(o.dim_sparsity(CFG["n_train"]))
    shards = [shard_of(data, CFG["n_train"], r, WORLD) for r in range(WORLD)]
    return data, o, shards


def global_steps(shards):
    """The steps of the rank processes as (kind, [per worker: global rows or range], lr) in worker order
    rank 0's hosted workers, then rank 1's -- the order the all-reduce adds in."""
    steps = []
    for ranges in ([(0, 1.0)], [(0, 1.0)], [(0, 1 / 3.0), (1 / 3.0, 1.0)]):
        glob = []
        for sh in shards:
            for lo, hi in ranges:
                a = sh.train_lo + (sh.n_train // 3 if lo else 0)
                b = sh.train_lo + (sh.n_train // 3 if hi != 1.0 else sh.n_train)
                glob.append((a, b))
        steps.append(("range", glob, CFG["lr_range"] * len(ranges) * WORLD))
    for step, (k, b) in enumerate(CFG["list_steps"]):
        glob = []
        for r, sh in enumerate(shards):
            glob += [(l.astype(np.int64) + sh.train_lo).astype(np.int32) for l in local_lists(r, step, k, b, sh.n_train)]
        steps.append(("list", glob, 0.5 * 100 / b))
    return steps


def test_one_ranking_one_dim_sparsity(ranks, problem):
    data, o, shards = problem
    np.testing.assert_array_equal(ranks[0]["ranks"], ranks[1]["ranks"])
    np.testing.assert_array_equal(ranks[0]["ds"], ranks[1]["ds"])
    np.testing.assert_array_equal(ranks[0]["ds"], o.ds.astype(np.float32))   # Main.scala:54-65 over the WHOLE train set
    assert sorted(ranks[0]["ranks"].tolist()) == list(range(data.dim + 1))    # a permutation
    with dsgd_amd.Engine(data.dim, CFG["lam"]) as eng:                        # the single process holding every row
        eng.load_csr(data.row_ptr, data.col, data.val, data.label)
        np.testing.assert_array_equal(eng.build_dim_sparsity(CFG["n_train"]), ranks[0]["ds"])
        np.testing.assert_array_equal(eng.column_ranks(), ranks[0]["ranks"])


def test_sync_steps_replicas_identical_and_equal_to_the_oracle_with_k_workers_times_world(ranks, problem):
    data, o, shards = problem
    steps = global_steps(shards)
    assert ranks[0]["w_hist"].shape == (len(steps), data.dim + 1)
    np.testing.assert_array_equal(ranks[0]["w_hist"], ranks[1]["w_hist"])     # bit-identical replicas, every step
    assert np.abs(ranks[0]["w_hist"][-1]).max() > 0
    with dsgd_amd.Engine(data.dim, CFG["lam"]) as single:
        single.load_csr(data.row_ptr, data.col, data.val, data.label)
        single.build_dim_sparsity(CFG["n_train"])
        w_prev = np.zeros(data.dim + 1, dtype=np.float32)
        for i, (kind, workers, lr) in enumerate(steps):
            w_ranks = ranks[0]["w_hist"][i].astype(np.float64)
            w0 = w_prev.astype(np.float64)
            w_ref = w0.copy()
            lists = [np.arange(a, b, dtype=np.int32) for a, b in workers] if kind == "range" else workers
            o.sync_step(w_ref, lists, lr)                                    # K = len(workers) = hosted workers x world
            shift = int(min(ranks[0]["shifts"][i], ranks[1]["shifts"][i]))
            if kind == "range":
                tol_v, n_near, near_part = orb.step_bound(o, w0, w_ref, workers, lr, shift, parts=True)
            else:
                tol_v, n_near, near_part = orb.list_bound(o, w0, w_ref, lists, lr, shift, parts=True)
            ratio, j = orb.worst_ratio(w_ranks, w_ref, tol_v)
            assert ratio <= 1.0, "step %d (%s): two ranks vs oracle: coordinate %d at %.3g x its bound" % (i, kind, j, ratio)
            n_act = int(ranks[0]["stats"][i][1] + ranks[1]["stats"][i][1])
            assert abs(n_act - o.last_stats["n_active"]) <= n_near
            assert int(ranks[0]["stats"][i][0] + ranks[1]["stats"][i][0]) == sum(len(l) for l in lists)
            tight = n_act == o.last_stats["n_active"] and orb.worst_ratio(w_ranks, w_ref, tol_v - near_part)[0] <= 1.0
            waivers.tight("world2:gates_as_the_oracle", tight, n_near > 0, "%d rows near the gate" % n_near)
            # the single process hosting ALL the workers, from the same weights: the same bound at its own shift
            single.set_weights(w_prev)
            st = single.sync_step_ranges(workers, lr) if kind == "range" else single.sync_step(workers, lr)
            shift1 = single.tuning_info()["fix_shift"]
            tol1, _ = (orb.step_bound(o, w0, w_ref, workers, lr, shift1) if kind == "range" else orb.list_bound(o, w0, w_ref, lists, lr, shift1))
            w_single = single.get_weights().astype(np.float64)
            assert orb.worst_ratio(w_single, w_ref, tol1)[0] <= 1.0
            assert (np.abs(w_single - w_ranks) <= tol_v + tol1).all()
            assert abs(st["n_active"] - n_act) <= 2 * n_near
            w_prev = ranks[0]["w_hist"][i]


def test_eval_tallies_are_the_sum_over_the_shards(ranks, problem):
    data, o, shards = problem
    np.testing.assert_array_equal(ranks[0]["eval"], ranks[1]["eval"])         # every rank reports the global numbers
    w = ranks[0]["w_hist"][-1].astype(np.float64)
    ev = ranks[0]["eval"]
    for off, (lo, hi) in ((0, (0, CFG["n_train"])), (5, (CFG["n_train"], data.n_rows))):
        loss_ref, acc_ref, counts_ref, mam = o.loss_acc(w, lo, hi)
        counts = [int(x) for x in ev[off + 2:off + 5]]
        assert sum(counts) == hi - lo
        waivers.tight("world2:tallies", counts == counts_ref and abs(ev[off] - loss_ref) <= 1e-6 and ev[off + 1] == acc_ref,
                      mam < 1e-5, "margin %.2g" % mam)


def test_async_exchange_of_two_replicas_equals_the_simulation(ranks, problem):
    """Replica r runs ONE deterministic worker on its own rows; every E local updates the replicas all-reduce what each
    subtracted since the last exchange and subtract their PEERS' part (core/Slave.scala:103-105,177-185, batched)."""
    from test_gpu_parity import GATE_EPS, hog_rows, tol

    data, o, shards = problem
    los = []
    for sh in shards:
        lo = orc.Oracle(data.dim, sh.csr.row_ptr, sh.csr.col, sh.csr.val, sh.csr.label, CFG["lam"])
        lo.set_dim_sparsity(o.ds)
        los.append(lo)
    w = [np.zeros(data.dim + 1) for _ in shards]
    w_prev = [x.copy() for x in w]
    it = [0, 0]
    exposed = False
    e_every, n_upd, batch = CFG["exch_every"], CFG["async_updates"], CFG["async_batch"]
    for rnd in range(n_upd // e_every):
        for r, sh in enumerate(shards):
            b0, e0 = CFG["async_range"]
            e0 = min(e0, sh.n_train)
            for _ in range(e_every):
                rows = hog_rows(CFG["async_seed"] + r, 0, it[r], b0, e0 - b0, batch, False)
                los[r].async_step(w[r], rows, 0.5)
                exposed = exposed or los[r].last_stats["min_abs_margin"] < GATE_EPS
                it[r] += 1
        d = [w_prev[r] - w[r] for r in range(WORLD)]
        for r in range(WORLD):
            w[r] = w[r] - sum(d[q] for q in range(WORLD) if q != r)
            w_prev[r] = w[r].copy()
    assert np.abs(w[0] - w[1]).max() <= 1e-12          # after an exchange the simulated replicas agree
    np.testing.assert_array_equal(ranks[0]["w_async"], ranks[1]["w_async"])   # ... and the real ones bit for bit
    assert np.abs(ranks[0]["w_async"]).max() > 0
    err = np.abs(ranks[0]["w_async"].astype(np.float64) - w[0]).max()
    waivers.tight("world2:async_exchange", err <= 4 * tol(w[0]), exposed, "a replayed row within 1e-5 of the gate, err %.3g" % err)
    # dsgd_async_start returned at once (the rounds are enqueued by a helper thread) and stayed pollable
    for r in range(WORLD):
        assert ranks[r]["async_meta"][0] < 0.5, ranks[r]["async_meta"]


def test_dense_two_ranks_equal_one_process_over_the_union(ranks):
    X, y, n_steps, bsz = dense_problem()
    np.testing.assert_array_equal(ranks[0]["w_dense"], ranks[1]["w_dense"])
    w_ref = np.zeros(X.shape[1])
    for s in range(n_steps):   # global mini-batch s = rank 0's rows then rank 1's
        w_ref, _, _ = dense_ref.step(X[s * 2 * bsz:(s + 1) * 2 * bsz], y[s * 2 * bsz:(s + 1) * 2 * bsz], w_ref, 0.5)
    scale = max(1.0, float(np.abs(w_ref).max()))
    assert np.abs(ranks[0]["w_dense"] - w_ref).max() <= 2e-6 * scale
    with dsgd_amd.DenseLogistic(X.shape[1]) as dl:
        dl.load(X, y)
        for s in range(n_steps):
            dl.step(s * 2 * bsz, (s + 1) * 2 * bsz, 0.5)
        dl.synchronize()
        assert np.abs(dl.get_weights() - ranks[0]["w_dense"]).max() <= 2e-6 * scale


def test_one_thread_driving_both_contexts_equals_the_two_processes(ranks, tmp_path):
    """dsgd_comm_init_all / dsgd_build_dim_sparsity_devices / dsgd_sync_step_devices / dsgd_sync_step_ranges_devices /
    dsgd_loss_acc_devices: ONE process and ONE thread drive two contexts (the reference's dev role, Main.scala:144-158);
    every collective goes through ncclGroupStart / ncclGroupEnd.  Same kernels, same sums, same order: the column
    ranking, every step's weights on both replicas, the summed statistics and the evaluation are those of the two rank
    processes BIT FOR BIT -- which the tests above hold to the oracle with K = workers x world."""
    env = seam_env()
    env["DSGD_TCOL"] = ranks[0]["_tcol"]   # (the same kernels as the rank processes of this round)
    proc = subprocess.run([sys.executable, os.path.join(HERE, "devices_worker.py"), str(tmp_path)], env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-4000:]
    d = dict(np.load(os.path.join(str(tmp_path), "devices.npz")))
    np.testing.assert_array_equal(d["ranks"][0], ranks[0]["ranks"])
    np.testing.assert_array_equal(d["ranks"][1], ranks[0]["ranks"])
    n_steps = ranks[0]["w_hist"].shape[0]
    assert d["w_hist"].shape[:2] == (n_steps, 2)
    for i in range(n_steps):
        np.testing.assert_array_equal(d["w_hist"][i][0], d["w_hist"][i][1])            # the replicas agree ...
        np.testing.assert_array_equal(d["w_hist"][i][0], ranks[0]["w_hist"][i])        # ... with the two processes
        assert list(d["stats"][i]) == [int(ranks[0]["stats"][i][0] + ranks[1]["stats"][i][0]),
                                       int(ranks[0]["stats"][i][1] + ranks[1]["stats"][i][1])]
    np.testing.assert_array_equal(d["eval"], ranks[0]["eval"])
    assert d["dup_refused"][0] == 1


@pytest.mark.parametrize("mode", ["strong", "weak"])
def test_bench_parity_gate_with_two_ranks(mode, tmp_path):
    """bench.py --gpus 2: the N > 1 parity gate EXECUTED (two ranks on the one device through the seam build + shim):
    every rank steps over the first --gate-rows rows of its shard through the in-library all-reduce, rank 0 hosts both
    shards in the oracle as world x workers workers and holds the update to the derived bound and the stated 1e-5
    tolerance; then the timed steps and the replica digest.  Strong mode: ONE data set split by SplitStrategy.vanilla."""
    import json

    env = seam_env({k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")})
    env["DSGD_BENCH_ONE_DEVICE"] = "1"
    shape = ["--scaling", "strong", "--rows-total", "60000"] if mode == "strong" else ["--rows", "30000"]
    gate_rows = 8000 if mode == "strong" else 12000   # (below / above the row count from which ranges take the streaming kernels)
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--workers", "2", "--gate-rows", str(gate_rows),
           "--steps", "3", "--warmup", "1", "--repeats", "2", "--clock-ramp", "0.05", "--no-cpu-baseline",
           "--detail", str(tmp_path / "detail.json")] + shape
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-4000:]
    last = proc.stdout.splitlines()[-1]
    compact = json.loads(last)                       # the LAST stdout line is the compact line of record ...
    assert len(last) < 8192 and compact["n_gpus"] == 2 and compact["scaling"] == mode and compact["replicas_bit_identical"] is True
    line = json.load(open(tmp_path / "detail.json"))   # ... and the full object is in the detail file
    assert abs(compact["value"] - line["value"]) <= 1e-5 * line["value"] and compact["parity_gate_max_rel_err"] <= 1e-5
    assert line["n_gpus"] == 2 and line["scaling"] == mode and line["replicas_bit_identical"] is True
    g = line["parity_gate"]
    assert g["world"] == 2 and g["workers_total"] == 4 and g["rows_per_rank"] == gate_rows and len(g["steps"]) == 2
    assert g["replicas_bit_identical_after_gate"] is True
    for st in g["steps"]:
        assert st["worst_err_over_bound"] <= 1.0 and st["max_rel_err"] <= 1e-5
        assert abs(st["n_active_engine"] - st["n_active_oracle"]) <= st["rows_near_gate"]
    if mode == "strong":
        assert line["config"]["train_rows_job"] == 48000 and line["config"]["train_rows_per_gpu"] == 24000
        assert abs(line["value"] - 48000 / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    else:
        assert abs(line["value"] - 2 * 24000 / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    assert len(line["repeats"]["ms_per_step"]) == 2
