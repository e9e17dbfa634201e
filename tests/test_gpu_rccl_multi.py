"""-m gpu, needs >= 2 visible gfx950 devices (skipped otherwise -- the build environment reaches one): the in-library
collective path over REAL RCCL / xGMI, which the one-GPU tests can only run through a host-staged, synchronous shim
(tests/test_gpu_world2.py) that cannot expose a stream-ordering bug.

  * two PROCESSES, one device each: 1,000 small steps (2 workers x batch 100 per rank) of a resident plan enqueued back to
    back without host synchronisation; the replicas must end bit-identical and on the oracle's weights with K = workers x
    world (core/Master.scala:194: the mean runs over all workers);
  * ONE process, one thread, two devices through the grouped entry points (dsgd_*_devices: ncclGroupStart / ncclGroupEnd
    around the all-reduces of both contexts): the same weights as the two processes, bit for bit."""

import os
import subprocess
import sys

import numpy as np
import pytest

import dsgd_amd
import waivers
from oracle import oracle as orc
from rccl2_worker import CFG, lists_of
from world2_common import shard_of


def n_devices():
    try:
        return dsgd_amd.device_count()
    except Exception:
        return 0


pytestmark = [pytest.mark.gpu, pytest.mark.skipif(n_devices() < 2, reason="needs two gfx950 devices (real RCCL refuses two ranks on one)")]

HERE = os.path.dirname(os.path.abspath(__file__))
WORLD = 2


@pytest.fixture(scope="module")
def two_processes(tmp_path_factory):
    wd = str(tmp_path_factory.mktemp("rccl2"))
    env = {k: v for k, v in os.environ.items() if k not in ("DSGD_LIB_PATH", "DSGD_RCCL_LIB")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "rccl2_worker.py"), str(r), str(WORLD), wd], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(WORLD)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=900)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-4000:])
    return [dict(np.load(os.path.join(wd, "out_%d.npz" % r))) for r in range(WORLD)]


def oracle_run():
    data = dsgd_amd.synth.generate(CFG["n_rows"], seed=CFG["seed"])
    o = orc.Oracle(data.dim, data.row_ptr, data.col, data.val, data.label, CFG["lam"])
    o.set_dim_sparsity(o.dim_sparsity(CFG["n_train"]))
    shards = [shard_of(data, CFG["n_train"], r, WORLD) for r in range(WORLD)]
    per_rank = [lists_of(r, shards[r].n_train) for r in range(WORLD)]
    w = np.zeros(data.dim + 1)
    exposed, active = 0, 0
    for s in range(CFG["steps"]):
        lists = [(l.astype(np.int64) + shards[r].train_lo).astype(np.int32) for r in range(WORLD) for l in per_rank[r][s]]
        o.sync_step(w, lists, 0.5)   # K = workers x world, rank order = the all-reduce's order
        exposed += o.last_stats["min_abs_margin"] < 1e-5
        active += o.last_stats["n_active"]
    return data, shards, per_rank, w, exposed, active


def test_a_thousand_steps_back_to_back_over_real_rccl(two_processes):
    r0, r1 = two_processes
    np.testing.assert_array_equal(r0["w"], r1["w"])                      # bit-identical replicas after 1,000 all-reduces
    data, shards, per_rank, w_ref, exposed, active = oracle_run()
    n = CFG["steps"] * CFG["workers"] * CFG["batch"]
    assert int(r0["stats"][0]) == n and int(r1["stats"][0]) == n
    err = np.abs(r0["w"].astype(np.float64) - w_ref).max()
    ok = err <= 1e-5 * max(1.0, np.abs(w_ref).max()) and int(r0["stats"][1] + r1["stats"][1]) == active
    waivers.tight("rccl2:thousand_steps", ok, exposed > 0, "%d steps with a row within 1e-5 of the gate, err %.3g" % (exposed, err))


def test_one_thread_two_devices_equals_the_two_processes(two_processes):
    data = dsgd_amd.synth.generate(CFG["n_rows"], seed=CFG["seed"])
    shards = [shard_of(data, CFG["n_train"], r, WORLD) for r in range(WORLD)]
    engines = [dsgd_amd.Engine(data.dim, CFG["lam"], device=r) for r in range(WORLD)]
    try:
        for eng, sh in zip(engines, shards):
            eng.load_csr(sh.csr.row_ptr, sh.csr.col, sh.csr.val, sh.csr.label)
        grp = dsgd_amd.EngineGroup(engines)
        grp.comm_init_all()
        grp.build_dim_sparsity([sh.n_train for sh in shards])
        per_rank = [lists_of(r, shards[r].n_train) for r in range(WORLD)]
        for s in range(50):
            grp.sync_step([per_rank[r][s] for r in range(WORLD)], 0.5)
        w = [e.get_weights() for e in engines]
    finally:
        for e in engines:
            e.close()
    np.testing.assert_array_equal(w[0], w[1])
    np.testing.assert_array_equal(w[0], two_processes[0]["w_req"])   # the rank processes' 50 per-request steps from w = 0
