#!/usr/bin/env python3
"""Round-5 probe (GPU box): what the boundary costs now.  Prints one JSON object.

  * dsgd_sync_step per request (3 x 100, 1 x 100, 4 x 200) at the C ABI's argument form, through the request kernel and with
    DSGD_CS_REQ=0 (the row-parallel kernels of round 4);
  * a plan per epoch: create (lists up, layout on the device), run, destroy -- 23,149 and 804,414 rows, 3 x 100;
  * the reference's random stream natively: one epoch's lists, both shapes, by thread count;
  * host.MasterSync.fit through plans: us per step, shuffles included."""

import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import dsgd_amd  # noqa: E402
from dsgd_amd import host  # noqa: E402

LAM = 1e-5


def requests(eng, n_train, k, b, n=300):
    rng = np.random.default_rng(1)
    size = -(-n_train // k)
    lists = [[(j * size + rng.permutation(min(size, n_train - j * size))[:b]).astype(np.int32) for j in range(k)] for _ in range(32)]
    for i in range(20):
        eng.sync_step(lists[i % 32], 0.0)
    t0 = time.perf_counter()
    for i in range(n):
        eng.sync_step(lists[i % 32], 0.0)
    dt = time.perf_counter() - t0
    return {"us_per_request_python_binding": 1e6 * dt / n, "kernel": eng.grad_kernel_name()}


def plan_cycle(eng, n_train, k, b, epochs=4):
    split = host.split_vanilla(n_train, k)
    mx = max(len(r) for r in split)
    rnd = host.JavaRandom(0)
    out = []
    for e in range(epochs):
        t0 = time.perf_counter()
        idx, offs, ns = host.epoch_lists(rnd, split, mx, b)
        t1 = time.perf_counter()
        plan = eng.plan_flat(idx, offs, ns, k)
        t2 = time.perf_counter()
        eng.plan_run(plan, 0, ns, 0.5 * 100 / b)
        t3 = time.perf_counter()
        eng.synchronize()
        t4 = time.perf_counter()
        info = plan.info()
        plan.destroy()
        t5 = time.perf_counter()
        out.append({"steps": ns, "shuffle_ms": 1e3 * (t1 - t0), "plan_create_ms": 1e3 * (t2 - t1), "run_enqueue_ms": 1e3 * (t3 - t2),
                    "run_wait_ms": 1e3 * (t4 - t3), "destroy_ms": 1e3 * (t5 - t4), "us_per_step_run": 1e6 * (t4 - t2) / ns,
                    "us_per_step_all": 1e6 * (t5 - t0) / ns, "kind": info["kind"], "device_built": info["device_built"],
                    "slot_stride": info["slot_stride"]})
    return out


def fit_leg(eng, n_train, n_rows, epochs, k=3, b=100):
    eng.set_weights(np.zeros(eng.dp, dtype=np.float32))
    res = {}
    for prefetch in (False, True):
        m = host.MasterSync(eng, n_train, n_rows, k, rnd=host.JavaRandom(0), plans=True, prefetch=prefetch)
        t0 = time.perf_counter()
        m.fit(np.zeros(eng.dp), epochs, b, 0.5, lambda losses: False)
        dt = time.perf_counter() - t0
        res["prefetch" if prefetch else "sequential"] = {
            "steps": m.steps_run, "batch_loop_us_per_step": 1e6 * m.batch_loop_s / max(1, m.steps_run), "shuffle_us_per_step_exposed": 1e6 * m.shuffle_s / max(1, m.steps_run),
            "fit_s": dt, "test_loss": m.test_losses[0]}
    return res


def with_a_communicator(eng, n_train, k=3, b=100, steps=300):
    """The reference's configuration when a communicator is attached (the N-GPU path: the column-slice kernel declines, a
    step is two row-parallel launches + one all-reduce of D + 1 floats + the update).  ONE rank here -- the all-reduce is
    RCCL's single-rank path -- so this is the per-step cost WITHOUT link time: the kernels, the launches, the collective's
    fixed cost."""
    rng = np.random.default_rng(2)
    size = -(-n_train // k)
    lists = [[(j * size + rng.permutation(min(size, n_train - j * size))[:b]).astype(np.int32) for j in range(k)] for _ in range(steps)]
    eng.comm_init(dsgd_amd.Engine.comm_unique_id(), 1, 0)
    try:
        plan = eng.plan(lists)
        eng.plan_run(plan, 0, 20, 0.0)
        eng.synchronize()
        t0 = time.perf_counter()
        eng.plan_run(plan, 0, steps, 0.0)
        eng.synchronize()
        dt = time.perf_counter() - t0
        kern = eng.grad_kernel_name()
        plan.destroy()
        t1 = time.perf_counter()
        for i in range(100):
            eng.sync_step(lists[i], 0.0)
        dr = time.perf_counter() - t1
    finally:
        eng.comm_destroy()
    return {"world": 1, "plan_us_per_step": 1e6 * dt / steps, "kernel": kern, "per_request_us": 1e6 * dr / 100}


def main():
    out = {"host_threads": host._host_lib().dsgd_host_threads(), "cpus": os.cpu_count()}
    # the random stream by thread count
    out["shuffle"] = {}
    for n in (18519, 643531):
        split = host.split_vanilla(n, 3)
        mx = max(len(r) for r in split)
        row = {}
        for th in ("1", "4", "16", "32", "64"):
            os.environ["DSGD_HOST_THREADS"] = th
            rnd = host.JavaRandom(0)
            ts = []
            for _ in range(4 if n < 100000 else 2):
                t0 = time.perf_counter()
                idx, offs, ns = host.epoch_lists(rnd, split, mx, 100)
                ts.append(time.perf_counter() - t0)
            row[th] = {"ms_min": 1e3 * min(ts), "ms_all": [round(1e3 * t, 3) for t in ts], "us_per_step": 1e6 * min(ts) / ns}
        os.environ.pop("DSGD_HOST_THREADS", None)
        out["shuffle"]["n_train=%d" % n] = row
    for n_rows in (23149, 804414):
        data = dsgd_amd.synth.generate(n_rows, seed=0)
        n_train = int(n_rows * 0.8)
        key = "rows=%d" % n_rows
        out[key] = {}
        for req in ("1", "0"):
            os.environ["DSGD_CS_REQ"] = req
            with dsgd_amd.Engine(data.dim, LAM) as eng:
                eng.load_csr(data.row_ptr, data.col, data.val, data.label)
                eng.build_dim_sparsity(n_train)
                out[key]["requests_cs_req=%s" % req] = {"%dx%d" % (k, b): requests(eng, n_train, k, b) for k, b in ((3, 100), (1, 100), (4, 200))}
                if req == "1":
                    out[key]["plan_cycle_3x100"] = plan_cycle(eng, n_train, 3, 100)
                    out[key]["plan_cycle_4x200"] = plan_cycle(eng, n_train, 4, 200, epochs=2)
                    out[key]["fit"] = fit_leg(eng, n_train, n_rows, 6 if n_rows < 100000 else 2)
                    try:
                        out[key]["with_communicator_3x100"] = with_a_communicator(eng, n_train)
                    except Exception as e:   # (no RCCL on the box: say so)
                        out[key]["with_communicator_3x100"] = {"error": str(e)[:200]}
        os.environ.pop("DSGD_CS_REQ", None)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
