#!/usr/bin/env python3
"""Latency of the drop-in boundary's per-request entry points WITH their host buffers (the S2 handlers of SURVEY.md
8(b): Slave.gradient / Slave.forward hand over `w` and the sample indices per call and get a dense vector back), next
to the device-resident forms the benchmark times.  PCIe-inclusive: every call below starts and ends in host memory.

    python tools/boundary_latency.py [rows]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402

import dsgd_amd  # noqa: E402
from dsgd_amd._lib import BatchStats, check  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
data = dsgd_amd.synth.generate(rows, seed=0)
n_train = int(rows * 0.8)
rng = np.random.default_rng(3)
out = {"rows": rows, "dim": data.dim, "vector_bytes": 4 * (data.dim + 1)}


def timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return 1e6 * (time.perf_counter() - t0) / reps


with dsgd_amd.Engine(data.dim, 1e-5) as eng:
    eng.load_csr(data.row_ptr, data.col, data.val, data.label)
    eng.build_dim_sparsity(n_train)
    w = np.zeros(data.dim + 1, dtype=np.float32)
    w[rng.choice(np.arange(1, data.dim + 1), 20000, replace=False)] = rng.normal(scale=0.05, size=20000).astype(np.float32)
    eng.set_weights(w)
    for b in (100, 4096):
        lists = [rng.choice(n_train, size=b, replace=False).astype(np.int32) for _ in range(64)]
        it = iter(range(10 ** 9))
        nxt = lambda: lists[next(it) % len(lists)]
        rec = {"batch": b}
        # Slave.gradient (core/Slave.scala:142-157): w and idx from the host, dense g back to the host
        rec["gradient_host_w_us"] = timed(lambda: eng.gradient(nxt(), w=w), 300)
        # the same with the weights already resident (w = NULL)
        rec["gradient_resident_w_us"] = timed(lambda: eng.gradient(nxt()), 300)
        # Slave.forward (core/Slave.scala:129-140): w and idx in, predictions out
        rec["forward_host_w_us"] = timed(lambda: eng.forward(nxt(), w=w), 300)
        # the fused step with host index lists (one worker): idx uploaded per call, nothing comes back but statistics
        rec["sync_step_host_idx_us"] = timed(lambda: eng.sync_step([nxt()], 0.0), 300)
        if b == 100:   # the reference's own configuration (application.conf:15,27): three hosted workers per request
            rec["sync_step_3x100_host_idx_us"] = timed(lambda: eng.sync_step([nxt(), nxt(), nxt()], 0.0), 300)
        # the same calls AT THE C ABI: the argument arrays built once (what a JNI / cgo caller hands over is already in this
        # form; the Python binding above spends ~11 us per call building them)
        for label, k in (("sync_step_abi_us", 1),) + ((("sync_step_3x100_abi_us", 3),) if b == 100 else ()):
            reqs = []
            for r in range(16):
                ls = [np.ascontiguousarray(nxt()) for _ in range(k)]
                reqs.append((ls, (C.c_void_p * k)(*[a.ctypes.data for a in ls]), (C.c_int64 * k)(*[len(a) for a in ls])))
            st = BatchStats()
            fn, ctx, kk, lr0, stp = eng._lib.dsgd_sync_step, eng._ctx, C.c_int32(k), C.c_float(0.0), C.byref(st)
            ctr = iter(range(10 ** 9))

            def call():
                r = reqs[next(ctr) % 16]
                check(fn(ctx, r[1], r[2], kk, lr0, stp))
            rec[label] = timed(call, 1000)
        # resident plan (what bench.py's sweep times): nothing crosses PCIe inside the loop
        plan = eng.plan([[l] for l in lists])
        eng.plan_run(plan, 0, len(lists), 0.0)
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            eng.plan_run(plan, 0, len(lists), 0.0)
        eng.synchronize()
        rec["resident_plan_step_us"] = 1e6 * (time.perf_counter() - t0) / (5 * len(lists))
        plan.destroy()
        out.setdefault("per_request", []).append(rec)
    out["loss_acc_test_split_us"] = timed(lambda: eng.loss_acc(n_train, rows), 50)
    out["get_weights_us"] = timed(lambda: eng.get_weights(), 200)
    out["set_weights_us"] = timed(lambda: eng.set_weights(w), 200)
print(json.dumps(out, indent=1))
