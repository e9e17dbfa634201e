#!/usr/bin/env python3
"""Tuning aid: where a small-batch step of dsgd_plan_kernel spends its time (DSGD_PLAN_PROF=1 cycle counters of
thread 0) and the wall time per step, for the reference's batch sizes; and the Hogwild engine's rate by worker count.

    python tools/plan_prof.py [rows]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DSGD_PLAN_PROF"] = "1"
import dsgd_amd  # noqa: E402
from dsgd_amd import host  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
data = dsgd_amd.synth.generate(rows, seed=0)
n_train = int(rows * 0.8)
out = {"rows": rows, "plan": [], "hogwild": []}
with dsgd_amd.Engine(data.dim, 1e-5) as eng:
    eng.load_csr(data.row_ptr, data.col, data.val, data.label)
    eng.build_dim_sparsity(n_train)
    rng = np.random.default_rng(1)
    for k, b, steps in ((1, 100, 400), (3, 100, 300), (4, 200, 200), (1, 1, 400), (1, 1000, 100)):
        size = -(-n_train // k)
        split = [np.arange(a, min(n_train, a + size)) for a in range(0, n_train, size)]
        lists = [[rng.choice(sp, size=b, replace=False).astype(np.int32) for sp in split] for _ in range(steps)]
        eng.set_weights(np.zeros(eng.dp, dtype=np.float32))
        plan = eng.plan(lists)
        eng.plan_run(plan, 0, min(20, steps), 0.5)
        eng.synchronize()
        eng.debug_cycles(reset=True)
        t0 = time.perf_counter()
        eng.plan_run(plan, 0, steps, 0.5)
        eng.synchronize()
        dt = time.perf_counter() - t0
        cyc = eng.debug_cycles(reset=True)
        plan.destroy()
        n = max(1, cyc[15])
        names = ("gather_dot", "barrier1", "gate_tables", "barrier2", "scatter", "barrier3", "requests", "sweep", "barrier4_collect")
        out["plan"].append({"workers": k, "batch": b, "steps": steps, "us_per_step": 1e6 * dt / steps,
                            "kernel": eng.grad_kernel_name(),
                            "cycles_per_step": {nm: cyc[i] / n for i, nm in enumerate(names)}})
    for workers, updates in ((1, 2000), (16, 8000), (64, 20000), (256, 40000)):
        eng.set_weights(np.zeros(eng.dp, dtype=np.float32))
        split = [(r.start, r.stop) for r in host.split_vanilla(n_train, workers)]
        t0 = time.perf_counter()
        eng.async_start(split, batch=100, lr=0.5, max_updates=updates, seed=1, positional_bug=False)
        eng.async_wait()
        dt = time.perf_counter() - t0
        u, _ = eng.async_updates()
        out["hogwild"].append({"workers": workers, "updates": int(u), "examples_per_s": u * 100 / dt,
                               "us_per_iteration_per_worker": 1e6 * dt * workers / max(1, u)})
print(json.dumps(out, indent=1))
