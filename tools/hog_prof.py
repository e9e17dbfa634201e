#!/usr/bin/env python3
"""One Hogwild run (BASELINE.json configs[3]) for rocprofv3: workers x batch on `rows` synthetic rows.

    python tools/hog_prof.py [rows] [workers] [updates] [batch]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dsgd_amd  # noqa: E402
from dsgd_amd import host  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8388608
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 256
updates = int(sys.argv[3]) if len(sys.argv) > 3 else 40000
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 100
data = dsgd_amd.synth.generate(rows, seed=0)
n_train = int(rows * 0.8)
with dsgd_amd.Engine(data.dim, 1e-5) as eng:
    eng.load_csr(data.row_ptr, data.col, data.val, data.label)
    eng.build_dim_sparsity(n_train)
    eng.loss_acc(0, n_train)  # layout + clocks
    split = [(r.start, r.stop) for r in host.split_vanilla(n_train, workers)]
    res = []
    for rep in range(2):
        eng.set_weights(np.zeros(eng.dp, dtype=np.float32))
        t0 = time.perf_counter()
        eng.async_start(split, batch=batch, lr=0.5, max_updates=updates, seed=1 + rep, positional_bug=False)
        eng.async_wait()
        dt = time.perf_counter() - t0
        u, _ = eng.async_updates()
        res.append({"updates": int(u), "ms": 1e3 * dt, "examples_per_s": u * batch / dt,
                    "us_per_iteration_per_worker": 1e6 * dt * workers / max(1, u)})
        cyc = eng.debug_cycles(reset=True)
        if cyc[15]:   # DSGD_PLAN_PROF=1: cycles of thread 0 of worker 0 per iteration, by phase
            names = ("batch_dot_gate_scatter", "update_head", "update_rounds", "reduce_and_drain", "scalars_and_weight_copy",
                     "next_tables", "next_requests")
            res[-1]["worker0_cycles_per_iteration"] = {nm: cyc[i] / cyc[15] for i, nm in enumerate(names)}
            res[-1]["worker0_update_rounds_per_iteration"] = cyc[12] / cyc[15]
            res[-1]["worker0_iterations"] = cyc[15]
    loss, acc, _ = eng.loss_acc(n_train, rows)
print(json.dumps({"rows": rows, "workers": workers, "batch": batch, "runs": res, "test_loss": loss, "test_acc": acc}))
