#!/usr/bin/env python3
"""Tuning aid for index-list batches beyond one workgroup (dsgd_mb_grad_kernel + the fused reduce): wall time per step
from resident plans for SURVEY.md 8(d)'s sweep sizes and the reference's multi-worker defaults, with the algorithmic
bytes (8 B per non-zero + 12 B per row) each step reads.  Run plain, or under `rocprofv3 --kernel-trace --stats` /
`--pmc` (tools/r03_visit.sh) for the per-kernel view.

    python tools/mb_prof.py [rows] [--quick]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dsgd_amd  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2000000
quick = "--quick" in sys.argv
single = "--single" in sys.argv   # one launch per step (the dispatch table of a trace then shows the cost of ONE step's launch)
only = [tuple(int(x) for x in a.split("=")[1].split("x")) for a in sys.argv if a.startswith("--only=")]   # e.g. --only=1x65536
data = dsgd_amd.synth.generate(rows, seed=0)
n_train = int(rows * 0.8)
out = {"rows": rows, "steps": []}
with dsgd_amd.Engine(data.dim, 1e-5) as eng:
    eng.load_csr(data.row_ptr, data.col, data.val, data.label)
    eng.build_dim_sparsity(n_train)
    rng = np.random.default_rng(1)
    cases = ((1, 65536, 40), (1, 4096, 100), (3, 100, 200), (4, 200, 200), (1, 100, 200), (1, 1000, 100), (1, 16384, 60), (8, 4096, 40))
    if only:
        cases = tuple(c for c in cases if (c[0], c[1]) in only)
    for k, b, steps in cases[:2] if quick else cases:
        size = -(-n_train // k)
        lists = []
        for _ in range(steps):
            lists.append([(a + rng.permutation(min(size, n_train - a))[:b]).astype(np.int32) for a in range(0, n_train, size)][:k])
        nnz = float(np.mean([sum(int((data.row_ptr[l + 1] - data.row_ptr[l]).sum()) for l in st) for st in lists[:8]]))
        eng.set_weights(np.zeros(eng.dp, dtype=np.float32))
        plan = eng.plan(lists)
        eng.plan_run(plan, 0, min(10, steps), 0.5 * 100 / b)
        eng.synchronize()
        eng.debug_cycles(reset=True)
        t0 = time.perf_counter()
        if single:
            for i in range(steps):
                eng.plan_run(plan, i, i + 1, 0.5 * 100 / b)
        else:
            eng.plan_run(plan, 0, steps, 0.5 * 100 / b)
        eng.synchronize()
        dt = (time.perf_counter() - t0) / steps
        cyc = eng.debug_cycles(reset=True)
        plan.destroy()
        alg = 8.0 * nnz + 12.0 * k * b
        out["steps"].append({"workers": k, "batch": b, "steps_per_launch": 1 if single else steps, "us_per_step": 1e6 * dt, "kernel": eng.grad_kernel_name(),
                             "algorithmic_bytes": alg, "GBps": alg / dt / 1e9, "frac_of_8TBps": alg / dt / 8e12,
                             "examples_per_s": k * b / dt, "fix_shift": eng.tuning_info()["fix_shift"]})
        if cyc[15]:   # DSGD_PLAN_PROF=1: cycles of wave 0 of workgroup 0 per launch, by phase
            names = ("issue_ids_copy_clear", "row_records", "first_items", "barrier_in", "pass_requests", "pass_dot",
                     "pass_scatter", "barrier_out", "write_partial")
            if "cs_step" in eng.grad_kernel_name():   # csrc/dsgd_cs.hpp: thread 0 of slice 0, cycles per STEP
                names = ("dot", "publish", "exchange", "scatter", "sweep", "reduce", "launch_setup", "launch_writeback")
            out["steps"][-1]["wave0_cycles_per_launch"] = {nm: cyc[i] / cyc[15] for i, nm in enumerate(names)}
print(json.dumps(out, indent=1))
