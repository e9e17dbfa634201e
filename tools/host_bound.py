#!/usr/bin/env python3
"""Is a resident plan's step rate bound by the host's launches?  Time until dsgd_plan_run has ENQUEUED n steps vs the
time until the GPU has finished them (2 M rows; 3 workers x 100 and 1 x 4,096)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dsgd_amd  # noqa: E402

data = dsgd_amd.synth.generate(2000000, seed=0)
n_train = 1600000
out = []
with dsgd_amd.Engine(data.dim, 1e-5) as eng:
    eng.load_csr(data.row_ptr, data.col, data.val, data.label)
    eng.build_dim_sparsity(n_train)
    rng = np.random.default_rng(1)
    for k, b, steps in ((3, 100, 2000), (1, 4096, 500)):
        size = -(-n_train // k)
        lists = [[(a + rng.permutation(min(size, n_train - a))[:b]).astype(np.int32) for a in range(0, n_train, size)][:k] for _ in range(steps)]
        plan = eng.plan(lists)
        eng.plan_run(plan, 0, 20, 0.5 * 100 / b)
        eng.synchronize()
        t0 = time.perf_counter()
        eng.plan_run(plan, 0, steps, 0.5 * 100 / b)
        t1 = time.perf_counter()
        eng.synchronize()
        t2 = time.perf_counter()
        plan.destroy()
        out.append({"workers": k, "batch": b, "steps": steps, "enqueue_us_per_step": 1e6 * (t1 - t0) / steps, "total_us_per_step": 1e6 * (t2 - t0) / steps})
print(json.dumps(out, indent=1))
