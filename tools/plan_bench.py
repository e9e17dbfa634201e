"""Time per small-batch step through resident plans (dsgd_plan_run).  (The `flags` loop dates from the hipGraph
experiment recorded in profiles/README.md: DSGD_F_NO_GRAPH is a reserved flag now, both passes run the same code.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dsgd_amd

data = dsgd_amd.synth.generate(200000, seed=0)
n_train = 160000
rng = np.random.default_rng(1)
for flags in (0, 1):
    eng = dsgd_amd.Engine(data.dim, 1e-5, flags=flags)
    eng.load_csr(data.row_ptr, data.col, data.val, data.label)
    eng.build_dim_sparsity(n_train)
    for k, b, steps in ((1, 100, 400), (3, 100, 400), (1, 4096, 100)):
        lists = [[rng.permutation(n_train)[:b].astype(np.int32) for _ in range(k)] for _ in range(steps)]
        plan = eng.plan(lists)
        eng.set_weights(np.zeros(eng.dp, dtype=np.float32))
        eng.plan_run(plan, 0, steps, 0.5); eng.synchronize()          # capture + first run
        w1 = eng.get_weights()
        eng.set_weights(np.zeros(eng.dp, dtype=np.float32))
        t0 = time.perf_counter()
        eng.plan_run(plan, 0, steps, 0.5); eng.synchronize()
        dt = time.perf_counter() - t0
        same = np.abs(eng.get_weights() - w1).max()
        print("flags %d  workers %d  batch %5d: %6.1f us/step   (replay vs first run: max |dw| %.2g)" % (flags, k, b, 1e6 * dt / steps, same), flush=True)
        plan.destroy()
    eng.close()
