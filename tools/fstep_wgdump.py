import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, dsgd_amd
rows = int(sys.argv[1])
data = dsgd_amd.synth.generate(rows, seed=0)
n_train = int(rows * 0.8)
with dsgd_amd.Engine(data.dim, 1e-5) as eng:
    eng.load_csr(data.row_ptr, data.col, data.val, data.label)
    eng.build_dim_sparsity(n_train)
    lr = 0.5 * 100 / n_train
    for _ in range(20):
        eng.sync_step_ranges([(0, n_train)], lr, asynchronous=True)
    eng.synchronize()
    for rep in range(6):
        for _ in range(3):
            eng.sync_step_ranges([(0, n_train)], lr, asynchronous=True)
        eng.synchronize()
        eng.debug_cycles(reset=True)
