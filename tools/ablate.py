import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dsgd_amd
rows = 3000000
data = dsgd_amd.synth.generate(rows, seed=0)
n_train = int(rows * 0.8)
alg = 8.0 * int(data.row_ptr[n_train]) + 12.0 * n_train
eng = dsgd_amd.Engine(data.dim, 1e-5)
eng.load_csr(data.row_ptr, data.col, data.val, data.label)
eng.build_dim_sparsity(n_train)
lr = 0.5 * 100 / n_train
for _ in range(8): eng.sync_step_ranges([(0, n_train)], lr)
w = eng.get_weights()
for dbg in [0, 32, 1]:
    os.environ["DSGD_DBG"] = str(dbg)
    eng.set_weights(w)
    eng.prof_enable(True); eng.prof_read(reset=True)
    for _ in range(5): eng.sync_step_ranges([(0, n_train)], 0.0)
    ms, n = eng.prof_read(reset=True)
    eng.loss_acc(0, n_train)
    t0 = time.perf_counter()
    for _ in range(5): eng.loss_acc(0, n_train)
    ev = (time.perf_counter() - t0) / 5 * 1e3
    print("dbg %2d  grad(+finalize+cold) %.3f ms %5.0f GB/s | eval(wall) %.3f ms %5.0f GB/s" % (dbg, ms, alg / ms / 1e6, ev, alg / ev / 1e6), flush=True)
