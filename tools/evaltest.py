import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import dsgd_amd
rows = 3000000
data = dsgd_amd.synth.generate(rows, seed=0)
n_train = int(rows * 0.8)
alg = 8.0 * int(data.row_ptr[n_train]) + 12.0 * n_train
nnz = np.diff(data.row_ptr)
print("row len: mean %.1f p50 %d p90 %d p99 %d max %d" % (nnz.mean(), np.percentile(nnz, 50), np.percentile(nnz, 90), np.percentile(nnz, 99), nnz.max()))
for skip in (0, 1):
    if skip: os.environ["DSGD_DBG_SKIPROWS"] = "1"
    eng = dsgd_amd.Engine(data.dim, 1e-5)
    eng.load_csr(data.row_ptr, data.col, data.val, data.label)
    eng.build_dim_sparsity(n_train)
    eng.loss_acc(0, n_train)
    t0 = time.perf_counter()
    for _ in range(10): eng.loss_acc(0, n_train)
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print("skip_rowphase=%d eval %.3f ms %.0f GB/s" % (skip, ms, alg / ms / 1e6))
    eng.close()
