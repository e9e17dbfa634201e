"""Minimal workload for rocprofv3 --pmc passes: load the default bench shard, run a few whole-shard steps, exit.
usage: python tools/pmc_step.py [rows] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsgd_amd

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8388608
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
data = dsgd_amd.synth.generate(rows, seed=0)
n_train = int(rows * 0.8)
eng = dsgd_amd.Engine(data.dim, 1e-5)
eng.load_csr(data.row_ptr, data.col, data.val, data.label)
eng.build_dim_sparsity(n_train)
lr = 0.5 * 100 / n_train
for _ in range(steps):
    eng.sync_step_ranges([(0, n_train)], lr)
eng.synchronize()
nnz, cold = eng.range_nnz(0, n_train)
print("rows %d train %d nnz %d cold %d kernel %s" % (rows, n_train, nnz, cold, eng.grad_kernel_name()))
