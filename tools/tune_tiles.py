#!/usr/bin/env python3
"""Tuning sweep on the GPU box: LDS tile split (hw, hg) of the tiled gradient kernel, and the
dot-only (evaluation) kernel, in GB/s of algorithmic CSR bytes (8 B/nnz + 12 B/row)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dsgd_amd  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 3000000
data = dsgd_amd.synth.generate(rows, seed=0)
n_train = int(rows * 0.8)
nnz_train = int(data.row_ptr[n_train])
alg = 8.0 * nnz_train + 12.0 * n_train
print("rows %d train %d nnz_train %d alg bytes %.1f MB" % (rows, n_train, nnz_train, alg / 1e6), flush=True)
B = n_train
lr = 0.5 * 100.0 / B

configs = [
    {"DSGD_STREAM": "0"},
    {"DSGD_STREAM": "1", "DSGD_PF_EARLY": "0"},
    {"DSGD_STREAM": "1", "DSGD_PF_EARLY": "1"},
    {"DSGD_STREAM": "1", "DSGD_PF_EARLY": "0", "DSGD_HW_S": "8192", "DSGD_HG_S": "22524"},
    {"DSGD_STREAM": "1", "DSGD_PF_EARLY": "0", "DSGD_HW_S": "2048", "DSGD_HG_S": "28668"},
    {"DSGD_STREAM": "1", "DSGD_PF_EARLY": "0", "DSGD_HW_S": "0", "DSGD_HG_S": "30716"},
]
if len(sys.argv) > 2:
    configs = [dict(kv.split("=") for kv in c.split(",")) for c in sys.argv[2:]]
for cfg in configs:
    for k in ("DSGD_STREAM", "DSGD_PF_EARLY", "DSGD_HW_S", "DSGD_HG_S", "DSGD_HW", "DSGD_HG"):
        os.environ.pop(k, None)
    os.environ.update(cfg)
    hw, hg = 0, 0
    print(cfg, flush=True)
    eng = dsgd_amd.Engine(data.dim, 1e-5)
    eng.load_csr(data.row_ptr, data.col, data.val, data.label)
    eng.build_dim_sparsity(n_train)
    eng.prof_enable(True)
    # (a) all rows active: w = 0 before every step
    zeros = np.zeros(data.dim + 1, dtype=np.float32)
    for _ in range(2):
        eng.set_weights(zeros)
        eng.sync_step_ranges([(0, n_train)], lr)
    eng.prof_read(reset=True)
    for _ in range(5):
        eng.set_weights(zeros)
        st = eng.sync_step_ranges([(0, n_train)], lr)
    ms_all, n = eng.prof_read(reset=True)
    # (b) training trajectory: 30 steps from w = 0
    eng.set_weights(zeros)
    fr = []
    ms_steps = []
    for i in range(30):
        st = eng.sync_step_ranges([(0, n_train)], lr)
        ms, _ = eng.prof_read(reset=True)
        fr.append(st["n_active"] / n_train)
        ms_steps.append(ms)
    loss, acc, _ = eng.loss_acc(n_train, rows)
    # (c) dot-only kernel: evaluation over the train rows
    eng.loss_acc(0, n_train)
    t0 = time.perf_counter()
    for _ in range(5):
        eng.loss_acc(0, n_train)
    ms_eval = (time.perf_counter() - t0) / 5 * 1e3
    print("hw %5d hg %5d | all-active %.3f ms %6.0f GB/s | steps 1/5/10/20/30: %s | active %s | eval(wall) %.3f ms %6.0f GB/s | test acc %.3f kernel %s" % (
        hw, hg, ms_all, alg / ms_all / 1e6,
        " ".join("%.3f" % ms_steps[i] for i in (0, 4, 9, 19, 29)),
        " ".join("%.2f" % fr[i] for i in (0, 4, 9, 19, 29)),
        ms_eval, alg / ms_eval / 1e6, acc, eng.grad_kernel_name()), flush=True)
    eng.close()
