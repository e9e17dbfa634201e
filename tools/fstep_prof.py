#!/usr/bin/env python3
"""Tuning aid for the chunked one-launch gradient (csrc/dsgd_fstep.hpp): whole-split steps of a row range, wall time per
step and -- with DSGD_PLAN_PROF=1 -- the shader-clock cycles thread 0 of every workgroup spent per phase (averaged over
the workgroups; ~2.35 GHz).  Run plain, or under `rocprofv3 --kernel-trace --stats` for the per-kernel view.

    [DSGD_PLAN_PROF=1] python tools/fstep_prof.py rows[,rows...] [workers] [steps]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dsgd_amd  # noqa: E402

sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "804414").split(",")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
warmup = int(sys.argv[4]) if len(sys.argv) > 4 else 20   # (100+: behind the measured re-cuts of the row chunks, 24 launches each)
NAMES = ("A_setup", "A_tiles", "B_setup", "B_tiles", "B_out", "C_setup", "C_tiles", "C_out")
out = []
for rows in sizes:
    data = dsgd_amd.synth.generate(rows, seed=0)
    n_train = int(rows * 0.8)
    size = -(-n_train // k)
    ranges = [(j * size, min(n_train, (j + 1) * size)) for j in range(k)]
    with dsgd_amd.Engine(data.dim, 1e-5) as eng:
        eng.load_csr(data.row_ptr, data.col, data.val, data.label)
        eng.build_dim_sparsity(n_train)
        lr = 0.5 * 100 / n_train * k
        for _ in range(warmup):
            eng.sync_step_ranges(ranges, lr, asynchronous=True)
            if _ % 10 == 9:
                eng.synchronize()
        eng.synchronize()
        eng.debug_cycles(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.sync_step_ranges(ranges, lr, asynchronous=True)
        eng.synchronize()
        dt = (time.perf_counter() - t0) / steps
        cyc = eng.debug_cycles(reset=True)
        row = {"rows": rows, "train_rows": n_train, "workers": k, "us_per_step": 1e6 * dt, "kernel": eng.grad_kernel_name(),
               "algorithmic_MB": (8.0 * int(data.row_ptr[n_train]) + 12.0 * n_train) / 1e6,
               "chunks_recut": eng.tuning_info().get("fstep_rebalances")}
        if cyc[15]:
            n = float(cyc[15])
            row["workgroups_per_launch"] = n / steps
            row["phase_us_avg_per_workgroup"] = {nm: round(cyc[i] / n / 2350.0, 2) for i, nm in enumerate(NAMES)}
            row["workgroup_total_us_avg"] = round(cyc[8] / n / 2350.0, 2)
            row["workgroup_total_us_slowest_ever"] = round(cyc[9] / 2350.0, 2)
            if cyc[11]:
                row["long_row_us_of_wave0"] = round(cyc[10] / float(cyc[11]) / 2350.0, 2)
        out.append(row)
        print(json.dumps(row), flush=True)
