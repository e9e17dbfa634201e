#!/usr/bin/env python3
"""Round-5 probe (GPU box): whole-split steps of row ranges of 10^4 .. 2 * 10^6 rows -- the chunked ONE-launch gradient
(csrc/dsgd_fstep.hpp) by rows per chunk, next to the three streaming launches / the row-wise kernel (DSGD_FSTEP=0).
Prints one JSON object: us per step (median of 5 x 20 steps, min), the gradient kernel's average duration."""

import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import dsgd_amd  # noqa: E402

LAM = 1e-5


def measure(data, n_train, ranges, env, steps=20, repeats=5):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        with dsgd_amd.Engine(data.dim, LAM) as eng:
            eng.load_csr(data.row_ptr, data.col, data.val, data.label)
            eng.build_dim_sparsity(n_train)
            lr = 0.5 * 100 / n_train * len(ranges)
            for _ in range(30):
                eng.sync_step_ranges(ranges, lr, asynchronous=True)
            eng.synchronize()
            eng.prof_enable(2)
            eng.prof_read(reset=True)
            ts = []
            for _ in range(repeats):
                eng.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    eng.sync_step_ranges(ranges, lr, asynchronous=True)
                eng.synchronize()
                ts.append((time.perf_counter() - t0) / steps)
            kinds = eng.prof_read_kinds()
            kernel_ms, n_launch = eng.prof_read(reset=True)
            eng.prof_enable(0)
            nnz = int(data.row_ptr[n_train])
            algo = 8.0 * nnz + 12.0 * n_train
            med = float(np.median(ts))
            return {"us_per_step": 1e6 * med, "us_min": 1e6 * min(ts), "kernel": eng.grad_kernel_name(),
                    "kernel_us_avg": 1e3 * kernel_ms / max(1, n_launch), "step_frac_hbm": algo / med / 8e12,
                    "fix_shift": eng.tuning_info()["fix_shift"], "kinds": kinds}
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    sizes = [int(x) for x in os.environ.get("PROBE_SIZES", "23149,100552,804414,2000000").split(",")]
    chunk_rows = os.environ.get("PROBE_ROWS", "128,256,512,1024,2048").split(",")
    out = {}
    for n_rows in sizes:
        data = dsgd_amd.synth.generate(n_rows, seed=0)
        n_train = int(n_rows * 0.8)
        row = {}
        for k in (1, 3):
            size = -(-n_train // k)
            ranges = [(j * size, min(n_train, (j + 1) * size)) for j in range(k)]
            r = {"three_launches": measure(data, n_train, ranges, {"DSGD_FSTEP": "0"})}
            for cr in chunk_rows:
                r["chunks_of_%s" % cr] = measure(data, n_train, ranges, {"DSGD_FSTEP": "1", "DSGD_FSTEP_ROWS": cr, "DSGD_FSTEP_MIN": "1000", "DSGD_FSTEP_MAX": "100000000"})
            row["workers=%d" % k] = r
            print(n_rows, k, {a: (round(b["us_per_step"], 1), round(b["kernel_us_avg"], 1), b["kernel"]) for a, b in r.items()}, file=sys.stderr, flush=True)
        out["rows=%d" % n_rows] = row
    print(json.dumps(out))


if __name__ == "__main__":
    main()
