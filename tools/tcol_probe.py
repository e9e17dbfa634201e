#!/usr/bin/env python3
"""Round-5 probe (GPU box): whole-split steps of row ranges of 10^3 .. 2 * 10^5 rows -- the column lists
(csrc/dsgd_tcol.hpp: dot + column-wise gradient + reduce) next to the row-wise kernel (DSGD_TCOL=0) and the chunked
launch (csrc/dsgd_fstep.hpp).  Prints one JSON object: us per step (median of 5 x 20 asynchronous steps, min), the
gradient launches' average duration by HIP events.

    python tools/tcol_probe.py [rows,rows,...] [variant,variant,...]
"""

import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import dsgd_amd  # noqa: E402

LAM = 1e-5
BIG = "100000000"
VARIANTS = {
    "rowwise": {"DSGD_TCOL": "0", "DSGD_FSTEP": "0", "DSGD_STREAM_MIN": BIG},
    "chunks": {"DSGD_TCOL": "0", "DSGD_FSTEP": "1", "DSGD_FSTEP_MIN": "1000"},
    "columns": {"DSGD_TCOL": "1", "DSGD_TCOL_MIN": "1", "DSGD_TCOL_MAX": BIG},
    "columns_share2048": {"DSGD_TCOL": "1", "DSGD_TCOL_MIN": "1", "DSGD_TCOL_MAX": BIG, "DSGD_TCOL_SHARE": "2048"},
    "columns_share4096": {"DSGD_TCOL": "1", "DSGD_TCOL_MIN": "1", "DSGD_TCOL_MAX": BIG, "DSGD_TCOL_SHARE": "4096"},
    "columns_share8192": {"DSGD_TCOL": "1", "DSGD_TCOL_MIN": "1", "DSGD_TCOL_MAX": BIG, "DSGD_TCOL_SHARE": "8192"},
    "product": {},
}


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
            kernel_ms, n_launch = eng.prof_read(reset=True)
            eng.prof_enable(0)
            # (the same loop without the event records: what a caller sees)
            ts2 = []
            for _ in range(repeats):
                eng.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    eng.sync_step_ranges(ranges, lr, asynchronous=True)
                eng.synchronize()
                ts2.append((time.perf_counter() - t0) / steps)
            nnz = int(data.row_ptr[n_train])
            algo = 8.0 * nnz + 12.0 * n_train
            med = float(np.median(ts2))
            loss, acc, _ = eng.loss_acc(n_train, data.n_rows)
            return {"us_per_step": 1e6 * med, "us_min": 1e6 * min(ts2), "us_per_step_with_events": 1e6 * float(np.median(ts)),
                    "kernel": eng.grad_kernel_name(), "grad_launches_us_avg": 1e3 * kernel_ms / max(1, n_launch),
                    "step_frac_hbm": algo / med / 8e12, "algorithmic_MB": algo / 1e6, "fix_shift": eng.tuning_info()["fix_shift"],
                    "test_loss_after": loss}
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def main():
    sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "23149,100552").split(",")]
    variants = (sys.argv[2] if len(sys.argv) > 2 else "rowwise,chunks,columns").split(",")
    workers = [int(x) for x in os.environ.get("PROBE_WORKERS", "1,3").split(",")]
    out = {}
    for n_rows in sizes:
        data = dsgd_amd.synth.generate(n_rows, seed=0)
        n_train = int(n_rows * 0.8)
        row = {}
        for k in workers:
            size = -(-n_train // k)
            ranges = [(j * size, min(n_train, (j + 1) * size)) for j in range(k)]
            r = {}
            for v in variants:
                try:
                    r[v] = measure(data, n_train, ranges, VARIANTS[v])
                except Exception as e:   # (a variant that fails must not cost the visit its other numbers)
                    r[v] = {"error": str(e)[:300]}
            row["workers=%d" % k] = r
            print(n_rows, k, {a: (round(b.get("us_per_step", -1), 1), round(b.get("grad_launches_us_avg", -1), 1), b.get("kernel"), round(b.get("test_loss_after", -1), 6))
                              for a, b in r.items()}, file=sys.stderr, flush=True)
        out["rows=%d" % n_rows] = row
    print(json.dumps(out))


if __name__ == "__main__":
    main()
