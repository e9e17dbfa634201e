#!/usr/bin/env python3
"""K8 tuning aid: wall time per dense mini-batch step (no per-launch events), best of three runs per batch size.
At commit 21f3295 the library also had the reduce + update fused into the step kernel's tail (DSGD_DENSE_FUSED): this
script produced profiles/r03_dense_fused_tail_ab.json there (fused measured slower; see profiles/README.md).

    python tools/dense_check.py [rows]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dsgd_amd  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1250000
dim = 4096
out = {"rows": rows, "dim": dim, "cases": []}
for fused in ("0",):
    with dsgd_amd.DenseLogistic(dim) as eng:
        eng.generate(rows, seed=0)
        for b, steps in ((4096, 400), (65536, 40), (1024, 400), (512, 400)):
            starts = [(i * b) % (rows - b) for i in range(steps + 10)]
            for st in starts[:10]:
                eng.step(st, st + b, 1.0)
            eng.synchronize()
            best = None
            for rep in range(3):
                t0 = time.perf_counter()
                for st in starts[10:]:
                    eng.step(st, st + b, 1.0)
                eng.synchronize()
                dt = (time.perf_counter() - t0) / steps
                best = dt if best is None else min(best, dt)
            out["cases"].append({"fused": int(fused), "batch": b, "us_per_step": 1e6 * best,
                                 "frac_hbm_peak": b * (4 * dim + 4) / best / 8e12})
        loss, acc = eng.loss(rows - 65536, rows)
        out["cases"].append({"fused": int(fused), "loss": loss, "acc": acc})
print(json.dumps(out, indent=1))
