#!/usr/bin/env python3
"""Does the library's CHOICE of gradient kernel for a whole-split step hold on shapes its thresholds were not tuned on?

For every (shape, rows): the step through the default dispatch and through every family it could have taken (forced by
the DSGD_* knobs, which a context reads when it is created): column lists (csrc/dsgd_tcol.hpp), row chunks
(csrc/dsgd_fstep.hpp), the three streaming launches (csrc/dsgd_kernels.hpp) and the row-wise kernel (csrc/dsgd_batch.hpp).
Prints one line per cell and a verdict: default within `tol` of the best eligible family.

    python tools/dispatch_table.py [--quick]      (tests/test_gpu_dispatch.py runs the same function)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dsgd_amd  # noqa: E402

FAMILIES = {
    "default": {},
    "column_lists": {"DSGD_TCOL_MIN": "1", "DSGD_TCOL_MAX": "100000000", "DSGD_TCOL_MAX_NNZ": "100000000000", "DSGD_FSTEP": "0"},
    "row_chunks": {"DSGD_TCOL": "0", "DSGD_FSTEP_MIN": "1"},
    "three_launches": {"DSGD_TCOL": "0", "DSGD_FSTEP": "0", "DSGD_STREAM_MIN": "1"},
    "row_wise": {"DSGD_TCOL": "0", "DSGD_FSTEP": "0", "DSGD_STREAM_MIN": "1000000000"},
}
KNOBS = sorted({k for env in FAMILIES.values() for k in env})
SHAPES = [{"zipf": 0.9, "nnz_mean": 40.0}, {"zipf": 1.3, "nnz_mean": 150.0}, {"zipf": 1.1, "nnz_mean": 75.0, "dim": 20000},
          {"zipf": 1.1, "nnz_mean": 75.0, "dim": 70000}]   # (the last: more cold columns than an LDS tile holds -- no row chunks)
ROWS = [2000, 20000, 80000, 200000, 800000]


def time_step(data, n_train, env, steps):
    saved = {k: os.environ.get(k) for k in KNOBS}
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        with dsgd_amd.Engine(data.dim, 1e-5) as eng:
            eng.load_csr(data.row_ptr, data.col, data.val, data.label)
            eng.build_dim_sparsity(n_train)
            lr = 0.5 * 100 / n_train
            for _ in range(10):
                eng.sync_step_ranges([(0, n_train)], lr, asynchronous=True)
            eng.synchronize()
            best = 1e30
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(steps):
                    eng.sync_step_ranges([(0, n_train)], lr, asynchronous=True)
                eng.synchronize()
                best = min(best, (time.perf_counter() - t0) / steps)
            return 1e6 * best, eng.grad_kernel_name()
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v


def table(shapes=SHAPES, rows=ROWS, tol=0.15, out=sys.stdout):
    worst, cells = 0.0, []
    for sh in shapes:
        for n in rows:
            data = dsgd_amd.synth.generate(n + 64, seed=3, dim=sh.get("dim", dsgd_amd.synth.RCV1_DIM), zipf=sh["zipf"], nnz_mean=sh["nnz_mean"])
            steps = 200 if n <= 100000 else 60
            res = {}
            for fam, env in FAMILIES.items():
                us, kern = time_step(data, n, env, steps)
                res[fam] = {"us_per_step": round(us, 2), "kernel": kern}
            # a forced family that ran the SAME kernel as another (not eligible: fell through) is that other family's time
            by_kernel = {}
            for fam, r in res.items():
                if fam != "default":
                    by_kernel.setdefault(r["kernel"], []).append(r["us_per_step"])
            best_kernel, best = min(((k, min(v)) for k, v in by_kernel.items()), key=lambda kv: kv[1])
            over = res["default"]["us_per_step"] / best - 1.0
            # the CHOICE is what is held to the best family: the same kernel timed twice differs by run-to-run noise only
            # (2 us on a 13 us step); a different kernel that looks too slow is timed again, both sides, before it counts
            if res["default"]["kernel"] == best_kernel:
                over = min(over, 0.0)
            else:
                best_fam = min((f for f in res if f != "default" and res[f]["kernel"] == best_kernel), key=lambda f: res[f]["us_per_step"])
                for _ in range(2):
                    if over <= tol:
                        break
                    res["default"]["us_per_step"] = min(res["default"]["us_per_step"], round(time_step(data, n, FAMILIES["default"], steps)[0], 2))
                    best = min(best, round(time_step(data, n, FAMILIES[best_fam], steps)[0], 2))
                    over = res["default"]["us_per_step"] / best - 1.0
            worst = max(worst, over)
            cell = {"shape": sh, "rows": n, "nnz_per_row": round(data.nnz / data.n_rows, 1), "families": res, "best_kernel": best_kernel,
                    "best_us": best, "default_over_best": round(over, 3), "ok": over <= tol}
            cells.append(cell)
            print(json.dumps(cell), file=out, flush=True)
    print(json.dumps({"worst_default_over_best": round(worst, 3), "tolerance": tol, "ok": worst <= tol}), file=out, flush=True)
    return cells, worst


if __name__ == "__main__":
    quick = "--quick" in sys.argv
    table(rows=[2000, 80000] if quick else ROWS)
