#!/usr/bin/env python3
"""host.MasterSync.fit at the reference's configuration (full = true shape, 3 workers x batch 100, application.conf:15,27),
two epochs without evaluation passes in the clock: what the batch loop costs per step with the epoch's lists drawn by the
device (Engine.plan_from_seed) and by the host (csrc/jrand.c; DSGD_DEVICE_LISTS=0)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dsgd_amd  # noqa: E402
from dsgd_amd import host  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 804414
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
data = dsgd_amd.synth.generate(rows, seed=0)
n_train = int(rows * 0.8)
out = []
with dsgd_amd.Engine(data.dim, 1e-5) as eng:
    eng.load_csr(data.row_ptr, data.col, data.val, data.label)
    eng.build_dim_sparsity(n_train)
    ws = {}
    for device_lists in (True, False, True):
        os.environ["DSGD_DEVICE_LISTS"] = "1" if device_lists else "0"
        m = host.MasterSync(eng, n_train, rows, 3, rnd=host.JavaRandom(0))
        m.local_loss = lambda test=False: 0.0          # (the evaluation passes are not what is being timed)
        m.local_accuracy = lambda test=False: 0.0
        t0 = time.perf_counter()
        m.fit(np.zeros(data.dim + 1, dtype=np.float32), epochs, 100, 0.5, lambda losses: False)
        dt = time.perf_counter() - t0
        ws[device_lists] = eng.get_weights()
        row = {"rows": rows, "lists": "device" if m.device_lists else "host", "epochs": epochs, "steps": m.steps_run, "fit_s": dt,
               "batch_loop_us_per_step": 1e6 * m.batch_loop_s / max(1, m.steps_run),
               "lists_us_per_step_not_hidden": 1e6 * m.shuffle_s / max(1, m.steps_run), "generator_state": m.rnd.seed}
        out.append(row)
        print(json.dumps(row), flush=True)
    print("same weights, same generator state either way:", bool(np.array_equal(ws[True], ws[False])), out[0]["generator_state"] == out[1]["generator_state"])
