"""Rank-less worker of tests/test_gpu_cs_device.py::test_the_exchange_gives_up_loudly_and_leaves_the_weights: runs on
the tests' seam build of the library (DSGD_LIB_PATH = tests/rccl_stub/libdsgd_hip_seam.so), the only build that has the
knob dsgd_test_cs_skip_publish."""

import ctypes as C
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import dsgd_amd  # noqa: E402
from dsgd_amd import _lib  # noqa: E402

assert os.environ.get("DSGD_LIB_PATH", "").endswith("libdsgd_hip_seam.so")
os.environ["DSGD_CS_REQ"] = "1"   # (per-request steps through the column-slice kernel are opt-in)
lib = _lib.load()
data = dsgd_amd.synth.generate(20000, seed=3)
n_train = 16000
rng = np.random.default_rng(1)
size = -(-n_train // 3)
steps = [[(j * size + rng.permutation(min(size, n_train - j * size))[:100]).astype(np.int32) for j in range(3)] for _ in range(6)]
w0 = np.zeros(data.dim + 1, dtype=np.float32)
hot = rng.choice(np.arange(1, data.dim + 1), size=5000, replace=False)
w0[hot] = rng.normal(scale=0.05, size=5000).astype(np.float32)
with dsgd_amd.Engine(data.dim, 1e-5) as eng:
    eng.load_csr(data.row_ptr, data.col, data.val, data.label)
    eng.build_dim_sparsity(n_train)
    plan = eng.plan(steps)
    # reference run
    eng.set_weights(w0)
    eng.plan_run(plan, 0, 6, 0.5)
    eng.synchronize()
    w_good = eng.get_weights()
    for from_step in (1, 3, 6):           # the first, a middle and the last step of the launch
        eng.set_weights(w0)
        eng.plan_run(plan, 0, 1, 0.0)     # (the weights slice-major, as between the launches of an epoch)
        eng.synchronize()
        assert lib.dsgd_test_cs_skip_publish(eng._ctx, C.c_int32(from_step)) == 0
        t0 = time.time()
        eng.plan_run(plan, 0, 6, 0.5)
        try:
            eng.synchronize()
            raise SystemExit("the launch with a silent slice was accepted (from step %d)" % from_step)
        except dsgd_amd.DsgdError as e:
            assert e.code == _lib.ESTATE and "exchange" in str(e), e
        assert time.time() - t0 < 60.0
        assert lib.dsgd_test_cs_skip_publish(eng._ctx, C.c_int32(0)) == 0
        assert np.array_equal(eng.get_weights(), w0), "weights moved by an aborted launch (from step %d)" % from_step
        eng.plan_run(plan, 0, 6, 0.5)     # the abort word is cleared: the next launch runs
        st = eng.synchronize()
        assert st["n_samples"] == 1800 and np.array_equal(eng.get_weights(), w_good)
    # a per-request step with a silent slice: the error comes back from the call itself, nothing applied
    eng.set_weights(w0)
    assert lib.dsgd_test_cs_skip_publish(eng._ctx, C.c_int32(1)) == 0
    try:
        eng.sync_step(steps[0], 0.5)
        raise SystemExit("the request with a silent slice was accepted")
    except dsgd_amd.DsgdError as e:
        assert e.code == _lib.ESTATE, e
    assert lib.dsgd_test_cs_skip_publish(eng._ctx, C.c_int32(0)) == 0
    assert np.array_equal(eng.get_weights(), w0)
    st = eng.sync_step(steps[0], 0.5)
    assert st["n_samples"] == 300 and eng.grad_kernel_name() == "dsgd_cs_request_kernel"
    plan.destroy()
print("CS_ABORT_OK")
