"""ctypes binding of include/dsgd.h -- the same symbols the JNI shim binds.

There is deliberately NO CPU fallback here: if libdsgd_hip.so is missing or no gfx950 device is
visible, calls fail loudly (DsgdError / OSError).
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB = os.environ.get("DSGD_LIB_PATH") or os.path.join(HERE, "lib", "libdsgd_hip.so")  # (override: A/B runs of two builds)

OK, EINVAL, ERANGE, ESTATE, EHIP, ERCCL, ENOMEM, EUNSUPPORTED = 0, -1, -2, -3, -4, -5, -6, -7
UNIQUE_ID_BYTES = 128

# every symbol include/dsgd.h declares (tests/test_abi.py checks the library exports them all)
SYMBOLS = [
    "dsgd_abi_version", "dsgd_last_error", "dsgd_device_count", "dsgd_create", "dsgd_destroy", "dsgd_load_csr",
    "dsgd_n_rows", "dsgd_set_dim_sparsity", "dsgd_build_dim_sparsity", "dsgd_set_weights", "dsgd_get_weights",
    "dsgd_gradient", "dsgd_apply", "dsgd_sync_step", "dsgd_sync_step_ranges", "dsgd_plan_create", "dsgd_plan_create_n", "dsgd_plan_create_from_seed", "dsgd_plan_read_lists", "dsgd_cache_trim",
    "dsgd_plan_destroy", "dsgd_plan_run", "dsgd_plan_info", "dsgd_plan_record", "dsgd_plan_read_record", "dsgd_sync_step_ranges_async", "dsgd_synchronize", "dsgd_forward",
    "dsgd_loss_acc", "dsgd_async_step", "dsgd_update_grad", "dsgd_async_start", "dsgd_async_updates",
    "dsgd_async_stop", "dsgd_async_wait", "dsgd_comm_unique_id", "dsgd_comm_init", "dsgd_comm_destroy",
    "dsgd_prof_enable", "dsgd_prof_read", "dsgd_prof_read_kinds", "dsgd_range_nnz", "dsgd_grad_kernel_name",
    "dsgd_device_ptrs", "dsgd_async_set_exchange", "dsgd_async_stats", "dsgd_async_set_trace", "dsgd_async_read_trace", "dsgd_async_read_trace_dots", "dsgd_tuning_info", "dsgd_column_ranks", "dsgd_debug_cycles",
    "dsgd_comm_init_all", "dsgd_build_dim_sparsity_devices", "dsgd_sync_step_devices", "dsgd_sync_step_ranges_devices",
    "dsgd_loss_acc_devices",
    "dsgd_dense_create", "dsgd_dense_destroy", "dsgd_dense_generate", "dsgd_dense_load", "dsgd_dense_set_weights",
    "dsgd_dense_get_weights", "dsgd_dense_step", "dsgd_dense_synchronize", "dsgd_dense_loss", "dsgd_dense_comm_init",
    "dsgd_dense_prof",
]


class Config(C.Structure):
    _fields_ = [
        ("n_features", C.c_int32),
        ("device", C.c_int32),
        ("lambda_", C.c_double),
        ("flags", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class BatchStats(C.Structure):
    _fields_ = [("n_samples", C.c_int64), ("n_active", C.c_int64)]


class DsgdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("dsgd error %d: %s" % (code, msg))
        self.code = code


class DsgdInvalidArgument(DsgdError, ValueError):
    """DSGD_EINVAL -- what the reference's `require` failures surface as (IllegalArgumentException)."""


class DsgdIndexError(DsgdError, IndexError):
    """DSGD_ERANGE -- IndexOutOfBoundsException on the reference side."""


_lib = None


def load():
    """dlopen libdsgd_hip.so (building it is __graft_entry__.build()'s job)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(HIP_LIB):
        raise OSError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % HIP_LIB)
    lib = C.CDLL(HIP_LIB)
    lib.dsgd_last_error.restype = C.c_char_p
    lib.dsgd_grad_kernel_name.restype = C.c_char_p
    lib.dsgd_grad_kernel_name.argtypes = [C.c_void_p]
    for name in SYMBOLS:
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if os.environ.get("DSGD_LIB_PATH"):   # an A/B run against an OLDER build: entry points added since are absent
                continue
            raise
        if name not in ("dsgd_last_error", "dsgd_grad_kernel_name"):
            fn.restype = C.c_int
    _lib = lib
    return lib


def check(rc):
    if rc == OK:
        return
    msg = load().dsgd_last_error().decode("utf-8", "replace")
    if rc == EINVAL:
        raise DsgdInvalidArgument(rc, msg)
    if rc == ERANGE:
        raise DsgdIndexError(rc, msg)
    raise DsgdError(rc, msg)


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f32(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if n is not None and a.shape != (n,):
        raise ValueError("expected %d floats, got shape %r" % (n, a.shape))
    return a


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)
