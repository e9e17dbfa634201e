"""The test-only collective shim (tests/rccl_stub) checked on its own, without a GPU: two PROCESSES attach to one
communicator and all-reduce host buffers (DSGD_RCCL_STUB_HOSTMEM=1 makes the shim treat its buffers as host memory).
What libdsgd_hip does with it on a real device is tests/test_gpu_world2.py."""

import ctypes as C
import multiprocessing as mp
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "rccl_stub", "rccl_stub.cpp")
LIB = os.path.join(HERE, "rccl_stub", "librccl_stub.so")


def build_stub():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", SRC, "-o", LIB, "-ldl", "-lpthread", "-lrt"])
    return LIB


SEAM_LIB = os.path.join(HERE, "rccl_stub", "libdsgd_hip_seam.so")


def build_seam():
    """The TEST build of libdsgd_hip: the product's sources compiled a second time with -DDSGD_TEST_COLLECTIVE_SEAM, the only
    build in which DSGD_RCCL_LIB can put the shim in RCCL's place (csrc/dsgd_hip.hip rccl::load).  Lives next to the
    shim, never under distributed-sgd_amd/lib; a rank process selects it with DSGD_LIB_PATH."""
    import importlib
    import sys

    root = os.path.dirname(HERE)
    if root not in sys.path:
        sys.path.insert(0, root)
    pkg = importlib.import_module("distributed-sgd_amd")
    return pkg._build.build_hip(out=SEAM_LIB, defines=("DSGD_TEST_COLLECTIVE_SEAM",))


def seam_env(env=None):
    """Environment of a rank process that runs the seam build over the shim."""
    env = dict(os.environ if env is None else env)
    env["DSGD_RCCL_LIB"] = build_stub()
    env["DSGD_LIB_PATH"] = build_seam()
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    # the rank processes hold 24,000 .. 50,000 train rows each: keep their row ranges on the STREAMING kernels (the product
    # switches to them at 131,072 rows) -- streaming kernels + all-reduce is the path bench.py --gpus N times
    env.setdefault("DSGD_STREAM_MIN", "8192")
    env.setdefault("DSGD_TCOL", "0")   # (the column lists would take ranges of that size: tests/test_gpu_tcol.py has them)
    return env


def test_the_product_library_has_no_collective_seam():
    """The shipping library must not let an environment variable replace RCCL: the variable's name does not occur in it
    -- and does in the tests' seam build."""
    product = os.path.join(os.path.dirname(HERE), "distributed-sgd_amd", "lib", "libdsgd_hip.so")
    assert b"DSGD_RCCL_LIB" not in open(product, "rb").read()
    assert b"DSGD_RCCL_LIB" in open(build_seam(), "rb").read()


class Uid(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def _rank(rank, world, uid_bytes, q):
    os.environ["DSGD_RCCL_STUB_HOSTMEM"] = "1"
    lib = C.CDLL(LIB)
    uid = Uid()
    C.memmove(C.byref(uid), uid_bytes, 128)
    comm = C.c_void_p()
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
    lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    assert lib.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0
    out = {}
    rng = np.random.default_rng(rank)
    for it in range(20):   # many collectives back to back: the two barriers per call keep the slots consistent
        a = rng.normal(size=47237).astype(np.float32)
        r = np.empty_like(a)
        assert lib.ncclAllReduce(a.ctypes.data, r.ctypes.data, a.size, 7, 0, comm, None) == 0
        out["f%d" % it] = (a, r)
    u = (np.arange(1000, dtype=np.uint32) * (rank + 1)).copy()
    assert lib.ncclAllReduce(u.ctypes.data, u.ctypes.data, u.size, 3, 0, comm, None) == 0   # in place
    t = np.asarray([1, 2, 3, 10 ** 12 * (rank + 1)], dtype=np.int64)
    assert lib.ncclAllReduce(t.ctypes.data, t.ctypes.data, 4, 4, 0, comm, None) == 0
    big = np.zeros(300000, dtype=np.float32)
    assert lib.ncclAllReduce(big.ctypes.data, big.ctypes.data, big.size, 7, 0, comm, None) != 0   # beyond the slot: an error, not a hang
    lib.ncclGetErrorString.restype = C.c_char_p
    assert b"slot" in lib.ncclGetErrorString(2)
    assert lib.ncclCommDestroy(comm) == 0
    q.put((rank, {k: v for k, v in out.items()}, u, t))


def test_two_processes_allreduce_through_the_shim():
    build_stub()
    lib = C.CDLL(LIB)
    uid = Uid()
    assert lib.ncclGetUniqueId(C.byref(uid)) == 0
    uid_bytes = bytes(uid)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rank, args=(r, 2, uid_bytes, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = {}
    for _ in ps:
        rank, out, u, t = q.get(timeout=120)
        res[rank] = (out, u, t)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for it in range(20):
        a0, r0 = res[0][0]["f%d" % it]
        a1, r1 = res[1][0]["f%d" % it]
        np.testing.assert_array_equal(r0, r1)          # every rank receives the identical sum ...
        np.testing.assert_array_equal(r0, a0 + a1)     # ... added in rank order
    np.testing.assert_array_equal(res[0][1], np.arange(1000, dtype=np.uint32) * 3)
    np.testing.assert_array_equal(res[1][2], np.asarray([2, 4, 6, 3 * 10 ** 12], dtype=np.int64))


def _one_thread_two_ranks(q):
    """ONE thread drives both ranks (dsgd_*_devices): inside ncclGroupStart / ncclGroupEnd nothing may block."""
    os.environ["DSGD_RCCL_STUB_HOSTMEM"] = "1"
    lib = C.CDLL(LIB)
    uid = Uid()
    assert lib.ncclGetUniqueId(C.byref(uid)) == 0
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
    lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    comms = [C.c_void_p(), C.c_void_p()]
    assert lib.ncclGroupStart() == 0
    for r in range(2):
        assert lib.ncclCommInitRank(C.byref(comms[r]), 2, uid, r) == 0      # returns at once: its peer is this thread's next call
    assert lib.ncclGroupEnd() == 0
    rng = np.random.default_rng(0)
    ok = True
    for it in range(5):
        a = [rng.normal(size=47237).astype(np.float32) for _ in range(2)]
        out = [np.empty_like(a[0]) for _ in range(2)]
        assert lib.ncclGroupStart() == 0
        for r in range(2):
            assert lib.ncclAllReduce(a[r].ctypes.data, out[r].ctypes.data, a[r].size, 7, 0, comms[r], None) == 0
        assert lib.ncclGroupEnd() == 0
        ok = ok and np.array_equal(out[0], out[1]) and np.array_equal(out[0], a[0] + a[1])
    # two all-reduces on ONE communicator inside one group: refused (one slot per rank), and the group still ends
    assert lib.ncclGroupStart() == 0
    x = np.ones(4, dtype=np.float32)
    assert lib.ncclAllReduce(x.ctypes.data, x.ctypes.data, 4, 7, 0, comms[0], None) == 0
    assert lib.ncclAllReduce(x.ctypes.data, x.ctypes.data, 4, 7, 0, comms[0], None) != 0
    # (rank 1 joins so that the recorded one can complete)
    y = np.ones(4, dtype=np.float32)
    assert lib.ncclAllReduce(y.ctypes.data, y.ctypes.data, 4, 7, 0, comms[1], None) == 0
    assert lib.ncclGroupEnd() == 0 and x[0] == 2.0 and y[0] == 2.0
    assert lib.ncclGroupEnd() != 0    # unbalanced
    for cm in comms:
        assert lib.ncclCommDestroy(cm) == 0
    q.put(ok)


def test_one_thread_drives_two_ranks_inside_a_group():
    build_stub()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_thread_two_ranks, args=(q,))
    p.start()
    assert q.get(timeout=120) is True
    p.join(60)
    assert p.exitcode == 0
