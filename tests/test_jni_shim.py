"""The JNI boundary (jni/dsgd_jni.cpp + scala/NativeSVM.scala) without a JDK: compile the shim against the stand-in
header tests/jni_stub/jni.h, compare its exported symbols with the Scala @native declarations (name mangling of a
Scala `object`: class NativeSVM$ -> `_00024`), and drive entry points with the stub's recording JNIEnv.

north_star: "Scala/Akka host code calls hand-written HIP kernels through a thin JNI C-ABI" behind
core/Slave.scala:129-157 and core/ml/SparseSVM.scala:11."""

import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, has_gpu
from dsgd_amd import _lib

SHIM = os.path.join(ROOT, "jni", "dsgd_jni.cpp")
SCALA = os.path.join(ROOT, "scala", "NativeSVM.scala")
PREFIX = "Java_epfl_distributed_core_ml_NativeSVM_00024_"

# Scala parameter type -> the JNI C type the shim must declare for it
JNI_TYPE = {"Int": "jint", "Long": "jlong", "Float": "jfloat", "Double": "jdouble", "Boolean": "jboolean",
            "Array[Long]": "jlongArray", "Array[Int]": "jintArray", "Array[Float]": "jfloatArray",
            "Array[Double]": "jdoubleArray", "Array[Byte]": "jbyteArray", "Array[Array[Int]]": "jobjectArray"}
JNI_RET = {"Unit": "void", "Long": "jlong", "Int": "jint"}


@pytest.fixture(scope="module")
def shim_lib(tmp_path_factory):
    _lib.load()  # libdsgd_hip.so is built by the session fixture
    out = str(tmp_path_factory.mktemp("jni") / "libdsgd_jni_check.so")
    libdir = os.path.dirname(_lib.HIP_LIB)
    cmd = ["g++", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter",
           "-I" + os.path.join(ROOT, "tests", "jni_stub"), "-I" + os.path.join(ROOT, "include"), SHIM, "-o", out,
           "-L" + libdir, "-l:libdsgd_hip.so", "-Wl,-rpath," + libdir]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert proc.returncode == 0, proc.stdout
    return out


def scala_natives():
    text = open(SCALA).read()
    body = text[text.index("object NativeSVM {"):]
    body = body[:body.index("\n}\n")]
    body = re.sub(r"//[^\n]*", "", body)
    out = {}
    for m in re.finditer(r"@native\s+def\s+(\w+)\s*\((.*?)\)\s*:\s*(\w+)", body, flags=re.S):
        params = [p.split(":", 1)[1].strip() for p in re.split(r",\s*(?![^\[]*\])", m.group(2).replace("\n", " ")) if p.strip()]
        out[m.group(1)] = (params, m.group(3))
    return out


def shim_signatures():
    text = open(SHIM).read()
    out = {}
    for m in re.finditer(r"JNIEXPORT\s+(\w+)\s+JNICALL\s+NATIVE\((\w+)\)\s*\((.*?)\)\s*\{", text, flags=re.S):
        params = [p.strip().split()[0] for p in m.group(3).replace("\n", " ").split(",")]
        out[m.group(2)] = (params, m.group(1))
    return out


def test_exported_symbols_are_the_scala_natives(shim_lib):
    nm = subprocess.run(["nm", "-D", "--defined-only", shim_lib], stdout=subprocess.PIPE, text=True).stdout
    exported = sorted(s for s in re.findall(r"\b(Java_\w+)", nm))
    natives = scala_natives()
    assert len(natives) >= 15
    assert exported == sorted(PREFIX + n for n in natives), (exported, sorted(natives))
    # no symbol binds to a Java-style static holder by accident (the round-1 bug: `..._NativeSVM_create`)
    assert not any(re.match(r"Java_epfl_distributed_core_ml_NativeSVM_[a-z]", s) for s in exported)


def test_signatures_match_parameter_by_parameter():
    natives, shim = scala_natives(), shim_signatures()
    assert sorted(natives) == sorted(shim)
    for name, (params, ret) in natives.items():
        c_params, c_ret = shim[name]
        assert c_params[0] == "JNIEnv*" and c_params[1] == "jobject", (name, c_params)  # instance method of NativeSVM$
        assert c_params[2:] == [JNI_TYPE[p] for p in params], (name, c_params, params)
        assert c_ret == JNI_RET[ret], (name, c_ret, ret)


def test_no_critical_regions_in_the_shim():
    text = re.sub(r"//[^\n]*", "", open(SHIM).read())
    assert "GetPrimitiveArrayCritical" not in text and "GetStringCritical" not in text


class JArray(C.Structure):
    _fields_ = [("length", C.c_int32), ("elem_size", C.c_int32), ("data", C.c_void_p)]


class Env(C.Structure):
    _fields_ = [("thrown_class", C.c_char * 128), ("thrown_message", C.c_char * 512), ("n_get", C.c_int),
                ("n_release", C.c_int), ("n_critical", C.c_int)]


def jarr(a):
    a = np.ascontiguousarray(a)
    return JArray(len(a), a.itemsize, a.ctypes.data_as(C.c_void_p)), a


def test_error_mapping_through_a_fake_env(shim_lib):
    lib = C.CDLL(shim_lib)
    env = Env()
    fn = getattr(lib, PREFIX + "create")
    fn.restype = C.c_int64
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_int32]
    h = fn(C.byref(env), None, 0, 1e-5, 0)  # n_features = 0: a `require` failure
    assert h == 0
    assert env.thrown_class == b"java/lang/IllegalArgumentException" and b"n_features" in env.thrown_message
    if not has_gpu():
        env = Env()
        assert fn(C.byref(env), None, 47236, 1e-5, 0) == 0  # no device: loud failure, no CPU fallback
        assert env.thrown_class == b"java/lang/RuntimeException"
    upd = getattr(lib, PREFIX + "updateGrad")
    upd.restype = None
    upd.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    env = Env()
    (k, _k), (v, _v) = jarr(np.arange(3, dtype=np.int32)), jarr(np.ones(2, dtype=np.float32))
    upd(C.byref(env), None, 0, C.byref(k), C.byref(v))
    assert env.thrown_class == b"java/lang/IllegalArgumentException" and env.n_get == 0  # lengths checked before any array is taken
    env = Env()
    (k, _k), (v, _v) = jarr(np.arange(3, dtype=np.int32)), jarr(np.ones(3, dtype=np.float32))
    upd(C.byref(env), None, 0, C.byref(k), C.byref(v))  # null context -> DSGD_EINVAL
    assert env.thrown_class == b"java/lang/IllegalArgumentException"
    assert env.n_get == 2 and env.n_release == 2 and env.n_critical == 0


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="no gfx950 device")
def test_shim_end_to_end_on_the_gpu(shim_lib):
    """create -> loadCsr -> buildDimSparsity -> syncStep -> getWeights through the JNI entry points equals the ctypes path."""
    import dsgd_amd

    lib = C.CDLL(shim_lib)
    data = dsgd_amd.synth.generate(2048, seed=3)
    env = Env()
    vp = C.c_void_p

    def call(name, restype, argtypes, *args):
        fn = getattr(lib, PREFIX + name)
        fn.restype, fn.argtypes = restype, [vp, vp] + argtypes
        r = fn(C.byref(env), None, *args)
        assert env.thrown_class == b"", (name, env.thrown_class, env.thrown_message)
        return r

    h = call("create", C.c_int64, [C.c_int32, C.c_double, C.c_int32], data.dim, 1e-5, 0)
    assert h != 0
    (rp, _1), (cl, _2), (vl, _3), (lb, _4) = jarr(data.row_ptr), jarr(data.col), jarr(data.val), jarr(data.label)
    call("loadCsr", None, [C.c_int64, vp, vp, vp, vp], h, C.byref(rp), C.byref(cl), C.byref(vl), C.byref(lb))
    call("buildDimSparsity", None, [C.c_int64, C.c_int64], h, 1600)
    rng = np.random.default_rng(0)
    lists = [rng.permutation(1600)[:100].astype(np.int32) for _ in range(3)]
    jl = [jarr(a) for a in lists]
    ptrs = (vp * 3)(*[C.cast(C.pointer(j[0]), vp) for j in jl])
    outer = JArray(3, 8, C.cast(ptrs, vp))
    n_active = call("syncStep", C.c_int64, [C.c_int64, vp, C.c_float], h, C.byref(outer), 0.5)
    w = np.zeros(data.dim + 1, dtype=np.float32)
    (wj, _w) = jarr(w)
    call("getWeights", None, [C.c_int64, vp], h, C.byref(wj))
    # an EPOCH as one plan through the shim (what the patched Master.fit calls: HipSVM.fitEpoch): 5 batches x 3 workers
    steps = [[rng.permutation(1600)[:100].astype(np.int32) for _ in range(3)] for _ in range(5)]
    flat = np.concatenate([a for st_ in steps for a in st_]).astype(np.int32)
    offs = np.arange(0, 1501, 100, dtype=np.int64)
    (fj, _f), (oj, _o) = jarr(flat), jarr(offs)
    plan = call("planCreate", C.c_int64, [C.c_int64, vp, vp, C.c_int32], h, C.byref(fj), C.byref(oj), 3)
    assert plan != 0
    call("planSynchronize", C.c_int64, [C.c_int64], h)   # (collects and clears the counters of the request above)
    call("planRun", None, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_float], h, plan, 0, 5, 0.5)
    act_epoch = call("planSynchronize", C.c_int64, [C.c_int64], h)
    call("planDestroy", None, [C.c_int64, C.c_int64], h, plan)
    w2 = np.zeros(data.dim + 1, dtype=np.float32)
    (w2j, _w2) = jarr(w2)
    call("getWeights", None, [C.c_int64, vp], h, C.byref(w2j))
    # offsets that do not describe nSteps x nWorkers lists are refused before anything is taken
    env_bad = Env()
    fnb = getattr(lib, PREFIX + "planCreate")
    fnb.restype, fnb.argtypes = C.c_int64, [vp, vp, C.c_int64, vp, vp, C.c_int32]
    assert fnb(C.byref(env_bad), None, h, C.byref(fj), C.byref(oj), 4) == 0
    assert env_bad.thrown_class == b"java/lang/IllegalArgumentException" and env_bad.n_get == 0
    # ... and offsets that end BEYOND the idx array (the lists would be read past the pinned JVM array)
    offs_long = offs.copy()
    offs_long[-1] += 7
    (olj, _ol) = jarr(offs_long)
    env_bad = Env()
    assert fnb(C.byref(env_bad), None, h, C.byref(fj), C.byref(olj), 3) == 0
    assert env_bad.thrown_class == b"java/lang/IllegalArgumentException", env_bad.thrown_class
    # ... and an epoch whose lists the DEVICE draws (HipSVM.fitEpochFromSeed): state = {generator state, batches out, draws out}
    from dsgd_amd import host
    split = host.split_vanilla(1600, 3)
    rnd = host.JavaRandom(0)
    state = np.asarray([rnd.seed, 0, 0], dtype=np.int64)
    sb = np.asarray([r.start for r in split], dtype=np.int64)
    se = np.asarray([r.stop for r in split], dtype=np.int64)
    (sj, _s), (bj, _b), (ej, _e) = jarr(state), jarr(sb), jarr(se)
    plan_s = call("planCreateFromSeed", C.c_int64, [C.c_int64, vp, vp, vp, C.c_int64, C.c_int32], h, C.byref(sj), C.byref(bj), C.byref(ej), 534, 100)
    idx_h, offs_h, n_h = host.epoch_lists(rnd, split, 534, 100)
    assert plan_s != 0 and _s[1] == n_h == 6 and _s[0] == rnd.seed and _s[2] == 6 * (533 + 533 + 531)
    call("planRun", None, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_float], h, plan_s, 0, int(_s[1]), 0.5)
    call("planSynchronize", C.c_int64, [C.c_int64], h)
    call("planDestroy", None, [C.c_int64, C.c_int64], h, plan_s)
    env_bad = Env()
    fns = getattr(lib, PREFIX + "planCreateFromSeed")
    fns.restype, fns.argtypes = C.c_int64, [vp, vp, C.c_int64, vp, vp, vp, C.c_int64, C.c_int32]
    assert fns(C.byref(env_bad), None, h, C.byref(sj), C.byref(bj), C.byref(ej), 534, 5000) == 0      # outside the device form
    assert env_bad.thrown_class == b"java/lang/UnsupportedOperationException", env_bad.thrown_class
    call("destroy", None, [C.c_int64], h)
    with dsgd_amd.Engine(data.dim, 1e-5) as eng:
        eng.load_csr(data.row_ptr, data.col, data.val, data.label)
        eng.build_dim_sparsity(1600)
        st = eng.sync_step(lists, 0.5)
        w_ref = eng.get_weights()
        p2 = eng.plan(steps)
        eng.synchronize()
        eng.plan_run(p2, 0, 5, 0.5)
        st2 = eng.synchronize()
        assert eng.grad_kernel_name() == "dsgd_cs_step_kernel"
        w_ref2 = eng.get_weights()
        p2.destroy()
    assert n_active == st["n_active"] == 300  # w = 0: every row is active
    assert np.abs(_w - w_ref).max() <= 1e-6 and np.abs(w_ref).max() > 0
    assert act_epoch == st2["n_active"] and np.array_equal(_w2, w_ref2)   # the same plan, the same kernel: the same bits
