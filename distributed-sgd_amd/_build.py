"""Build the native pieces in-tree (the .so files travel to the GPU box with the snapshot).

    libdsgd_hip.so    hipcc --offload-arch=gfx950   csrc/dsgd_hip.hip   (the product)
    libdsgd_synth.so  gcc -fopenmp                  csrc/synth.c        (synthetic RCV1-like data)
    libdsgd_rcv1.so   gcc                           csrc/rcv1.c         (RCV1-v2 text files -> CSR, Dataset.rcv1 semantics)
    libdsgd_host.so   gcc -pthread                  csrc/jrand.c        (java.util.Random / scala.util.Random.shuffle: the epoch's index lists)

hipcc cross-compiles gfx950 without a GPU, so this runs in the build container too.

The HIP runtime is referenced by its unversioned name (DT_NEEDED "libamdhip64.so"): a process
that already holds a HIP runtime under that name -- e.g. one that imported torch, whose wheel
bundles an un-SONAMEd libamdhip64.so -- keeps exactly ONE runtime; otherwise the loader finds
/opt/rocm/lib/libamdhip64.so through the RUNPATH.  Two HIP runtimes in one process would each
own separate streams and device state.
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")

HIP_SRC = os.path.join(CSRC, "dsgd_hip.hip")
HIP_LIB = os.path.join(LIBDIR, "libdsgd_hip.so")
SYNTH_SRC = os.path.join(CSRC, "synth.c")
SYNTH_LIB = os.path.join(LIBDIR, "libdsgd_synth.so")
RCV1_SRC = os.path.join(CSRC, "rcv1.c")
RCV1_LIB = os.path.join(LIBDIR, "libdsgd_rcv1.so")
JRAND_SRC = os.path.join(CSRC, "jrand.c")
JRAND_LIB = os.path.join(LIBDIR, "libdsgd_host.so")


def _stale(target: str, *sources: str) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd: list[str]) -> None:
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), proc.stdout))
    if proc.stdout.strip() and os.environ.get("DSGD_BUILD_VERBOSE"):
        print(proc.stdout)


def hipcc_path() -> str:
    for cand in (os.path.join(ROCM, "bin", "hipcc"), shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked in %s/bin and PATH)" % ROCM)


def build_hip(force: bool = False, out: str | None = None, defines: tuple = ()) -> str:
    """libdsgd_hip.so.  `out` / `defines`: a second build of the same sources under another name -- the tests' seam
    build (tests/rccl_stub/libdsgd_hip_seam.so, -DDSGD_TEST_COLLECTIVE_SEAM); the product is always built without."""
    header = os.path.join(HERE, "..", "include", "dsgd.h")
    target = out or HIP_LIB
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp"))   # every device header the .hip includes
    if not force and not _stale(target, HIP_SRC, *headers, header, __file__):
        return target
    os.makedirs(os.path.dirname(target), exist_ok=True)
    stub_dir = tempfile.mkdtemp(prefix="dsgd_stub_")
    stub = os.path.join(stub_dir, "libamdhip64.so")
    empty = os.path.join(stub_dir, "empty.c")
    with open(empty, "w") as f:
        f.write("/* link-time stand-in so that DT_NEEDED records the unversioned name */\n")
    _run(["gcc", "-shared", "-fPIC", "-o", stub, empty])
    cmd = [
        hipcc_path(),
        "--offload-arch=gfx950",
        "-O3",
        "-std=c++17",
        "-fPIC",
        "-shared",
        "-munsafe-fp-atomics",  # fp32 atomicAdd -> global_atomic_add_f32, not a CAS loop
        "-Wall",
        "-Wno-unused-function",
        *["-D" + d for d in defines],
        HIP_SRC,
        "-o",
        target,
        "-L" + stub_dir,
        "-Wl,-rpath," + os.path.join(ROCM, "lib"),
        "-ldl",
        "-lpthread",
    ]
    try:
        _run(cmd)
    finally:
        shutil.rmtree(stub_dir, ignore_errors=True)
    return target


def build_synth(force: bool = False) -> str:
    if not force and not _stale(SYNTH_LIB, SYNTH_SRC, __file__):
        return SYNTH_LIB
    os.makedirs(LIBDIR, exist_ok=True)
    _run(["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-std=c11", "-Wall", SYNTH_SRC, "-o", SYNTH_LIB, "-lm"])
    return SYNTH_LIB


def build_rcv1(force: bool = False) -> str:
    if not force and not _stale(RCV1_LIB, RCV1_SRC, __file__):
        return RCV1_LIB
    os.makedirs(LIBDIR, exist_ok=True)
    _run(["gcc", "-O2", "-fPIC", "-shared", "-std=gnu11", "-Wall", RCV1_SRC, "-o", RCV1_LIB])
    return RCV1_LIB


def build_host(force: bool = False) -> str:
    """libdsgd_host.so: the reference's random stream (java.util.Random + scala.util.Random.shuffle) for the host mirrors."""
    if not force and not _stale(JRAND_LIB, JRAND_SRC, __file__):
        return JRAND_LIB
    os.makedirs(LIBDIR, exist_ok=True)
    _run(["gcc", "-O2", "-pthread", "-fPIC", "-shared", "-std=gnu11", "-Wall", JRAND_SRC, "-o", JRAND_LIB])
    return JRAND_LIB


def build_all(force: bool = False) -> dict:
    return {"hip": build_hip(force), "synth": build_synth(force), "rcv1": build_rcv1(force), "host": build_host(force)}


if __name__ == "__main__":
    out = build_all(force="--force" in sys.argv)
    for k, v in out.items():
        print(k, v)
