"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/dsgd.h declares; without a GPU the product fails loudly (no CPU fallback)."""

import ctypes as C
import os
import re

import pytest

import dsgd_amd
from dsgd_amd import _lib
from conftest import ROOT, has_gpu


def header_symbols():
    text = open(os.path.join(ROOT, "include", "dsgd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dsgd_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), "libdsgd_hip.so does not export %s" % name
    assert sorted(_lib.SYMBOLS) == names
    assert lib.dsgd_abi_version() == 1


def test_hip_runtime_is_referenced_unversioned():
    # one HIP runtime per process: DT_NEEDED must be the unversioned name (see _build.py)
    import subprocess

    out = subprocess.run(["readelf", "-d", _lib.HIP_LIB], stdout=subprocess.PIPE, text=True).stdout
    needed = re.findall(r"NEEDED.*\[(.*?)\]", out)
    assert "libamdhip64.so" in needed
    assert not any(n.startswith("librccl") for n in needed)  # RCCL is resolved lazily


def test_code_object_targets_gfx950_only():
    data = open(_lib.HIP_LIB, "rb").read()
    # offload-bundle entry ids name the ISA of every embedded code object
    # (plain arch strings also occur in rocPRIM's host-side dispatch tables, so look at the bundle ids only)
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", data))
    assert targets == {b"gfx950"}, targets
    assert b"sm_90" not in data and b"nvptx" not in data


def test_argument_errors_do_not_need_a_device():
    lib = _lib.load()
    ctx = C.c_void_p()
    assert lib.dsgd_create(None, C.byref(ctx)) == _lib.EINVAL
    cfg = _lib.Config(0, 0, 1e-5, 0, 0)
    assert lib.dsgd_create(C.byref(cfg), C.byref(ctx)) == _lib.EINVAL
    assert b"n_features" in lib.dsgd_last_error()
    assert lib.dsgd_get_weights(None, None) == _lib.EINVAL
    assert lib.dsgd_destroy(None) == _lib.OK


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback_without_gpu():
    assert dsgd_amd.device_count() == 0
    with pytest.raises(dsgd_amd.DsgdError) as ei:
        dsgd_amd.Engine(47236, 1e-5)
    assert ei.value.code == _lib.EUNSUPPORTED


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing in the package may import, link or dlopen it."""
    pkg_dir = os.path.join(ROOT, "distributed-sgd_amd")
    pat = re.compile(r"(from|import)\s+oracle|liboracle|\borc_[a-z]|oracle[/\\.](py|c|so)|ref_dict")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".c", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not pat.search(text), (dirpath, f, pat.search(text).group(0))
