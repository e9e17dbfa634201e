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


def _kernel_notes(tmp_path):
    """{kernel name: {metadata key: int}} of the gfx950 code object inside the shipped library."""
    import subprocess

    llvm = "/opt/rocm/lib/llvm/bin"
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "dev.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", _lib.HIP_LIB, fat])
    subprocess.check_call([llvm + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    notes = subprocess.run([llvm + "/llvm-readelf", "--notes", co], stdout=subprocess.PIPE, text=True, check=True).stdout
    out, cur = {}, {}
    for line in notes.splitlines():   # (a kernel's keys are sorted: .name sits in the middle of its block, "- " opens one)
        if re.match(r"\s+- \.", line):
            cur = {}
        m = re.search(r"\.(\w+):\s+(\S+)", line)
        if m:
            if m.group(1) == "name":
                out[m.group(2)] = cur
            elif m.group(2).isdigit():
                cur[m.group(1)] = int(m.group(2))
    return out


def test_column_slice_kernels_spill_nothing(tmp_path):
    """csrc/dsgd_cs.hpp keeps the NEXT step's slots in registers under the exchange.  A spilled register is reloaded with a
    vector-memory operation, and those retire in order BEHIND the prefetch: every reload waits the prefetch out (the
    round's first forms had 150-195 spills and ran 15.6 us per 3 x 100 step instead of 6.2).  512-lane shapes run two
    waves per SIMD: 256 registers each."""
    kn = {k: v for k, v in _kernel_notes(tmp_path).items() if "dsgd_cs_step_kernel" in k}
    assert len(kn) == 3, sorted(kn)
    for k, v in kn.items():
        # (the 80 bytes of private segment are the set-up's staging of the slice's dimSparsity piece: once per launch,
        #  in front of the first barrier -- the step loop touches no scratch)
        assert v["vgpr_spill_count"] == 0 and v["private_segment_fixed_size"] <= 80, (k, v)
        assert v.get("agpr_count", 0) == 0, (k, v)          # (values parked in accumulation registers need the value: a wait)
        lanes = int(re.search(r"ILi(\d+)E", k).group(1))
        assert v["vgpr_count"] <= (256 if lanes == 512 else 512), (k, v)


def test_column_list_kernels_spill_nothing_and_keep_two_workgroups_per_cu(tmp_path):
    """csrc/dsgd_tcol.hpp: every request of the dot kernel and of the gradient kernel is issued before the first is used
    -- a scratch reload would retire behind them.  The dot kernel's 1024-lane workgroups run two per CU (64 registers),
    the gradient kernel's 4-entry-piece form too; its 8-entry form (shares above 4,096 entries) one and a half."""
    kn = {k: v for k, v in _kernel_notes(tmp_path).items() if "dsgd_tc_" in k}
    names = sorted(kn)
    for want in ("dsgd_tc_dot_kernel", "dsgd_tc_grad_kernel", "dsgd_tc_count_kernel", "dsgd_tc_scan_kernel", "dsgd_tc_shares_kernel", "dsgd_tc_fill_kernel"):
        assert any(want in k for k in names), (want, names)
    for k, v in kn.items():
        assert v["vgpr_spill_count"] == 0 and v["private_segment_fixed_size"] == 0, (k, v)
        if "dsgd_tc_dot_kernel" in k or "dsgd_tc_grad_kernelILi1E" in k:
            assert v["vgpr_count"] <= 64, (k, v)
        if "dsgd_tc_grad_kernelILi2E" in k:
            assert v["vgpr_count"] <= 96, (k, v)


def test_register_budget_of_the_async_engine_and_the_concurrent_loss_check(tmp_path):
    """MasterAsync checks the loss WHILE the persistent lock-free engine runs (core/MasterAsync.scala:96-162): the
    evaluation kernel must become resident beside workgroups that never leave their CU.  Per SIMD the engine holds
    2 waves, the check 1 wave (256-lane blocks, csrc/dsgd_hip.hip dsgd_loss_acc): their ALLOCATED VGPRs (granule 8)
    must fit the 512-register file, and their LDS the 160 KiB of a CU."""
    import subprocess

    llvm = "/opt/rocm/lib/llvm/bin"
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "dev.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", _lib.HIP_LIB, fat])
    subprocess.check_call([llvm + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    notes = subprocess.run([llvm + "/llvm-readelf", "--notes", co], stdout=subprocess.PIPE, text=True, check=True).stdout
    vg = {}
    name = None
    for line in notes.splitlines():
        m = re.search(r"\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"\.vgpr_count:\s+(\d+)", line)
        if m and name:
            vg[name] = int(m.group(1))
    alloc = lambda n: -(-n // 8) * 8
    # (<true, .> is the tuning variant with phase counters, <., true> the traced form: parity runs, no concurrent check)
    hog = [v for k, v in vg.items() if "dsgd_hogwild_kernelILb0ELb0E" in k]
    evals = [v for k, v in vg.items() if "dsgd_eval_kernel" in k]
    assert len(hog) == 1 and len(evals) == 4
    assert 2 * alloc(hog[0]) + alloc(max(evals)) <= 512, (hog, evals)
    for k, v in vg.items():   # 1024-lane workgroups: 4 waves per SIMD
        if "dsgd_plan_kernel" in k or "dsgd_wseg_kernel" in k or "dsgd_fstep_kernel" in k:
            assert v <= 128, (k, v)
