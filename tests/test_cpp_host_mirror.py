"""include/dsgd.hpp (the C++ host mirror of SparseSVM / Slave / Master over the C ABI) compiled with g++ and driven by
tests/cpp/host_mirror_test.cpp: CPU checks here, the known-answer runs through the device under -m gpu."""

import json
import os
import subprocess

import numpy as np
import pytest

import dsgd_amd
from dsgd_amd import _lib, host
from conftest import ROOT, has_gpu


def build(tmp_path):
    exe = str(tmp_path / "host_mirror_test")
    libdir = os.path.dirname(_lib.HIP_LIB)
    _lib.load()  # make sure the library is built
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp"), "-o", exe, "-L", libdir, "-ldsgd_hip",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    subprocess.run(cmd, check=True)
    return exe


def test_cpp_mirror_compiles_and_passes_its_cpu_checks(tmp_path):
    r = subprocess.run([build(tmp_path), "cpu"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    assert "all checks passed" in r.stderr


@pytest.mark.gpu
@pytest.mark.skipif(not has_gpu(), reason="no gfx950 device")
def test_cpp_mirror_known_answers_and_fit_equal_the_python_mirror(tmp_path):
    r = subprocess.run([build(tmp_path), "gpu"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout.strip().splitlines()[-1])
    # the same fit through the Python mirror (same java.util.Random stream, same engine)
    from test_oracle_golden import KAT_ROWS

    data = dsgd_amd.synth.from_rows(6, KAT_ROWS)
    with dsgd_amd.Engine(6, 0.1) as eng:
        eng.load_csr(data.row_ptr, data.col, data.val, data.label)
        eng.build_dim_sparsity(4)
        m = host.MasterSync(eng, 4, 6, node_count=2, rnd=host.JavaRandom(0))
        s = m.fit(np.zeros(7), 2, 2, 0.25, host.EarlyStopping.no_improvement(5, 0.01))
    assert got["updates"] == s.updates == 2
    np.testing.assert_allclose(got["weights"], s.grad, rtol=0, atol=1e-7)
    np.testing.assert_allclose(got["losses"], m.losses, rtol=0, atol=1e-7)
    np.testing.assert_allclose(got["test_losses"], m.test_losses, rtol=0, atol=1e-7)
