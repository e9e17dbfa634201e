"""The reference-side patch (scala/patch/dsgd-hip-backend.diff) must keep applying: it is dry-run -- and then really
applied -- against a copy of the reference tree, so it cannot rot silently.  (No JVM here: whether it COMPILES is for a
maintainer with sbt; what this pins is that every hunk still finds its context, that the patch touches exactly the seams
INTEGRATION.md names, that the file it adds is scala/NativeSVM.scala byte for byte, and that every @native the patched
code can reach is exported by the JNI shim -- tests/test_jni_shim.py compares those two lists.)"""

import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference"
PATCH = os.path.join(ROOT, "scala", "patch", "dsgd-hip-backend.diff")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")


def test_patch_applies_to_the_reference_tree(tmp_path):
    tree = tmp_path / "ref"
    shutil.copytree(REF, tree, ignore=shutil.ignore_patterns(".git"))
    dry = subprocess.run(["patch", "-p1", "--dry-run", "-i", PATCH], cwd=tree, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert dry.returncode == 0 and "FAILED" not in dry.stdout and "fuzz" not in dry.stdout, dry.stdout
    real = subprocess.run(["patch", "-p1", "-i", PATCH], cwd=tree, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert real.returncode == 0, real.stdout
    touched = sorted(re.findall(r"patching file (\S+)", real.stdout))
    base = "src/main/scala/epfl/distributed/"
    assert touched == sorted(["src/main/resources/application.conf", base + "Main.scala", base + "utils/Config.scala",
                              base + "core/Slave.scala", base + "core/Master.scala", base + "core/MasterAsync.scala",
                              base + "core/ml/NativeSVM.scala"])
    added = (tree / base / "core/ml/NativeSVM.scala").read_text()
    assert added == open(os.path.join(ROOT, "scala", "NativeSVM.scala")).read()
    # the seams of INTEGRATION.md, each behind `dsgd.backend = hip` with the JVM path left as it was
    slave = (tree / base / "core/Slave.scala").read_text()
    master = (tree / base / "core/Master.scala").read_text()
    masync = (tree / base / "core/MasterAsync.scala").read_text()
    main = (tree / base / "Main.scala").read_text()
    conf = (tree / "src/main/resources/application.conf").read_text()
    assert "backend = jvm" in conf and "${?DSGD_BACKEND}" in conf
    assert 'backend: String = "jvm"' in (tree / base / "utils/Config.scala").read_text()
    assert "new HipSVM(config.lambda, dimSparsity, data, trainData.length)" in main and "new SparseSVM(config.lambda, dimSparsity)" in main
    for call in ("h.gradientBatch(w, samplesIdx)", "h.forwardBatch(w, samplesIdx)", "h.asyncStepBatch(sampleIdx, learningRate)",
                 "h.updateGrad(request.gradUpdate)", "h.setWeights(request.weights)"):
        assert call in slave, call
    assert "model.backward(w, x, y)" in slave and "model.regularize(grad, w)" in slave      # the JVM bodies are still there
    # the resident (dev) master hands a whole EPOCH over as one plan: the lists drawn first, in the reference's own order
    assert "h.fitEpoch(batches, learningRate)" in master and "worker.gradient(req)" in master
    assert "split.map(Random.shuffle(_)).toSeq.take(nWorkers).map(_.slice(batch, batch + batchSize))" in master
    assert master.count("split.map(Random.shuffle(_))") == 2      # ... and the per-request loop still draws them itself
    assert 'Kamon.timer("master.sync.batch.duration").record(perBatch)' in master
    assert "h.lossAndAccuracy(weights, lo, hi)" in master
    # resident (dev) mode: the loss check evaluates the device weights in place -- a snapshot written back would discard
    # the updates the slave threads applied since it was taken
    assert "NativeSVM.lossAcc(ctx, if (resident) null else DenseKeys.fromVec(w), rowBegin, rowEnd, out)" in added
    assert "HipSVM.isResident(model)" in masync and "_.update(request.gradUpdate)" in masync
    # every method the patched code calls on HipSVM exists in the file the patch adds
    used = set(re.findall(r"\bh\.(\w+)\(", slave + master + masync + main)) | {"weights"}
    defined = set(re.findall(r"def (\w+)\(", added)) | set(re.findall(r"var (\w+)\s*:", added))
    assert used <= defined | {"resident"}, used - defined


def test_patch_is_a_plain_unified_diff():
    text = open(PATCH).read()
    assert text.count("\ndiff -ruN ") + text.startswith("diff -ruN ") == 7
    assert "/root/" not in text and "/tmp/" not in text      # relative a/ b/ paths only: applies with -p1 anywhere
