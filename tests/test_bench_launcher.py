"""CPU: `python bench.py --gpus N` without a launcher spawns its own N ranks (one per GPU, torch.distributed.run's
environment) -- checked here as far as a machine without a GPU allows: both ranks start, rendezvous over gloo on
127.0.0.1, and fail LOUDLY for want of a gfx950 device (no CPU fallback), and the parent reports the failure."""

import os
import subprocess
import sys

import pytest

from conftest import ROOT, has_gpu


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode of the self-spawning launcher")
def test_self_spawned_ranks_fail_loudly_without_a_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                           "--rows", "1024"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert proc.returncode != 0
    assert not any(l.startswith("{") for l in proc.stdout.splitlines())   # no JSON line from a run that measured nothing
    for rank in (0, 1):
        assert "rank %d: no gfx950 device %d visible" % (rank, rank) in proc.stderr, proc.stderr[-2000:]


def test_launcher_environment_is_respected():
    """Under torch.distributed.run the script must NOT spawn again: WORLD_SIZE in the environment wins, and a mismatch
    with --gpus is an error before any device work."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=120)
    assert proc.returncode != 0 and "--gpus 2 but WORLD_SIZE=4" in proc.stderr
