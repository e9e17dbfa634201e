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


_RAMP_RANK = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
import bench
rank = int(os.environ["RANK"])
dist.init_process_group(backend="gloo", rank=rank, world_size=2)
n = [0]
def group():                      # rank 1 is slower and noisier: alone it would leave the ramp at a different point
    n[0] += 1
    time.sleep(0.002 if rank == 0 else 0.004 + 0.003 * (n[0] % 3))
    dist.barrier()                # stands for the all-reduce inside every step: unmatched calls would hang here
groups = bench.clock_ramp(group, 0.15, 2, dist)
alone = bench.clock_ramp(lambda: time.sleep(0.001), 0.02)
dist.barrier()
dist.destroy_process_group()
print("RAMP", rank, groups, n[0], alone)
"""


def test_ranks_leave_the_clock_ramp_together(tmp_path):
    """bench.py's clock ramp ends on a timing criterion; with N > 1 every ramp step carries an all-reduce, so ranks
    that left after different numbers of steps would leave unmatched collectives behind (a hang at the end of the
    run).  Two gloo ranks with different timings must report the same number of groups."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    script = tmp_path / "ramp_rank.py"
    script.write_text(_RAMP_RANK)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    lines = []
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
        lines.append([l for l in so.splitlines() if l.startswith("RAMP")][0].split())
    g0, g1 = int(lines[0][2]), int(lines[1][2])
    assert g0 == g1 and g0 == int(lines[0][3]) == int(lines[1][3]) and g0 >= 3, lines
    assert int(lines[0][4]) >= 3    # a single rank needs no process group


def test_committed_bench_line_keeps_the_contract():
    """The full object of the bench line the last GPU visit produced (profiles/r06_bench_detail.json): the keys, types and internal arithmetic of
    the driver's contract -- whole-job examples/s from the timed steps (median of the repeats), the dominant kernel's
    roofline fraction from its algorithmic bytes and its measured duration, a bounded CPU baseline, nothing quoted against
    a baseline that was never published -- and that every quoted configuration carries the parity of the kernel that
    produced it."""
    import json

    d = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_detail.json")))
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                     ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(d[key], typ), key
    assert d["unit"] == "examples/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None            # BASELINE.md publishes no number for this metric
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    # value = rows per step x GPUs / time per step; the time is the MEDIAN of >= 5 repeats of the timed K steps
    rows = d["config"]["train_rows_per_gpu"] * d["n_gpus"]
    assert abs(d["value"] - rows / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    rp = d["repeats"]
    assert rp["n"] >= 5 and rp["statistic"] == "median" and len(rp["ms_per_step"]) == rp["n"]
    assert sorted(rp["ms_per_step"])[rp["n"] // 2] == d["ms_per_step"] and rp["ms_per_step_min"] <= d["ms_per_step"] <= rp["ms_per_step_max"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = algorithmic bytes per launch / measured launch duration (HIP events inside the timed region)
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms_avg"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    assert r["kernel_launches"] == d["steps"] * rp["n"]
    assert r["traffic"] is None or 0.5 * r["algorithmic_bytes_per_launch"] < r["traffic"] < 1.5 * r["algorithmic_bytes_per_launch"]
    assert 0.0 < r["frac"] <= 1.0 and 0.0 < r["step"]["frac"] <= r["frac"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["unit"] == "examples/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # the parity gate ran on the benchmarked configuration: inside the derived bound AND the stated 1e-5 tolerance
    assert d["parity_gate_rows"] == d["config"]["train_rows_per_gpu"]
    for st in d["parity_gate"]["steps"]:
        assert st["worst_err_over_bound"] <= 1.0 and st["max_rel_err"] <= st["stated_tolerance"] == 1e-5
    # every sweep row: the parity step IS step 0 of the timed plan, so the checked kernel is the timed kernel
    assert {(s["workers"], s["batch"]) for s in d["sweep"]} >= {(1, 100), (3, 100), (4, 200), (1, 4096), (1, 65536)}
    for s in d["sweep"]:
        pz = s["parity"]
        assert pz["kernel"] == s["kernel"] and pz["checked"] == "step 0 of the timed plan"
        assert pz["worst_err_over_bound"] <= 1.0 and pz["max_rel_err"] <= pz["stated_tolerance"] == 1e-5
        assert abs(pz["n_active_engine"] - pz["n_active_oracle"]) <= pz["rows_near_gate"]
        assert 0.0 < s["frac_hbm_peak"] < 1.0
    by = {(s["workers"], s["batch"]): s for s in d["sweep"]}
    # the reference's own batch sizes run through the column-slice kernel, below 10 us per step from a resident plan
    for cfg in ((3, 100), (4, 200)):
        assert by[cfg]["kernel"] == "dsgd_cs_step_kernel" and by[cfg]["us_per_step"] < 10.0
    # the 256-worker lock-free run: traced replay (can fail: both negative controls rejected) 
    hw = d["hogwild"]
    tr = hw["traced_replay"]
    assert tr["workers"] == hw["workers"] == 256 and tr["batch"] == 100 and tr["accounting_agrees"] is True and tr["controls_rejected"] is True
    assert tr["gates_are"].startswith("engine-recorded")                 # (what the accounting replay does NOT re-derive ...
    gc = tr["gate_check"]                                                 #  ... and the statement that checks EVERY one of them, at 256 workers:
    assert gc["ok"] is True and gc["gate_rows_checked"] == gc["rows"] - gc["empty_rows"] == gc["updates"] * tr["batch"]   # rule on the recorded x . w,
    assert gc["rule_violations"] == 0 and gc["range_violations"] == 0 and gc["max_window"] >= 200                        # that x . w inside its range)
    assert all(v["rejected"] for v in gc["negative_controls"].values()) and len(gc["negative_controls"]) == 2
    g4 = tr["gate_check_4_workers"]                                       #  ... beside round 5's few-worker statement
    assert g4["ok"] is True and g4["updates_checked"] > 0 and g4["differ_at_both_ends"] <= g4["explained_by_in_flight_or_resolution"]
    assert tr["checkpoint"]["account_err_over_tol"] <= 1.0
    assert all(v["rejected"] for v in tr["negative_controls"].values()) and len(tr["negative_controls"]) >= 2
    assert hw["atomics_per_s"] > 0 and 0.0 < hw["frac_hbm_peak"] < 1.0
    # the reference's own data-set sizes next to the headline, gated and with roofline fields
    shapes = {rs["rows"]: rs for rs in d["reference_shapes"]}
    assert set(shapes) == {804414, 23149, 100552}   # (the last: one GPU of eight's share of RCV1, SplitStrategy.scala:13-14)
    for rs in shapes.values():
        assert rs["parity_gate"]["max_rel_err"] <= 1e-5 and rs["parity_gate"]["worst_err_over_bound"] <= 1.0
        assert rs["whole_shard"]["repeats"] >= 5 and 0.0 < rs["roofline"]["step"]["frac"] <= rs["roofline"]["frac"] <= 1.0
        assert all(s["parity"]["kernel"] == s["kernel"] for s in rs["sweep"])
    # the whole-split step at N = 23,149 runs as column lists (csrc/dsgd_tcol.hpp), at N = 804,414 as row chunks
    assert shapes[23149]["whole_shard"]["kernel"] == "dsgd_tc_grad_kernel" and shapes[804414]["whole_shard"]["kernel"] == "dsgd_fstep_kernel"
    assert shapes[100552]["whole_shard"]["kernel"] == "dsgd_fstep_kernel"      # (6 M non-zeros: beyond the column lists since round 6)
    # wall-clock to the oracle's target loss THROUGH host.MasterSync.fit (what a patched Master.fit runs), evaluation passes
    # inside the clock, for the batch sizes of SURVEY.md 8(d); the reference's configuration with its forced replay
    tt = d["time_to_target"]
    assert tt["through"].startswith("host.MasterSync.fit")
    assert {(c["workers"], c["batch"]) for c in tt["configs"]} >= {(3, 100), (4, 200), (1, 4096), (1, 65536)}
    assert tt["fastest"] is not None and tt["fastest"]["time_to_target_s"] > 0
    assert all("evaluation" in c and (c["time_to_target_s"] is None) == (c["engine_epochs"] is None) for c in tt["configs"])
    ref = {(c["workers"], c["batch"]): c for c in tt["configs"]}[(3, 100)]
    assert ref["divergent_rows_all_near_gate"] is True and ref["forced_replay_account_err_over_tol"] <= 1.0
    # an epoch as ONE plan: the fit's cost per 3 x 100 step, shuffle included, and its forced replay over 10 epochs
    fit = d["fit"]
    assert fit["kernel"] == ["dsgd_cs_step_kernel"] and 0.0 < fit["batch_loop_us_per_step"] < 15.0 < fit["per_request_us_per_step"]
    fr = fit["forced_replay_10_epochs"]
    assert fr["accounting_agrees"] is True and fr["account_err_over_tol"] <= 1.0 and fr["divergent_rows_all_near_gate"] is True
