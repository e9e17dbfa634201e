#!/usr/bin/env python3
"""bench.py -- RCV1 examples/s of the synchronous SGD hot path on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...            (no launcher: spawns its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one synchronous SGD step of the reference's Master.fit batch closure
(core/Master.scala:184-197) over one batch of synthetic RCV1-like rows: per-worker gated
sub-gradient sum + support-only regulariser (core/Slave.scala:142-157), mean over workers
(an RCCL all-reduce when N > 1), w <- w - lr * mean.  The batch is the worker's whole train
shard (batch-size >= split size), i.e. every step streams the resident CSR shard once --
the configuration in which the path is HBM-bound; a batch sweep including
the reference's default batch-size 100 is reported in "sweep".

Inputs are generated on the host, uploaded once and resident in HBM before the timed region.
Prints ONE JSON line (rank 0).
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12      # B/s, spec (MI355X_MICROARCH.md "Chip-level parameters")
HBM_MEASURED = 6.29e12  # B/s, float4-copy ceiling from the same table
LR0, LAMBDA = 0.5, 1e-5  # application.conf:18,21 (learning-rate is per batch of 100, application.conf:15)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=int(os.environ.get("DSGD_BENCH_ROWS", 8388608)),
                    help="rows per GPU.  8388608 (5.1 GB of CSR, BASELINE.md section 3) keeps the stream in HBM; "
                         "804414 = RCV1 full=true (DatasetTests.scala:18) fits mostly in the 256 MiB Infinity Cache")
    ap.add_argument("--workers", type=int, default=1, help="virtual workers (node-count share) per GPU")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-parity-gate", action="store_true", help="skip the whole-shard oracle comparison (profiling runs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget for each CPU baseline leg")
    ap.add_argument("--clock-ramp", type=float, default=0.5, help="seconds of untimed lr=0 steps before the warmup steps")
    return ap.parse_args()


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N copies of this command, one rank per GPU, with the
    environment torch.distributed.run would have set; rank 0 prints the JSON line."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # poll all ranks: a rank that dies before the rendezvous would leave the others waiting for it forever
    rc = 0
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            time.sleep(5.0)   # let the survivors report, then end exactly the processes started here
            for p in procs:
                if p.poll() is None:
                    p.terminate()
        time.sleep(0.2)
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def clock_ramp(run_group, nominal_s, world=1, dist=None):
    """Run groups of untimed steps until two groups in a row took within 1.5 % of the fastest group seen and at least
    `nominal_s` have passed, or 6 x `nominal_s` at the latest.  Returns the number of groups.  With more than one rank
    every step carries an all-reduce, so the ranks must leave after the SAME number of groups: they decide on the
    same numbers (the slowest rank's group time, the longest elapsed time), agreed through `dist`."""
    t_ramp = time.perf_counter()
    best, settled, groups = float("inf"), 0, 0
    while True:
        t_g = time.perf_counter()
        run_group()
        now = time.perf_counter()
        g, elapsed = now - t_g, now - t_ramp
        if world > 1:
            import torch

            t = torch.tensor([g, elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            g, elapsed = float(t[0].item()), float(t[1].item())
        groups += 1
        settled = settled + 1 if g <= 1.015 * best else 0
        best = min(best, g)
        if (elapsed >= nominal_s and settled >= 2) or elapsed >= 6.0 * nominal_s:
            return groups


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    # torch is plumbing only: rendezvous, barrier, the max-over-ranks reduction, cuda.synchronize
    import torch  # imported BEFORE libdsgd_hip so the process holds exactly one HIP runtime
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # one node: no need to resolve the container's hostname
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)

    import dsgd_amd

    if dsgd_amd.device_count() <= local_rank:
        raise SystemExit("rank %d: no gfx950 device %d visible (%d found): libdsgd_hip has no CPU fallback"
                         % (rank, local_rank, dsgd_amd.device_count()))

    def barrier():
        if world > 1:
            dist.barrier()

    def sync_all(eng):
        eng.synchronize()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    t_gen = time.time()
    data = dsgd_amd.synth.generate(args.rows, seed=args.seed, row0=rank * args.rows)
    t_gen = time.time() - t_gen
    n_train = int(args.rows * 0.8)  # Main.scala:52
    nnz_train = int(data.row_ptr[n_train])
    bytes_per_row = (8.0 * nnz_train + 12.0 * n_train) / n_train

    eng = dsgd_amd.Engine(data.dim, LAMBDA, device=local_rank)
    t_up = time.time()
    eng.load_csr(data.row_ptr, data.col, data.val, data.label)
    t_up = time.time() - t_up
    if world > 1:
        uid = [dsgd_amd.Engine.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        eng.comm_init(uid[0], world, rank)
    eng.build_dim_sparsity(n_train)

    k = args.workers
    # the reference SUMS the gated sub-gradients of a batch (core/Slave.scala:153), so its step length
    # scales with batch-size; keep the per-sample step of the defaults (0.5 per 100 samples)
    LR = LR0 * 100.0 / (n_train / k)
    size = -(-n_train // k)
    ranges = [(b, min(n_train, b + size)) for b in range(0, n_train, size)]  # SplitStrategy.vanilla

    # ---- parity gate on the BENCHMARKED configuration: the same whole-shard step, the same ranges, every row ----
    # Two steps of the timed configuration against the fp64 oracle (OpenMP restatement on the host cores), each from
    # identical weights, checked per coordinate against the DERIVED bound of oracle/bounds.py (fixed-point grid of
    # the shift this launch really uses + rows within 1e-5 of the gate) -- no blanket tolerance.  Must pass before any
    # timing is reported.
    parity = None
    if rank == 0 and world == 1 and not args.no_parity_gate:
        from oracle import bounds as orb  # checker only
        from oracle import oracle as orc

        t_gate = time.time()
        o = orc.Oracle(data.dim, data.row_ptr, data.col, data.val, data.label, LAMBDA)
        o.set_dim_sparsity(o.dim_sparsity(n_train))
        parity = {"rows": n_train, "ranges": len(ranges), "steps": []}
        for _ in range(2):
            w0 = eng.get_weights()
            w_ref = w0.astype(np.float64)
            st = eng.sync_step_ranges(ranges, LR)
            shift = eng.tuning_info()["fix_shift"]
            w_before = w_ref.copy()
            if len(ranges) == 1:
                act_ref = o.sync_step_range_omp(w_ref, 0, n_train, LR)
            else:
                o.sync_step(w_ref, [np.arange(a, b, dtype=np.int32) for a, b in ranges], LR)
                act_ref = o.last_stats["n_active"]
            tol, n_near = orb.step_bound(o, w_before, w_ref, ranges, LR, shift)
            w = eng.get_weights().astype(np.float64)
            ratio, j = orb.worst_ratio(w, w_ref, tol)
            rel = float(np.abs(w - w_ref).max()) / max(1.0, float(np.abs(w_ref).max()))
            parity["steps"].append({"fix_shift": shift, "n_active_engine": st["n_active"], "n_active_oracle": int(act_ref),
                                    "rows_near_gate": int(n_near), "max_rel_err": rel, "worst_err_over_bound": ratio,
                                    "worst_coordinate": j})
            if not ratio <= 1.0:
                raise SystemExit("parity gate failed on the benchmarked configuration: coordinate %d is %.3g x its "
                                 "derived bound (shift %d, max rel err %.3e)" % (j, ratio, shift, rel))
            if abs(st["n_active"] - act_ref) > n_near:
                raise SystemExit("parity gate failed: active rows %d vs oracle %d with only %d rows near the gate"
                                 % (st["n_active"], act_ref, n_near))
        parity["seconds"] = round(time.time() - t_gate, 1)
    eng.set_weights(np.zeros(data.dim + 1, dtype=np.float32))

    # ---- clock ramp (setup, not warmup): the parity gate and the layout leave the GPU idle for seconds while the host
    # works; the first milliseconds afterwards run at idle clocks (measured: 2.9 ms per step straight after the gate vs
    # 0.84 ms once the clocks are up -- 3 warmup steps are 3 ms, far less than the ramp).  Steps with lr = 0 (weights
    # unchanged) for a fixed wall time bring the clocks up; the W warmup steps and the K timed steps follow unchanged.
    # (some boxes of the pool need longer than others: the ramp goes on -- up to 6x the nominal time -- until two groups
    #  of steps in a row ran within 1.5 % of the fastest group seen)
    def ramp_group():
        for _ in range(8):
            eng.sync_step_ranges(ranges, 0.0, asynchronous=True)
        sync_all(eng)

    t_ramp = time.perf_counter()
    clock_ramp(ramp_group, args.clock_ramp, world, dist if world > 1 else None)
    ramp_s = time.perf_counter() - t_ramp
    eng.set_weights(np.zeros(data.dim + 1, dtype=np.float32))

    # ---- timed region ------------------------------------------------------------------------------
    for _ in range(args.warmup):
        eng.sync_step_ranges(ranges, LR, asynchronous=True)
    sync_all(eng)
    eng.prof_enable(2)   # HIP events around the DOMINANT kernel only: two event records per step inside the clock
    eng.prof_read(reset=True)
    barrier()
    sync_all(eng)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.sync_step_ranges(ranges, LR, asynchronous=True)
    sync_all(eng)
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms, n_launch = eng.prof_read(reset=True)
    # (outside the clock) the two cold-stream kernels, bracketed the same way over a few more steps
    eng.prof_enable(1)
    for _ in range(5):
        eng.sync_step_ranges(ranges, LR, asynchronous=True)
    sync_all(eng)
    kinds = eng.prof_read_kinds()
    eng.prof_read(reset=True)
    eng.prof_enable(0)
    last = eng.sync_step_ranges(ranges, 0.0)   # (outside the clock) gate statistics of the state the timed steps ended in
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    loss, acc, _ = eng.loss_acc(n_train, data.n_rows)
    value = world * n_train * args.steps / dt
    replicas_identical = None
    if world > 1:
        # every rank applied the same all-reduced gradient to the same weights: the replicas must agree BIT FOR BIT
        # (SURVEY.md 8(e)); compared through a 64-bit digest of the fp32 words
        import hashlib

        digest = int.from_bytes(hashlib.sha256(eng.get_weights().tobytes()).digest()[:7], "little")
        dmin = torch.tensor([digest], dtype=torch.int64)
        dmax = torch.tensor([digest], dtype=torch.int64)
        dist.all_reduce(dmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(dmax, op=dist.ReduceOp.MAX)
        replicas_identical = bool(dmin.item() == dmax.item())
        if not replicas_identical:
            raise SystemExit("rank %d: weight replicas diverged after the timed steps" % rank)

    out = {
        "metric": "RCV1 examples/sec (sync SGD, sparse hinge-SVM gradient step)",
        "value": value,
        "unit": "examples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "rcv1-synth sync SGD: %d rows/GPU (D=47236, nnz/row=%.1f), 80/20 split, whole-shard batch "
                        "B=%d rows/GPU/step, %d worker(s)/GPU, lr=0.5*100/B=%.3g, lambda=%g" %
                        (args.rows, data.nnz / data.n_rows, n_train, k, LR, LAMBDA),
            "rows_per_gpu": args.rows,
            "train_rows_per_gpu": n_train,
            "workers_per_gpu": k,
            "parallelism": "dp%d (row-partitioned, RCCL all-reduce of the %d-float gradient)" % (world, data.dim + 1),
            "seed": args.seed,
        },
        "active_fraction_after": last["n_active"] / max(1, last["n_samples"]),
        "test_loss_after": loss,
        "test_acc_after": acc,
        "setup_s": {"generate": round(t_gen, 2), "upload": round(t_up, 2), "clock_ramp": round(ramp_s, 2)},
    }
    if replicas_identical is not None:
        out["replicas_bit_identical"] = replicas_identical
    if parity is not None:
        out["parity_gate_rows"] = parity["rows"]
        out["parity_gate_max_rel_err"] = max(p["max_rel_err"] for p in parity["steps"])
        out["parity_gate"] = parity
    out["config"].update({"tuning": eng.tuning_info(),
                          "env_overrides": {k: v for k, v in os.environ.items() if k.startswith("DSGD_")}})

    # ---- roofline of the dominant kernel (the gradient kernel) ---------------------------------------
    # Algorithmic bytes: 8 B per non-zero + 12 B per row.  In the split layout the cold entries are
    # read by the two cold-stream kernels, so the main kernel is credited with the hot entries and the per-row bytes only.
    launches_per_step = n_launch / max(1, args.steps)
    nnz_int, cold_int = eng.range_nnz(0, n_train)
    alg_bytes = (8.0 * (nnz_train - cold_int) + 12.0 * n_train) / max(1.0, launches_per_step)  # per launch
    achieved = alg_bytes / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0
    step_s = dt / args.steps
    # HBM traffic of the dominant kernel comes from a SEPARATE rocprofv3 --pmc pass (tools/pmc_pass.sh: FETCH_SIZE,
    # x2 gfx950 correction) recorded in profiles/traffic.json -- it cannot be collected inside this run
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(str(args.rows))
            traffic_source = tj.get("source", "profiles/traffic.json (committed rocprofv3 --pmc FETCH_SIZE pass)")
        except Exception:
            traffic = None
    out["roofline"] = {
        "bound": "hbm",
        "kernel": eng.grad_kernel_name(),
        "achieved": achieved / 1e9,
        "peak": HBM_PEAK / 1e9,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK,
        "frac_of_measured_copy_peak": achieved / HBM_MEASURED,
        "traffic": traffic,
        "traffic_source": traffic_source,
        # the PHYSICAL fraction: HBM bytes the kernel really moved (rocprofv3 FETCH_SIZE pass) / its duration / peak --
        # below the algorithmic one because 16-bit column ranks make the stream smaller than 8 B per non-zero
        "physical_frac": (traffic / (kernel_ms * 1e-3) / HBM_PEAK) if (traffic and kernel_ms > 0) else None,
        "physical_frac_of_measured_copy_peak": (traffic / (kernel_ms * 1e-3) / HBM_MEASURED) if (traffic and kernel_ms > 0) else None,
        "algorithmic_bytes_per_launch": alg_bytes,
        "algorithmic_bytes_per_example": bytes_per_row,
        "kernel_ms_avg": kernel_ms,
        "kernel_launches": n_launch,
        "kernel_share_of_nonzeros": (nnz_train - cold_int) / max(1, nnz_train),
        # the whole step (all kernels, launch gaps included) against the same roofline
        "step": {"algorithmic_bytes": bytes_per_row * n_train, "ms": 1e3 * step_s,
                 "achieved": bytes_per_row * n_train / step_s / 1e9, "frac": bytes_per_row * n_train / step_s / HBM_PEAK},
        "other_kernels": {
            name: {"ms_avg": kinds[name][0], "launches": kinds[name][1],
                   "algorithmic_bytes_per_launch": 8.0 * cold_int / max(1.0, kinds[name][1] / 5.0),
                   "achieved": (8.0 * cold_int / max(1.0, kinds[name][1] / 5.0)) / (kinds[name][0] * 1e-3) / 1e9,
                   "measured": "5 untimed steps after the timed region"}
            for name in ("cdot", "cgrad") if kinds[name][1] > 0
        },
    }

    # ---- batch sweep incl. the reference's default batch-size (N=1 only) ------------------------------
    if rank == 0 and world == 1 and not args.no_sweep:
        out["sweep"] = sweep(eng, data, n_train, bytes_per_row, with_parity=not args.no_parity_gate)

    # ---- the other configurations of BASELINE.json, reported beside the headline (N=1 only) --------------------
    if rank == 0 and world == 1 and not args.no_sweep:
        out["eval_pass"] = eval_pass(eng, n_train, bytes_per_row)
        out["hogwild"] = hogwild(eng, n_train)
        if not args.no_parity_gate:
            out["hogwild"]["oracle_band"] = hogwild_band(dsgd_amd, local_rank)
        out["dense_logistic"] = dense_logistic(dsgd_amd, local_rank)

    # ---- CPU baseline on this box's host cores (rank 0, N=1 only) --------------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], out["cpu_literal"] = cpu_baseline(data, n_train, args.cpu_seconds)
        # second half of BASELINE.json's metric ("epochs to target hinge loss"); the target is defined by the oracle, so
        # it is computed in this leg and shown at the top level as well
        out["cpu_baseline"]["epochs_to_target"] = epochs_to_target(dsgd_amd, local_rank)
        e = out["cpu_baseline"]["epochs_to_target"]
        out["epochs_to_target"] = {k: e[k] for k in ("config", "target_test_loss", "engine_epochs", "oracle_epochs", "max_epochs")}

    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def sweep(eng, data, n_train, bytes_per_row, with_parity=True):
    """examples/s for index-list batches from resident plans: the reference's real defaults (3 workers x batch 100,
    application.conf:15,27; 4 x 200, kube/config-sync.yaml) and SURVEY.md 8(d)'s sweep B in {100, 200, 4096, 65536}.
    Every row carries `parity`: ONE step of that shape from non-zero weights against the fp64 oracle under the derived
    per-coordinate bound of oracle/bounds.py (outside the timed loop; the oracle is the checker, never the thing timed)."""
    LR = LR0
    res = []
    rng = np.random.default_rng(123)
    o = None
    if with_parity:
        from oracle import bounds as orb  # checker only
        from oracle import oracle as orc

        o = orc.Oracle(data.dim, data.row_ptr, data.col, data.val, data.label, LAMBDA)
        o.set_dim_sparsity(o.dim_sparsity(n_train))
        w_nz = np.zeros(eng.dp, dtype=np.float32)
        hot = rng.choice(np.arange(1, data.dim + 1), size=6000, replace=False)
        w_nz[hot] = rng.normal(scale=0.05, size=6000).astype(np.float32)
    for k, b, steps in ((1, 100, 300), (3, 100, 300), (4, 200, 200), (1, 200, 200), (1, 4096, 100), (1, 65536, 20)):
        if k * b > n_train:
            continue
        size = -(-n_train // k)
        split = [(a, min(n_train, a + size)) for a in range(0, n_train, size)]   # SplitStrategy.vanilla
        lists = [[(lo + rng.permutation(hi - lo)[:b]).astype(np.int32) for lo, hi in split] for _ in range(steps)]
        lr = LR * 100.0 / b   # the reference sums the batch (core/Slave.scala:153): keep the per-sample step of the defaults
        entry = {"workers": k, "batch": b, "steps": steps}
        if o is not None:
            eng.set_weights(w_nz)
            w0 = w_nz.astype(np.float64)
            w_ref = w0.copy()
            st = eng.sync_step(lists[0], lr)
            # (one hosted worker with lists of up to 192 rows runs the persistent one-workgroup kernel, which derives
            # its fixed-point shift per batch: 30 - ceil(log2 B); everything else reports the shift of its launch)
            kern = eng.grad_kernel_name()
            shift = 30 - int(np.ceil(np.log2(b))) if "plan_kernel" in kern else eng.tuning_info()["fix_shift"]
            o.sync_step(w_ref, lists[0], lr)
            tol, n_near = orb.list_bound(o, w0, w_ref, lists[0], lr, shift)
            ratio, j = orb.worst_ratio(eng.get_weights(), w_ref, tol)
            entry["parity"] = {"worst_err_over_bound": ratio, "fix_shift": shift, "rows_near_gate": int(n_near),
                               "n_active_engine": st["n_active"], "n_active_oracle": int(o.last_stats["n_active"]),
                               "max_abs_err": float(np.abs(eng.get_weights() - w_ref).max()), "kernel": kern}
            if not ratio <= 1.0 or abs(st["n_active"] - o.last_stats["n_active"]) > n_near:
                raise SystemExit("sweep parity failed for %d x %d: %r" % (k, b, entry["parity"]))
        eng.set_weights(np.zeros(eng.dp, dtype=np.float32))
        plan = eng.plan(lists)
        eng.plan_run(plan, 0, min(10, steps), lr)
        eng.synchronize()
        t0 = time.perf_counter()
        eng.plan_run(plan, 0, steps, lr)
        eng.synchronize()
        dt = time.perf_counter() - t0
        plan.destroy()
        nnz = float(np.mean([sum(int((data.row_ptr[l + 1] - data.row_ptr[l]).sum()) for l in st_) for st_ in lists[:8]]))
        alg = 8.0 * nnz + 12.0 * k * b
        entry.update({"examples_per_s": k * b * steps / dt, "us_per_step": 1e6 * dt / steps, "kernel": eng.grad_kernel_name(),
                      "algorithmic_bytes_per_step": alg, "frac_hbm_peak": alg * steps / dt / HBM_PEAK})
        res.append(entry)
    return res


def eval_pass(eng, n_train, bytes_per_row):
    """Master.localLoss/localAccuracy over the train rows (core/Master.scala:100-107): examples/s of one pass."""
    eng.loss_acc(0, n_train)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        eng.loss_acc(0, n_train)
    dt = (time.perf_counter() - t0) / reps
    return {"rows": n_train, "examples_per_s": n_train / dt, "ms": 1e3 * dt,
            "frac_hbm_peak": n_train / dt * bytes_per_row / HBM_PEAK, "note": "wall time incl. launch + readback"}


def hogwild(eng, n_train, workers=256, batch=100, updates=60000):
    """BASELINE.json configs[3]: asynchronous mode, one workgroup per worker, lock-free atomicAdd into ONE
    device-resident w (core/Slave.scala:79-111); reference defaults batch-size 100, learning-rate 0.5."""
    from dsgd_amd import host

    eng.set_weights(np.zeros(eng.dp, dtype=np.float32))
    split = [(r.start, r.stop) for r in host.split_vanilla(n_train, workers)]
    # a short untimed run first (clocks, first touch of the engine's buffers), then the timed one from w = 0;
    # wall time from async_start to the end of async_wait, i.e. launch and join included
    eng.async_start(split, batch=batch, lr=LR0, max_updates=updates // 4, seed=7, positional_bug=False)
    eng.async_wait()
    eng.set_weights(np.zeros(eng.dp, dtype=np.float32))
    t0 = time.perf_counter()
    eng.async_start(split, batch=batch, lr=LR0, max_updates=updates, seed=1, positional_bug=False)
    eng.async_wait()
    dt = time.perf_counter() - t0
    st = eng.async_stats()
    u = st["updates"]
    loss, acc, _ = eng.loss_acc(n_train, eng.n_rows)
    return {"workers": len(split), "batch": batch, "updates": int(u), "examples_per_s": u * batch / dt,
            "updates_per_s": u / dt, "ms": 1e3 * dt, "test_loss_after": loss, "test_acc_after": acc,
            # lane-level atomicAdd(w[j], -delta_j) counted on the device (the coordinates the updates
            # really moved); the memory system sees them coalesced per 128-byte line: profiles/ (TCP_TCC_ATOMIC_*)
            "atomics_per_s": st["atomics"] / dt, "atomics_per_update": st["atomics"] / max(1, u),
            "active_fraction": st["active"] / max(1, st["samples"]),
            "note": "one lock-free workgroup per worker on ONE device-resident w; wall time incl. launch and join; a single "
                    "end-of-run evaluation of a constant-step lock-free run fluctuates by several points (see oracle_band)"}


def hogwild_band(dsgd_amd, device, workers=256, batch=100, rows=100000, checkpoints=(2048, 4096, 6144, 8192), n_seeds=3):
    """Parity evidence for the BENCHMARKED Hogwild shape (256 workers x batch 100) on a shard small enough for the
    oracle: the band comes from the ORACLE (oracle/hogwild_band.py: sequential / stale-round / constant-delay replays of
    core/Slave.scala:92-101, several sampling seeds each; test loss and accuracy averaged over the second half of the
    checkpoints, |w| at the end); the engine runs in segments ending at the same checkpoints, three seeds."""
    from dsgd_amd import host
    from oracle import hogwild_band as hb  # checker only
    from oracle import oracle as orc

    data = dsgd_amd.synth.generate(rows, seed=13)
    n_train = int(rows * 0.8)
    o = orc.Oracle(data.dim, data.row_ptr, data.col, data.val, data.label, LAMBDA)
    o.set_dim_sparsity(o.dim_sparsity(n_train))
    split = [(r.start, r.stop) for r in host.split_vanilla(n_train, workers)]
    ev = (n_train, data.n_rows)
    t0 = time.perf_counter()
    band = hb.band(o, split, batch, list(checkpoints), LR0, ev, n_seeds=n_seeds)
    t_oracle = time.perf_counter() - t0
    runs = []
    with dsgd_amd.Engine(data.dim, LAMBDA, device=device) as eng:
        eng.load_csr(data.row_ptr, data.col, data.val, data.label)
        eng.build_dim_sparsity(n_train)
        for seed in (5, 6, 7):
            eng.set_weights(np.zeros(data.dim + 1, dtype=np.float32))
            curve, prev, total = [], 0, 0
            for c, target in enumerate(checkpoints):
                eng.async_start(split, batch=batch, lr=LR0, max_updates=target - prev, seed=seed + 7919 * c, positional_bug=False)
                eng.async_wait()
                total += eng.async_updates()[0]
                prev = target
                loss, acc, _ = eng.loss_acc(*ev)
                curve.append((total, loss, acc))
            summ = hb.summarise(curve, eng.get_weights().astype(np.float64))
            summ.update(inside=hb.inside(band, summ), updates=total, end_acc=curve[-1][2], end_loss=curve[-1][1])
            runs.append(summ)
    ok = all(all(r["inside"].values()) for r in runs)
    out = {"rows": rows, "workers": workers, "batch": batch, "checkpoints": list(checkpoints), "oracle_seconds": round(t_oracle, 1),
           "band": {q: {k: band[q][k] for k in ("lo", "hi", "oracle_min", "oracle_max", "by_mode")} for q in ("loss", "acc", "wnorm")},
           "margin": band["margin"], "oracle_end_of_run_acc_spread": band["end_of_run_acc_spread"],
           "engine_runs": runs, "inside": ok}
    if not ok:
        raise SystemExit("Hogwild parity failed: an engine run left the oracle's band: %r" % out)
    return out


def dense_logistic(dsgd_amd, device, rows=1250000, dim=4096):
    """BASELINE.json configs[4] on ONE GPU's share: 10 M x 4096 over 8 GPUs = 1.25 M rows (20.5 GB) per GPU, generated
    on the device; mini-batch steps of contiguous 4,096 / 65,536-row slices.  No reference counterpart (the optional
    dense variant of north_star); roofline 16,388 B per example."""
    res = {"rows": rows, "dim": dim, "bytes_per_example": 4 * dim + 4, "batches": [],
           "note": "dsgd_dense_step_kernel: row block read once, forward and gradient products from registers (v_fma: an fp32 "
                   "GEMV fills 1/16 of an MFMA at the vector rate -- csrc/dsgd_dense.hpp); no reference counterpart"}
    with dsgd_amd.DenseLogistic(dim, device=device) as eng:
        t0 = time.perf_counter()
        eng.generate(rows, seed=device)
        res["generate_s"] = round(time.perf_counter() - t0, 2)
        for b, steps in ((65536, 40), (4096, 200)):
            starts = [(i * b) % (rows - b) for i in range(steps + 5)]
            for st in starts[:5]:
                eng.step(st, st + b, 1.0)
            eng.synchronize()
            eng.prof(True)
            t0 = time.perf_counter()
            for st in starts[5:]:
                eng.step(st, st + b, 1.0)
            eng.synchronize()
            dt = time.perf_counter() - t0
            kms, kn = eng.prof(False)
            ex = b * steps / dt
            res["batches"].append({"batch": b, "steps": steps, "examples_per_s": ex, "us_per_step": 1e6 * dt / steps,
                                   "frac_hbm_peak": ex * (4 * dim + 4) / HBM_PEAK, "kernel_ms_avg": kms,
                                   "kernel_frac_hbm_peak": (b * (4 * dim + 4) / (kms * 1e-3) / HBM_PEAK) if kms > 0 else None})
        loss, acc = eng.loss(rows - 65536, rows)
        res["loss_after"], res["acc_after"] = loss, acc
    # the variant with the forward product on the matrix cores (configs[4]: "mini-batch GEMV via MFMA"), same data
    os.environ["DSGD_DENSE_MFMA"] = "1"
    try:
        with dsgd_amd.DenseLogistic(dim, device=device) as eng:
            eng.generate(rows, seed=device)
            b, steps = 65536, 40
            starts = [(i * b) % (rows - b) for i in range(steps + 5)]
            for st in starts[:5]:
                eng.step(st, st + b, 1.0)
            eng.synchronize()
            eng.prof(True)
            t0 = time.perf_counter()
            for st in starts[5:]:
                eng.step(st, st + b, 1.0)
            eng.synchronize()
            dt = time.perf_counter() - t0
            kms, _ = eng.prof(False)
            res["mfma_variant"] = {"batch": b, "steps": steps, "examples_per_s": b * steps / dt, "us_per_step": 1e6 * dt / steps,
                                   "kernel_ms_avg": kms,
                                   "kernel_frac_hbm_peak": (b * (4 * dim + 4) / (kms * 1e-3) / HBM_PEAK) if kms > 0 else None,
                                   "note": "dsgd_dense_step_mfma_kernel: v_mfma_f32_16x16x4_f32 forward product (1/16 of each "
                                           "instruction useful for a matrix-vector product), gradient product on the VALU"}
    finally:
        del os.environ["DSGD_DENSE_MFMA"]
    return res


def l3_bytes():
    """Total L3 of the host (all sockets), from sysfs; None if unknown."""
    try:
        seen, total = set(), 0
        base = "/sys/devices/system/cpu"
        for cpu in os.listdir(base):
            p = os.path.join(base, cpu, "cache", "index3")
            if not os.path.isdir(p):
                continue
            key = open(os.path.join(p, "shared_cpu_list")).read().strip()
            if key in seen:
                continue
            seen.add(key)
            sz = open(os.path.join(p, "size")).read().strip()
            total += int(sz[:-1]) * (1024 if sz.endswith("K") else 1024 * 1024) if sz[-1] in "KM" else int(sz)
        return total or None
    except (OSError, ValueError):
        return None


def cpu_baseline(data, n_train, budget_s):
    LR = LR0
    """The oracle timed on the host cores: (B) OpenMP CSR restatement on all cores, same whole-shard
    step on a bounded sample; (A) literal per-sample sparse-map restatement, one thread, B=100."""
    from oracle import oracle as orc

    n_s = min(n_train, 2000000)   # 1.2 GB of CSR: beyond the host's L3 (the 200,000-row sample of round 2 was cache-resident)
    sub = data.rows(0, n_s)
    o = orc.Oracle(sub.dim, sub.row_ptr, sub.col, sub.val, sub.label, LAMBDA)
    o.set_dim_sparsity(o.dim_sparsity(n_s))
    w = np.zeros(sub.dim + 1)
    o.sync_step_range_omp(w, 0, n_s, LR)  # warm
    t0, steps = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s and steps < 200:
        o.sync_step_range_omp(w, 0, n_s, LR)
        steps += 1
    dt = time.perf_counter() - t0
    fast = {
        "value": n_s * steps / dt,
        "unit": "examples/s",
        "cores": orc.num_threads(),
        "kind": "port",
        "sample": "OpenMP CSR fp64 restatement (oracle.c orc_sync_step_range_omp): %d whole-shard steps over the "
                  "first %d train rows of the same workload, %.1f s" % (steps, n_s, dt),
        "host_cpus": os.cpu_count(),
        "sample_bytes": int(8 * sub.nnz + 12 * n_s),
        "sample_exceeds_l3": bool(8 * sub.nnz + 12 * n_s > l3_bytes()) if l3_bytes() else None,
        "host_l3_bytes": l3_bytes(),
    }
    # literal: Slave.gradient as one single-threaded request of 100 samples (core/Slave.scala:142)
    rng = np.random.default_rng(5)
    w = np.zeros(sub.dim + 1)
    t0, steps = time.perf_counter(), 0
    while time.perf_counter() - t0 < min(budget_s, 6.0):
        o.sync_step(w, [rng.permutation(n_s)[:100].astype(np.int32)], LR, literal=True)
        steps += 1
    dt = time.perf_counter() - t0
    lit = {
        "value": 100 * steps / dt,
        "unit": "examples/s",
        "cores": 1,
        "kind": "port",
        "sample": "literal per-sample sparse-vector restatement (oracle.c orc_lit_sync_step), 1 worker, "
                  "batch-size 100, %d steps, %.1f s" % (steps, dt),
    }
    return fast, lit


def epochs_to_target(dsgd_amd, device):
    """Second half of BASELINE.json's metric on configs[1]'s shape (full=false: 23,149 rows, 80/20 split, 3 workers,
    batch-size 100, lr 0.5, lambda 1e-5, max-epochs 10 -- application.conf): the target is the ORACLE's test loss
    after the 10 epochs; reported: epochs until the engine's test loss is at or below it, and both loss curves.
    Both sides get the same per-batch index lists (one fresh shuffle of every split per batch, Master.scala:184)."""
    from oracle import oracle as orc  # checker / CPU baseline only

    n_rows, k, batch, lr, epochs = 23149, 3, 100, LR0, 10
    n_train = int(n_rows * 0.8)
    data = dsgd_amd.synth.generate(n_rows, seed=0)
    size = -(-n_train // k)
    split = [np.arange(b, min(n_train, b + size)) for b in range(0, n_train, size)]
    rng = np.random.default_rng(0)
    lists = [[[rng.permutation(sp)[b:b + batch].astype(np.int32) for sp in split] for b in range(0, size, batch)]
             for _ in range(epochs)]
    o = orc.Oracle(data.dim, data.row_ptr, data.col, data.val, data.label, LAMBDA)
    o.set_dim_sparsity(o.dim_sparsity(n_train))
    w = np.zeros(data.dim + 1)
    ref_curve = []
    t0 = time.perf_counter()
    for ep in lists:
        for step in ep:
            o.sync_step(w, step, lr)
        ref_curve.append(o.loss_acc(w, n_train, n_rows)[0])
    t_ref = time.perf_counter() - t0
    curve = []
    with dsgd_amd.Engine(data.dim, LAMBDA, device=device) as eng:
        eng.load_csr(data.row_ptr, data.col, data.val, data.label)
        eng.build_dim_sparsity(n_train)
        eng.sync_step(lists[0][0], 0.0)  # layout + first launches outside the clock
        eng.set_weights(np.zeros(data.dim + 1, dtype=np.float32))
        t0 = time.perf_counter()
        for ep in lists:
            for step in ep:
                eng.sync_step(step, lr)
            curve.append(eng.loss_acc(n_train, n_rows)[0])
        t_eng = time.perf_counter() - t0
    # Target: the BEST test loss the oracle reaches within max-epochs (the curve of a constant-step run is noisy: its
    # last-epoch loss is worse than its epoch-2 loss, which made "epochs to the last-epoch loss" trivially 2 for both
    # sides in round 2).  The hinge part moves in steps of 1/n_test (predictions are -1/0/+1): one test row of slack.
    target = min(ref_curve)
    slack = 1.0 / (n_rows - n_train)
    reached = next((i + 1 for i, l in enumerate(curve) if l <= target + slack), None)
    reached_ref = next((i + 1 for i, l in enumerate(ref_curve) if l <= target + slack), None)
    half = 0.5 * 1.0   # loss at w = 0 is exactly 1 (every prediction is 0): first epoch at or below half of it
    return {"config": "23149 rows, 3 workers x batch 100, lr 0.5, lambda 1e-5, 10 epochs", "target_test_loss": target,
            "target": "min over the oracle's epochs", "slack": slack,
            "epochs_to_half_initial_loss": {"engine": next((i + 1 for i, l in enumerate(curve) if l <= half), None),
                                            "oracle": next((i + 1 for i, l in enumerate(ref_curve) if l <= half), None)},
            "max_curve_difference": float(np.abs(np.asarray(curve) - np.asarray(ref_curve)).max()),
            "max_epochs": epochs, "engine_epochs": reached, "oracle_epochs": reached_ref,
            "engine_test_loss": curve, "oracle_test_loss": ref_curve,
            "engine_s": t_eng, "oracle_s": t_ref, "steps_per_epoch": len(lists[0])}


if __name__ == "__main__":
    main()
