"""One rank of tests/test_gpu_world2.py: a process of its own on device 0, its shard of the rows, the in-library
collective path through the test-only shim (DSGD_RCCL_LIB = tests/rccl_stub/librccl_stub.so, honoured only by the
tests' seam build of the library, DSGD_LIB_PATH = tests/rccl_stub/libdsgd_hip_seam.so).
usage: python world2_worker.py <rank> <world> <workdir>"""

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import dsgd_amd  # noqa: E402
from world2_common import CFG, dense_problem, local_lists, shard_of  # noqa: E402


def exchange_uid(wd, name, rank, make):
    path = os.path.join(wd, name)
    if rank == 0:
        uid = make()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(path + ".tmp", path)
        return uid
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > 120:
            raise RuntimeError("no unique id from rank 0")
        time.sleep(0.01)
    return open(path, "rb").read()


def main():
    rank, world, wd = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    assert os.environ.get("DSGD_RCCL_LIB"), "the worker must run with the shim selected explicitly"
    assert os.environ.get("DSGD_LIB_PATH", "").endswith("libdsgd_hip_seam.so"), "... through the tests' seam build of the library"
    data = dsgd_amd.synth.generate(CFG["n_rows"], seed=CFG["seed"])
    sh = shard_of(data, CFG["n_train"], rank, world)
    out = {}
    uid = exchange_uid(wd, "uid_sparse.bin", rank, dsgd_amd.Engine.comm_unique_id)
    with dsgd_amd.Engine(data.dim, CFG["lam"]) as eng:
        eng.load_csr(sh.csr.row_ptr, sh.csr.col, sh.csr.val, sh.csr.label)
        eng.comm_init(uid, world, rank)
        out["ds"] = eng.build_dim_sparsity(sh.n_train)          # feature counts all-reduced (Main.scala:57-60 counts the WHOLE train set)
        out["ranks"] = eng.column_ranks()                       # column counts all-reduced: one ranking for all replicas
        ntl = sh.n_train
        w_hist, shifts, stats = [], [], []
        # whole-range steps: one hosted worker per rank, then two
        for step, ranges in enumerate(([(0, ntl)], [(0, ntl)], [(0, ntl // 3), (ntl // 3, ntl)])):
            st = eng.sync_step_ranges(ranges, CFG["lr_range"] * len(ranges) * world)
            out["range_kernel"] = np.asarray(eng.grad_kernel_name())
            w_hist.append(eng.get_weights())
            shifts.append(eng.tuning_info()["fix_shift"])
            stats.append([st["n_samples"], st["n_active"]])
        # index-list steps (the plan kernel is not eligible with a communicator: a collective sits inside the step)
        for step, (k, b) in enumerate(CFG["list_steps"]):
            lists = local_lists(rank, step, k, b, ntl)
            st = eng.sync_step(lists, 0.5 * 100 / b)
            w_hist.append(eng.get_weights())
            shifts.append(eng.tuning_info()["fix_shift"])
            stats.append([st["n_samples"], st["n_active"]])
        out["w_hist"] = np.stack(w_hist)
        out["shifts"] = np.asarray(shifts)
        out["stats"] = np.asarray(stats)
        # evaluation: tallies summed over the ranks (three tallies + the row count)
        l_tr, a_tr, c_tr = eng.loss_acc(0, ntl)
        l_te, a_te, c_te = eng.loss_acc(ntl, sh.csr.n_rows)
        out["eval"] = np.asarray([l_tr, a_tr] + list(c_tr) + [l_te, a_te] + list(c_te), dtype=np.float64)
        # asynchronous mode across ranks: one deterministic worker per rank, exchange every E local updates
        eng.set_weights(np.zeros(data.dim + 1, dtype=np.float32))
        eng.async_set_exchange(CFG["exch_every"])
        b0, e0 = CFG["async_range"]
        t0 = time.perf_counter()
        eng.async_start([(b0, min(e0, ntl))], batch=CFG["async_batch"], lr=0.5, max_updates=CFG["async_updates"],
                        seed=CFG["async_seed"] + rank, positional_bug=False)
        t_start = time.perf_counter() - t0
        polls = 0
        while eng.async_updates()[1]:
            polls += 1
            time.sleep(0.001)
        eng.async_wait()
        u, running = eng.async_updates()
        assert u == CFG["async_updates"] and not running
        out["w_async"] = eng.get_weights()
        out["async_meta"] = np.asarray([t_start, polls], dtype=np.float64)
    # K8 with two ranks: the all-reduce of the D gradient sums
    uid2 = exchange_uid(wd, "uid_dense.bin", rank, dsgd_amd.Engine.comm_unique_id)
    X, y, n_steps, bsz = dense_problem()
    with dsgd_amd.DenseLogistic(X.shape[1]) as dl:
        mine = np.concatenate([np.arange(s * 2 * bsz + rank * bsz, s * 2 * bsz + (rank + 1) * bsz) for s in range(n_steps)])
        dl.load(X[mine], y[mine])
        dl.comm_init(uid2, world, rank)
        for s in range(n_steps):
            dl.step(s * bsz, (s + 1) * bsz, 0.5)
        dl.synchronize()
        out["w_dense"] = dl.get_weights()
    np.savez(os.path.join(wd, "out_%d.npz" % rank), **out)
    print("rank %d done" % rank, flush=True)


if __name__ == "__main__":
    main()
