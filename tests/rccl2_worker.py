"""One rank of tests/test_gpu_rccl_multi.py: REAL RCCL, one device per rank.  1,000 small synchronous steps of a resident
plan enqueued back to back WITHOUT any host synchronisation in between: the all-reduce of every step sits on the library's
stream between the reduce kernel in front of it and the update kernel behind it (csrc/dsgd_hip.hip finish_pre /
finish_collective / finish_post) -- an ordering bug would show as diverging replicas or a wrong mean.
usage: python rccl2_worker.py <rank> <world> <workdir>"""

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import dsgd_amd  # noqa: E402
from world2_common import shard_of  # noqa: E402

CFG = {"n_rows": 60000, "n_train": 48000, "seed": 21, "lam": 1e-5, "steps": 1000, "batch": 100, "workers": 2}


def lists_of(rank, n_train_local):
    rng = np.random.default_rng(7000 + rank)
    size = -(-n_train_local // CFG["workers"])
    return [[(min(j * size, n_train_local - 1) + rng.permutation(min(size, n_train_local - j * size))[:CFG["batch"]]).astype(np.int32)
             for j in range(CFG["workers"])] for _ in range(CFG["steps"])]


def main():
    rank, world, wd = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    assert not os.environ.get("DSGD_LIB_PATH"), "this test runs the PRODUCT library over real RCCL"
    data = dsgd_amd.synth.generate(CFG["n_rows"], seed=CFG["seed"])
    sh = shard_of(data, CFG["n_train"], rank, world)
    path = os.path.join(wd, "uid.bin")
    if rank == 0:
        uid = dsgd_amd.Engine.comm_unique_id()
        with open(path + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(path + ".tmp", path)
    else:
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > 120:
                raise RuntimeError("no unique id from rank 0")
            time.sleep(0.01)
        uid = open(path, "rb").read()
    with dsgd_amd.Engine(data.dim, CFG["lam"], device=rank) as eng:
        eng.load_csr(sh.csr.row_ptr, sh.csr.col, sh.csr.val, sh.csr.label)
        eng.comm_init(uid, world, rank)
        eng.build_dim_sparsity(sh.n_train)
        plan = eng.plan(lists_of(rank, sh.n_train))
        eng.plan_run(plan, 0, CFG["steps"], 0.5)      # 1,000 x (gradient, reduce, all-reduce, update): no host sync inside
        st = eng.synchronize()
        w_plan = eng.get_weights()
        plan.destroy()
        # the same first 50 lists once more as per-request steps from w = 0 (what the one-thread test repeats through the
        # grouped entry points: the same kernel path, so the comparison there is bit for bit)
        eng.set_weights(np.zeros(data.dim + 1, dtype=np.float32))
        for ls in lists_of(rank, sh.n_train)[:50]:
            eng.sync_step(ls, 0.5)
        np.savez(os.path.join(wd, "out_%d.npz" % rank), w=w_plan, stats=np.asarray([st["n_samples"], st["n_active"]]),
                 w_req=eng.get_weights())
    print("rank %d done" % rank, flush=True)


if __name__ == "__main__":
    main()
