"""tests/test_gpu_world2.py: ONE process, ONE thread, two contexts on device 0 -- the dsgd_*_devices entry points
(include/dsgd.h: the reference's dev role runs the master and every slave in one JVM, Main.scala:144-158) over the test-only
collective shim, which records the calls inside ncclGroupStart / ncclGroupEnd and runs them phase by phase.  The same
steps as the two rank PROCESSES of tests/world2_worker.py: the results must be theirs bit for bit.
usage: python devices_worker.py <workdir>"""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import dsgd_amd  # noqa: E402
from world2_common import CFG, local_lists, shard_of  # noqa: E402


def main():
    wd = sys.argv[1]
    assert os.environ.get("DSGD_RCCL_LIB") and os.environ.get("DSGD_LIB_PATH", "").endswith("libdsgd_hip_seam.so")
    world = 2
    data = dsgd_amd.synth.generate(CFG["n_rows"], seed=CFG["seed"])
    shards = [shard_of(data, CFG["n_train"], r, world) for r in range(world)]
    engines = [dsgd_amd.Engine(data.dim, CFG["lam"]) for _ in range(world)]
    out = {}
    try:
        for eng, sh in zip(engines, shards):
            eng.load_csr(sh.csr.row_ptr, sh.csr.col, sh.csr.val, sh.csr.label)
        grp = dsgd_amd.EngineGroup(engines)
        grp.comm_init_all()
        grp.build_dim_sparsity([sh.n_train for sh in shards])
        out["ranks"] = np.stack([e.column_ranks() for e in engines])
        ntl = [sh.n_train for sh in shards]
        w_hist, stats = [], []
        for ranges in ([(0, 1.0)], [(0, 1.0)], [(0, 1 / 3.0), (1 / 3.0, 1.0)]):
            per = [[(n // 3 if lo else 0, n // 3 if hi != 1.0 else n) for lo, hi in ranges] for n in ntl]
            st = grp.sync_step_ranges(per, CFG["lr_range"] * len(ranges) * world)
            w_hist.append(np.stack([e.get_weights() for e in engines]))
            stats.append([st["n_samples"], st["n_active"]])
        for step, (k, b) in enumerate(CFG["list_steps"]):
            st = grp.sync_step([local_lists(r, step, k, b, ntl[r]) for r in range(world)], 0.5 * 100 / b)
            w_hist.append(np.stack([e.get_weights() for e in engines]))
            stats.append([st["n_samples"], st["n_active"]])
        out["w_hist"] = np.stack(w_hist)        # [step, engine, D + 1]
        out["stats"] = np.asarray(stats)
        l_tr, a_tr, c_tr = grp.loss_acc([(0, n) for n in ntl])
        l_te, a_te, c_te = grp.loss_acc([(ntl[r], shards[r].csr.n_rows) for r in range(world)])
        out["eval"] = np.asarray([l_tr, a_tr] + list(c_tr) + [l_te, a_te] + list(c_te), dtype=np.float64)
        # a context of the group refuses the per-context call that would block this thread inside a collective? (it would
        # hang, not fail -- nothing to test here); what IS refused: a context twice, contexts without a communicator
        try:
            dsgd_amd.EngineGroup([engines[0], engines[0]]).sync_step_ranges([[(0, 10)], [(0, 10)]], 0.1)
            out["dup_refused"] = np.asarray([0])
        except dsgd_amd.DsgdInvalidArgument:
            out["dup_refused"] = np.asarray([1])
    finally:
        for e in engines:
            e.close()
    np.savez(os.path.join(wd, "devices.npz"), **out)
    print("devices worker done", flush=True)


if __name__ == "__main__":
    main()
