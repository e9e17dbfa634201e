#!/usr/bin/env python3
"""What the JVM reference should log on a small exported workload -- the only route to pinning the oracle without a
JVM in this image.  Test infrastructure: runs the Main.scala scenario (Main.scala:32-118) with
the fp64 CPU oracle behind the host mirror (host.MasterSync with the java.util.Random / scala.util.Random.shuffle
stream of seed 0) and writes

    tests/golden/jvm_expected/data/            the workload as RCV1-v2 text (lyrl2004_vectors_train.dat, qrels, ...)
    tests/golden/jvm_expected/expected.json    initial loss / accuracy, test loss per epoch, final test loss /
                                               accuracy, and the final weights as {key: fp64 value}

A maintainer with a JVM diffs them against the log of

    cd <reference>; ln -s <repo>/tests/golden/jvm_expected/data data
    DSGD_NODE_COUNT=3 DSGD_BATCH_SIZE=100 DSGD_LEARNING_RATE=0.5 DSGD_LAMBDA=1e-5 DSGD_MAX_EPOCHS=3 DSGD_FULL=false sbt run

("initial loss", "loss after epoch", "final weights: idx:value ...", "final test loss / accuracy", Main.scala:72-117).
Agreement pins oracle.c / ref_dict.py on backward, regularize, forward, loss and the batch closure -- everything the
reference's own tests leave unpinned; a disagreement in the batch ORDER only (weights close, not equal) would point at
the shuffle mirror instead.  tests/test_oracle_golden.py regenerates expected.json and compares it with the committed
file, so the oracle cannot drift silently.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ROWS, NODES, BATCH, LR, LAM, EPOCHS = 400, 3, 100, 0.5, 1e-5, 3


def compute(write_data=False):
    import dsgd_amd
    from dsgd_amd import host, rcv1
    from oracle import oracle as orc
    from oracle_backend import OracleBackend

    out = os.path.join(HERE, "jvm_expected")
    data = dsgd_amd.synth.generate(ROWS, seed=0)
    if write_data:
        rcv1.export(os.path.join(out, "data"), data, n_train_file=ROWS)
    n_train = int(ROWS * 0.8)                                   # Main.scala:52
    o = orc.Oracle(data.dim, data.row_ptr, data.col, data.val, data.label, LAM)
    o.set_dim_sparsity(o.dim_sparsity(n_train))                # Main.scala:54-65
    w0 = np.zeros(data.dim + 1)
    l0, a0, _, _ = o.loss_acc(w0, 0, n_train)
    ob = OracleBackend(o)
    m = host.MasterSync(ob, n_train, ROWS, node_count=NODES, rnd=host.JavaRandom(0))
    st = m.fit(w0, EPOCHS, BATCH, LR, host.EarlyStopping.no_improvement(5, 0.01))
    w1 = np.asarray(ob.get_weights(), dtype=np.float64)
    l1, a1, _, _ = o.loss_acc(w1, n_train, ROWS)
    return {
        "config": {"rows": ROWS, "node_count": NODES, "batch_size": BATCH, "learning_rate": LR, "lambda": LAM,
                   "max_epochs": EPOCHS, "full": False, "seed": 0},
        "initial_loss": l0, "initial_accuracy": a0,
        "train_loss_per_epoch_newest_first": list(m.losses), "test_loss_per_epoch_newest_first": list(m.test_losses),
        "batches": len(ob.steps), "final_test_loss": l1, "final_test_accuracy": a1,
        "final_weights": {str(int(k)): float(w1[k]) for k in np.flatnonzero(np.abs(w1) > 1e-20)},
    }


if __name__ == "__main__":
    exp = compute(write_data=True)
    with open(os.path.join(HERE, "jvm_expected", "expected.json"), "w") as f:
        json.dump(exp, f, indent=1, sort_keys=True)
    print("batches %d, %d final weights, final test loss %.9f accuracy %.4f" % (
        exp["batches"], len(exp["final_weights"]), exp["final_test_loss"], exp["final_test_accuracy"]))
