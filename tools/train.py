"""The "dev" role of the reference's Main (Main.scala:32-118, 153-159: master and node-count slaves in one process) on
one MI355X: same `dsgd { ... }` keys / DSGD_* variables, same scenario -- load, 80/20 split, dimSparsity, initial
loss and accuracy, fit (sync or async, early stopping by patience / conv-delta), final weights, final test loss and
accuracy -- with the HIP engine hosting all the workers.

    DSGD_DATA_PATH=/data/rcv1 python tools/train.py [--conf application.conf] [--synthetic ROWS] [--device 0] [--weights-out w.txt]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dsgd_amd
from dsgd_amd import host, rcv1

ap = argparse.ArgumentParser()
ap.add_argument("--conf", help="application.conf to read the dsgd{...} block from (DSGD_* variables override)")
ap.add_argument("--synthetic", type=int, default=0, help="use N synthetic RCV1-like rows instead of data-path")
ap.add_argument("--device", type=int, default=0)
ap.add_argument("--weights-out", help="write the `idx:value` line of Main.scala:114 here instead of logging it")
a = ap.parse_args()


def log(msg, *args):  # logback's pattern is not reproduced; the messages are
    print(time.strftime("%H:%M:%S"), msg.replace("{}", "%s") % args if args else msg, flush=True)


cfg = host.Config.load(open(a.conf).read() if a.conf else None)
log("config: {}", cfg)
log("loading data in: {}", "synthetic(%d)" % a.synthetic if a.synthetic else cfg.data_path)
t0 = time.time()
data = dsgd_amd.synth.generate(a.synthetic, seed=0) if a.synthetic else rcv1.load(cfg.data_path, full=cfg.full)
log("data loaded: {} ({}s)", data.n_rows, round(time.time() - t0, 2))
n_train = int(data.n_rows * 0.8)                                              # Main.scala:52
metrics = host.Metrics()
with dsgd_amd.Engine(data.dim, cfg.lambda_, device=a.device) as eng:
    eng.load_csr(data.row_ptr, data.col, data.val, data.label)
    t0 = time.time()
    eng.build_dim_sparsity(n_train)                                            # Main.scala:54-65
    log("dim sparsity ({}s)", round(time.time() - t0, 3))
    w0 = np.zeros(data.dim + 1, dtype=np.float32)
    eng.set_weights(w0)
    l0, a0, _ = eng.loss_acc(0, n_train)                                       # distributedLoss / Accuracy over the train split
    log("initial loss: {}", l0)
    log("initial accuracy: {}", a0)
    stop = host.EarlyStopping.no_improvement(cfg.patience, cfg.conv_delta, None)
    t0 = time.time()
    if cfg.async_:
        master = host.MasterAsync(eng, n_train, data.n_rows, cfg.node_count, log=log, poll_s=0.001)
        state = master.fit(w0, cfg.max_epochs, cfg.batch_size, cfg.learning_rate, stop, cfg.check_every, cfg.leaky_loss)
    else:
        master = host.MasterSync(eng, n_train, data.n_rows, cfg.node_count, rnd=host.JavaRandom(0), log=log, metrics=metrics)
        state = master.fit(w0, cfg.max_epochs, cfg.batch_size, cfg.learning_rate, stop)
    log("fit ({}s)", round(time.time() - t0, 3))
    w1 = np.asarray(state.grad, dtype=np.float32)
    line = host.format_final_weights(w1)
    if a.weights_out:
        open(a.weights_out, "w").write(line + "\n")
        log("final weights: {} entries written to {}", int(np.count_nonzero(w1)), a.weights_out)
    else:
        log("final weights: {}", line)
    l1, a1, _ = eng.loss_acc(n_train, data.n_rows, w=w1)                       # localLoss / localAccuracy on testData
    log("final test loss: {}", l1)
    log("final test accuracy: {}", a1)
    if cfg.record:
        for ln in metrics.influx_lines():
            print(ln)
