"""A GPU worker for an UNMODIFIED Scala master: the slave role of Main.scala:131-152 with the HIP engine behind the
reference's own gRPC protocol (distributed-sgd_amd/wire.py).  Configuration = the reference's `dsgd { ... }` keys and
DSGD_* environment overrides (resources/application.conf), e.g.

    DSGD_MASTER_HOST=10.0.0.1 DSGD_MASTER_PORT=4000 DSGD_NODE_HOST=10.0.0.7 DSGD_NODE_PORT=4001 \\
    DSGD_DATA_PATH=/data/rcv1 python tools/run_worker.py [--conf application.conf] [--synthetic ROWS] [--device 0]

Like every node of the reference it loads the whole data set (Main.scala:44-52: train = first 80 %), builds
dimSparsity from the train rows (Main.scala:54-65), registers with the master and serves until interrupted."""
import argparse, os, signal, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsgd_amd
from dsgd_amd import host, rcv1, wire

ap = argparse.ArgumentParser()
ap.add_argument("--conf", help="application.conf to read the dsgd{...} block from (DSGD_* variables override)")
ap.add_argument("--synthetic", type=int, default=0, help="use N synthetic RCV1-like rows instead of data-path")
ap.add_argument("--device", type=int, default=0)
a = ap.parse_args()
cfg = host.Config.load(open(a.conf).read() if a.conf else None)
if cfg.role() != "slave":
    sys.exit("this process is a worker: set master-host/master-port (DSGD_MASTER_HOST/PORT) to the master's address")
data = dsgd_amd.synth.generate(a.synthetic, seed=0) if a.synthetic else rcv1.load(cfg.data_path, full=cfg.full)
n_train = int(data.n_rows * 0.8)                                   # Main.scala:52 (the slave serves the train rows)
eng = dsgd_amd.Engine(data.dim, cfg.lambda_, device=a.device)
eng.load_csr(data.row_ptr[:n_train + 1], data.col[:data.row_ptr[n_train]], data.val[:data.row_ptr[n_train]], data.label[:n_train])
eng.build_dim_sparsity(n_train)
worker = wire.SlaveWorker(eng, data.dim, host_=cfg.host, port=cfg.port, master=(cfg.master_host, cfg.master_port),
                          asynchronous=cfg.async_).start()
print("worker %s:%d registered with %s:%d (%d train rows, %s mode)" % (cfg.host, worker.port, cfg.master_host, cfg.master_port,
                                                                        n_train, "async" if cfg.async_ else "sync"), flush=True)
done = threading.Event()
signal.signal(signal.SIGINT, lambda *x: done.set())
signal.signal(signal.SIGTERM, lambda *x: done.set())
done.wait()
worker.stop()
for line in worker.metrics.influx_lines("node=%s:%d" % (cfg.host, worker.port)):
    print(line)
eng.close()
