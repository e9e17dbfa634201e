"""Write the synthetic workload of bench.py / the tests as RCV1-v2 text files, so that anyone with a JVM can run the
reference itself on identical data:

    python tools/export_rcv1_text.py /tmp/data --rows 23149          # configs[0]/[1] shape (full = false reads the
    cd <reference>; ln -s /tmp/data data; DSGD_NODE_COUNT=1 sbt run  #  train file only)

With --rows 804414 the five files have the official sizes (23,149 + 4 test parts)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dsgd_amd

ap = argparse.ArgumentParser()
ap.add_argument("folder")
ap.add_argument("--rows", type=int, default=23149)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
data = dsgd_amd.synth.generate(a.rows, seed=a.seed)
dsgd_amd.rcv1.export(a.folder, data)
back = dsgd_amd.rcv1.load(a.folder, full=a.rows > dsgd_amd.rcv1.N_TRAIN_OFFICIAL)
assert back.n_rows == a.rows and (back.label == data.label).all() and (back.col == data.col).all()
print("wrote %d rows, %d non-zeros to %s (read back: labels and keys identical, max |value diff| %.3g)" % (
    a.rows, data.nnz, a.folder, float(abs(back.val - data.val).max()) if data.nnz else 0.0))
