"""Re-derive tests/golden/kat.json with the dict-based restatement (oracle/ref_dict.py) and compare."""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_dict as rd


def close(a, expect, tol):
    assert set(a.map) == {int(k) for k in expect}, (a, expect)
    for k, v in expect.items():
        assert abs(a.map[int(k)] - v) < tol, (k, a.map[int(k)], v)


def check(kat=None):
    kat = kat or json.load(open(os.path.join(HERE, "kat.json")))
    rows = [({int(k): v for k, v in m.items()}, y) for m, y in kat["rows"]]
    k1 = kat["kat1"]
    data = [(rd.Sparse(dict(m), 6), y) for m, y in rows]
    ds = rd.dim_sparsity(data)
    close(ds, k1["ds"], 1e-12)
    model = rd.SparseSVM(k1["lambda"], ds)
    w = rd.Sparse({}, 6)
    assert rd.local_loss(model, w, data) == k1["initial"]["loss"] and rd.local_accuracy(model, w, data) == k1["initial"]["acc"]
    for st in k1["steps"]:
        close(rd.slave_gradient(model, data, w, k1["batches"][0]), st["g0"], 1e-9)
        close(rd.slave_gradient(model, data, w, k1["batches"][1]), st["g1"], 1e-9)
        w = rd.master_sync_step(model, data, w, k1["batches"], k1["lr"])
        close(w, st["w"], 1e-9)
        assert abs(rd.local_loss(model, w, data) - st["loss"]) < 1e-8 and abs(rd.local_accuracy(model, w, data) - st["acc"]) < 1e-12
    k2 = kat["kat2"]
    data = [(rd.Sparse(dict(m), 6), y) for m, y in rows[:k2["n_rows"]]]
    model = rd.SparseSVM(k2["lambda"], rd.dim_sparsity(data))
    w = rd.Sparse({}, 6)
    close(rd.slave_gradient(model, data, w, k2["batches"][0]), k2["step0"]["g0"], 1e-9)
    close(rd.slave_gradient(model, data, w, k2["batches"][1]), k2["step0"]["g1"], 1e-9)
    w = rd.master_sync_step(model, data, w, k2["batches"], k2["lr"])
    close(w, k2["step0"]["w"], 1e-9)
    assert abs(rd.local_loss(model, w, data) - k2["step0"]["loss"]) < 1e-9 and rd.local_accuracy(model, w, data) == k2["step0"]["acc"]
    for _ in range(2):
        assert not rd.slave_gradient(model, data, w, k2["batches"][0]).map and not rd.slave_gradient(model, data, w, k2["batches"][1]).map
        w2 = rd.master_sync_step(model, data, w, k2["batches"], k2["lr"])
        assert w2 == w
    return True


if __name__ == "__main__":
    check()
    print("kat.json agrees with oracle/ref_dict.py")
