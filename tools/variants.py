"""A/B runs of the whole-shard gradient step under different environment knobs, one subprocess per variant.
usage: python tools/variants.py [--rows N] "" "DSGD_FIX_BOUND=0" "DSGD_HSPLIT=16384 DSGD_FIX_SHIFT=18" ...   ("" = defaults)"""
import os, subprocess, sys, time

def child(rows):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import dsgd_amd
    data = dsgd_amd.synth.generate(rows, seed=0)
    n_train = int(rows * 0.8)
    alg = 8.0 * int(data.row_ptr[n_train]) + 12.0 * n_train
    eng = dsgd_amd.Engine(data.dim, 1e-5)
    eng.load_csr(data.row_ptr, data.col, data.val, data.label)
    eng.build_dim_sparsity(n_train)
    lr = 0.0 if os.environ.get("VAR_ALL_ACTIVE") else 0.5 * 100 / n_train   # lr 0: w stays 0, every row active
    for _ in range(6): eng.sync_step_ranges([(0, n_train)], lr)
    eng.synchronize()
    eng.prof_enable(True); eng.prof_read(reset=True)
    t0 = time.perf_counter()
    n = 20
    for _ in range(n): eng.sync_step_ranges([(0, n_train)], lr)
    eng.synchronize()
    step = (time.perf_counter() - t0) / n * 1e3
    kinds = eng.prof_read_kinds()
    ms, cnt = eng.prof_read(reset=True)
    eng.loss_acc(0, n_train)
    t0 = time.perf_counter()
    for _ in range(5): loss, acc, _t = eng.loss_acc(0, n_train)
    ev = (time.perf_counter() - t0) / 5 * 1e3
    print("cdot %.3f cgrad %.3f | kernel %.3f ms %5.0f GB/s | step %.3f ms %.3f Gex/s | eval %.3f ms %5.0f GB/s | loss %.6f acc %.4f | %s" % (
        kinds["cdot"][0], kinds["cgrad"][0], ms, alg / ms / 1e6, step, n_train / step / 1e6, ev, alg / ev / 1e6, loss, acc, eng.grad_kernel_name()), flush=True)

if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--child":
        child(int(args[1])); sys.exit(0)
    rows = 8388608
    if args and args[0] == "--rows":
        rows = int(args[1]); args = args[2:]
    for v in (args or [""]):
        env = dict(os.environ)
        for kv in v.split():
            k, val = kv.split("=", 1); env[k] = val
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(rows)], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=600)
        print("[%-28s] %s" % (v or "defaults", r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "no output rc=%d" % r.returncode), flush=True)
