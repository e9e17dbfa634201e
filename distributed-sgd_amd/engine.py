"""Engine: one libdsgd_hip context (one MI355X) as a Python object.

Thin, typed face over the C ABI of include/dsgd.h; all arithmetic happens in the HIP library.
Dense vectors are numpy float32 arrays of D+1 slots indexed by key (see include/dsgd.h).
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import BatchStats, Config, check, f32, i32, ptr


class Plan:
    """Resident schedule of index lists (steps x workers); see dsgd_plan_create."""

    def __init__(self, engine, handle, n_steps, n_workers, n_samples):
        self.engine, self.handle, self.n_steps, self.n_workers, self.n_samples = engine, handle, n_steps, n_workers, n_samples

    def destroy(self):
        if self.handle:
            check(_lib.load().dsgd_plan_destroy(self.engine._ctx, self.handle))
            self.handle = None

    def info(self):
        """How the plan will run: kind ('column_slices', 'one_workgroup', 'virtual_tiles', 'row_parallel', 'not_laid_out')
        and, for column slices, the layout's shape; see dsgd_plan_info."""
        v = (C.c_int32 * 8)()
        check(_lib.load().dsgd_plan_info(self.engine._ctx, self.handle, v, C.c_int32(8)))
        kinds = {0: "not_laid_out", 1: "column_slices", 2: "one_workgroup", 3: "virtual_tiles", 4: "row_parallel"}
        return {"kind": kinds.get(int(v[0]), "?"), "slices": int(v[1]), "slot_stride": int(v[2]), "row_stride": int(v[3]),
                "col_list_stride": int(v[4]), "slots_per_lane": int(v[5]), "device_built": bool(v[6]), "record_words": int(v[7])}

    def record(self, on=True):
        """Keep the gate decision of every row and the regulariser scalar of every step this plan runs (column slices)."""
        check(_lib.load().dsgd_plan_record(self.engine._ctx, self.handle, C.c_int32(1 if on else 0)))

    def read_record(self, step_begin=0, step_end=None):
        """(mask, s): mask bool [steps, words * 32] -- bit r of a step = row r of the step (workers in order, each list in
        order) was active; s float32 [steps] -- the regulariser scalar the step used."""
        step_end = self.n_steps if step_end is None else step_end
        lib = _lib.load()
        mw = C.c_int32(0)
        check(lib.dsgd_plan_read_record(self.engine._ctx, self.handle, C.c_int64(step_begin), C.c_int64(step_end), None, None, C.byref(mw)))
        n, m = step_end - step_begin, mw.value
        mask = np.zeros((n, max(m, 1)), dtype=np.uint32)
        s = np.zeros(n, dtype=np.float32)
        check(lib.dsgd_plan_read_record(self.engine._ctx, self.handle, C.c_int64(step_begin), C.c_int64(step_end), ptr(mask), ptr(s), None))
        bits = ((mask[:, :, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & 1).astype(bool).reshape(n, 32 * max(m, 1))
        return bits, s


class Engine:
    def __init__(self, n_features, lam, device=0, flags=0):
        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        self.dim = int(n_features)
        self.dp = self.dim + 1
        self.lam = float(lam)
        cfg = Config(self.dim, int(device), self.lam, int(flags), 0)
        check(self._lib.dsgd_create(C.byref(cfg), C.byref(self._ctx)))
        self.n_rows = 0
        self.nnz = 0

    # -- lifecycle -----------------------------------------------------------------------------
    def close(self):
        if self._ctx:
            self._lib.dsgd_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data ------------------------------------------------------------------------------------
    def load_csr(self, row_ptr, col, val, label):
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
        col = i32(col)
        val = f32(val)
        label = np.ascontiguousarray(label, dtype=np.int8)
        n_rows = len(row_ptr) - 1
        if len(label) != n_rows:
            raise ValueError("label has %d entries for %d rows" % (len(label), n_rows))
        if n_rows >= 1 and (len(col) < row_ptr[-1] or len(val) < row_ptr[-1]):
            raise ValueError("col/val shorter than row_ptr[-1]")
        check(self._lib.dsgd_load_csr(self._ctx, C.c_int64(n_rows), ptr(row_ptr), ptr(col), ptr(val), ptr(label)))
        self.n_rows, self.nnz = n_rows, int(row_ptr[-1])

    def set_dim_sparsity(self, ds):
        check(self._lib.dsgd_set_dim_sparsity(self._ctx, ptr(f32(ds, self.dp))))

    def build_dim_sparsity(self, n_train):
        out = np.zeros(self.dp, dtype=np.float32)
        check(self._lib.dsgd_build_dim_sparsity(self._ctx, C.c_int64(n_train), ptr(out)))
        return out

    def set_weights(self, w):
        check(self._lib.dsgd_set_weights(self._ctx, ptr(f32(w, self.dp))))

    def get_weights(self):
        out = np.zeros(self.dp, dtype=np.float32)
        check(self._lib.dsgd_get_weights(self._ctx, ptr(out)))
        return out

    # -- synchronous path ------------------------------------------------------------------------
    def gradient(self, idx, w=None):
        idx = i32(idx)
        g = np.zeros(self.dp, dtype=np.float32)
        st = BatchStats()
        wv = None if w is None else f32(w, self.dp)
        check(self._lib.dsgd_gradient(self._ctx, ptr(wv), ptr(idx), C.c_int64(len(idx)), ptr(g), C.byref(st)))
        return g, {"n_samples": st.n_samples, "n_active": st.n_active}

    def apply(self, g_mean, lr):
        check(self._lib.dsgd_apply(self._ctx, ptr(f32(g_mean, self.dp)), C.c_float(lr)))

    def sync_step(self, idx_per_worker, lr):
        lists = [i32(a) for a in idx_per_worker]
        k = len(lists)
        ptrs = (C.c_void_p * max(k, 1))(*[ptr(a) for a in lists])
        ns = (C.c_int64 * max(k, 1))(*[len(a) for a in lists])
        st = BatchStats()
        check(self._lib.dsgd_sync_step(self._ctx, ptrs, ns, C.c_int32(k), C.c_float(lr), C.byref(st)))
        return {"n_samples": st.n_samples, "n_active": st.n_active}

    def sync_step_ranges(self, ranges, lr, asynchronous=False):
        k = len(ranges)
        rb = (C.c_int64 * max(k, 1))(*[int(r[0]) for r in ranges])
        re_ = (C.c_int64 * max(k, 1))(*[int(r[1]) for r in ranges])
        if asynchronous:
            check(self._lib.dsgd_sync_step_ranges_async(self._ctx, rb, re_, C.c_int32(k), C.c_float(lr)))
            return None
        st = BatchStats()
        check(self._lib.dsgd_sync_step_ranges(self._ctx, rb, re_, C.c_int32(k), C.c_float(lr), C.byref(st)))
        return {"n_samples": st.n_samples, "n_active": st.n_active}

    def synchronize(self):
        st = BatchStats()
        check(self._lib.dsgd_synchronize(self._ctx, C.byref(st)))
        return {"n_samples": st.n_samples, "n_active": st.n_active}

    def plan(self, steps):
        """steps: list (per step) of lists (per worker) of index arrays."""
        n_steps = len(steps)
        n_workers = len(steps[0]) if n_steps else 0
        flat, offs = [], [0]
        for s in steps:
            if len(s) != n_workers:
                raise ValueError("every step needs the same number of workers")
            for a in s:
                a = i32(a)
                flat.append(a)
                offs.append(offs[-1] + len(a))
        idx = np.concatenate(flat) if flat else np.zeros(0, np.int32)
        offsets = np.asarray(offs, dtype=np.int64)
        h = C.c_void_p()
        check(self._lib.dsgd_plan_create_n(self._ctx, ptr(idx), C.c_int64(len(idx)), ptr(offsets), C.c_int64(n_steps), C.c_int32(n_workers), C.byref(h)))
        return Plan(self, h, n_steps, n_workers, int(offsets[-1]))

    def plan_flat(self, idx, offsets, n_steps, n_workers):
        """A plan from the flat form: idx = all lists concatenated (step-major, worker-minor), offsets = n_steps * n_workers
        + 1 prefix offsets (what host.epoch_lists returns: one epoch of Master.fit)."""
        idx = i32(idx)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        if len(offsets) != n_steps * n_workers + 1 or (n_steps and int(offsets[-1]) != len(idx)):
            raise ValueError("offsets do not describe %d x %d lists over %d entries" % (n_steps, n_workers, len(idx)))
        h = C.c_void_p()
        check(self._lib.dsgd_plan_create_n(self._ctx, ptr(idx), C.c_int64(len(idx)), ptr(offsets), C.c_int64(n_steps), C.c_int32(n_workers), C.byref(h)))
        return Plan(self, h, n_steps, n_workers, int(offsets[-1]))

    def plan_from_seed(self, jstate, split, max_samples, batch_size):
        """One epoch of Master.fit as a plan whose lists the DEVICE draws, draw for draw the reference's stream
        (dsgd_plan_create_from_seed).  jstate: java.util.Random's internal 48-bit state in front of the epoch; split: the
        workers' row ranges (SplitStrategy.vanilla).  Returns (plan or None, n_steps, new jstate, draws); raises
        DsgdError with code EUNSUPPORTED when the device form does not apply (draw the lists on the host then)."""
        sb = np.asarray([r.start if isinstance(r, range) else r[0] for r in split], dtype=np.int64)
        se = np.asarray([r.stop if isinstance(r, range) else r[1] for r in split], dtype=np.int64)
        st = C.c_uint64(int(jstate))
        h = C.c_void_p()
        n_steps, draws = C.c_int64(0), C.c_int64(0)
        check(self._lib.dsgd_plan_create_from_seed(self._ctx, C.byref(st), ptr(sb), ptr(se), C.c_int32(len(sb)), C.c_int64(max_samples),
                                                   C.c_int32(batch_size), C.byref(h), C.byref(n_steps), C.byref(draws)))
        if not h.value:
            return None, 0, int(st.value), 0
        total = int(sum(min(batch_size, int(e - b) - s_ * batch_size) for s_ in range(n_steps.value) for b, e in zip(sb, se)))
        return Plan(self, h, n_steps.value, len(sb), total), n_steps.value, int(st.value), draws.value

    def plan_lists(self, plan):
        """(idx, offsets) of a plan as the device holds them (tests)."""
        n_lists = plan.n_steps * plan.n_workers
        offsets = np.zeros(n_lists + 1, dtype=np.int64)
        check(self._lib.dsgd_plan_read_lists(self._ctx, plan.handle, None, C.c_int64(0), ptr(offsets), C.c_int64(n_lists + 1)))
        idx = np.zeros(int(offsets[-1]), dtype=np.int32)
        check(self._lib.dsgd_plan_read_lists(self._ctx, plan.handle, ptr(idx), C.c_int64(len(idx)), None, C.c_int64(0)))
        return idx, offsets

    def cache_trim(self, keep_bytes=0):
        """Give the device blocks of destroyed plans back (all but keep_bytes); returns the bytes still held."""
        held = C.c_int64(0)
        check(self._lib.dsgd_cache_trim(self._ctx, C.c_int64(keep_bytes), C.byref(held)))
        return held.value

    def plan_run(self, plan, step_begin, step_end, lr):
        check(self._lib.dsgd_plan_run(self._ctx, plan.handle, C.c_int64(step_begin), C.c_int64(step_end), C.c_float(lr)))

    # -- evaluation --------------------------------------------------------------------------------
    def forward(self, idx, w=None):
        idx = i32(idx)
        pred = np.zeros(len(idx), dtype=np.float32)
        wv = None if w is None else f32(w, self.dp)
        check(self._lib.dsgd_forward(self._ctx, ptr(wv), ptr(idx), C.c_int64(len(idx)), ptr(pred)))
        return pred

    def loss_acc(self, row_begin, row_end, w=None):
        loss, acc = C.c_double(0), C.c_double(0)
        counts = (C.c_int64 * 3)()
        wv = None if w is None else f32(w, self.dp)
        check(self._lib.dsgd_loss_acc(self._ctx, ptr(wv), C.c_int64(row_begin), C.c_int64(row_end), C.byref(loss), C.byref(acc), counts))
        return loss.value, acc.value, list(counts)

    # -- asynchronous path -------------------------------------------------------------------------
    def async_step(self, idx, lr, want_delta=False):
        idx = i32(idx)
        delta = np.zeros(self.dp, dtype=np.float32) if want_delta else None
        st = BatchStats()
        check(self._lib.dsgd_async_step(self._ctx, ptr(idx), C.c_int64(len(idx)), C.c_float(lr), ptr(delta), C.byref(st)))
        return delta, {"n_samples": st.n_samples, "n_active": st.n_active}

    def update_grad(self, keys, values):
        keys, values = i32(keys), f32(values)
        if len(keys) != len(values):
            raise ValueError("keys / values length mismatch")
        check(self._lib.dsgd_update_grad(self._ctx, ptr(keys), ptr(values), C.c_int64(len(keys))))

    def async_start(self, assigned_ranges, batch, lr, max_updates, seed=0, positional_bug=True):
        k = len(assigned_ranges)
        rb = (C.c_int64 * max(k, 1))(*[int(r[0]) for r in assigned_ranges])
        re_ = (C.c_int64 * max(k, 1))(*[int(r[1]) for r in assigned_ranges])
        check(self._lib.dsgd_async_start(self._ctx, rb, re_, C.c_int32(k), C.c_int32(batch), C.c_float(lr),
                                         C.c_int64(max_updates), C.c_uint64(seed), C.c_int32(1 if positional_bug else 0)))

    def async_set_exchange(self, every_updates):
        """Cross-GPU asynchronous mode: exchange the summed updates every `every_updates` local updates (0 = off)."""
        check(self._lib.dsgd_async_set_exchange(self._ctx, C.c_int64(every_updates)))

    def async_updates(self):
        n, running = C.c_int64(0), C.c_int32(0)
        check(self._lib.dsgd_async_updates(self._ctx, C.byref(n), C.byref(running)))
        return n.value, bool(running.value)

    def async_stop(self):
        check(self._lib.dsgd_async_stop(self._ctx))

    def async_wait(self):
        check(self._lib.dsgd_async_wait(self._ctx))

    def async_stats(self):
        """Counters of the lock-free engine (updates, samples, active rows, lane-level weight atomics) and its regulariser
        scalar s = 2 lambda (w . ds): as kept incrementally on the device / recomputed from the weights as they are now."""
        cnt = (C.c_int64 * 4)()
        a, b = C.c_double(0), C.c_double(0)
        check(self._lib.dsgd_async_stats(self._ctx, cnt, C.byref(a), C.byref(b)))
        return {"updates": cnt[0], "samples": cnt[1], "active": cnt[2], "atomics": cnt[3], "s_engine": a.value, "s_exact": b.value}

    def async_set_trace(self, capacity):
        """Attach a trace of `capacity` update records to the following lock-free runs (0 detaches it)."""
        check(self._lib.dsgd_async_set_trace(self._ctx, C.c_int64(capacity)))

    def async_read_trace(self):
        """The recorded updates of the last run in commit order (record i = update number i + 1) as a dict of arrays:
        worker, it (the worker's iteration = the sampler's key), read_at (update count its weights were read at), s (the
        regulariser scalar it used), n_active, mask (bool [n, batch]: the gate decision of every sampled row);
        see dsgd_async_set_trace."""
        n, mw = C.c_int64(0), C.c_int32(0)
        check(self._lib.dsgd_async_read_trace(self._ctx, None, None, None, None, None, None, C.c_int64(0), C.byref(n), C.byref(mw)))
        k, m = n.value, mw.value
        worker = np.zeros(k, dtype=np.int32)
        it = np.zeros(k, dtype=np.uint32)
        read_at = np.zeros(k, dtype=np.int64)
        s = np.zeros(k, dtype=np.float32)
        n_active = np.zeros(k, dtype=np.int32)
        mask = np.zeros((k, m), dtype=np.uint32)
        if k:
            check(self._lib.dsgd_async_read_trace(self._ctx, ptr(worker), ptr(it), ptr(read_at), ptr(s), ptr(n_active), ptr(mask),
                                                  C.c_int64(k), None, None))
        bits = ((mask[:, :, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & 1).astype(bool).reshape(k, 32 * m)
        # what every decision was taken on: the x . w of each sampled row, and the update count known to be in the weights
        bo = C.c_int32(0)
        check(self._lib.dsgd_async_read_trace_dots(self._ctx, None, None, C.c_int64(0), C.byref(bo)))
        seen_from = np.zeros(k, dtype=np.int64)
        dots = np.zeros((k, max(1, bo.value)), dtype=np.float32)
        if k:
            check(self._lib.dsgd_async_read_trace_dots(self._ctx, ptr(seen_from), ptr(dots), C.c_int64(k), None))
        return {"worker": worker, "it": it, "read_at": read_at, "s": s, "n_active": n_active, "mask": bits,
                "seen_from": seen_from, "dot": dots}

    # -- multi-GPU ---------------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
        check(_lib.load().dsgd_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id, world_size, rank):
        if len(unique_id) != _lib.UNIQUE_ID_BYTES:
            raise ValueError("unique id must be %d bytes" % _lib.UNIQUE_ID_BYTES)
        check(self._lib.dsgd_comm_init(self._ctx, C.c_char_p(unique_id), C.c_int32(world_size), C.c_int32(rank)))

    def comm_destroy(self):
        check(self._lib.dsgd_comm_destroy(self._ctx))

    # -- profiling ---------------------------------------------------------------------------------
    def prof_enable(self, on=True):
        """0 / False: off; 1 / True: bracket every profiled kernel; 2: the dominant gradient kernel only."""
        check(self._lib.dsgd_prof_enable(self._ctx, C.c_int32(int(on))))

    def prof_read(self, reset=True):
        ms, n = C.c_double(0), C.c_int64(0)
        check(self._lib.dsgd_prof_read(self._ctx, C.byref(ms), C.byref(n), C.c_int32(1 if reset else 0)))
        return ms.value, n.value

    def prof_read_kinds(self):
        """{kind: (avg ms, launches)} of the main / cold x.w / cold gradient kernels of the split layout."""
        ms = (C.c_double * 3)()
        n = (C.c_int64 * 3)()
        check(self._lib.dsgd_prof_read_kinds(self._ctx, ms, n))
        return {k: (ms[i], n[i]) for i, k in enumerate(("main", "cdot", "cgrad"))}

    def range_nnz(self, row_begin, row_end):
        """(non-zeros, of which in the cold stream) of rows [row_begin, row_end) as held internally."""
        a, b = C.c_int64(), C.c_int64()
        check(self._lib.dsgd_range_nnz(self._ctx, C.c_int64(row_begin), C.c_int64(row_end), C.byref(a), C.byref(b)))
        return a.value, b.value

    def tuning_info(self):
        v = (C.c_int32 * 7)()
        if self._lib.dsgd_tuning_info(self._ctx, v, C.c_int32(7)) != 0:   # (an older build in an A/B run: six slots)
            check(self._lib.dsgd_tuning_info(self._ctx, v, C.c_int32(6)))
        return dict(zip(("stream_mode", "hsplit", "fix_shift", "cold_packed", "plan_kernel", "fix_bound", "fstep_rebalances"), [int(x) for x in v]))

    def column_ranks(self):
        """Internal frequency rank of every key (D + 1 entries); identical on all ranks of a communicator."""
        v = np.zeros(self.dp, dtype=np.int32)
        check(self._lib.dsgd_column_ranks(self._ctx, ptr(v)))
        return v

    def debug_cycles(self, reset=True):
        v = (C.c_uint64 * 16)()
        check(self._lib.dsgd_debug_cycles(self._ctx, v, C.c_int32(1 if reset else 0)))
        return [int(x) for x in v]

    def grad_kernel_name(self):
        return self._lib.dsgd_grad_kernel_name(self._ctx).decode()


class EngineGroup:
    """Several contexts (one per device, each with its own rows) driven by ONE host thread: the dsgd_*_devices entry
    points of include/dsgd.h (the reference's dev role runs the master and every slave in one JVM, Main.scala:144-158).
    Arrays over workers are context-major."""

    def __init__(self, engines):
        self.engines = list(engines)
        self._lib = _lib.load()
        self._ctxs = (C.c_void_p * len(self.engines))(*[e._ctx for e in self.engines])
        self.n = len(self.engines)

    def comm_init_all(self):
        check(self._lib.dsgd_comm_init_all(self._ctxs, C.c_int32(self.n)))

    def build_dim_sparsity(self, n_train_per_engine):
        nt = (C.c_int64 * self.n)(*[int(v) for v in n_train_per_engine])
        check(self._lib.dsgd_build_dim_sparsity_devices(self._ctxs, C.c_int32(self.n), nt))

    def sync_step(self, lists_per_engine, lr):
        """lists_per_engine: per engine, the index lists of its hosted workers (the same number for every engine)."""
        k = len(lists_per_engine[0])
        flat = [i32(a) for ls in lists_per_engine for a in ls]
        if len(lists_per_engine) != self.n or any(len(ls) != k for ls in lists_per_engine):
            raise ValueError("every engine needs the same number of workers")
        ptrs = (C.c_void_p * len(flat))(*[ptr(a) for a in flat])
        ns = (C.c_int64 * len(flat))(*[len(a) for a in flat])
        st = BatchStats()
        check(self._lib.dsgd_sync_step_devices(self._ctxs, C.c_int32(self.n), ptrs, ns, C.c_int32(k), C.c_float(lr), C.byref(st)))
        return {"n_samples": st.n_samples, "n_active": st.n_active}

    def sync_step_ranges(self, ranges_per_engine, lr):
        k = len(ranges_per_engine[0])
        if len(ranges_per_engine) != self.n or any(len(r) != k for r in ranges_per_engine):
            raise ValueError("every engine needs the same number of workers")
        flat = [r for rs in ranges_per_engine for r in rs]
        rb = (C.c_int64 * len(flat))(*[int(r[0]) for r in flat])
        re_ = (C.c_int64 * len(flat))(*[int(r[1]) for r in flat])
        st = BatchStats()
        check(self._lib.dsgd_sync_step_ranges_devices(self._ctxs, C.c_int32(self.n), rb, re_, C.c_int32(k), C.c_float(lr), C.byref(st)))
        return {"n_samples": st.n_samples, "n_active": st.n_active}

    def loss_acc(self, range_per_engine):
        rb = (C.c_int64 * self.n)(*[int(r[0]) for r in range_per_engine])
        re_ = (C.c_int64 * self.n)(*[int(r[1]) for r in range_per_engine])
        loss, acc = C.c_double(0), C.c_double(0)
        counts = (C.c_int64 * 3)()
        check(self._lib.dsgd_loss_acc_devices(self._ctxs, C.c_int32(self.n), rb, re_, C.byref(loss), C.byref(acc), counts))
        return loss.value, acc.value, list(counts)


def device_count():
    return _lib.load().dsgd_device_count()
