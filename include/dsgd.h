/*
 * dsgd.h -- C ABI of libdsgd_hip: the MI355X (gfx950) engine for the hot path of
 * zifeo/distributed-sgd (sparse-SVM gradient step, synchronous aggregate+update,
 * asynchronous "Hogwild" update, prediction and loss/accuracy evaluation).
 *
 * The reference has NO native/FFI interface (it is 100 % Scala on the JVM); this header is the
 * boundary the new engine introduces behind the reference's natural seams.  Every entry point
 * names the reference code whose body it replaces; citations are relative to
 * /root/reference/src/main/scala/epfl/distributed/ unless they start with "proto.proto"
 * (src/main/protobuf/proto.proto).  The JNI / ctypes stubs a maintainer would add on the
 * reference side are shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross the boundary
 *   - every function returns int: 0 = DSGD_OK, < 0 = DSGD_E*; nothing throws or aborts.
 *     dsgd_last_error() returns a thread-local message for the last failing call.
 *     (The JNI shim maps DSGD_EINVAL / DSGD_ERANGE to IllegalArgumentException /
 *     IndexOutOfBoundsException -- what `require` at math/Vec.scala:129 and
 *     math/Sparse.scala:16,63 throw -- and the rest to RuntimeException.)
 *   - dense vectors (w, g, ds, delta) have n_features + 1 float slots indexed by KEY: feature
 *     ids are 1-based (utils/Dataset.scala:30 uses the file's ids as map keys), the
 *     dimSparsity vector uses 0-based keys (Main.scala:60-62), so slot 0 and slot D both exist.
 *     "absent from the map" == 0.0f.
 *   - host buffers are owned by the caller; the library copies in / out.  Pointers ending in
 *     `_dev` are device pointers (HIP) on the context's device.
 *   - a context may be called from several host threads (the reference calls its model from an
 *     8-thread pool, utils/Pool.scala:13); calls on one context are serialised internally.
 *   - arithmetic is IEEE fp32 on the device ("fp32 CSR-SpMV gradient kernel" of BASELINE.json);
 *     the reference is fp64.  Stated tolerance: tests/test_gpu_parity.py.
 */
#ifndef DSGD_H
#define DSGD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSGD_ABI_VERSION 1

enum {
  DSGD_OK = 0,
  DSGD_EINVAL = -1,       /* bad argument / a reference `require` would have failed            */
  DSGD_ERANGE = -2,       /* sample index outside the loaded data                              */
  DSGD_ESTATE = -3,       /* call order: no data loaded, async already running, ...            */
  DSGD_EHIP = -4,         /* HIP runtime error (message has hipGetErrorString)                 */
  DSGD_ERCCL = -5,        /* RCCL error                                                        */
  DSGD_ENOMEM = -6,
  DSGD_EUNSUPPORTED = -7  /* no gfx950 device / feature not available                          */
};

/* dsgd_config.flags: none defined (must be 0) */
#define DSGD_F_DEFAULT 0u

typedef struct dsgd_ctx dsgd_ctx;

/* MODEL WIDTH.  Any D >= 1 gives the same results; what a wide model costs (measured, whole-split steps of 800,000
 * RCV1-like rows of 75 non-zeros, profiles/r06_dispatch_table.txt): the matrix is split by column frequency into the
 * 18,396 hottest columns (weights and gradient of a workgroup in LDS, 16-bit ranks in the stream) and the rest, whose
 * weights / gradient must fit ONE LDS tile of 36,796 words for the one-launch row chunks (csrc/dsgd_fstep.hpp):
 *   D <= 55,191   (RCV1: 47,236)  row chunks: 97 us per step at D = 47,236, 108 us at D = 20,000
 *   D <= 83,931                   the three streaming launches, cold columns beyond the tile gathered from L2 and added
 *                                 through 64-bit global atomics: 155 us at D = 70,000 (+ 50 %)
 *   D  > 83,931                   more than 65,536 cold columns: 32-bit cold column words (8 instead of 6 bytes per cold
 *                                 non-zero, 7.5 % of RCV1-like non-zeros), otherwise as the line above
 * dsgd_grad_kernel_name() says which family ran; tests/test_gpu_dispatch.py holds the choice to the best family.    */
typedef struct {
  int32_t n_features; /* D; 47236 for RCV1 (utils/Dataset.scala:16)                              */
  int32_t device;     /* HIP device ordinal                                                      */
  double lambda;      /* SparseSVM.lambda (core/ml/SparseSVM.scala:11; application.conf:21)      */
  uint32_t flags;
  uint32_t reserved;
} dsgd_config;

/* counters a batch call reports back (the reference increments Kamon counters per SAMPLE:
 * core/Slave.scala:131,145,90 -- the host side does counter.increment(n_samples)) */
typedef struct {
  int64_t n_samples;  /* rows whose gradient / prediction was computed                          */
  int64_t n_active;   /* rows with y * (x . w) >= 0 (core/ml/SparseSVM.scala:27-28)             */
} dsgd_batch_stats;

int dsgd_abi_version(void);
const char* dsgd_last_error(void);
/* number of visible gfx950 devices (0 if none / no HIP runtime) */
int dsgd_device_count(void);

/* ---- lifecycle: replaces `new SparseSVM(lambda, dimSparsity)` + the data array handed to
 *      `new Slave(node, master, data, model, async)` (Main.scala:68,138,148-149) ------------- */
int dsgd_create(const dsgd_config* cfg, dsgd_ctx** out);
int dsgd_destroy(dsgd_ctx* ctx);

/* data: Array[(Vec, Int)] (utils/Dataset.scala:11) as CSR.  col ids are the reference's keys (1-based feature ids),
 * each at most once per row in any order (a row is a Map: DSGD_EINVAL for a repeated key), label +1/-1.
 * Copied to HBM once; resident afterwards.  Indices in later calls refer to it,
 * exactly as GradientRequest.samples / ForwardRequest.samples index Slave.data
 * (core/Slave.scala:134,149; proto.proto:51-63). */
int dsgd_load_csr(dsgd_ctx* ctx, int64_t n_rows, const int64_t* row_ptr, const int32_t* col_1based, const float* val,
                  const int8_t* label);
int dsgd_n_rows(dsgd_ctx* ctx, int64_t* n_rows, int64_t* nnz);

/* SparseSVM.dimSparsity: either given (dense, 0-based keys as Main.scala:62 builds them) ...  */
int dsgd_set_dim_sparsity(dsgd_ctx* ctx, const float* ds /* D+1 */);
/* ... or built on the device from rows [0, n_train) exactly as Main.scala:54-65 does
 * (including the off-by-one: count of feature f lands on key f-1).  ds_out may be NULL.       */
int dsgd_build_dim_sparsity(dsgd_ctx* ctx, int64_t n_train, float* ds_out /* D+1 or NULL */);

/* resident weights (GradState.grad holds the WEIGHTS: core/ml/GradState.scala:6-10)           */
int dsgd_set_weights(dsgd_ctx* ctx, const float* w /* D+1 */);
int dsgd_get_weights(dsgd_ctx* ctx, float* w_out /* D+1 */);

/* ---- synchronous path ----------------------------------------------------------------------
 * SlaveImpl.gradient (core/Slave.scala:142-157): g = regularize(sum_i backward(w, x_i, y_i), w).
 * w == NULL uses the resident weights (no transfer); otherwise w (D+1 floats) replaces them,
 * like the full `weights` every GradientRequest carries (proto.proto:60-63).
 * n == 0 fails with DSGD_EINVAL (Vec.sum requires a non-empty list, math/Vec.scala:129).      */
int dsgd_gradient(dsgd_ctx* ctx, const float* w, const int32_t* idx, int64_t n, float* g_out /* D+1 */,
                  dsgd_batch_stats* stats /* may be NULL */);

/* Master.fit batch closure, update half (core/Master.scala:194-197): w <- w - lr * g_mean      */
int dsgd_apply(dsgd_ctx* ctx, const float* g_mean /* D+1 */, float lr);

/* The whole batch closure (core/Master.scala:184-197) for n_workers workers hosted by this
 * context: per-worker regularised sums, MEAN over workers, w <- w - lr * mean.  Index lists are
 * what `split.map(Random.shuffle(_)).slice(batch, batch + batchSize)` produced on the host.
 * If a communicator is attached (dsgd_comm_init) the mean runs over n_workers * world_size
 * workers with one RCCL all-reduce of the summed gradient.
 * 40 us per call at the reference's sizes (two row-parallel launches).  Hand an epoch's lists over as ONE plan instead
 * (dsgd_plan_create / dsgd_plan_run below: 5 us per step) wherever they are known ahead -- Master.fit knows them.  With
 * DSGD_CS_REQ=1 such a request runs as ONE launch of the column-slice kernel, which lays the lists out itself
 * (csrc/dsgd_cs.hpp: dsgd_cs_request_kernel); measured SLOWER (126 us: the set-up is a latency chain), so off by default. */
int dsgd_sync_step(dsgd_ctx* ctx, const int32_t* const* idx_per_worker, const int64_t* n_per_worker,
                   int32_t n_workers, float lr, dsgd_batch_stats* stats /* may be NULL */);

/* Same, for batches that are whole contiguous row ranges (batch-size >= split size makes
 * slice(0, B) of the shuffled split the entire split; a sum does not depend on the order).
 * By the rows of the step: 512 .. 98,303 COLUMN LISTS (csrc/dsgd_tcol.hpp: the ranges' entries sorted by column once
 * per configuration; dot kernel -> one bit per row -> column-wise exact sums: no per-workgroup partials; DSGD_TCOL_MIN /
 * _MAX); beyond them ROW CHUNKS (csrc/dsgd_fstep.hpp: cold x.w, hot tiles and cold gradient of a chunk of rows in ONE
 * launch; DSGD_FSTEP_MIN); below 512 the row-wise kernel.  Layouts are cached per (ranges) configuration (eight each).
 * Every path accumulates the same fixed-point integers: same gate decisions => the same bits.                          */
int dsgd_sync_step_ranges(dsgd_ctx* ctx, const int64_t* row_begin, const int64_t* row_end, int32_t n_workers,
                          float lr, dsgd_batch_stats* stats /* may be NULL */);

/* Asynchronous launch variants: enqueue on the context's stream and return; results/errors are
 * collected by dsgd_synchronize().  steps x workers index lists live in a resident plan so that
 * no host->device traffic happens between steps (timed loops, hipGraph replay).                 */
typedef struct dsgd_plan dsgd_plan;
/* idx: concatenation of all lists; offsets: n_steps * n_workers + 1 prefix offsets into idx.
 * A plan is RESIDENT and laid out for the device WHEN IT IS CREATED (data and dimSparsity present; otherwise at its first
 * run), by the device itself, on a stream of its own beside the launch stream: dsgd_plan_create returns once the lists are
 * staged and one 16-byte read-back has fixed the layout's strides; the rest of the set-up overlaps whatever the launch
 * stream is running (the next epoch's plan can be created while this epoch's steps run).  Steps of the reference's own size
 * (<= 8 hosted workers, <= 1,024 rows per step: application.conf:15,27) become COLUMN SLICES -- dsgd_plan_run then runs
 * ALL steps of [step_begin, step_end) in ONE persistent launch (5 us per 3 x 100 step, 9 us per 4 x 200; a launch of a
 * single step 11 us: hand over as many steps per call as are known); larger steps are laid out over the device's streams
 * (16 bytes per 8 non-zeros) and, up to DSGD_VT_PACK_MB (default 2048), copied in that order: two launches per step
 * (18 us at 4,096 rows).  One epoch of Master.fit (core/Master.scala:179-199) = one plan: the device blocks of a destroyed
 * plan are kept by the context for the next one (DSGD_CACHE_MB, default 8192), dsgd_plan_destroy does not synchronise.
 * The same lists through dsgd_sync_step (per request) cost 40 us per call.                                            */
int dsgd_plan_create(dsgd_ctx* ctx, const int32_t* idx, const int64_t* offsets, int64_t n_steps, int32_t n_workers,
                     dsgd_plan** out);
/* The same with the length of idx stated: DSGD_EINVAL unless offsets[n_steps * n_workers] == n_idx -- the form a binding
 * that holds idx as a managed array must use (the lists are read up to offsets[last]: a JVM array shorter than that
 * would be read past its end).  The JNI shim, the Python binding and include/dsgd.hpp all go through this one.       */
int dsgd_plan_create_n(dsgd_ctx* ctx, const int32_t* idx, int64_t n_idx, const int64_t* offsets, int64_t n_steps,
                       int32_t n_workers, dsgd_plan** out);
/* One EPOCH of Master.fit as a plan whose lists are DRAWN BY THE DEVICE, draw for draw the reference's random stream
 * (core/Master.scala:184: for every batch every worker's whole split is reshuffled -- scala.util.Random.shuffle over
 * java.util.Random -- and sliced; 1.38 G draws per epoch of RCV1 at full = true, which a host reproduces in 0.19 s on 32
 * threads while the epoch's 2,146 steps run in 10 ms).  *jstate is java.util.Random's 48-bit internal state (what
 * `new java.util.Random(seed)` holds: (seed ^ 0x5DEECE66D) & (2^48 - 1)) in front of the epoch's first draw; on success it
 * is the state behind the last draw of the steps emitted (*draws_out raw values later), exactly as the JVM's generator
 * would stand.  split k = rows [split_begin[k], split_end[k]) (SplitStrategy.vanilla, the caller's); the steps are those of
 * `0 until max_samples by batch_size` up to the first one that hands some worker an EMPTY slice (*n_steps_out of them: the
 * reference's slave throws there, math/Vec.scala:129; 0 steps: *out = NULL).  csrc/dsgd_shuffle.hpp: the raw stream is
 * scanned for rejection candidates by every lane of the device at once, the host walks the ~10^5 candidates (the one
 * sequential part), and every (batch, worker) list is traced backwards through its Fisher-Yates by one workgroup straight
 * into the plan's index buffer.  DSGD_EUNSUPPORTED (nothing drawn, *jstate untouched): batch_size > 1,024, a split of more
 * than 2^20 rows, or a stream outside the device form's limits -- draw the lists on the host then (the Python / C++ host
 * mirrors do: csrc/jrand.c) and use dsgd_plan_create_n.  tests/test_gpu_shuffle.py: equal to csrc/jrand.c entry for entry. */
int dsgd_plan_create_from_seed(dsgd_ctx* ctx, uint64_t* jstate, const int64_t* split_begin, const int64_t* split_end,
                               int32_t n_splits, int64_t max_samples, int32_t batch_size, dsgd_plan** out,
                               int64_t* n_steps_out, int64_t* draws_out);
/* the lists of a plan as the device holds them: the first n entries and the first n_offsets prefix offsets (tests) */
int dsgd_plan_read_lists(dsgd_ctx* ctx, dsgd_plan* plan, int32_t* idx_out, int64_t n, int64_t* offsets_out, int64_t n_offsets);
int dsgd_plan_destroy(dsgd_ctx* ctx, dsgd_plan* plan);
/* The device blocks destroyed plans leave with the context (up to DSGD_CACHE_MB, default 8192 MiB, read at dsgd_create;
 * a block no plan took again within ~5 plans is freed by itself): give back all but keep_bytes of them now (blocks the
 * launch stream is still reading stay); *held_out (may be NULL) = bytes still held.  For hosts that run several
 * contexts on one GPU (the dev role's JVM workers, Main.scala:144-158) and want the memory between fits.            */
int dsgd_cache_trim(dsgd_ctx* ctx, int64_t keep_bytes, int64_t* held_out);
int dsgd_plan_run(dsgd_ctx* ctx, dsgd_plan* plan, int64_t step_begin, int64_t step_end, float lr);
/* How a plan will run (nothing in the reference; benchmarks, tests): vals[0] = 1 column slices (dsgd_cs_step_kernel),
 * 2 the one-workgroup kernel, 3 virtual tiles, 4 the row-parallel kernels, 0 not laid out yet; [1] slices, [2..4] slot /
 * row / column-list strides, [5] slots per lane, [6] 1 if the device laid it out, [7] words per step of the record.
 * n <= 8 slots.                                                                                                       */
int dsgd_plan_info(dsgd_ctx* ctx, dsgd_plan* plan, int32_t* vals, int32_t n);
/* Parity aid for LONG synchronous runs (nothing in the reference; tests/test_gpu_cs_device.py, tests/test_gpu_host.py, tests/test_sync_replay.py, bench.py): with the record
 * on, every step a column-slice plan runs leaves the GATE DECISION of each of its rows (bit r of the step's words: row r
 * of the step -- workers in order, each worker's list in order -- had y (x . w) >= 0, core/ml/SparseSVM.scala:27-28) and
 * the regulariser scalar s = 2 lambda (w . ds) it used (SparseSVM.scala:31).  fp32 against fp64 decides a row on the gate
 * differently once in a few thousand steps and a constant-step run then goes its own way; with the engine's decisions on
 * record the oracle REPLAYS the trajectory exactly (oracle/sync_replay.py): the final weights must agree to rounding, and
 * every decision that differs from the oracle's own must be a row whose margin lies inside the fp32 bound.
 * dsgd_plan_record(on = 0) drops the record.  dsgd_plan_read_record copies the words of steps [step_begin, step_end)
 * (*mask_words_out words each; gate_mask / s_used may be NULL to query the width).                                    */
int dsgd_plan_record(dsgd_ctx* ctx, dsgd_plan* plan, int32_t on);
int dsgd_plan_read_record(dsgd_ctx* ctx, dsgd_plan* plan, int64_t step_begin, int64_t step_end, uint32_t* gate_mask,
                          float* s_used, int32_t* mask_words_out);
int dsgd_sync_step_ranges_async(dsgd_ctx* ctx, const int64_t* row_begin, const int64_t* row_end, int32_t n_workers,
                                float lr);
int dsgd_synchronize(dsgd_ctx* ctx, dsgd_batch_stats* stats_accum /* may be NULL */);

/* ---- evaluation --------------------------------------------------------------------------
 * SlaveImpl.forward (core/Slave.scala:129-140): pred_i = -signum(x_i . w) in {-1, 0, +1}       */
int dsgd_forward(dsgd_ctx* ctx, const float* w /* or NULL */, const int32_t* idx, int64_t n, float* pred_out /* n */);

/* Master.localLoss / localAccuracy (core/Master.scala:100-107; core/ml/SparseSVM.scala:16-23)
 * over rows [row_begin, row_end): loss = lambda*|w|^2 + mean_i max(0, 1 - y_i p_i), acc =
 * mean_i [p_i == y_i].  counts (may be NULL) gets the exact integer tallies
 * {#p==y, #p==0, #p==-y}.  With a communicator attached, tallies are summed over ranks.        */
int dsgd_loss_acc(dsgd_ctx* ctx, const float* w /* or NULL */, int64_t row_begin, int64_t row_end, double* loss,
                  double* acc, int64_t* counts /* 3 or NULL */);

/* ---- asynchronous ("Hogwild") path ---------------------------------------------------------
 * one iteration of Slave.asyncTask (core/Slave.scala:92-101) on the resident weights with the
 * given sample list: grad = MEAN_i backward; delta = lr * regularize(grad, w); w -= delta.
 * delta_out (D+1, may be NULL) receives what Slave.scala:103-105 would gossip.                 */
int dsgd_async_step(dsgd_ctx* ctx, const int32_t* idx, int64_t n, float lr, float* delta_out,
                    dsgd_batch_stats* stats /* may be NULL */);

/* SlaveImpl.updateGrad / MasterAsync.updateGrad / GradState.update
 * (core/Slave.scala:177-185, core/MasterAsync.scala:164-177, core/ml/GradState.scala:8):
 * w[key[i]] -= dv[i].  Keys as in the wire message Sparse.map (proto.proto:28-31); a key outside [0, D] fails with
 * DSGD_ERANGE before anything is applied.  May be called WHILE the lock-free engine runs (the reference's handler runs
 * concurrently with asyncTask): the update is applied with atomic adds on a side stream and folded into the engine's
 * regulariser scalar; the call returns when it has been applied and never waits for the engine.                    */
int dsgd_update_grad(dsgd_ctx* ctx, const int32_t* key, const float* dv, int64_t nnz);

/* SlaveImpl.startAsync (core/Slave.scala:159-175) for n_workers lock-free workers sharing ONE
 * device-resident weight vector: every worker (a workgroup) loops
 *   draw `batch` of its assigned rows -> mean gated gradient on a snapshot -> regularize ->
 *   atomicAdd(w[j], -lr * g_j)
 * until dsgd_async_stop or until max_updates mini-batch updates have been applied in total
 * (MasterAsync counts UPDATES, maxSteps = N * maxEpochs: core/MasterAsync.scala:83,171).
 * assigned_begin/end: worker k samples rows [assigned_begin[k], assigned_end[k]) (the
 * SplitStrategy.vanilla ranges); sampling is `shuffle take batch` (Slave.scala:87) or a single
 * uniform draw when batch == 1 (Slave.scala:84).  positional_bug != 0 reproduces
 * Slave.scala:87's indexing of `data` by POSITION (rows 0 .. n_k-1) instead of by assigned id. */
int dsgd_async_start(dsgd_ctx* ctx, const int64_t* assigned_begin, const int64_t* assigned_end, int32_t n_workers,
                     int32_t batch, float lr, int64_t max_updates, uint64_t seed, int32_t positional_bug);
int dsgd_async_updates(dsgd_ctx* ctx, int64_t* updates, int32_t* running);
/* Asynchronous mode ACROSS GPUs: with a communicator attached, every context runs its own
 * single-w engine on its own rows and the replicas exchange updates every `every_updates` LOCAL mini-batch updates:
 * one all-reduce of what each replica subtracted since the last exchange, after which every replica subtracts its
 * peers' part -- the batched form of the reference's gossip (core/Slave.scala:103-105 sends every update to every
 * peer, :177-185 / core/MasterAsync.scala:164-177 subtract it).  0 (default) = no exchange.  Every rank must use
 * the same period and the same finite max_updates (the ranks enqueue the same number of collectives).            */
int dsgd_async_set_exchange(dsgd_ctx* ctx, int64_t every_updates);
int dsgd_async_stop(dsgd_ctx* ctx); /* SlaveImpl.stopAsync, core/Slave.scala:187-195 */
/* Measurement aid (nothing in the reference).  counters (4, may be NULL): mini-batch updates applied, rows whose
 * gradient was computed, rows the gate let through, lane-level atomicAdd(w[j], -delta_j) performed (SURVEY.md 8(d):
 * "additionally report atomics/s").  s_engine / s_exact (may be NULL): the engine's incrementally kept regulariser scalar
 * s = 2 lambda (w . ds) as the device holds it (one atomic add per mini-batch and per dsgd_update_grad call, re-derived
 * from the weights every few thousand iterations), and the same quantity recomputed from the weights as they are now.
 * While the engine runs the two differ by the updates in flight; counters [1..3] are flushed by every worker each 16 of
 * its iterations and when it leaves: exact once the engine is joined, up to 16 mini-batches per worker behind before
 * (counters[0], what MasterAsync polls, is exact at any time).                                                      */
int dsgd_async_stats(dsgd_ctx* ctx, int64_t* counters, double* s_engine, double* s_exact);
int dsgd_async_wait(dsgd_ctx* ctx); /* block until max_updates reached */
/* Parity aid for the MANY-worker lock-free engine (nothing in the reference; tests/test_gpu_hogwild_trace.py, bench.py):
 * with a trace of `capacity` records attached (0 detaches it), every mini-batch update of the following engine runs
 * leaves one record at index (its commit number - 1): the worker that made it, that worker's iteration number (the key
 * of the engine's replayable sampler), the update count its weights were read at, the regulariser
 * scalar s = 2 lambda (w . ds) it used, and the GATE DECISIONS of its mini-batch (bit t of the mask = row t of the
 * sample was active, core/ml/SparseSVM.scala:27-28).  A constant-step lock-free run is chaotic -- no replay that
 * re-decides the gates can follow it -- but with the engine's own decisions on record the oracle recomputes every update
 * exactly (oracle/hogwild_replay.py): the final weights must agree to rounding, and the recorded decisions are held
 * to the margins of the replayed weights at `read_at`.  A many-worker parity statement that can fail.
 * dsgd_async_read_trace copies the first min(n, recorded) records of the LAST run out (engine joined); gate_mask holds
 * *mask_words_out = ceil(batch / 32) words per record; *n_out = records available (both may be NULL; n = 0 queries them). */
int dsgd_async_set_trace(dsgd_ctx* ctx, int64_t capacity);
int dsgd_async_read_trace(dsgd_ctx* ctx, int32_t* worker, uint32_t* iteration, int64_t* read_at, float* s_used,
                          int32_t* n_active, uint32_t* gate_mask, int64_t n, int64_t* n_out, int32_t* mask_words_out);
/* ... and what every recorded decision was TAKEN ON (round 6): dots holds *batch_out floats per record, entry t = the
 * fp32 x . w row t of the update's sample was gated on; seen_from[i] <= read_at[i] is an update count read (returning
 * atomic) before the iteration requested any weight: every update with a commit number <= seen_from had fully landed in
 * the weights the iteration saw (0 for a launch's first iteration: only the run's starting weights are known to be in).
 * With these the oracle checks EVERY decision of a run of any worker count (oracle/hogwild_replay.gate_check_recorded_dots):
 * decision == !(y d < 0) for the recorded d (core/ml/SparseSVM.scala:27-28), and d inside the range x . w can take over
 * the replayed weights between seen_from and the updates still in flight at the commit (core/Slave.scala:92).       */
int dsgd_async_read_trace_dots(dsgd_ctx* ctx, int64_t* seen_from, float* dots, int64_t n, int32_t* batch_out);

/* ---- multi-GPU (one process per GPU) ---------------------------------------
 * The synchronous master's aggregate (core/Master.scala:190-194: Future.sequence barrier +
 * Vec.mean) becomes ONE ncclAllReduce(sum, float, D+1) over xGMI on the context's stream.
 * unique_id is the 128-byte ncclUniqueId produced on rank 0 and distributed by the host
 * (torch.distributed / MPI / files).                                                           */
#define DSGD_UNIQUE_ID_BYTES 128
int dsgd_comm_unique_id(char* id_out /* DSGD_UNIQUE_ID_BYTES */);
int dsgd_comm_init(dsgd_ctx* ctx, const char* unique_id, int32_t world_size, int32_t rank);
int dsgd_comm_destroy(dsgd_ctx* ctx);

/* ---- several GPUs driven by ONE host thread --------------------------------------------------------------------
 * The reference's dev role runs the master and every slave in ONE JVM (Main.scala:144-158); SURVEY.md 8(b) lists the
 * fused step as dsgd_sync_step(ctx, idx_per_dev, n_per_dev, lr).  A collective blocks its caller until every rank has
 * joined, so one thread cannot call the per-context entry points one after the other; these take all the contexts of
 * the node at once (one context per device, every one with its own rows): each context's kernels in front of a
 * collective are enqueued first, then all the collectives inside one ncclGroupStart / ncclGroupEnd, then what follows
 * -- the kernels, sums and summation order of N processes with one GPU each (replicas bit-identical).
 * Arrays over workers are context-major: worker k of context i at [i * workers_per_ctx + k].                       */
int dsgd_comm_init_all(dsgd_ctx* const* ctxs, int32_t n_ctx);   /* rank i = ctxs[i]; replaces dsgd_comm_unique_id + dsgd_comm_init */
/* column ranking + dimSparsity with both counts summed over the contexts (Main.scala:54-65 over the whole train set)  */
int dsgd_build_dim_sparsity_devices(dsgd_ctx* const* ctxs, int32_t n_ctx, const int64_t* n_train_per_ctx);
int dsgd_sync_step_devices(dsgd_ctx* const* ctxs, int32_t n_ctx, const int32_t* const* idx_per_worker,
                           const int64_t* n_per_worker, int32_t workers_per_ctx, float lr, dsgd_batch_stats* stats /* summed; may be NULL */);
int dsgd_sync_step_ranges_devices(dsgd_ctx* const* ctxs, int32_t n_ctx, const int64_t* row_begin, const int64_t* row_end,
                                  int32_t workers_per_ctx, float lr, dsgd_batch_stats* stats /* summed; may be NULL */);
/* Master.localLoss / localAccuracy over rows [row_begin[i], row_end[i]) of every context, tallies summed               */
int dsgd_loss_acc_devices(dsgd_ctx* const* ctxs, int32_t n_ctx, const int64_t* row_begin, const int64_t* row_end, double* loss,
                          double* acc, int64_t* counts /* 3 or NULL */);

/* ---- introspection for benchmarks ---------------------------------------------------------
 * average device time (ms) of the dominant gradient kernel over its launches since the last
 * reset, measured with HIP events on the launch stream; n_launches may be NULL.               */
int dsgd_prof_enable(dsgd_ctx* ctx, int32_t on /* 0 off, 1 all profiled kernels, 2 the dominant kernel only */);
int dsgd_prof_read(dsgd_ctx* ctx, double* grad_kernel_ms_avg, int64_t* n_launches, int32_t reset);
/* Split layout only: average duration (ms) and launch count of {main gradient kernel, cold x.w kernel, cold
 * gradient kernel} since the last reset of dsgd_prof_read.  Measurement aid; nothing in the reference. */
int dsgd_prof_read_kinds(dsgd_ctx* ctx, double* ms_avg3, int64_t* n_launches3);

/* Non-zeros of rows [row_begin, row_end) as held internally (empty rows count one explicit zero), and how many of
 * them live in the cold stream of the split layout (0 in the other layouts).  Used by bench.py to attribute the
 * algorithmic bytes of a step to the kernel that reads them.  Nothing in the reference. */
int dsgd_range_nnz(dsgd_ctx* ctx, int64_t row_begin, int64_t row_end, int64_t* nnz, int64_t* cold_nnz);

/* Tuning state that changes numerics or layout, for benchmark records and the derived error bound of the tests:
 * vals[0] = layout generation (4 = matrix split by column rank, the only one), [1] = hot/cold split rank, [2] =
 * fixed-point shift of the last gradient launch (whole ranges and index lists alike; not the one-workgroup plan
 * kernel, which derives 30 - ceil(log2 batch) per batch), [3] = 1 if the cold stream is packed, [4] = 1 if small
 * batches run in the persistent plan kernel, [5] = 1 if the data-dependent fixed-point bound is enabled, [6] = how often
 * the row chunks of the last chunked launch's configuration have been re-cut by their workgroups' measured durations (0-2;
 * only with DSGD_FSTEP_REBALANCE=1: measured at 1 %, off by default -- the re-cut moves rows between workgroups, it changes no bit of any sum).
 * n = number of slots the caller provides (<= 7).                                                                 */
int dsgd_tuning_info(dsgd_ctx* ctx, int32_t* vals, int32_t n);

/* Layout introspection (tests of the multi-GPU path): the internal frequency rank of every key, D + 1 entries.  With a
 * communicator attached the ranking is derived from the column counts summed over the ranks: identical on all of them. */
int dsgd_column_ranks(dsgd_ctx* ctx, int32_t* rank_of_key /* D+1 */);

/* Tuning aid (DSGD_PLAN_PROF=1 in the environment at dsgd_create): shader-clock cycles thread 0 of the small-batch
 * plan kernel spent in the nine phases of a batch ([0..8]: gather+dot, barrier, gate+tables, barrier, scatter,
 * barrier, requests, sweep, barrier+collect) and the number of steps ([15]), accumulated since the last reset;
 * 16 words, all zeros when the aid is off.                                                                       */
int dsgd_debug_cycles(dsgd_ctx* ctx, uint64_t* out16, int32_t reset);

/* name of the gradient kernel variant in use (for matching rocprofv3 kernel-trace rows)        */
const char* dsgd_grad_kernel_name(dsgd_ctx* ctx);
/* raw device pointers (float[D+1], the library's internal column order: dsgd_column_ranks) for hosts that own the
 * collective (e.g. torch.distributed).  The call brings the resident weights into that vector (a column-slice run keeps
 * them slice-major elsewhere until some entry point needs them): *w_dev is current for work enqueued on *stream until the
 * next dsgd_plan_run / dsgd_sync_step; ask again after those.                                                          */
int dsgd_device_ptrs(dsgd_ctx* ctx, void** w_dev, void** g_dev, void** stream);

/* ---- K8: dense logistic mini-batch step (BASELINE.json configs[4]) ------------------------------------------
 * NO REFERENCE COUNTERPART: the reference has one model, the sparse hinge "SVM" (core/ml/SparseSVM.scala:11;
 * Main.scala:67 "could use another model").  This is the optional dense variant north_star names: a second, small
 * object next to dsgd_ctx.  X is n_rows x D fp32 row-major in HBM (D a multiple of 512, <= 8192), labels in {0, 1}:
 *   z = X w, p = sigmoid(z), loss = mean(softplus(z) - y z), g = X^T (p - y) / B, w <- w - lr g
 * over the contiguous rows [row_begin, row_end) of the (pre-shuffled) shard -- one mini-batch.  With a communicator
 * the gradient sums are all-reduced and B is the global batch.  Oracle: oracle/dense_ref.py (fp64, pinned by finite
 * differences -- parity unpinned by construction).                                                                */
typedef struct dsgd_dense dsgd_dense;
int dsgd_dense_create(int32_t n_features, int32_t device, dsgd_dense** out);
int dsgd_dense_destroy(dsgd_dense* d);
/* synthetic shard generated on the device: x ~ N(0,1)/sqrt(D), y = [x . w* + noise > 0]; the planted w* is the same
 * for every seed (one problem, many shards), the seed (= rank) varies the rows and the label noise                  */
int dsgd_dense_generate(dsgd_dense* d, int64_t n_rows, uint64_t seed);
/* host-provided data (tests): X n_rows x D row-major, y n_rows                                                    */
int dsgd_dense_load(dsgd_dense* d, int64_t n_rows, const float* X, const float* y);
int dsgd_dense_set_weights(dsgd_dense* d, const float* w /* D */);
int dsgd_dense_get_weights(dsgd_dense* d, float* w_out /* D */);
/* one mini-batch step, enqueued on the object's stream (dsgd_dense_synchronize collects)                          */
int dsgd_dense_step(dsgd_dense* d, int64_t row_begin, int64_t row_end, float lr);
int dsgd_dense_synchronize(dsgd_dense* d);
/* mean logistic loss and accuracy of the resident weights over rows [row_begin, row_end) (no update)               */
int dsgd_dense_loss(dsgd_dense* d, int64_t row_begin, int64_t row_end, double* loss, double* acc);
int dsgd_dense_comm_init(dsgd_dense* d, const char* unique_id, int32_t world_size, int32_t rank);
/* average device time (ms) of the step kernel since the last reset (HIP events on the object's stream)             */
int dsgd_dense_prof(dsgd_dense* d, int32_t enable, double* kernel_ms_avg, int64_t* n_launches);

#ifdef __cplusplus
}
#endif
#endif /* DSGD_H */
