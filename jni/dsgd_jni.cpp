// JNI shim between the reference's Scala classes and libdsgd_hip (include/dsgd.h).
//
// Build on a box with a JDK:
//   g++ -std=c++17 -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../include
//       dsgd_jni.cpp -o libdsgd_jni.so -L../distributed-sgd_amd/lib -ldsgd_hip        (one command line)
// This image has no JDK; tests/test_jni_shim.py compiles this file against tests/jni_stub/jni.h (a stand-in that
// declares the slice of the JNI API used here), checks the exported symbol set against the @native declarations of
// scala/NativeSVM.scala and drives the entry points with a recording fake JNIEnv.
//
// Symbol names.  The natives are declared in `object NativeSVM` (scala/NativeSVM.scala), i.e. on the JVM class
// `epfl.distributed.core.ml.NativeSVM$`; JNI mangles '$' as `_00024`, and a method of a Scala object is an INSTANCE
// method of the module singleton, so every entry point takes (JNIEnv*, jobject self, ...).
//
// Array passing.  Arrays are taken with Get<Type>ArrayElements / Release<Type>ArrayElements (which may pin or copy
// and put no restriction on what runs in between), never with GetPrimitiveArrayCritical: every dsgd_* call below
// takes the context mutex and blocks on the GPU, which is exactly what a JNI critical region must not do (the
// reference calls its model from an 8-thread pool, utils/Pool.scala:13 -- a thread parked inside a critical region
// stalls every collector).  Array lengths are read before any array is taken.
//
// Error mapping (include/dsgd.h): DSGD_EINVAL -> IllegalArgumentException (what `require` throws at
// math/Vec.scala:129 and math/Sparse.scala:16), DSGD_ERANGE -> IndexOutOfBoundsException
// (math/Sparse.scala:63), everything else -> RuntimeException.
#include <jni.h>

#include <vector>

#include "dsgd.h"

#define NATIVE(name) Java_epfl_distributed_core_ml_NativeSVM_00024_##name

namespace {
jint raise(JNIEnv* env, int rc) {
  const char* cls = rc == DSGD_EINVAL   ? "java/lang/IllegalArgumentException"
                    : rc == DSGD_ERANGE ? "java/lang/IndexOutOfBoundsException"
                                        : "java/lang/RuntimeException";
  env->ThrowNew(env->FindClass(cls), dsgd_last_error());
  return rc;
}
inline dsgd_ctx* ctx(jlong h) { return reinterpret_cast<dsgd_ctx*>(h); }

// RAII views of primitive arrays.  `mode` 0 copies changes back (outputs), JNI_ABORT discards them (inputs).
// (each view keeps its TYPED array reference: in the JDK's jni.h jlongArray, jintArray, ... are distinct classes derived
// from _jarray, and Get/Release<Type>ArrayElements take exactly their own)
#define DSGD_ELEMS(Name, T, A)                                                                       \
  struct Name##Elems {                                                                               \
    JNIEnv* env;                                                                                     \
    A arr;                                                                                           \
    T* p;                                                                                            \
    jint mode;                                                                                       \
    Name##Elems(JNIEnv* e, A a, jint m)                                                              \
        : env(e), arr(a), p(a ? e->Get##Name##ArrayElements(a, nullptr) : nullptr), mode(m) {}       \
    ~Name##Elems() {                                                                                 \
      if (p) env->Release##Name##ArrayElements(arr, p, mode);                                        \
    }                                                                                                \
    Name##Elems(const Name##Elems&) = delete;                                                        \
    Name##Elems& operator=(const Name##Elems&) = delete;                                             \
  };
DSGD_ELEMS(Long, jlong, jlongArray)
DSGD_ELEMS(Int, jint, jintArray)
DSGD_ELEMS(Float, jfloat, jfloatArray)
DSGD_ELEMS(Byte, jbyte, jbyteArray)
#undef DSGD_ELEMS
}  // namespace

extern "C" {

// new SparseSVM(lambda, dimSparsity) + the data array given to `new Slave(...)` (Main.scala:68,138,148)
JNIEXPORT jlong JNICALL NATIVE(create)(JNIEnv* env, jobject, jint nFeatures, jdouble lambda, jint device) {
  dsgd_config cfg{};
  cfg.n_features = nFeatures;
  cfg.device = device;
  cfg.lambda = lambda;
  dsgd_ctx* c = nullptr;
  int rc = dsgd_create(&cfg, &c);
  if (rc) {
    raise(env, rc);
    return 0;
  }
  return reinterpret_cast<jlong>(c);
}

JNIEXPORT void JNICALL NATIVE(destroy)(JNIEnv*, jobject, jlong h) { dsgd_destroy(ctx(h)); }

// Array[(Vec, Int)] flattened by the Scala side to CSR (utils/Dataset.scala:11)
JNIEXPORT void JNICALL NATIVE(loadCsr)(JNIEnv* env, jobject, jlong h, jlongArray rowPtr, jintArray col, jfloatArray val,
                                       jbyteArray label) {
  const jsize nRows = env->GetArrayLength(label);
  int rc;
  {
    LongElems rp(env, rowPtr, JNI_ABORT);
    IntElems c(env, col, JNI_ABORT);
    FloatElems v(env, val, JNI_ABORT);
    ByteElems y(env, label, JNI_ABORT);
    rc = dsgd_load_csr(ctx(h), nRows, reinterpret_cast<const int64_t*>(rp.p), reinterpret_cast<const int32_t*>(c.p), v.p,
                       reinterpret_cast<const int8_t*>(y.p));
  }
  if (rc) raise(env, rc);
}

// Main.scala:54-65 on the device
JNIEXPORT void JNICALL NATIVE(buildDimSparsity)(JNIEnv* env, jobject, jlong h, jlong nTrain) {
  int rc = dsgd_build_dim_sparsity(ctx(h), nTrain, nullptr);
  if (rc) raise(env, rc);
}

// SlaveImpl.gradient (core/Slave.scala:142-157); w may be null = use the resident weights.
// Returns the number of active samples so that the caller can counter.increment(n) (Slave.scala:145,150).
JNIEXPORT jlong JNICALL NATIVE(gradient)(JNIEnv* env, jobject, jlong h, jfloatArray w, jintArray idx, jfloatArray gOut) {
  dsgd_batch_stats st{};
  const jsize n = env->GetArrayLength(idx);
  int rc;
  {
    FloatElems wv(env, w, JNI_ABORT);
    IntElems iv(env, idx, JNI_ABORT);
    FloatElems gv(env, gOut, 0);
    rc = dsgd_gradient(ctx(h), wv.p, reinterpret_cast<const int32_t*>(iv.p), n, gv.p, &st);
  }
  if (rc) raise(env, rc);
  return st.n_active;
}

// SlaveImpl.forward (core/Slave.scala:129-140)
JNIEXPORT void JNICALL NATIVE(forward)(JNIEnv* env, jobject, jlong h, jfloatArray w, jintArray idx, jfloatArray predOut) {
  const jsize n = env->GetArrayLength(idx);
  int rc;
  {
    FloatElems wv(env, w, JNI_ABORT);
    IntElems iv(env, idx, JNI_ABORT);
    FloatElems pv(env, predOut, 0);
    rc = dsgd_forward(ctx(h), wv.p, reinterpret_cast<const int32_t*>(iv.p), n, pv.p);
  }
  if (rc) raise(env, rc);
}

// Master.fit batch closure (core/Master.scala:184-197) for the workers hosted by this process
JNIEXPORT jlong JNICALL NATIVE(syncStep)(JNIEnv* env, jobject, jlong h, jobjectArray idxPerWorker, jfloat lr) {
  const jsize k = env->GetArrayLength(idxPerWorker);
  // index lists are small (batch-size entries): copied out with GetIntArrayRegion
  std::vector<std::vector<int32_t>> lists(static_cast<size_t>(k));
  std::vector<const int32_t*> ptrs(static_cast<size_t>(k));
  std::vector<int64_t> ns(static_cast<size_t>(k));
  for (jsize i = 0; i < k; ++i) {
    jintArray a = static_cast<jintArray>(env->GetObjectArrayElement(idxPerWorker, i));
    const jsize n = a ? env->GetArrayLength(a) : 0;
    lists[i].resize(n > 0 ? n : 1);
    if (n > 0) env->GetIntArrayRegion(a, 0, n, reinterpret_cast<jint*>(lists[i].data()));
    ptrs[i] = lists[i].data();
    ns[i] = n;
    if (a) env->DeleteLocalRef(a);
  }
  dsgd_batch_stats st{};
  int rc = dsgd_sync_step(ctx(h), ptrs.data(), ns.data(), k, lr, &st);
  if (rc) raise(env, rc);
  return st.n_active;
}

// ---- an EPOCH of Master.fit as ONE resident plan (core/Master.scala:179-199) ---------------------------------------
// The epoch loop draws `split.map(Random.shuffle(_)).slice(batch, batch + batchSize)` for every batch (:184) -- nothing else
// consumes the generator inside the loop -- so the patched Master.fit draws the epoch's lists first, in the reference's
// own order, and hands them over flattened: idx = all lists concatenated (batch-major, worker-minor), offsets = nSteps *
// nWorkers + 1 prefix offsets.  planRun(0, nSteps) then runs the whole epoch in ONE launch of the column-slice kernel
// (5 us per 3 x 100 batch against 40 us per syncStep call); planCreate lays the lists out on the device beside whatever is
// running, so the next epoch's plan can be created while this epoch's batches run.
JNIEXPORT jlong JNICALL NATIVE(planCreate)(JNIEnv* env, jobject, jlong h, jintArray idx, jlongArray offsets, jint nWorkers) {
  const jsize nOff = env->GetArrayLength(offsets);
  if (nWorkers < 1 || nOff < 1 || (nOff - 1) % nWorkers != 0) {
    env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "offsets must hold nSteps * nWorkers + 1 entries");
    return 0;
  }
  dsgd_plan* plan = nullptr;
  int rc;
  {
    IntElems iv(env, idx, JNI_ABORT);
    LongElems ov(env, offsets, JNI_ABORT);
    // (the stated length: offsets that end beyond the pinned array are refused instead of read)
    rc = dsgd_plan_create_n(ctx(h), reinterpret_cast<const int32_t*>(iv.p), (int64_t)env->GetArrayLength(idx),
                            reinterpret_cast<const int64_t*>(ov.p), (nOff - 1) / nWorkers, nWorkers, &plan);
  }
  if (rc) {
    raise(env, rc);
    return 0;
  }
  return reinterpret_cast<jlong>(plan);
}

// The same epoch with its lists DRAWN BY THE DEVICE, draw for draw scala.util.Random's stream (dsgd_plan_create_from_seed:
// core/Master.scala:184 costs 1.38 G draws per epoch of RCV1 -- seconds on the JVM, 14 ms here).  state = {java.util.Random's
// internal 48-bit seed in front of the epoch (HipSVM reads and writes it by reflection), out: batches emitted, out: raw
// values consumed}; on return state[0] is where the JVM's generator would stand behind the epoch's last shuffle.  Returns 0
// with state[1] = 0 when the first batch already hands a worker an empty slice.  An epoch outside the device form's limits
// raises UnsupportedOperationException: the caller draws the lists itself and uses planCreate.
JNIEXPORT jlong JNICALL NATIVE(planCreateFromSeed)(JNIEnv* env, jobject, jlong h, jlongArray state, jlongArray splitBegin,
                                                   jlongArray splitEnd, jlong maxSamples, jint batchSize) {
  const jsize n = env->GetArrayLength(splitBegin);
  if (env->GetArrayLength(state) < 3 || n < 1 || env->GetArrayLength(splitEnd) != n) {
    env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "state must hold 3 entries, splitBegin / splitEnd one per worker");
    return 0;
  }
  dsgd_plan* plan = nullptr;
  int rc;
  {
    LongElems sv(env, state, 0);
    LongElems bv(env, splitBegin, JNI_ABORT);
    LongElems ev(env, splitEnd, JNI_ABORT);
    uint64_t js = (uint64_t)sv.p[0];
    int64_t n_steps = 0, draws = 0;
    rc = dsgd_plan_create_from_seed(ctx(h), &js, reinterpret_cast<const int64_t*>(bv.p), reinterpret_cast<const int64_t*>(ev.p), (int32_t)n,
                                    (int64_t)maxSamples, (int32_t)batchSize, &plan, &n_steps, &draws);
    if (rc == DSGD_OK) {
      sv.p[0] = (jlong)js;
      sv.p[1] = (jlong)n_steps;
      sv.p[2] = (jlong)draws;
    }
  }
  if (rc == DSGD_EUNSUPPORTED) {
    env->ThrowNew(env->FindClass("java/lang/UnsupportedOperationException"), dsgd_last_error());
    return 0;
  }
  if (rc) {
    raise(env, rc);
    return 0;
  }
  return reinterpret_cast<jlong>(plan);
}

// the batches [stepBegin, stepEnd) of the plan; enqueued -- planSynchronize (or anything that reads the weights) waits
JNIEXPORT void JNICALL NATIVE(planRun)(JNIEnv* env, jobject, jlong h, jlong plan, jlong stepBegin, jlong stepEnd, jfloat lr) {
  int rc = dsgd_plan_run(ctx(h), reinterpret_cast<dsgd_plan*>(plan), stepBegin, stepEnd, lr);
  if (rc) raise(env, rc);
}

// waits for the batches enqueued so far; returns the number of ACTIVE samples among them (errors of the run surface here)
JNIEXPORT jlong JNICALL NATIVE(planSynchronize)(JNIEnv* env, jobject, jlong h) {
  dsgd_batch_stats st{};
  int rc = dsgd_synchronize(ctx(h), &st);
  if (rc) raise(env, rc);
  return st.n_active;
}

JNIEXPORT void JNICALL NATIVE(planDestroy)(JNIEnv* env, jobject, jlong h, jlong plan) {
  int rc = dsgd_plan_destroy(ctx(h), reinterpret_cast<dsgd_plan*>(plan));
  if (rc) raise(env, rc);
}

// the same closure when every worker's batch is its whole split (batch-size >= split size): contiguous row ranges
JNIEXPORT jlong JNICALL NATIVE(syncStepRanges)(JNIEnv* env, jobject, jlong h, jlongArray rowBegin, jlongArray rowEnd,
                                              jfloat lr) {
  const jsize k = env->GetArrayLength(rowBegin);
  if (env->GetArrayLength(rowEnd) != k) {
    env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "rowBegin / rowEnd length mismatch");
    return 0;
  }
  dsgd_batch_stats st{};
  int rc;
  {
    LongElems rb(env, rowBegin, JNI_ABORT);
    LongElems re(env, rowEnd, JNI_ABORT);
    rc = dsgd_sync_step_ranges(ctx(h), reinterpret_cast<const int64_t*>(rb.p), reinterpret_cast<const int64_t*>(re.p), k,
                               lr, &st);
  }
  if (rc) raise(env, rc);
  return st.n_active;
}

// Master.localLoss / localAccuracy (core/Master.scala:100-107): out = {loss, accuracy}
JNIEXPORT void JNICALL NATIVE(lossAcc)(JNIEnv* env, jobject, jlong h, jfloatArray w, jlong rowBegin, jlong rowEnd,
                                       jdoubleArray out) {
  double la[2] = {0, 0};
  int rc;
  {
    FloatElems wv(env, w, JNI_ABORT);
    rc = dsgd_loss_acc(ctx(h), wv.p, rowBegin, rowEnd, &la[0], &la[1], nullptr);
  }
  if (rc) {
    raise(env, rc);
    return;
  }
  env->SetDoubleArrayRegion(out, 0, 2, la);
}

// Slave.asyncTask body (core/Slave.scala:92-101); deltaOut receives what Slave.scala:103-105 gossips
JNIEXPORT void JNICALL NATIVE(asyncStep)(JNIEnv* env, jobject, jlong h, jintArray idx, jfloat lr, jfloatArray deltaOut) {
  const jsize n = env->GetArrayLength(idx);
  int rc;
  {
    IntElems iv(env, idx, JNI_ABORT);
    FloatElems dv(env, deltaOut, 0);
    rc = dsgd_async_step(ctx(h), reinterpret_cast<const int32_t*>(iv.p), n, lr, dv.p, nullptr);
  }
  if (rc) raise(env, rc);
}

// SlaveImpl.updateGrad / MasterAsync.updateGrad (core/Slave.scala:177-185, core/MasterAsync.scala:164-177)
JNIEXPORT void JNICALL NATIVE(updateGrad)(JNIEnv* env, jobject, jlong h, jintArray keys, jfloatArray values) {
  const jsize n = env->GetArrayLength(keys);
  if (env->GetArrayLength(values) != n) {
    env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "keys / values length mismatch");
    return;
  }
  int rc;
  {
    IntElems kv(env, keys, JNI_ABORT);
    FloatElems vv(env, values, JNI_ABORT);
    rc = dsgd_update_grad(ctx(h), reinterpret_cast<const int32_t*>(kv.p), vv.p, n);
  }
  if (rc) raise(env, rc);
}

// SlaveImpl.startAsync (core/Slave.scala:159-175): the persistent lock-free engine on ONE device-resident w
JNIEXPORT void JNICALL NATIVE(asyncStart)(JNIEnv* env, jobject, jlong h, jlongArray assignedBegin, jlongArray assignedEnd,
                                          jint batch, jfloat lr, jlong maxUpdates, jlong seed, jboolean positionalBug) {
  const jsize k = env->GetArrayLength(assignedBegin);
  if (env->GetArrayLength(assignedEnd) != k) {
    env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "assignedBegin / assignedEnd length mismatch");
    return;
  }
  int rc;
  {
    LongElems ab(env, assignedBegin, JNI_ABORT);
    LongElems ae(env, assignedEnd, JNI_ABORT);
    rc = dsgd_async_start(ctx(h), reinterpret_cast<const int64_t*>(ab.p), reinterpret_cast<const int64_t*>(ae.p), k, batch,
                          lr, maxUpdates, static_cast<uint64_t>(seed), positionalBug ? 1 : 0);
  }
  if (rc) raise(env, rc);
}

// mini-batch updates applied so far (MasterAsync counts these: core/MasterAsync.scala:83,171)
JNIEXPORT jlong JNICALL NATIVE(asyncUpdates)(JNIEnv* env, jobject, jlong h) {
  int64_t u = 0;
  int32_t running = 0;
  int rc = dsgd_async_updates(ctx(h), &u, &running);
  if (rc) raise(env, rc);
  return u;
}

// SlaveImpl.stopAsync (core/Slave.scala:187-195)
JNIEXPORT void JNICALL NATIVE(asyncStop)(JNIEnv* env, jobject, jlong h) {
  int rc = dsgd_async_stop(ctx(h));
  if (rc) raise(env, rc);
}

JNIEXPORT void JNICALL NATIVE(asyncWait)(JNIEnv* env, jobject, jlong h) {
  int rc = dsgd_async_wait(ctx(h));
  if (rc) raise(env, rc);
}

JNIEXPORT void JNICALL NATIVE(setWeights)(JNIEnv* env, jobject, jlong h, jfloatArray w) {
  int rc;
  {
    FloatElems wv(env, w, JNI_ABORT);
    rc = dsgd_set_weights(ctx(h), wv.p);
  }
  if (rc) raise(env, rc);
}

JNIEXPORT void JNICALL NATIVE(getWeights)(JNIEnv* env, jobject, jlong h, jfloatArray wOut) {
  int rc;
  {
    FloatElems wv(env, wOut, 0);
    rc = dsgd_get_weights(ctx(h), wv.p);
  }
  if (rc) raise(env, rc);
}

// ---- several GPUs driven by ONE JVM thread (the dev role: master + every slave in one JVM, Main.scala:144-158) ----------
// ctxs: one context per device (create / loadCsr each), rank i = ctxs[i]; arrays over workers are context-major.
namespace {
std::vector<dsgd_ctx*> ctx_list(JNIEnv* env, jlongArray ctxs) {
  const jsize n = env->GetArrayLength(ctxs);
  std::vector<jlong> h(static_cast<size_t>(n > 0 ? n : 0));
  if (n > 0) env->GetLongArrayRegion(ctxs, 0, n, h.data());
  std::vector<dsgd_ctx*> out;
  for (jlong v : h) out.push_back(ctx(v));
  return out;
}
}  // namespace

JNIEXPORT void JNICALL NATIVE(commInitAll)(JNIEnv* env, jobject, jlongArray ctxs) {
  std::vector<dsgd_ctx*> c = ctx_list(env, ctxs);
  int rc = dsgd_comm_init_all(c.data(), static_cast<int32_t>(c.size()));
  if (rc) raise(env, rc);
}

// Main.scala:54-65 over the WHOLE train set: column ranking and feature counts summed over the contexts
JNIEXPORT void JNICALL NATIVE(buildDimSparsityDevices)(JNIEnv* env, jobject, jlongArray ctxs, jlongArray nTrain) {
  std::vector<dsgd_ctx*> c = ctx_list(env, ctxs);
  if (env->GetArrayLength(nTrain) != static_cast<jsize>(c.size())) {
    env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "one nTrain per context");
    return;
  }
  int rc;
  {
    LongElems nt(env, nTrain, JNI_ABORT);
    rc = dsgd_build_dim_sparsity_devices(c.data(), static_cast<int32_t>(c.size()), reinterpret_cast<const int64_t*>(nt.p));
  }
  if (rc) raise(env, rc);
}

// Master.fit batch closure (core/Master.scala:184-197) over every device of the node: idxPerWorker holds
// contexts x workersPerCtx lists; the mean runs over all of them (one all-reduce inside)
JNIEXPORT jlong JNICALL NATIVE(syncStepDevices)(JNIEnv* env, jobject, jlongArray ctxs, jobjectArray idxPerWorker,
                                               jint workersPerCtx, jfloat lr) {
  std::vector<dsgd_ctx*> c = ctx_list(env, ctxs);
  const jsize k = env->GetArrayLength(idxPerWorker);
  if (workersPerCtx < 1 || k != static_cast<jsize>(c.size()) * workersPerCtx) {
    env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "idxPerWorker must hold contexts x workersPerCtx lists");
    return 0;
  }
  std::vector<std::vector<int32_t>> lists(static_cast<size_t>(k));
  std::vector<const int32_t*> ptrs(static_cast<size_t>(k));
  std::vector<int64_t> ns(static_cast<size_t>(k));
  for (jsize i = 0; i < k; ++i) {
    jintArray a = static_cast<jintArray>(env->GetObjectArrayElement(idxPerWorker, i));
    const jsize n = a ? env->GetArrayLength(a) : 0;
    lists[i].resize(n > 0 ? n : 1);
    if (n > 0) env->GetIntArrayRegion(a, 0, n, reinterpret_cast<jint*>(lists[i].data()));
    ptrs[i] = lists[i].data();
    ns[i] = n;
    if (a) env->DeleteLocalRef(a);
  }
  dsgd_batch_stats st{};
  int rc = dsgd_sync_step_devices(c.data(), static_cast<int32_t>(c.size()), ptrs.data(), ns.data(), workersPerCtx, lr, &st);
  if (rc) raise(env, rc);
  return st.n_active;
}

JNIEXPORT jlong JNICALL NATIVE(syncStepRangesDevices)(JNIEnv* env, jobject, jlongArray ctxs, jlongArray rowBegin, jlongArray rowEnd,
                                                     jint workersPerCtx, jfloat lr) {
  std::vector<dsgd_ctx*> c = ctx_list(env, ctxs);
  const jsize k = env->GetArrayLength(rowBegin);
  if (workersPerCtx < 1 || env->GetArrayLength(rowEnd) != k || k != static_cast<jsize>(c.size()) * workersPerCtx) {
    env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "rowBegin / rowEnd must hold contexts x workersPerCtx ranges");
    return 0;
  }
  dsgd_batch_stats st{};
  int rc;
  {
    LongElems rb(env, rowBegin, JNI_ABORT);
    LongElems re(env, rowEnd, JNI_ABORT);
    rc = dsgd_sync_step_ranges_devices(c.data(), static_cast<int32_t>(c.size()), reinterpret_cast<const int64_t*>(rb.p),
                                       reinterpret_cast<const int64_t*>(re.p), workersPerCtx, lr, &st);
  }
  if (rc) raise(env, rc);
  return st.n_active;
}

// Master.localLoss / localAccuracy with the tallies summed over the contexts; out = {loss, accuracy}
JNIEXPORT void JNICALL NATIVE(lossAccDevices)(JNIEnv* env, jobject, jlongArray ctxs, jlongArray rowBegin, jlongArray rowEnd,
                                             jdoubleArray out) {
  std::vector<dsgd_ctx*> c = ctx_list(env, ctxs);
  if (env->GetArrayLength(rowBegin) != static_cast<jsize>(c.size()) || env->GetArrayLength(rowEnd) != static_cast<jsize>(c.size())) {
    env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "one row range per context");
    return;
  }
  double la[2] = {0, 0};
  int rc;
  {
    LongElems rb(env, rowBegin, JNI_ABORT);
    LongElems re(env, rowEnd, JNI_ABORT);
    rc = dsgd_loss_acc_devices(c.data(), static_cast<int32_t>(c.size()), reinterpret_cast<const int64_t*>(rb.p),
                               reinterpret_cast<const int64_t*>(re.p), &la[0], &la[1], nullptr);
  }
  if (rc) {
    raise(env, rc);
    return;
  }
  env->SetDoubleArrayRegion(out, 0, 2, la);
}

}  // extern "C"
