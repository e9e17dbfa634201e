package epfl.distributed.core.ml

import epfl.distributed.math.Vec
import spire.math.Number

/** JVM face of libdsgd_hip (include/dsgd.h) -- the binding a maintainer of zifeo/distributed-sgd adds.
  *
  * `SparseSVM` keeps its constructor and its six methods (core/ml/SparseSVM.scala:11-31); `HipSVM` below is a
  * subclass whose batch-level entry points replace the bodies of `SlaveImpl.gradient` / `forward`
  * (core/Slave.scala:129-157), of the `Slave.asyncTask` iteration and `updateGrad` (core/Slave.scala:79-111,177-185), of
  * `Master.localLoss` / `localAccuracy` (core/Master.scala:100-107) and of the `Master.fit` batch closure
  * (core/Master.scala:179-199) -- a whole epoch of it as ONE resident plan (`fitEpoch`) -- when `dsgd.backend = hip` (a new
  * OPTIONAL key; every existing `dsgd { ... }` key of
  * application.conf is untouched).  scala/patch/dsgd-hip-backend.diff is the patch that wires it in (it adds this file
  * as src/main/scala/epfl/distributed/core/ml/NativeSVM.scala).
  */
object NativeSVM {
  // The natives live on the module class `NativeSVM$`: their JNI names are
  // Java_epfl_distributed_core_ml_NativeSVM_00024_<name>(JNIEnv*, jobject self, ...) -- jni/dsgd_jni.cpp exports
  // exactly these (tests/test_jni_shim.py compares the two lists).
  System.loadLibrary("dsgd_jni") // jni/dsgd_jni.cpp, links libdsgd_hip.so

  @native def create(nFeatures: Int, lambda: Double, device: Int): Long
  @native def destroy(ctx: Long): Unit
  @native def loadCsr(ctx: Long, rowPtr: Array[Long], col: Array[Int], value: Array[Float], label: Array[Byte]): Unit
  @native def buildDimSparsity(ctx: Long, nTrain: Long): Unit
  @native def gradient(ctx: Long, w: Array[Float], idx: Array[Int], gOut: Array[Float]): Long
  @native def forward(ctx: Long, w: Array[Float], idx: Array[Int], predOut: Array[Float]): Unit
  @native def syncStep(ctx: Long, idxPerWorker: Array[Array[Int]], lr: Float): Long
  @native def syncStepRanges(ctx: Long, rowBegin: Array[Long], rowEnd: Array[Long], lr: Float): Long
  // an epoch of Master.fit as ONE resident plan (core/Master.scala:179-199): idx = the epoch's lists concatenated
  // (batch-major, worker-minor), offsets = nSteps * nWorkers + 1 prefix offsets
  @native def planCreate(ctx: Long, idx: Array[Int], offsets: Array[Long], nWorkers: Int): Long
  @native def planCreateFromSeed(ctx: Long, state: Array[Long], splitBegin: Array[Long], splitEnd: Array[Long], maxSamples: Long, batchSize: Int): Long
  @native def planRun(ctx: Long, plan: Long, stepBegin: Long, stepEnd: Long, lr: Float): Unit
  @native def planSynchronize(ctx: Long): Long
  @native def planDestroy(ctx: Long, plan: Long): Unit
  @native def lossAcc(ctx: Long, w: Array[Float], rowBegin: Long, rowEnd: Long, out: Array[Double]): Unit
  @native def asyncStep(ctx: Long, idx: Array[Int], lr: Float, deltaOut: Array[Float]): Unit
  @native def updateGrad(ctx: Long, keys: Array[Int], values: Array[Float]): Unit
  @native def asyncStart(ctx: Long, assignedBegin: Array[Long], assignedEnd: Array[Long], batch: Int, lr: Float,
                         maxUpdates: Long, seed: Long, positionalBug: Boolean): Unit
  @native def asyncUpdates(ctx: Long): Long
  @native def asyncStop(ctx: Long): Unit
  @native def asyncWait(ctx: Long): Unit
  @native def setWeights(ctx: Long, w: Array[Float]): Unit
  @native def getWeights(ctx: Long, wOut: Array[Float]): Unit
  // several GPUs driven by ONE thread of this JVM (dev role: master + every slave in one JVM, Main.scala:144-158): one
  // context per device, rank i = ctxs(i); arrays over workers are context-major (include/dsgd.h, dsgd_*_devices)
  @native def commInitAll(ctxs: Array[Long]): Unit
  @native def buildDimSparsityDevices(ctxs: Array[Long], nTrain: Array[Long]): Unit
  @native def syncStepDevices(ctxs: Array[Long], idxPerWorker: Array[Array[Int]], workersPerCtx: Int, lr: Float): Long
  @native def syncStepRangesDevices(ctxs: Array[Long], rowBegin: Array[Long], rowEnd: Array[Long], workersPerCtx: Int, lr: Float): Long
  @native def lossAccDevices(ctxs: Array[Long], rowBegin: Array[Long], rowEnd: Array[Long], out: Array[Double]): Unit
}

/** Dense float[D+1] indexed by KEY is the exchange format: feature ids are 1-based map keys
  * (utils/Dataset.scala:30), dimSparsity uses 0-based keys (Main.scala:60-62), so both slot 0 and slot D exist. */
object DenseKeys {
  def fromVec(v: Vec): Array[Float] = {
    val a = new Array[Float](v.size + 1)
    v.map.foreach { case (k, n) => a(k) = n.toDouble.toFloat }
    a
  }
  def toVec(a: Array[Float], size: Int): Vec =
    Vec(a.iterator.zipWithIndex.collect { case (x, k) if x != 0f => k -> Number(x.toDouble) }.toMap, size)
}

class HipSVM(lambda: Number, dimSparsity: Vec, data: Array[(Vec, Int)], nTrain: Int, device: Int = 0)
    extends SparseSVM(lambda, dimSparsity) {

  private val dim = data(0)._1.size
  private val ctx = NativeSVM.create(dim, lambda.toDouble, device)

  {
    // Array[(Vec, Int)] -> CSR, columns ascending (the order of the text files, utils/Dataset.scala:19-34)
    val rowPtr = new Array[Long](data.length + 1)
    val cols   = Array.newBuilder[Int]
    val vals   = Array.newBuilder[Float]
    val labels = new Array[Byte](data.length)
    var nnz    = 0L
    data.zipWithIndex.foreach {
      case ((x, y), i) =>
        x.map.toSeq.sortBy(_._1).foreach { case (k, n) => cols += k; vals += n.toDouble.toFloat; nnz += 1 }
        rowPtr(i + 1) = nnz
        labels(i) = y.toByte
    }
    NativeSVM.loadCsr(ctx, rowPtr, cols.result(), vals.result(), labels)
    NativeSVM.buildDimSparsity(ctx, nTrain)
  }

  /** body of SlaveImpl.gradient (core/Slave.scala:142-157) */
  def gradientBatch(w: Vec, samplesIdx: Seq[Int]): Vec = {
    val g = new Array[Float](dim + 1)
    NativeSVM.gradient(ctx, DenseKeys.fromVec(w), samplesIdx.toArray, g)
    DenseKeys.toVec(g, dim)
  }

  /** body of SlaveImpl.forward (core/Slave.scala:129-140) */
  def forwardBatch(w: Vec, samplesIdx: Seq[Int]): Seq[Number] = {
    val p = new Array[Float](samplesIdx.size)
    NativeSVM.forward(ctx, DenseKeys.fromVec(w), samplesIdx.toArray, p)
    p.map(x => Number(x.toDouble))
  }

  /** Master.localLoss / localAccuracy over a row range (core/Master.scala:100-107).  In resident mode the weights on
    * the device ARE the model's weights and are evaluated IN PLACE (a null `w`: dsgd_loss_acc then writes nothing): the
    * `w` a caller holds is a snapshot of them, and writing it back would discard every update the slave threads applied
    * since it was taken (MasterAsync's loss check runs while they do, core/MasterAsync.scala:96-162). */
  def lossAndAccuracy(w: Vec, rowBegin: Int, rowEnd: Int): (Number, Double) = {
    val out = new Array[Double](2)
    NativeSVM.lossAcc(ctx, if (resident) null else DenseKeys.fromVec(w), rowBegin, rowEnd, out)
    (Number(out(0)), out(1))
  }

  /** dev mode (Main.scala "launch: master + slaves"): master and slaves live in ONE JVM and share this object, i.e. ONE
    * device context holds every row and the weights can stay on the device between batches / iterations. */
  @volatile var resident: Boolean = false

  def setWeights(w: Vec): Unit = NativeSVM.setWeights(ctx, DenseKeys.fromVec(w))

  def weights(): Vec = {
    val a = new Array[Float](dim + 1)
    NativeSVM.getWeights(ctx, a)
    DenseKeys.toVec(a, dim)
  }

  /** the whole batch closure of Master.fit (core/Master.scala:184-197) for the workers hosted by this context: per-worker
    * regularised sums, mean over the workers, w <- w - learningRate * mean.  An empty list fails as Vec.sum does. */
  def syncStep(idxPerWorker: Seq[Seq[Int]], learningRate: Double): Long =
    NativeSVM.syncStep(ctx, idxPerWorker.map(_.toArray).toArray, learningRate.toFloat)

  /** ONE EPOCH of Master.fit's batch loop (core/Master.scala:179-199) for the workers hosted by this context, as one resident
    * plan: `batches(b)(k)` = the sample list of worker k in batch b, i.e. what
    * `split.map(Random.shuffle(_)).slice(batch, batch + batchSize)` (:184) yields, drawn by the caller in the reference's own
    * order.  All batches run in ONE launch of the column-slice kernel (5 us per 3 x 100 batch; `syncStep` per batch costs
    * 40 us plus the JNI array traffic); the weights stay on the device.  Returns the number of active samples of the epoch.
    * An empty list fails as Vec.sum does (math/Vec.scala:129) -- before anything runs. */
  def fitEpoch(batches: Seq[Seq[Seq[Int]]], learningRate: Double): Long = {
    val plan = createEpochPlan(batches)
    try runEpochPlan(plan, batches.size, learningRate)
    finally NativeSVM.planDestroy(ctx, plan)
  }

  /** The two halves of `fitEpoch`, for a master that lays the NEXT epoch's plan out while this epoch's batches run
    * (planCreate works on a stream of its own beside the launch stream). */
  def createEpochPlan(batches: Seq[Seq[Seq[Int]]]): Long = {
    require(batches.nonEmpty, "an epoch needs at least one batch")
    val nWorkers = batches.head.size
    require(batches.forall(_.size == nWorkers), "every batch needs one list per worker")
    val total   = batches.iterator.map(_.iterator.map(_.size).sum).sum
    val idx     = new Array[Int](total)
    val offsets = new Array[Long](batches.size * nWorkers + 1)
    var at, i   = 0
    batches.foreach(_.foreach { list =>
      list.foreach { r => idx(at) = r; at += 1 }
      i += 1
      offsets(i) = at
    })
    NativeSVM.planCreate(ctx, idx, offsets, nWorkers)
  }

  /** `fitEpoch` with the epoch's sample lists DRAWN BY THE DEVICE -- draw for draw what
    * `split.map(Random.shuffle(_)).slice(batch, batch + batchSize)` (core/Master.scala:184) would have drawn from `rnd`, which is
    * left exactly where those shuffles would have left it.  The reference reshuffles every worker's whole split for EVERY
    * batch: 1.38 G draws per epoch of RCV1 (seconds on the JVM) for 644 K list entries; the device scans the raw stream,
    * the library walks the rejection candidates, and every list is traced backwards through its Fisher-Yates by one
    * workgroup (14 ms).  `split(k)` must be the contiguous range SplitStrategy.vanilla hands worker k.  Returns the number
    * of batches run -- fewer than `0 until maxSamples by batchSize` holds only if a batch would hand some worker an empty
    * slice (the reference's slave throws there: the caller raises the same error) -- or None when the epoch is outside the
    * device form (batches beyond 1,024 rows, splits beyond 2^20 rows): the caller then draws the lists itself (`fitEpoch`). */
  def fitEpochFromSeed(rnd: java.util.Random, split: Seq[Range], maxSamples: Int, batchSize: Int, learningRate: Double): Option[Int] = {
    if (split.exists(r => r.step != 1 || r.isEmpty)) return None
    val seedField = classOf[java.util.Random].getDeclaredField("seed")   // (an AtomicLong; 48 bits, already scrambled)
    seedField.setAccessible(true)
    val seed  = seedField.get(rnd).asInstanceOf[java.util.concurrent.atomic.AtomicLong]
    val state = Array(seed.get(), 0L, 0L)
    val plan =
      try NativeSVM.planCreateFromSeed(ctx, state, split.map(_.start.toLong).toArray, split.map(r => (r.last + 1).toLong).toArray, maxSamples.toLong, batchSize)
      catch { case _: UnsupportedOperationException => return None }
    seed.set(state(0))
    if (plan != 0L) {
      try runEpochPlan(plan, state(1).toInt, learningRate)
      finally NativeSVM.planDestroy(ctx, plan)
    }
    Some(state(1).toInt)
  }

  def runEpochPlan(plan: Long, nBatches: Int, learningRate: Double): Long = {
    NativeSVM.planRun(ctx, plan, 0, nBatches, learningRate.toFloat)
    NativeSVM.planSynchronize(ctx)
  }

  def destroyEpochPlan(plan: Long): Unit = NativeSVM.planDestroy(ctx, plan)

  /** one iteration of Slave.asyncTask (core/Slave.scala:92-101) on the device-resident weights; returns the update that
    * is gossiped (core/Slave.scala:103-105) */
  def asyncStepBatch(samplesIdx: Seq[Int], learningRate: Double): Vec = {
    val d = new Array[Float](dim + 1)
    NativeSVM.asyncStep(ctx, samplesIdx.toArray, learningRate.toFloat, d)
    DenseKeys.toVec(d, dim)
  }

  /** SlaveImpl.updateGrad / MasterAsync.updateGrad (core/Slave.scala:177-185): w <- w - delta; may arrive while the
    * lock-free engine runs */
  def updateGrad(delta: Vec): Unit = {
    val entries = delta.map.toSeq
    NativeSVM.updateGrad(ctx, entries.map(_._1).toArray, entries.map(_._2.toDouble.toFloat).toArray)
  }

  /** the persistent lock-free engine: `assigned` = one contiguous row range per worker (SplitStrategy.vanilla), all of
    * them updating ONE device-resident weight vector (BASELINE configs[3]); maxUpdates as MasterAsync counts them
    * (core/MasterAsync.scala:83,171) */
  def asyncStart(assigned: Seq[(Int, Int)], batchSize: Int, learningRate: Double, maxUpdates: Long, seed: Long): Unit =
    NativeSVM.asyncStart(ctx, assigned.map(_._1.toLong).toArray, assigned.map(_._2.toLong).toArray, batchSize,
                         learningRate.toFloat, maxUpdates, seed, false)

  def asyncUpdates(): Long = NativeSVM.asyncUpdates(ctx)

  def asyncStop(): Unit = NativeSVM.asyncStop(ctx)

  def asyncWait(): Unit = NativeSVM.asyncWait(ctx)

  def close(): Unit = NativeSVM.destroy(ctx)
}

object HipSVM {
  def isResident(model: SparseSVM): Boolean = model match {
    case h: HipSVM => h.resident
    case _         => false
  }
}
