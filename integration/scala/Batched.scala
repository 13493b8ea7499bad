// Scala shim a maintainer of jonnylaw/bayesian_dlms would add (SOURCE ONLY here: no scalac / sbt / JDK in the build
// container or on the GPU box; tests/test_integration_sources.py checks it against the JNI glue and the C header, and
// tests/cpp/jni_glue_check.cpp drives the glue itself on the GPU).
//
// It offers the reference's own whole-series calls over N series, with the reference's argument order and per-series
// semantics:
//   KalmanFilter.filterDlm(mod, ys, p)             KalmanFilter.scala:291-294   -> Batched#filterDlm
//   Smoothing.backwardsSmoother(mod)(kfStates)     Smoothing.scala:57-64        -> Batched#backwardsSmoother
//   Smoothing.ffbsDlm(mod, ys, p)                  Smoothing.scala:173-180      -> Batched#ffbsDlm
//   SvdFilter.filterDlm(mod, ys, p)                SvdFilter.scala:158-161      -> Batched#svdFilterDlm
//   SvdSampler.ffbsDlm(mod, ys, p)                 SvdSampler.scala:79-82       -> Batched#svdFfbsDlm
//   GibbsSampling.sample / sampleSvd(mod, priorV, priorW, initParams, ys)   Gibbs.scala:165-180,203-217 -> Batched#gibbsSample(Svd)
//   GibbsWishart.sample(mod, priorV, priorW, initParams, ys)                GibbsWishart.scala:65-80    -> Batched#gibbsWishartSample
// plus the fused filter + smoother (Batched#filterSmooth, the metric path) and the log-likelihood.
//
// How data moves.  The model closures are evaluated on the host exactly once per distinct time / time increment,
// Breeze's column-major `DenseMatrix.data` is flattened, `Option[Double]` becomes NaN.  EVERYTHING then lives in
// engine-owned device buffers (Native.bufferAlloc): observations are uploaded once through a bounded direct staging
// buffer (`stagingBytes`, default 64 MiB -- no direct buffer ever approaches the JVM's 2 GB limit, all offsets are Long),
// results stay in HBM (10^4 x 10^3 x d = 13 gives 2 x 14.6 GB of records) and KfState / SmoothingState / SamplingState
// objects are materialised lazily per (series, t) or per series from small downloads: 10^4 x 10^3 eager KfStates would be
// ~10^7 JVM objects (> 30 GB).
package com.github.jonnylaw.dlm.gpu

import java.nio.{ByteBuffer, ByteOrder, DoubleBuffer}
import breeze.linalg.{DenseMatrix, DenseVector}
import com.github.jonnylaw.dlm._

/** One @native method per export of include/dlm_engine.h (integration/jni/dlm_jni.cpp).  Addresses are raw: a device
  * pointer from bufferAlloc, or the address of a direct buffer (Native#address); descriptors are long arrays:
  *   model  = {d, p, T, N, F, fStride, G, nG, gIndex, dt}
  *   params = {V, vStride, W, wStride, m0, m0Stride, C0, c0Stride, vTStride, wTStride}
  *   opts   = {flags, mem, seed, seriesOffset}
  * A class, not an object: the JNI symbols are Java_com_github_jonnylaw_dlm_gpu_Native_<method>. */
final class Native private[gpu] () {
  @native def engineCreate(device: Int): Long
  @native def engineDestroy(h: Long): Unit
  @native def lastError(h: Long): String
  @native def version(): String
  @native def lastVariant(h: Long): String
  @native def engineSetStream(h: Long, stream: Long): Unit
  @native def engineSync(h: Long): Unit
  @native def engineWaitStream(h: Long, stream: Long): Unit
  @native def streamWaitEngine(h: Long, stream: Long): Unit
  @native def lastTiming(h: Long): Array[Double]
  @native def lastCounters(h: Long): Array[Long]
  @native def address(directBuffer: java.nio.Buffer): Long
  @native def bufferAlloc(h: Long, bytes: Long): Long
  @native def bufferFree(h: Long, dev: Long): Unit
  @native def bufferUpload(h: Long, dstDev: Long, dstOffset: Long, srcHost: Long, bytes: Long): Unit
  @native def bufferDownload(h: Long, srcDev: Long, srcOffset: Long, dstHost: Long, bytes: Long): Unit
  @native def bufferFill(h: Long, dstDev: Long, dstOffset: Long, byteValue: Int, bytes: Long): Unit
  @native def deviceMemInfo(h: Long): Array[Long]
  @native def packedRecordDoubles(d: Int): Int
  @native def unpackRecords(h: Long, d: Int, count: Long, packed: Long, opts: Array[Long], dense: Long): Unit
  @native def filter(h: Long, model: Array[Long], params: Array[Long], opts: Array[Long], y: Long, filt: Long, prior: Long, fq: Long, status: Long): Unit
  @native def smooth(h: Long, model: Array[Long], params: Array[Long], opts: Array[Long], filt: Long, smooth: Long, status: Long): Unit
  @native def filterSmooth(h: Long, model: Array[Long], params: Array[Long], opts: Array[Long], y: Long, filt: Long, smooth: Long, status: Long): Unit
  @native def loglik(h: Long, model: Array[Long], params: Array[Long], opts: Array[Long], y: Long, loglik: Long, status: Long): Unit
  @native def simulate(h: Long, model: Array[Long], params: Array[Long], opts: Array[Long], x: Long, y: Long, status: Long): Unit
  /** filtWs = 0L: the forward pass's records are not wanted (the engine keeps them to itself; where the batch shares V, W, C0 on a regular grid it does not produce them) */
  @native def ffbs(h: Long, model: Array[Long], params: Array[Long], opts: Array[Long], y: Long, z: Long, filtWs: Long, theta: Long, cond: Long, stats: Long, status: Long): Unit
  @native def statsLen(d: Int, p: Int, flags: Int): Int
  @native def backwardSample(h: Long, model: Array[Long], params: Array[Long], opts: Array[Long], y: Long, filt: Long, z: Long, theta: Long, cond: Long, stats: Long, status: Long): Unit
  @native def svdFilter(h: Long, model: Array[Long], params: Array[Long], opts: Array[Long], y: Long, svdRec: Long, status: Long): Unit
  @native def svdFfbs(h: Long, model: Array[Long], params: Array[Long], opts: Array[Long], y: Long, z: Long, svdWs: Long, theta: Long, stats: Long, status: Long): Unit
  @native def dinvgammaStep(h: Long, d: Int, p: Int, n: Int, stats: Long, alphaV: Double, betaV: Double, alphaW: Double, betaW: Double, iteration: Long, opts: Array[Long], vOut: Long, wOut: Long): Unit
  @native def ar1Ffbs(h: Long, n: Int, t: Int, y: Long, v: Long, vStride: Long, sv: Long, svStride: Long, z: Long, opts: Array[Long], filt: Long, theta: Long, status: Long): Unit
  @native def ouFfbs(h: Long, n: Int, t: Int, times: Long, y: Long, v: Long, vStride: Long, sv: Long, svStride: Long, z: Long, opts: Array[Long], filt: Long, theta: Long, status: Long): Unit
  @native def statsPool(h: Long, stats: Long, n: Int, l: Int, pooled: Long, opts: Array[Long]): Unit
  @native def commUniqueId(): Array[Byte]
  @native def commInitRank(h: Long, nranks: Int, rank: Int, id: Array[Byte]): Unit
  @native def gibbsSuffstatsAllreduce(h: Long, statsDev: Long, count: Long): Unit
}

object Native {
  lazy val lib: Native = { System.loadLibrary("dlm_jni"); new Native }
  val MemDevice = 0L
  val MemHost = 1L
  // dlm_options.flags (include/dlm_engine.h)
  val SmootherCompatQ1 = 1 << 0   // literal Smoothing.scala:44 (J X J)
  val SvdRawWQ2 = 1 << 1          // literal SvdFilter.filterDlm (raw W as sqrt W)
  val SvdSamplerQ9 = 1 << 2       // literal SvdSampler.step
  val ForceGeneric = 1 << 3
  val StatsOuter = 1 << 4         // GibbsWishart statistics
  val FfbsSimSmooth = 1 << 6      // Durbin-Koopman simulation smoother instead of backward sampling
  val PackedSym = 1 << 7
  val ModelUnchanged = 1 << 8     // a PROMISE: F, G and the time grid are those of this engine's previous call (its structure analysis is reused); the engine verifies it with a device checksum and fails the call if it does not hold
  val CountSteps = 1 << 9         // count the steps that took a short path (Native.lastCounters)
  val LoglikLiteralQ7 = 1 << 11   // logLikelihood: KalmanFilter.likelihood as written (the transition density of the filtered means, KalmanFilter.scala:299-306)
  val TrustModelUnchanged = 1 << 10 // with ModelUnchanged: skip the device checksum (only for callers that compared the tables themselves)
  val SvdPerSeries = 1 << 25      // SVD filter: every series its own decompositions (default: once per call where V, W, C0 are shared; the same bits)
  val SamplerPerSeries = 1 << 26  // ffbs: every series its own J_t, H_t, chol(H_t) (default: once per call where V, W, C0 are shared -- the pooled Gibbs samplers; the same draws, bit for bit)
  val DrawEig = 1 << 27           // ffbs: draw with the reference's own factor, theta = h + E sqrt(Lambda) z from eigSym(H) (MultivariateGaussianSvd.scala:13-22), instead of the Cholesky factor
  val SmootherPerSeries = 1 << 28 // filterSmooth with SmootherCompatQ1: every series its own J_t, S_t (default: once per call where V, W, C0 are shared; the same records, bit for bit)
  val NoSteady = 1 << 22          // every step recomputes the covariance recursion, also once it has settled (the reference's arithmetic, step for step)
  /** every reference quirk switched on: results are the reference's arithmetic, not the textbook's (SURVEY Q1 / Q2 / Q9) */
  val LiteralReference = SmootherCompatQ1 | SvdRawWQ2 | SvdSamplerQ9
}

/** An engine-owned device allocation; `free()` (or closing the owning Batched) releases it. */
final class DeviceBuffer private[gpu] (owner: Batched, val bytes: Long) {
  private[gpu] var ptr: Long = Native.lib.bufferAlloc(owner.handle, bytes)
  def free(): Unit = if (ptr != 0L) { Native.lib.bufferFree(owner.handle, ptr); ptr = 0L }
}

final class Batched(device: Int = 0, stagingBytes: Int = 64 << 20) extends AutoCloseable {
  private val nat = Native.lib
  private[gpu] val handle: Long = nat.engineCreate(device)
  private val staging: ByteBuffer = ByteBuffer.allocateDirect(stagingBytes).order(ByteOrder.nativeOrder)
  private val stagingAddr: Long = nat.address(staging)
  def close(): Unit = nat.engineDestroy(handle)   // releases every buffer still allocated
  def lastVariant: String = nat.lastVariant(handle)

  // ---- staging: host arrays <-> device buffers through the bounded direct buffer ---------------------------------
  private def alloc(bytes: Long): DeviceBuffer = new DeviceBuffer(this, math.max(bytes, 8L))
  private def upload(dst: DeviceBuffer, dstOffset: Long, xs: Array[Double]): Unit = {
    val per = stagingBytes / 8
    var done = 0
    while (done < xs.length) {
      val n = math.min(per, xs.length - done)
      staging.clear(); staging.asDoubleBuffer.put(xs, done, n)
      nat.bufferUpload(handle, dst.ptr, dstOffset + 8L * done, stagingAddr, 8L * n)
      done += n
    }
  }
  private def uploadInts(dst: DeviceBuffer, dstOffset: Long, xs: Array[Int]): Unit = {
    require(4L * xs.length <= stagingBytes, "time grid longer than the staging buffer")
    staging.clear(); staging.asIntBuffer.put(xs)
    nat.bufferUpload(handle, dst.ptr, dstOffset, stagingAddr, 4L * xs.length)
  }
  private[gpu] def download(src: DeviceBuffer, srcOffsetDoubles: Long, n: Int): Array[Double] = {
    val out = new Array[Double](n)
    val per = stagingBytes / 8
    var done = 0
    while (done < n) {
      val k = math.min(per, n - done)
      nat.bufferDownload(handle, src.ptr, 8L * (srcOffsetDoubles + done), stagingAddr, 8L * k)
      staging.clear(); staging.asDoubleBuffer.get(out, done, k)
      done += k
    }
    out
  }
  private def downloadInts(src: DeviceBuffer, n: Int): Array[Int] = {
    val out = new Array[Int](n)
    var done = 0
    val per = stagingBytes / 4
    while (done < n) {
      val k = math.min(per, n - done)
      nat.bufferDownload(handle, src.ptr, 4L * done, stagingAddr, 4L * k)
      staging.clear(); staging.asIntBuffer.get(out, done, k)
      done += k
    }
    out
  }

  // ---- the model on the device ----------------------------------------------------------------------------------------
  /** Observations of N series on one time grid plus the materialised model tables, resident on the device. */
  final class DeviceSeries private[Batched] (val mod: Dlm, val times: Array[Double], val n: Int, val d: Int, val p: Int,
                                             private[Batched] val y: DeviceBuffer, private[Batched] val tables: DeviceBuffer,
                                             private[Batched] val model: Array[Long], val dts: Array[Double]) {
    val t: Int = times.length
    /** time of record k (record 0 is the initial state at t0 - 1, KalmanFilter.initialiseState) */
    def timeOf(k: Int): Double = if (k == 0) times(0) - dts(0) else times(k - 1)
    def free(): Unit = { y.free(); tables.free() }
  }

  /** Evaluate the closures once: F_t = mod.f(time_t) (d x p), G_k = mod.g(dt_k) per distinct dt; upload everything. */
  def upload(mod: Dlm, ys: Vector[Vector[Data]]): DeviceSeries = {
    require(ys.nonEmpty && ys.head.nonEmpty, "empty observations")   // the reference throws on t0.get (KalmanFilter.scala:116-117)
    val times = ys.head.map(_.time).toArray
    require(ys.forall(_.length == times.length), "all series of a batch share one time grid")
    val t0 = times.min - 1.0                                           // KalmanFilter.initialiseState
    val dts = (t0 +: times.init).zip(times).map { case (a, b) => b - a }
    val fs = times.map(mod.f)
    val constF = fs.forall(_ == fs.head)
    val fFlat = if (constF) fs.head.data else fs.flatMap(_.data)
    val uniq = dts.distinct
    val gFlat = uniq.flatMap(x => mod.g(x).data)
    val gIndex = dts.map(x => uniq.indexOf(x))
    val (d, p, n, t) = (fs.head.rows, fs.head.cols, ys.length, times.length)
    def pad(b: Long) = (b + 255L) & ~255L
    val offG = pad(8L * fFlat.length); val offGi = offG + pad(8L * gFlat.length); val offDt = offGi + pad(4L * t)
    val tables = alloc(offDt + pad(8L * t))
    upload(tables, 0L, fFlat); upload(tables, offG, gFlat); upload(tables, offDt, dts)
    uploadInts(tables, offGi, gIndex)
    val yDev = alloc(8L * n * t * p)
    // observations: one series at a time through the staging buffer, Option -> NaN
    val row = new Array[Double](t * p)
    var i = 0
    while (i < n) {
      var k = 0
      for (obs <- ys(i); o <- obs.observation.data) { row(k) = o.getOrElse(Double.NaN); k += 1 }
      upload(yDev, 8L * i * t * p, row)
      i += 1
    }
    val unit = dts.forall(_ == 1.0)
    val model = Array[Long](d, p, t, n, tables.ptr, if (constF) 0L else d.toLong * p, tables.ptr + offG, uniq.length,
                            if (uniq.length > 1) tables.ptr + offGi else 0L, if (unit) 0L else tables.ptr + offDt)
    new DeviceSeries(mod, times, n, d, p, yDev, tables, model, dts)
  }

  /** DlmParameters on the device: one shared set (strides 0) or one per series. */
  final class DeviceParameters private[Batched] (val buf: DeviceBuffer, val n: Int, val d: Int, val p: Int, shared: Boolean) {
    private val (pp, dd) = (p.toLong * p, d.toLong * d)
    private[Batched] def vPtr = buf.ptr
    private[Batched] def wPtr = vPtr + 8L * n * pp
    private[Batched] def m0Ptr = wPtr + 8L * n * dd
    private[Batched] def c0Ptr = m0Ptr + 8L * n * d
    private[Batched] def desc: Array[Long] =
      if (shared) Array[Long](vPtr, 0L, wPtr, 0L, m0Ptr, 0L, c0Ptr, 0L, 0L, 0L)
      else Array[Long](vPtr, pp, wPtr, dd, m0Ptr, d.toLong, c0Ptr, dd, 0L, 0L)
    def free(): Unit = buf.free()
  }
  def uploadParameters(ps: Vector[DlmParameters], shared: Boolean): DeviceParameters = {
    val (d, p, n) = (ps.head.m0.length, ps.head.v.rows, ps.length)
    val buf = alloc(8L * n * (p * p + d * d + d + d * d))
    val dp = new DeviceParameters(buf, n, d, p, shared)
    upload(buf, 0L, ps.flatMap(_.v.data).toArray)
    upload(buf, dp.wPtr - dp.vPtr, ps.flatMap(_.w.data).toArray)
    upload(buf, dp.m0Ptr - dp.vPtr, ps.flatMap(_.m0.data).toArray)
    upload(buf, dp.c0Ptr - dp.vPtr, ps.flatMap(_.c0.data).toArray)
    dp
  }
  def uploadParameters(p: DlmParameters): DeviceParameters = uploadParameters(Vector(p), shared = true)

  private def opts(flags: Int, seed: Long = 0L, seriesOffset: Long = 0L) = Array[Long](flags.toLong & 0xffffffffL, Native.MemDevice, seed, seriesOffset)
  private def checkStatus(status: DeviceBuffer, n: Int): Array[Int] = { val s = downloadInts(status, n); status.free(); s }

  // ---- results living on the device --------------------------------------------------------------------------------------
  /** [N][T+1][d + d*d] state records (mean, covariance) in HBM; objects are built on demand. */
  final class Records private[Batched] (val ys: DeviceSeries, private[Batched] val buf: DeviceBuffer, val status: Array[Int]) {
    private val rec = ys.d + ys.d * ys.d
    private def offset(i: Int, k: Int): Long = (i.toLong * (ys.t + 1) + k) * rec
    /** (mean, covariance) of series i at record k (0 = the initial state at t0 - 1) */
    def apply(i: Int, k: Int): (DenseVector[Double], DenseMatrix[Double]) = {
      val a = download(buf, offset(i, k), rec)
      (DenseVector(a.take(ys.d)), new DenseMatrix(ys.d, ys.d, a.drop(ys.d)))
    }
    /** all T+1 records of series i: one contiguous download */
    def series(i: Int): Vector[(Double, DenseVector[Double], DenseMatrix[Double])] = {
      val a = download(buf, offset(i, 0), (ys.t + 1) * rec)
      Vector.tabulate(ys.t + 1) { k =>
        val o = k * rec
        (ys.timeOf(k), DenseVector(a.slice(o, o + ys.d)), new DenseMatrix(ys.d, ys.d, a.slice(o + ys.d, o + rec)))
      }
    }
    /** Smoothing.SmoothingState objects of series i (for smoothed records) */
    def smoothingStates(i: Int): Vector[SmoothingState] = series(i).map { case (time, m, c) => SmoothingState(time, m, c, m, c) }
    def free(): Unit = buf.free()
  }

  /** The complete output of KalmanFilter(...).filter for N series: (m, C), (a, R) and (f, Q) records. */
  final class Filtered private[Batched] (val ys: DeviceSeries, val p: DeviceParameters, val posterior: Records, val prior: Records,
                                         private[Batched] val fq: DeviceBuffer) {
    /** the reference's KfState objects of series i; `dropInit` as filterTraverse / filterDlm do (Filter.scala:32-36) */
    def kfStates(i: Int, dropInit: Boolean = true): Vector[KfState] = {
      val (post, pri) = (posterior.series(i), prior.series(i))
      val q = ys.p + ys.p * ys.p
      val f = download(fq, i.toLong * (ys.t + 1) * q, (ys.t + 1) * q)
      val all = Vector.tabulate(ys.t + 1) { k =>
        val ft = if (k == 0) None else Some(DenseVector(f.slice(k * q, k * q + ys.p)))
        val qt = if (k == 0) None else Some(new DenseMatrix(ys.p, ys.p, f.slice(k * q + ys.p, (k + 1) * q)))
        KfState(post(k)._1, post(k)._2, post(k)._3, pri(k)._2, pri(k)._3, ft, qt)
      }
      if (dropInit) all.tail else all
    }
    def free(): Unit = { posterior.free(); prior.free(); fq.free() }
  }

  /** One draw of theta_0 .. theta_T per series, on the device. */
  final class Sampled private[Batched] (val ys: DeviceSeries, private[Batched] val theta: DeviceBuffer, val status: Array[Int]) {
    def sample(i: Int, k: Int): DenseVector[Double] = DenseVector(download(theta, (i.toLong * (ys.t + 1) + k) * ys.d, ys.d))
    /** Smoothing.SamplingState objects of series i.  Only (time, sample) carry information -- what the Gibbs steps read
      * (Gibbs.scala:141-142); mean / cov / at1 / rt1 repeat the sample and empty matrices. */
    def samplingStates(i: Int): Vector[SamplingState] = {
      val a = download(theta, i.toLong * (ys.t + 1) * ys.d, (ys.t + 1) * ys.d)
      val e = DenseMatrix.zeros[Double](0, 0)
      Vector.tabulate(ys.t + 1) { k => val s = DenseVector(a.slice(k * ys.d, (k + 1) * ys.d)); SamplingState(ys.timeOf(k), s, s, e, s, e) }
    }
    def free(): Unit = theta.free()
  }

  // ---- the reference's calls ------------------------------------------------------------------------------------------------
  /** KalmanFilter.filterDlm(mod, ys, p) for N series (KalmanFilter.scala:291-294). */
  def filterDlm(mod: Dlm, ys: Vector[Vector[Data]], p: DlmParameters): Filtered = filterDlm(upload(mod, ys), uploadParameters(p))
  def filterDlm(ys: DeviceSeries, p: DeviceParameters): Filtered = {
    val rec = 8L * ys.n * (ys.t + 1) * (ys.d + ys.d * ys.d)
    val (post, pri, fq, st) = (alloc(rec), alloc(rec), alloc(8L * ys.n * (ys.t + 1) * (ys.p + ys.p * ys.p)), alloc(4L * ys.n))
    nat.filter(handle, ys.model, p.desc, opts(0), ys.y.ptr, post.ptr, pri.ptr, fq.ptr, st.ptr)
    val status = checkStatus(st, ys.n)
    new Filtered(ys, p, new Records(ys, post, status), new Records(ys, pri, status), fq)
  }

  /** Smoothing.backwardsSmoother(mod)(kfStates) (Smoothing.scala:57-64): takes what filterDlm returned (the T+1 records
    * including the initial state).  literal = true reproduces Smoothing.scala:44's J X J (SURVEY Q1). */
  def backwardsSmoother(kf: Filtered, literal: Boolean = false): Records = {
    val ys = kf.ys
    val (sm, st) = (alloc(8L * ys.n * (ys.t + 1) * (ys.d + ys.d * ys.d)), alloc(4L * ys.n))
    nat.smooth(handle, ys.model, kf.p.desc, opts(if (literal) Native.SmootherCompatQ1 else 0), kf.posterior.buf.ptr, sm.ptr, st.ptr)
    new Records(ys, sm, checkStatus(st, ys.n))
  }

  /** KalmanFilter(...).filter followed by Smoothing.backwardsSmoother in one fused call (the metric path):
    * returns (filtered records, smoothed records). */
  def filterSmooth(mod: Dlm, ys: Vector[Vector[Data]], p: DlmParameters): (Records, Records) = filterSmooth(upload(mod, ys), uploadParameters(p))
  def filterSmooth(ys: DeviceSeries, p: DeviceParameters, flags: Int = 0): (Records, Records) = {
    val rec = 8L * ys.n * (ys.t + 1) * (ys.d + ys.d * ys.d)
    val (filt, sm, st) = (alloc(rec), alloc(rec), alloc(4L * ys.n))
    nat.filterSmooth(handle, ys.model, p.desc, opts(flags), ys.y.ptr, filt.ptr, sm.ptr, st.ptr)
    val status = checkStatus(st, ys.n)
    (new Records(ys, filt, status), new Records(ys, sm, status))
  }

  /** Sum over each series of KalmanFilter.conditionalLikelihood(f_t, Q_t, y_t) (KalmanFilter.scala:138-153). */
  def logLikelihood(ys: DeviceSeries, p: DeviceParameters, flags: Int = 0): Array[Double] = {
    val (ll, st) = (alloc(8L * ys.n), alloc(4L * ys.n))
    nat.loglik(handle, ys.model, p.desc, opts(flags), ys.y.ptr, ll.ptr, st.ptr)
    st.free()
    val out = download(ll, 0L, ys.n); ll.free(); out
  }

  /** KalmanFilter.likelihood(mod, ys)(p) as the reference writes it (KalmanFilter.scala:299-306) -- what
    * MetropolisHastings.dlm evaluates (MetropolisHastings.scala:134, :205): the transition density of the filtered means,
    * sum_t log N(m_t; g(dt_t) m_(t-1), W dt_t) (KalmanFilter.logLikelihood :175-183).  With one DlmParameters per series
    * (DeviceParameters built from a Vector) a whole population of proposals is evaluated in one launch. */
  def likelihood(ys: DeviceSeries, p: DeviceParameters): Array[Double] = logLikelihood(ys, p, Native.LoglikLiteralQ7)

  /** Smoothing.ffbsDlm(mod, ys, p) for N series (Smoothing.scala:173-180).  The reference's draws cannot be seeded
    * (SURVEY Q3); here a draw is a pure function of (seed, series index). */
  def ffbsDlm(mod: Dlm, ys: Vector[Vector[Data]], p: DlmParameters, seed: Long = 0L): Sampled = ffbsDlm(upload(mod, ys), uploadParameters(p), seed)
  def ffbsDlm(ys: DeviceSeries, p: DeviceParameters, seed: Long): Sampled = {
    val (ws, th, st) = (alloc(8L * ys.n * (ys.t + 1) * (ys.d + ys.d * ys.d)), alloc(8L * ys.n * (ys.t + 1) * ys.d), alloc(4L * ys.n))
    nat.ffbs(handle, ys.model, p.desc, opts(0, seed), ys.y.ptr, 0L, ws.ptr, th.ptr, 0L, 0L, st.ptr)
    ws.free()
    new Sampled(ys, th, checkStatus(st, ys.n))
  }

  /** SvdFilter.filterDlm(mod, ys, p) (SvdFilter.scala:158-161): records (m_t, dc_t, uc_t), C_t = uc diag(dc^2) uc^T.
    * literal = true hands the raw W to the time update as the reference does (SURVEY Q2). */
  final class SvdFiltered private[Batched] (val ys: DeviceSeries, private[Batched] val buf: DeviceBuffer, val status: Array[Int]) {
    private val rec = 2 * ys.d + ys.d * ys.d
    /** (time, mt, dc, uc) of series i, initial state dropped as filterTraverse does */
    def states(i: Int): Vector[(Double, DenseVector[Double], DenseVector[Double], DenseMatrix[Double])] = {
      val a = download(buf, i.toLong * (ys.t + 1) * rec, (ys.t + 1) * rec)
      Vector.tabulate(ys.t) { j =>
        val o = (j + 1) * rec
        (ys.timeOf(j + 1), DenseVector(a.slice(o, o + ys.d)), DenseVector(a.slice(o + ys.d, o + 2 * ys.d)), new DenseMatrix(ys.d, ys.d, a.slice(o + 2 * ys.d, o + rec)))
      }
    }
    def free(): Unit = buf.free()
  }
  def svdFilterDlm(mod: Dlm, ys: Vector[Vector[Data]], p: DlmParameters, literal: Boolean = false): SvdFiltered = {
    val (dys, dp) = (upload(mod, ys), uploadParameters(p))
    val (rec, st) = (alloc(8L * dys.n * (dys.t + 1) * (2 * dys.d + dys.d * dys.d)), alloc(4L * dys.n))
    nat.svdFilter(handle, dys.model, dp.desc, opts(if (literal) Native.SvdRawWQ2 else 0), dys.y.ptr, rec.ptr, st.ptr)
    new SvdFiltered(dys, rec, checkStatus(st, dys.n))
  }

  /** SvdSampler.ffbsDlm(mod, ys, p) (SvdSampler.scala:79-82); literal = true: SURVEY Q2 + Q9. */
  def svdFfbsDlm(mod: Dlm, ys: Vector[Vector[Data]], p: DlmParameters, seed: Long = 0L, literal: Boolean = false): Sampled = {
    val (dys, dp) = (upload(mod, ys), uploadParameters(p))
    val (ws, th, st) = (alloc(8L * dys.n * (dys.t + 1) * (2 * dys.d + dys.d * dys.d)), alloc(8L * dys.n * (dys.t + 1) * dys.d), alloc(4L * dys.n))
    val fl = if (literal) Native.SvdRawWQ2 | Native.SvdSamplerQ9 else 0
    nat.svdFfbs(handle, dys.model, dp.desc, opts(fl, seed), dys.y.ptr, 0L, ws.ptr, th.ptr, 0L, st.ptr)
    ws.free()
    new Sampled(dys, th, checkStatus(st, dys.n))
  }

  // ---- Gibbs samplers: one chain per series, advanced together ----------------------------------------------------------------
  /** State of all chains after an iteration: per-series (V, W) on the device, fetched on demand. */
  final class GibbsState private[Batched] (val iteration: Int, val ys: DeviceSeries, private[Batched] val p: DeviceParameters, init: DlmParameters) {
    /** GibbsSampling.State.p of series i */
    def params(i: Int): DlmParameters = {
      val v = download(p.buf, i.toLong * ys.p * ys.p, ys.p * ys.p)
      val w = download(p.buf, (p.wPtr - p.vPtr) / 8L + i.toLong * ys.d * ys.d, ys.d * ys.d)
      DlmParameters(new DenseMatrix(ys.p, ys.p, v), new DenseMatrix(ys.d, ys.d, w), init.m0, init.c0)
    }
  }

  /** GibbsSampling.sample(mod, priorV, priorW, initParams, observations) (Gibbs.scala:165-180) for N series with
    * independent parameters.  Every iteration is dinvGammaStep (Gibbs.scala:134-151): FFBS with the sufficient
    * statistics accumulated on the device, then the d-Inverse-Gamma draws on the device too (Native#dinvgammaStep) --
    * nothing crosses PCIe per iteration.  svd = true is sampleSvd / stepSvd (Gibbs.scala:182-217). */
  def gibbsSample(mod: Dlm, priorV: InverseGamma, priorW: InverseGamma, initParams: DlmParameters, ys: Vector[Vector[Data]],
                  seed: Long = 0L, svd: Boolean = false, flags: Int = 0): Iterator[GibbsState] = {
    val dys = upload(mod, ys)
    val dp = uploadParameters(Vector.fill(dys.n)(initParams), shared = false)
    val l = nat.statsLen(dys.d, dys.p, 0)
    val recD = if (svd) 2 * dys.d + dys.d * dys.d else dys.d + dys.d * dys.d
    val (ws, stats, st) = (alloc(8L * dys.n * (dys.t + 1) * recD), alloc(8L * dys.n * l), alloc(4L * dys.n))
    Iterator.from(0).map { it =>
      val o = opts(flags, seed * 1000003L + it)
      if (svd) nat.svdFfbs(handle, dys.model, dp.desc, o, dys.y.ptr, 0L, ws.ptr, 0L, stats.ptr, st.ptr)
      else nat.ffbs(handle, dys.model, dp.desc, o, dys.y.ptr, 0L, ws.ptr, 0L, 0L, stats.ptr, st.ptr)
      nat.dinvgammaStep(handle, dys.d, dys.p, dys.n, stats.ptr, priorV.shape, priorV.scale, priorW.shape, priorW.scale, it.toLong, opts(0, seed), dp.vPtr, dp.wPtr)
      new GibbsState(it, dys, dp, initParams)
    }
  }
  def gibbsSampleSvd(mod: Dlm, priorV: InverseGamma, priorW: InverseGamma, initParams: DlmParameters, ys: Vector[Vector[Data]], seed: Long = 0L): Iterator[GibbsState] =
    gibbsSample(mod, priorV, priorW, initParams, ys, seed, svd = true)

  /** GibbsWishart.sample(mod, priorV, priorW, initParams, observations) (GibbsWishart.scala:65-80): the outer-product
    * statistics come back from the device (N x (2p + d^2 + 1) doubles per iteration), W and V are drawn with the
    * reference's own InverseWishart / InverseGamma (order theta, W, V as wishartStep), and go back up. */
  def gibbsWishartSample(mod: Dlm, priorV: InverseGamma, priorW: InverseWishart, initParams: DlmParameters, ys: Vector[Vector[Data]],
                         seed: Long = 0L): Iterator[GibbsState] = {
    val dys = upload(mod, ys)
    val dp = uploadParameters(Vector.fill(dys.n)(initParams), shared = false)
    val (d, p, n) = (dys.d, dys.p, dys.n)
    val l = nat.statsLen(d, p, Native.StatsOuter)
    val (ws, stats, st) = (alloc(8L * n * (dys.t + 1) * (d + d * d)), alloc(8L * n * l), alloc(4L * n))
    Iterator.from(0).map { it =>
      nat.ffbs(handle, dys.model, dp.desc, opts(Native.StatsOuter, seed * 1000003L + it), dys.y.ptr, 0L, ws.ptr, 0L, 0L, stats.ptr, st.ptr)
      val s = download(stats, 0L, n * l)
      val vs = new Array[Double](n * p * p); val wsNew = new Array[Double](n * d * d)
      var i = 0
      while (i < n) {
        val o = i * l
        val tcount = s(o + l - 1)
        val outer = new DenseMatrix(d, d, s.slice(o + 2 * p, o + 2 * p + d * d))
        val w = InverseWishart(priorW.nu + tcount, priorW.psi + outer).draw
        System.arraycopy(w.data, 0, wsNew, i * d * d, d * d)
        var j = 0
        while (j < p) { vs(i * p * p + j * p + j) = InverseGamma(priorV.shape + 0.5 * s(o + p + j), priorV.scale + 0.5 * s(o + j)).draw; j += 1 }
        i += 1
      }
      upload(dp.buf, 0L, vs); upload(dp.buf, dp.wPtr - dp.vPtr, wsNew)
      new GibbsState(it, dys, dp, initParams)
    }
  }
}
