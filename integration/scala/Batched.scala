// Scala shim a maintainer of jonnylaw/bayesian_dlms would add (SOURCE ONLY here: no scalac/sbt in
// the build container).  It offers the reference's own calls over N series, evaluates the model
// closures on the host exactly once per distinct time / time increment, flattens Breeze's
// column-major `DenseMatrix.data`, encodes `Option[Double]` as NaN and hands direct buffers to
// the JNI layer (integration/jni/dlm_jni.cpp -> include/dlm_engine.h).
//
// KfState / SmoothingState objects are materialised lazily per (series, t) from the flat
// DoubleBuffer: 10^4 x 10^3 eager KfStates would be ~10^7 JVM objects (> 30 GB).
package com.github.jonnylaw.dlm.gpu

import java.nio.{ByteBuffer, ByteOrder, DoubleBuffer, IntBuffer}
import breeze.linalg.{DenseMatrix, DenseVector}
import com.github.jonnylaw.dlm._

object Native {
  System.loadLibrary("dlm_jni")
  @native def create(device: Int): Long
  @native def destroy(h: Long): Unit
  @native def filterSmooth(h: Long, d: Int, p: Int, t: Int, n: Int, f: DoubleBuffer, fStride: Long,
      g: DoubleBuffer, nG: Int, gIndex: IntBuffer, dt: DoubleBuffer,
      v: DoubleBuffer, vs: Long, w: DoubleBuffer, ws: Long, m0: DoubleBuffer, ms: Long, c0: DoubleBuffer, cs: Long,
      y: DoubleBuffer, flags: Int, filt: DoubleBuffer, smooth: DoubleBuffer, status: IntBuffer): Unit
  @native def logLikelihood(h: Long, d: Int, p: Int, t: Int, n: Int, f: DoubleBuffer, fStride: Long,
      g: DoubleBuffer, nG: Int, gIndex: IntBuffer, dt: DoubleBuffer,
      v: DoubleBuffer, vs: Long, w: DoubleBuffer, ws: Long, m0: DoubleBuffer, ms: Long, c0: DoubleBuffer, cs: Long,
      y: DoubleBuffer, flags: Int, loglik: DoubleBuffer, status: IntBuffer): Unit
  @native def filter(h: Long, d: Int, p: Int, t: Int, n: Int, f: DoubleBuffer, fStride: Long,
      g: DoubleBuffer, nG: Int, gIndex: IntBuffer, dt: DoubleBuffer,
      v: DoubleBuffer, vs: Long, w: DoubleBuffer, ws: Long, m0: DoubleBuffer, ms: Long, c0: DoubleBuffer, cs: Long,
      y: DoubleBuffer, flags: Int, filt: DoubleBuffer, prior: DoubleBuffer, fq: DoubleBuffer, status: IntBuffer): Unit
  @native def ffbs(h: Long, d: Int, p: Int, t: Int, n: Int, f: DoubleBuffer, fStride: Long,
      g: DoubleBuffer, nG: Int, gIndex: IntBuffer, dt: DoubleBuffer,
      v: DoubleBuffer, vs: Long, w: DoubleBuffer, ws: Long, m0: DoubleBuffer, ms: Long, c0: DoubleBuffer, cs: Long,
      y: DoubleBuffer, flags: Int, seed: Long, seriesOffset: Long,
      filtWs: DoubleBuffer, theta: DoubleBuffer, stats: DoubleBuffer, status: IntBuffer): Unit
}

final class Batched(device: Int = 0) extends AutoCloseable {
  private val h = Native.create(device)
  def close(): Unit = Native.destroy(h)

  private def dbuf(n: Long): DoubleBuffer =
    ByteBuffer.allocateDirect((n * 8).toInt).order(ByteOrder.nativeOrder).asDoubleBuffer
  private def dbuf(xs: Array[Double]): DoubleBuffer = { val b = dbuf(xs.length); b.put(xs); b.rewind(); b }

  /** Evaluate the closures once: F_t = mod.f(time_t) (d x p), G_k = mod.g(dt_k) per distinct dt. */
  private case class Tables(d: Int, p: Int, times: Array[Double], f: DoubleBuffer, fStride: Long,
                            g: DoubleBuffer, nG: Int, gIndex: IntBuffer, dt: DoubleBuffer)
  private def materialise(mod: Dlm, times: Array[Double]): Tables = {
    val t0 = times.min - 1.0                                   // KalmanFilter.initialiseState
    val dts = (t0 +: times.init).zip(times).map { case (a, b) => b - a }
    val fs = times.map(mod.f)
    val constF = fs.forall(_ == fs.head)
    val f = dbuf(if (constF) fs.head.data else fs.flatMap(_.data))
    val uniq = dts.distinct.sorted
    val gs = uniq.map(mod.g)
    val gi = ByteBuffer.allocateDirect(4 * times.length).order(ByteOrder.nativeOrder).asIntBuffer
    dts.foreach(x => gi.put(uniq.indexOf(x))); gi.rewind()
    Tables(fs.head.rows, fs.head.cols, times, f, if (constF) 0L else fs.head.size.toLong,
           dbuf(gs.flatMap(_.data)), uniq.length, gi, dbuf(dts))
  }

  /** Result of the fused filter + smoother: flat records, KfState-like views on demand. */
  final class FilterSmoothResult(val d: Int, val n: Int, val times: Array[Double],
                                 val filt: DoubleBuffer, val smooth: DoubleBuffer) {
    private val rec = d + d * d
    private def mat(b: DoubleBuffer, off: Int) = { val a = new Array[Double](d * d); b.position(off); b.get(a); new DenseMatrix(d, d, a) }
    private def vec(b: DoubleBuffer, off: Int) = { val a = new Array[Double](d); b.position(off); b.get(a); DenseVector(a) }
    /** (m_t, C_t) of series i at record t (0 = initial state at t0 - 1) */
    def filtered(i: Int, t: Int) = { val o = (i * (times.length + 1) + t) * rec; (vec(filt, o), mat(filt, o + d)) }
    /** Smoothing.SmoothingState-like (mean, covariance) */
    def smoothed(i: Int, t: Int) = { val o = (i * (times.length + 1) + t) * rec; (vec(smooth, o), mat(smooth, o + d)) }
  }

  /** Sum over each series of KalmanFilter.conditionalLikelihood(f_t, Q_t, y_t): log p(y | V, W), one value per series. */
  def logLikelihood(mod: Dlm, ys: Vector[Vector[Data]], p: DlmParameters): Array[Double] = {
    require(ys.nonEmpty && ys.head.nonEmpty, "empty observations")
    val times = ys.head.map(_.time).toArray
    val tb = materialise(mod, times)
    val (n, t) = (ys.length, times.length)
    val y = dbuf(n.toLong * t * tb.p)
    for (s <- ys; d <- s; o <- d.observation.data) y.put(o.getOrElse(Double.NaN))
    y.rewind()
    val ll = dbuf(n.toLong)
    val status = ByteBuffer.allocateDirect(4 * n).order(ByteOrder.nativeOrder).asIntBuffer
    Native.logLikelihood(h, tb.d, tb.p, t, n, tb.f, tb.fStride, tb.g, tb.nG, tb.gIndex, tb.dt,
      dbuf(p.v.data), 0L, dbuf(p.w.data), 0L, dbuf(p.m0.data), 0L, dbuf(p.c0.data), 0L, y, 0, ll, status)
    Array.tabulate(n)(ll.get)
  }

  /** KalmanFilter(advanceState(p, mod.g)).filter + Smoothing.backwardsSmoother for N series
    * (same argument order as KalmanFilter.filterDlm(mod, ys, p)). */
  def filterSmooth(mod: Dlm, ys: Vector[Vector[Data]], p: DlmParameters): FilterSmoothResult = {
    require(ys.nonEmpty && ys.head.nonEmpty, "empty observations")   // the reference throws on t0.get
    val times = ys.head.map(_.time).toArray
    val tb = materialise(mod, times)
    val (n, t) = (ys.length, times.length)
    val y = dbuf(n.toLong * t * tb.p)
    for (s <- ys; d <- s; o <- d.observation.data) y.put(o.getOrElse(Double.NaN))
    y.rewind()
    val rec = tb.d + tb.d * tb.d
    val (filt, smooth) = (dbuf(n.toLong * (t + 1) * rec), dbuf(n.toLong * (t + 1) * rec))
    val status = ByteBuffer.allocateDirect(4 * n).order(ByteOrder.nativeOrder).asIntBuffer
    Native.filterSmooth(h, tb.d, tb.p, t, n, tb.f, tb.fStride, tb.g, tb.nG, tb.gIndex, tb.dt,
      dbuf(p.v.data), 0L, dbuf(p.w.data), 0L, dbuf(p.m0.data), 0L, dbuf(p.c0.data), 0L,
      y, 0, filt, smooth, status)
    new FilterSmoothResult(tb.d, n, times, filt, smooth)
  }
}
