"""Python handle over the C ABI.  numpy arrays => DLM_MEM_HOST (the engine stages them),
torch CUDA tensors => DLM_MEM_DEVICE (zero-copy; PyTorch is only the allocator here)."""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Union

import numpy as np

from . import _lib
from .dlm import DlmParameters, MaterialisedModel


class EngineError(RuntimeError):
    pass


def _cm(a):
    a = np.asarray(a, dtype=np.float64)
    return np.ascontiguousarray(a.T).reshape(-1) if a.ndim == 2 else np.ascontiguousarray(a).reshape(-1)


def _cm_stream(a):
    """One matrix, or a [T] stream of matrices -> flat column-major array(s)."""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 3:
        return np.ascontiguousarray(np.transpose(a, (0, 2, 1))).reshape(-1)
    return _cm(a)


def pack_params(params: Union[DlmParameters, Sequence[DlmParameters]], N: int):
    """DlmParameters (shared) or a sequence of N DlmParameters (per series) -> flat column-major arrays and strides
    in doubles: (V, v_stride, W, w_stride, m0, m0_stride, C0, c0_stride, v_tstride, w_tstride).  The last two are
    the per-time strides of V_t / W_t streams (0 = time-invariant)."""
    plist = [params] if isinstance(params, DlmParameters) else list(params)
    if not isinstance(params, DlmParameters) and len(plist) != N:
        raise ValueError(f"need {N} DlmParameters, got {len(plist)}")
    p0 = plist[0]
    vts = p0.v.shape[-1] ** 2 if p0.v.ndim == 3 else 0
    wts = p0.w.shape[-1] ** 2 if p0.w.ndim == 3 else 0
    if any((q.v.ndim == 3) != (vts != 0) or (q.w.ndim == 3) != (wts != 0) for q in plist):
        raise ValueError("either every series has time-varying V (W) or none")
    if isinstance(params, DlmParameters):
        return (_cm_stream(p0.v), 0, _cm_stream(p0.w), 0, _cm(p0.m0), 0, _cm(p0.c0), 0, vts, wts)
    V = np.stack([_cm_stream(q.v) for q in plist]); W = np.stack([_cm_stream(q.w) for q in plist])
    m0 = np.stack([_cm(q.m0) for q in plist]); C0 = np.stack([_cm(q.c0) for q in plist])
    return (V.reshape(-1), V.shape[1], W.reshape(-1), W.shape[1], m0.reshape(-1), m0.shape[1],
            C0.reshape(-1), C0.shape[1], vts, wts)


class _Host:
    mem = _lib.DLM_MEM_HOST

    @staticmethod
    def put(a, dtype=np.float64):
        return None if a is None else np.ascontiguousarray(a, dtype=dtype)

    @staticmethod
    def empty(shape, dtype=np.float64):
        return np.empty(shape, dtype=dtype)

    @staticmethod
    def ptr(a):
        return None if a is None else ctypes.c_void_p(a.ctypes.data)


class _Device:
    mem = _lib.DLM_MEM_DEVICE

    def __init__(self, device):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device)

    def put(self, a, dtype=np.float64):
        if a is None:
            return None
        t = self.torch
        if isinstance(a, t.Tensor):
            want = t.float64 if dtype == np.float64 else t.int32
            return a.to(device=self.device, dtype=want).contiguous()
        return t.as_tensor(np.ascontiguousarray(a, dtype=dtype), device=self.device)

    def empty(self, shape, dtype=np.float64):
        t = self.torch
        return t.empty(shape, dtype=t.float64 if dtype == np.float64 else t.int32, device=self.device)

    @staticmethod
    def ptr(a):
        return None if a is None else ctypes.c_void_p(a.data_ptr())


class Engine:
    """One engine per GPU (not thread-safe), mirroring `dlm_engine`."""

    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        h = ctypes.c_void_p()
        rc = self.lib.dlm_engine_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise EngineError(f"dlm_engine_create(device={device}) failed with {rc}: no usable HIP device")
        self.h = h
        self.device = int(device)
        self._keep = []

    def close(self):
        if getattr(self, "h", None):
            self.lib.dlm_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers -------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise EngineError(f"engine call failed ({rc}): {self.lib.dlm_last_error(self.h).decode()}")

    @property
    def last_variant(self) -> str:
        return self.lib.dlm_last_variant(self.h).decode()

    def set_stream(self, stream_ptr: Optional[int]):
        self._check(self.lib.dlm_engine_set_stream(self.h, ctypes.c_void_p(stream_ptr or 0)))

    def sync(self):
        self._check(self.lib.dlm_engine_sync(self.h))
        self._keep.clear()

    def _backend(self, y):
        if isinstance(y, np.ndarray) or y is None:
            return _Host()
        be = _Device(self.device)
        # Device mode takes torch tensors that were produced (or whose memory was last used) on torch's current stream,
        # while the engine launches on its own: order the engine's stream behind it (an event, no host wait).  The other
        # direction needs nothing for synchronous calls (the engine stream is drained on return); after DLM_OPT_ASYNC
        # calls sync() -- or stream_wait_engine() -- before torch touches the outputs.
        self._check(self.lib.dlm_engine_wait_stream(self.h, ctypes.c_void_p(be.torch.cuda.current_stream(be.device).cuda_stream or 0)))
        return be

    def stream_wait_engine(self, stream_ptr: Optional[int] = None):
        """Make `stream_ptr` (default: torch's current stream) wait for the engine work submitted so far."""
        if stream_ptr is None:
            import torch
            stream_ptr = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.dlm_stream_wait_engine(self.h, ctypes.c_void_p(stream_ptr or 0)))

    def _hold(self, flags, *objs):
        """Under DLM_OPT_ASYNC the kernels may still be reading the staged model / parameter tables and the inputs when
        the call returns: keep them alive until sync()."""
        if flags & _lib.OPT_ASYNC:
            self._keep.append(objs)

    # -- engine-owned device buffers (what a caller without torch uses; also exercised by the tests) ----------------
    def buffer_alloc(self, nbytes: int) -> int:
        p = ctypes.c_void_p()
        self._check(self.lib.dlm_buffer_alloc(self.h, int(nbytes), ctypes.byref(p)))
        return int(p.value)

    def buffer_free(self, ptr: int):
        self._check(self.lib.dlm_buffer_free(self.h, ctypes.c_void_p(ptr)))

    def buffer_upload(self, ptr: int, a: np.ndarray, offset: int = 0):
        a = np.ascontiguousarray(a)
        self._check(self.lib.dlm_buffer_upload(self.h, ctypes.c_void_p(ptr), int(offset), ctypes.c_void_p(a.ctypes.data), a.nbytes))

    def buffer_download(self, ptr: int, out: np.ndarray, offset: int = 0):
        assert out.flags["C_CONTIGUOUS"]
        self._check(self.lib.dlm_buffer_download(self.h, ctypes.c_void_p(ptr), int(offset), ctypes.c_void_p(out.ctypes.data), out.nbytes))
        return out

    def mem_info(self):
        f, t = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self.lib.dlm_device_mem_info(self.h, ctypes.byref(f), ctypes.byref(t)))
        return int(f.value), int(t.value)

    def prepare(self, mat: MaterialisedModel, params, N: int, be, flags=0, seed=0, series_offset=0):
        """Build the descriptors; returns (model, params, opts, keepalive list)."""
        packed = pack_params(params, N) if not isinstance(params, tuple) else params
        V, vs, W, ws, m0, m0s, C0, c0s = packed[:8]
        vts, wts = (packed[8], packed[9]) if len(packed) > 8 else (0, 0)
        if vts and np.size(V) // max(1, (N if vs else 1)) != mat.T * vts or wts and np.size(W) // max(1, (N if ws else 1)) != mat.T * wts:
            raise ValueError("time-varying V / W need one matrix per observation")
        items = dict(F=(mat.F, np.float64), G=(mat.G, np.float64), gi=(mat.g_index, np.int32), dt=(mat.dt, np.float64),
                     V=(V, np.float64), W=(W, np.float64), m0=(m0, np.float64), C0=(C0, np.float64))
        if isinstance(be, _Device) and not any(isinstance(a, be.torch.Tensor) for a, _ in items.values()):
            # device mode with a host-side model: ONE upload for all the small tables instead of eight
            parts, offs, pos = [], {}, 0
            for name, (a, dt) in items.items():
                if a is None:
                    continue
                b = np.ascontiguousarray(a, dtype=dt).reshape(-1).view(np.uint8)
                offs[name] = pos
                parts.append(b)
                pad = (-b.size) % 256
                if pad:
                    parts.append(np.zeros(pad, dtype=np.uint8))
                pos += b.size + pad
            host = np.concatenate(parts)
            # The staged tables of the call before are kept: the same bytes are not uploaded again, and the same MODEL bytes (F, G,
            # time grid -- the parameters may move, as in a Gibbs loop) let the engine reuse its analysis of their structure
            # (DLM_OPT_MODEL_UNCHANGED: two stream round trips per call less).  Not under DLM_OPT_ASYNC, whose tables may still be read.
            mend = offs.get("V", host.size)
            mkey = (mat.d, mat.p, mat.T, mat.n_g, mat.f_stride, host[:mend].tobytes())
            cache = getattr(self, "_staged", None)
            oflags = flags
            small = host.size <= (1 << 20)          # per-series or per-step parameter tables are not worth comparing: uploaded as before
            if not small:
                cache = None
            if cache is not None and not (flags & _lib.OPT_ASYNC) and cache["dev"] == be.device:
                if cache["mkey"] == mkey:   # compared byte for byte right here: the engine's device checksum of the promise is not needed
                    oflags |= _lib.OPT_MODEL_UNCHANGED | _lib.OPT_TRUST_MODEL_UNCHANGED
                blob = cache["blob"] if (cache["host"].size == host.size and np.array_equal(cache["host"], host)) else None
            else:
                blob = None
            if blob is None:
                blob = be.torch.as_tensor(host, device=be.device)
            self._staged = None if ((flags & _lib.OPT_ASYNC) or not small) else {"host": host, "blob": blob, "mkey": mkey, "dev": be.device}
            base = blob.data_ptr()
            bufs = {"blob": blob}
            P = lambda name: (base + offs[name]) if name in offs else None
            md = _lib.ModelDesc(mat.d, mat.p, mat.T, N, P("F"), mat.f_stride, P("G"), mat.n_g, P("gi"), P("dt"))
            pd = _lib.ParamsDesc(P("V"), vs, P("W"), ws, P("m0"), m0s, P("C0"), c0s, vts, wts)
            self._hold(flags, bufs)
            return md, pd, _lib.Options(oflags, be.mem, seed, series_offset), bufs
        self._staged = None
        bufs = {name: be.put(a, dt) for name, (a, dt) in items.items()}
        P = lambda a: (be.ptr(a).value if a is not None else None)
        md = _lib.ModelDesc(mat.d, mat.p, mat.T, N, P(bufs["F"]), mat.f_stride, P(bufs["G"]), mat.n_g,
                            P(bufs["gi"]), P(bufs["dt"]))
        pd = _lib.ParamsDesc(P(bufs["V"]), vs, P(bufs["W"]), ws, P(bufs["m0"]), m0s, P(bufs["C0"]), c0s, vts, wts)
        op = _lib.Options(flags, be.mem, seed, series_offset)
        self._hold(flags, bufs)
        return md, pd, op, bufs

    # -- entry points --------------------------------------------------------------
    def filter(self, mat, params, y, *, want_prior=False, want_fq=False, flags=0, out=None):
        """out: an existing record buffer [N][T+1][rec] to write into (its previous content must not matter)."""
        be = self._backend(y)
        N = int(y.shape[0]); d, p, T = mat.d, mat.p, mat.T
        rec = d + d * d
        yb = be.put(y).reshape(N, T, p)
        md, pd, op, keep = self.prepare(mat, params, N, be, flags)
        self._hold(flags, yb)
        filt = out if out is not None else be.empty((N, T + 1, self.lib.dlm_packed_record_doubles(d) if flags & _lib.OPT_PACKED_SYM else rec))
        prior = be.empty((N, T + 1, rec)) if want_prior else None
        fq = be.empty((N, T + 1, p + p * p)) if want_fq else None
        status = be.empty((N,), np.int32)
        self._check(self.lib.dlm_filter_batch(self.h, md, pd, be.ptr(yb), op, be.ptr(filt), be.ptr(prior),
                                              be.ptr(fq), be.ptr(status)))
        return {"filt": filt, "prior": prior, "fq": fq, "status": status}

    def loglik(self, mat, params, y, *, flags=0):
        """Per-series prediction-error log-likelihood (dlm_loglik_batch); nothing but [N] numbers leaves the GPU."""
        be = self._backend(y)
        N = int(y.shape[0]); p, T = mat.p, mat.T
        yb = be.put(y).reshape(N, T, p)
        md, pd, op, keep = self.prepare(mat, params, N, be, flags)
        ll = be.empty((N,))
        status = be.empty((N,), np.int32)
        self._check(self.lib.dlm_loglik_batch(self.h, md, pd, be.ptr(yb), op, be.ptr(ll), be.ptr(status)))
        return {"loglik": ll, "status": status}

    def ar1_ffbs(self, y, v, sv, *, z=None, seed=0, series_offset=0, want_filt=True, want_theta=True, times=None):
        """Scalar AR(1) FFBS, one lane per series (dlm_ar1_ffbs_batch; FilterAr.scala:15-82).  y [N][T] (NaN = missing),
        v [N][T] or [T] per-step observation variances, sv [N][3] or [3] = (phi, mu, sigma_eta).  With `times` [T] it is
        the Ornstein-Uhlenbeck variant on that grid (dlm_ou_ffbs_batch; FilterOu.scala:7-79)."""
        be = self._backend(y)
        N, T = int(y.shape[0]), int(y.shape[1])
        yb = be.put(y).reshape(N, T)
        vv = np.asarray(v, dtype=np.float64) if not hasattr(v, "data_ptr") else v
        if np.ndim(vv) == 0:
            vv = np.full(T, float(vv))
        v_stride = T if vv.ndim == 2 else 0
        svv = np.asarray(sv, dtype=np.float64) if not hasattr(sv, "data_ptr") else sv
        sv_stride = 3 if svv.ndim == 2 else 0
        vb, sb, zb = be.put(vv), be.put(svv), be.put(z)
        filt = be.empty((N, T + 1, 2)) if want_filt else None
        theta = be.empty((N, T + 1)) if want_theta else None
        status = be.empty((N,), np.int32)
        op = _lib.Options(0, be.mem, seed, series_offset)
        if times is None:
            self._check(self.lib.dlm_ar1_ffbs_batch(self.h, N, T, be.ptr(yb), be.ptr(vb), v_stride, be.ptr(sb), sv_stride,
                                                    be.ptr(zb), op, be.ptr(filt), be.ptr(theta), be.ptr(status)))
        else:
            tb = be.put(np.asarray(times, dtype=np.float64))
            self._check(self.lib.dlm_ou_ffbs_batch(self.h, N, T, be.ptr(tb), be.ptr(yb), be.ptr(vb), v_stride, be.ptr(sb),
                                                   sv_stride, be.ptr(zb), op, be.ptr(filt), be.ptr(theta), be.ptr(status)))
        return {"filt": filt, "theta": theta, "status": status}

    def dinvgamma_step(self, d, p, stats, prior_v, prior_w, *, iteration, seed=0, series_offset=0):
        """GibbsSampling.dinvGammaStep on the device (dlm_dinvgamma_step_batch): stats [N][2p + d + 1] from an FFBS call
        -> (V [N][p*p], W [N][d*d]) dense diagonal matrices, ready as the next call's per-series parameters.
        prior_v / prior_w: (shape, scale) pairs or objects with .shape / .scale."""
        be = self._backend(stats)
        N = int(stats.shape[0])
        sb = be.put(stats)
        av, bv = (prior_v.shape, prior_v.scale) if hasattr(prior_v, "scale") else prior_v
        aw, bw = (prior_w.shape, prior_w.scale) if hasattr(prior_w, "scale") else prior_w
        V = be.empty((N, p * p)); W = be.empty((N, d * d))
        op = _lib.Options(0, be.mem, seed, series_offset)
        self._check(self.lib.dlm_dinvgamma_step_batch(self.h, d, p, N, be.ptr(sb), float(av), float(bv), float(aw), float(bw),
                                                      int(iteration), op, be.ptr(V), be.ptr(W)))
        return V, W

    def simulate(self, mat, params, N, *, seed=0, series_offset=0, device=False, want_x=True):
        """Dlm.simulateRegular over the model's time grid for N series (dlm_simulate_batch): (x [N][T+1][d], y [N][T][p])."""
        be = _Device(self.device) if device else _Host()
        d, p, T = mat.d, mat.p, mat.T
        md, pd, op, keep = self.prepare(mat, params, N, be, 0, seed, series_offset)
        x = be.empty((N, T + 1, d)) if want_x else None
        y = be.empty((N, T, p))
        status = be.empty((N,), np.int32)
        self._check(self.lib.dlm_simulate_batch(self.h, md, pd, op, be.ptr(x), be.ptr(y), be.ptr(status)))
        return {"x": x, "y": y, "status": status}

    def smooth(self, mat, params, filt, *, flags=0):
        be = self._backend(filt)
        N = int(filt.shape[0]); d, T = mat.d, mat.T
        md, pd, op, keep = self.prepare(mat, params, N, be, flags)
        fb = be.put(filt)
        smooth = be.empty((N, T + 1, d + d * d))
        status = be.empty((N,), np.int32)
        self._check(self.lib.dlm_smooth_batch(self.h, md, pd, be.ptr(fb), op, be.ptr(smooth), be.ptr(status)))
        return {"smooth": smooth, "status": status}

    def filter_smooth(self, mat, params, y, *, flags=0, out=None, want_filt=True):
        """Fused filter + smoother.  want_filt=False keeps the filtered records inside the engine (packed on the
        structured fast path) and returns only the smoothed moments.  flags & OPT_PACKED_SYM: both outputs are packed
        records [mean | lower triangle by rows] (dlm_packed_record_doubles(d) doubles; unpack_records expands them)."""
        be = self._backend(y)
        N = int(y.shape[0]); d, p, T = mat.d, mat.p, mat.T
        recw = self.lib.dlm_packed_record_doubles(d) if flags & _lib.OPT_PACKED_SYM else d + d * d
        yb = be.put(y).reshape(N, T, p)
        md, pd, op, keep = self.prepare(mat, params, N, be, flags)
        self._hold(flags, yb)
        filt = (out["filt"] if out else be.empty((N, T + 1, recw))) if want_filt else None
        smooth = out["smooth"] if out else be.empty((N, T + 1, recw))
        status = out["status"] if out else be.empty((N,), np.int32)
        self._check(self.lib.dlm_filter_smooth_batch(self.h, md, pd, be.ptr(yb), op, be.ptr(filt),
                                                     be.ptr(smooth), be.ptr(status)))
        return {"filt": filt, "smooth": smooth, "status": status}

    def unpack_records(self, d, packed):
        """Packed records [..., dlm_packed_record_doubles(d)] -> dense [..., d + d*d] (dlm_unpack_records)."""
        be = self._backend(packed)
        pb = be.put(packed)
        count = int(np.prod(pb.shape[:-1]))
        dense = be.empty(tuple(pb.shape[:-1]) + (d + d * d,))
        op = _lib.Options(0, be.mem, 0, 0)
        self._check(self.lib.dlm_unpack_records(self.h, int(d), count, be.ptr(pb), op, be.ptr(dense)))
        return dense

    def last_timing(self):
        """(forward_ms, backward_ms) of the last fused call, from HIP events on the engine stream."""
        ms = (ctypes.c_double * 2)()
        self._check(self.lib.dlm_last_timing(self.h, ctypes.byref(ms)))
        return float(ms[0]), float(ms[1])

    def last_counters(self):
        """(forward steady steps, backward steady steps, series on the shared-covariance path, series on their own
        recursion) of the last call made with DLM_OPT_COUNT_STEPS."""
        c = (ctypes.c_uint64 * 4)()
        self._check(self.lib.dlm_last_counters(self.h, ctypes.byref(c)))
        return tuple(int(v) for v in c)

    def ffbs(self, mat, params, y, *, z=None, seed=0, series_offset=0, flags=0, want_theta=True,
             want_cond=False, want_stats=True, filt=None, want_filt=True):
        """FFBS (filt=None) or backward sampling from existing filter records (filt given).  want_filt=False: the forward
        pass's records are not wanted (dlm_ffbs_batch with filt_ws = NULL)."""
        be = self._backend(y)
        N = int(y.shape[0]); d, p, T = mat.d, mat.p, mat.T
        rec = d + d * d
        yb = be.put(y).reshape(N, T, p)
        zb = be.put(z)
        md, pd, op, keep = self.prepare(mat, params, N, be, flags, seed, series_offset)
        self._hold(flags, yb, zb)
        theta = be.empty((N, T + 1, d)) if want_theta else None
        cond = be.empty((N, T + 1, rec)) if want_cond else None
        L = self.lib.dlm_stats_len(d, p, flags & ~_lib.OPT_FFBS_SIMSMOOTH)
        stats = be.empty((N, L)) if want_stats else None
        status = be.empty((N,), np.int32)
        if filt is None:
            ws = be.empty((N, T + 1, rec)) if want_filt else None
            self._check(self.lib.dlm_ffbs_batch(self.h, md, pd, be.ptr(yb), be.ptr(zb), op, be.ptr(ws),
                                                be.ptr(theta), be.ptr(cond), be.ptr(stats), be.ptr(status)))
        else:
            ws = be.put(filt)
            self._check(self.lib.dlm_backward_sample_batch(self.h, md, pd, be.ptr(yb), be.ptr(ws), be.ptr(zb),
                                                           op, be.ptr(theta), be.ptr(cond), be.ptr(stats),
                                                           be.ptr(status)))
        return {"theta": theta, "cond": cond, "stats": stats, "filt": ws, "status": status}

    def svd_filter(self, mat, params, y, *, flags=0):
        be = self._backend(y)
        N = int(y.shape[0]); d, p, T = mat.d, mat.p, mat.T
        yb = be.put(y).reshape(N, T, p)
        md, pd, op, keep = self.prepare(mat, params, N, be, flags)
        rec = be.empty((N, T + 1, 2 * d + d * d))
        status = be.empty((N,), np.int32)
        self._check(self.lib.dlm_svd_filter_batch(self.h, md, pd, be.ptr(yb), op, be.ptr(rec), be.ptr(status)))
        return {"svd": rec, "status": status}

    def svd_ffbs(self, mat, params, y, *, z=None, seed=0, series_offset=0, flags=0, want_stats=True):
        be = self._backend(y)
        N = int(y.shape[0]); d, p, T = mat.d, mat.p, mat.T
        yb = be.put(y).reshape(N, T, p)
        zb = be.put(z)
        md, pd, op, keep = self.prepare(mat, params, N, be, flags, seed, series_offset)
        ws = be.empty((N, T + 1, 2 * d + d * d))
        theta = be.empty((N, T + 1, d))
        L = self.lib.dlm_stats_len(d, p, flags)
        stats = be.empty((N, L)) if want_stats else None
        status = be.empty((N,), np.int32)
        self._check(self.lib.dlm_svd_ffbs_batch(self.h, md, pd, be.ptr(yb), be.ptr(zb), op, be.ptr(ws),
                                                be.ptr(theta), be.ptr(stats), be.ptr(status)))
        return {"svd": ws, "theta": theta, "stats": stats, "status": status}

    def stats_pool(self, stats):
        be = self._backend(stats)
        N, L = int(stats.shape[0]), int(stats.shape[1])
        sb = be.put(stats)
        pooled = be.empty((L,))
        op = _lib.Options(0, be.mem, 0, 0)
        self._check(self.lib.dlm_stats_pool(self.h, be.ptr(sb), N, L, be.ptr(pooled), op))
        return pooled

    # -- RCCL ----------------------------------------------------------------------
    def comm_unique_id(self) -> bytes:
        buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
        rc = self.lib.dlm_comm_unique_id(buf)
        if rc != 0:
            raise EngineError(f"dlm_comm_unique_id failed ({rc})")
        return buf.raw

    def comm_init_rank(self, nranks: int, rank: int, uid: bytes):
        self._check(self.lib.dlm_comm_init_rank(self.h, nranks, rank, ctypes.create_string_buffer(uid, _lib.COMM_ID_BYTES)))

    def allreduce_stats(self, stats_dev):
        """In-place RCCL sum of a device fp64 tensor."""
        self._check(self.lib.dlm_gibbs_suffstats_allreduce(self.h, ctypes.c_void_p(stats_dev.data_ptr()),
                                                           stats_dev.numel()))
        return stats_dev
