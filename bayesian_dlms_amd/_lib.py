"""ctypes binding of the C ABI (include/dlm_engine.h).  Fails loudly when the HIP engine
library is missing or cannot be loaded: there is no CPU fallback in the product path."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# DLM_ENGINE_LIB selects an alternative build of the same C ABI (kernel A/B experiments)
LIB_PATH = os.environ.get("DLM_ENGINE_LIB") or os.path.join(HERE, "libdlm_engine.so")

DLM_MEM_DEVICE, DLM_MEM_HOST = 0, 1
OPT_SMOOTHER_COMPAT_Q1 = 1 << 0
OPT_SVD_RAW_W_Q2 = 1 << 1
OPT_SVD_SAMPLER_Q9 = 1 << 2
OPT_FORCE_GENERIC = 1 << 3
OPT_STATS_OUTER = 1 << 4
OPT_ASYNC = 1 << 5
OPT_FFBS_SIMSMOOTH = 1 << 6
OPT_PACKED_SYM = 1 << 7
OPT_MODEL_UNCHANGED = 1 << 8
OPT_COUNT_STEPS = 1 << 9
OPT_TRUST_MODEL_UNCHANGED = 1 << 10
OPT_LOGLIK_LITERAL_Q7 = 1 << 11
OPT_NO_LANE = 1 << 16
OPT_NO_SAMPLER16 = 1 << 17
OPT_NO_WAVE = 1 << 18
OPT_FORCE_WAVE = 1 << 19
OPT_NO_SPARSE_F = 1 << 20
OPT_NO_SMALL_BATCH = 1 << 21
OPT_NO_STEADY = 1 << 22
OPT_NO_PIPE = 1 << 23
OPT_SHARED_COV = 1 << 24
OPT_SVD_PER_SERIES = 1 << 25
OPT_SAMPLER_PER_SERIES = 1 << 26
OPT_DRAW_EIG = 1 << 27
OPT_SMOOTHER_PER_SERIES = 1 << 28
OPT_TEST_FAIL_AFTER_TABLES = 1 << 30
ST_NONFINITE, ST_NOT_PD, ST_NOCONV = 1, 2, 4
COMM_ID_BYTES = 128

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)


class ModelDesc(ctypes.Structure):
    _fields_ = [("d", ctypes.c_int32), ("p", ctypes.c_int32), ("T", ctypes.c_int32), ("N", ctypes.c_int32),
                ("F", ctypes.c_void_p), ("f_stride", ctypes.c_int64),
                ("G", ctypes.c_void_p), ("n_g", ctypes.c_int32), ("g_index", ctypes.c_void_p),
                ("dt", ctypes.c_void_p)]


class ParamsDesc(ctypes.Structure):
    _fields_ = [("V", ctypes.c_void_p), ("v_stride", ctypes.c_int64),
                ("W", ctypes.c_void_p), ("w_stride", ctypes.c_int64),
                ("m0", ctypes.c_void_p), ("m0_stride", ctypes.c_int64),
                ("C0", ctypes.c_void_p), ("c0_stride", ctypes.c_int64),
                ("v_tstride", ctypes.c_int64), ("w_tstride", ctypes.c_int64)]


class Options(ctypes.Structure):
    _fields_ = [("flags", ctypes.c_uint32), ("mem", ctypes.c_int32),
                ("seed", ctypes.c_uint64), ("series_offset", ctypes.c_uint64)]


# every symbol include/dlm_engine.h declares: (name, restype, argtypes)
_V = ctypes.c_void_p
_MP, _PP, _OP = ctypes.POINTER(ModelDesc), ctypes.POINTER(ParamsDesc), ctypes.POINTER(Options)
SYMBOLS = [
    ("dlm_engine_create", ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_V)]),
    ("dlm_engine_destroy", None, [_V]),
    ("dlm_last_error", ctypes.c_char_p, [_V]),
    ("dlm_version", ctypes.c_char_p, []),
    ("dlm_engine_set_stream", ctypes.c_int, [_V, _V]),
    ("dlm_engine_sync", ctypes.c_int, [_V]),
    ("dlm_last_variant", ctypes.c_char_p, [_V]),
    ("dlm_engine_wait_stream", ctypes.c_int, [_V, _V]),
    ("dlm_stream_wait_engine", ctypes.c_int, [_V, _V]),
    ("dlm_buffer_alloc", ctypes.c_int, [_V, ctypes.c_uint64, ctypes.POINTER(_V)]),
    ("dlm_buffer_free", ctypes.c_int, [_V, _V]),
    ("dlm_buffer_upload", ctypes.c_int, [_V, _V, ctypes.c_uint64, _V, ctypes.c_uint64]),
    ("dlm_buffer_download", ctypes.c_int, [_V, _V, ctypes.c_uint64, _V, ctypes.c_uint64]),
    ("dlm_buffer_fill", ctypes.c_int, [_V, _V, ctypes.c_uint64, ctypes.c_int, ctypes.c_uint64]),
    ("dlm_device_mem_info", ctypes.c_int, [_V, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    ("dlm_packed_record_doubles", ctypes.c_int32, [ctypes.c_int32]),
    ("dlm_unpack_records", ctypes.c_int, [_V, ctypes.c_int32, ctypes.c_int64, _V, _OP, _V]),
    ("dlm_filter_batch", ctypes.c_int, [_V, _MP, _PP, _V, _OP, _V, _V, _V, _V]),
    ("dlm_loglik_batch", ctypes.c_int, [_V, _MP, _PP, _V, _OP, _V, _V]),
    ("dlm_simulate_batch", ctypes.c_int, [_V, _MP, _PP, _OP, _V, _V, _V]),
    ("dlm_dinvgamma_step_batch", ctypes.c_int, [_V, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _V, ctypes.c_double,
                                                ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_uint64, _OP, _V, _V]),
    ("dlm_ou_ffbs_batch", ctypes.c_int, [_V, ctypes.c_int32, ctypes.c_int32, _V, _V, _V, ctypes.c_int64, _V, ctypes.c_int64,
                                         _V, _OP, _V, _V, _V]),
    ("dlm_ar1_ffbs_batch", ctypes.c_int, [_V, ctypes.c_int32, ctypes.c_int32, _V, _V, ctypes.c_int64, _V, ctypes.c_int64,
                                          _V, _OP, _V, _V, _V]),
    ("dlm_smooth_batch", ctypes.c_int, [_V, _MP, _PP, _V, _OP, _V, _V]),
    ("dlm_filter_smooth_batch", ctypes.c_int, [_V, _MP, _PP, _V, _OP, _V, _V, _V]),
    ("dlm_last_timing", ctypes.c_int, [_V, ctypes.POINTER(ctypes.c_double * 2)]),
    ("dlm_last_counters", ctypes.c_int, [_V, ctypes.POINTER(ctypes.c_uint64 * 4)]),
    ("dlm_ffbs_batch", ctypes.c_int, [_V, _MP, _PP, _V, _V, _OP, _V, _V, _V, _V, _V]),
    ("dlm_stats_len", ctypes.c_int32, [ctypes.c_int32, ctypes.c_int32, ctypes.c_uint32]),
    ("dlm_backward_sample_batch", ctypes.c_int, [_V, _MP, _PP, _V, _V, _V, _OP, _V, _V, _V, _V]),
    ("dlm_svd_filter_batch", ctypes.c_int, [_V, _MP, _PP, _V, _OP, _V, _V]),
    ("dlm_svd_ffbs_batch", ctypes.c_int, [_V, _MP, _PP, _V, _V, _OP, _V, _V, _V, _V]),
    ("dlm_stats_pool", ctypes.c_int, [_V, _V, ctypes.c_int32, ctypes.c_int32, _V, _OP]),
    ("dlm_comm_unique_id", ctypes.c_int, [ctypes.c_char_p]),
    ("dlm_comm_init_rank", ctypes.c_int, [_V, ctypes.c_int32, ctypes.c_int32, ctypes.c_char_p]),
    ("dlm_gibbs_suffstats_allreduce", ctypes.c_int, [_V, _V, ctypes.c_int64]),
]

_lib = None


class EngineLibraryMissing(RuntimeError):
    pass


def load():
    """Load libdlm_engine.so; raise EngineLibraryMissing if it is absent (no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch wheels bundle their own libamdhip64.so.7 / librccl.so.1.  Two HIP runtimes in one
    # process cannot both own the GPU, so when torch is installed it is imported first and
    # the engine library binds to the runtime torch has already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise EngineLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -m bayesian_dlms_amd.build` "
            "(hipcc, gfx950).  The engine has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
