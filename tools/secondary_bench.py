#!/usr/bin/env python3
"""Secondary measurements (not the headline): FFBS/Gibbs (C3), d=40 multivariate (C4), SVD filter (C5).
Prints one JSON object per config.  Usage: python tools/secondary_bench.py [c3] [c4] [c5]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import seasonal_c2, simulate
from bayesian_dlms_amd.dlm import Dlm, DlmParameters, materialise
from bayesian_dlms_amd.engine import Engine

def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

def main():
    which = sys.argv[1:] or ["c3", "c4", "c5"]
    eng = Engine(0); dev = torch.device("cuda", 0)
    if "c3" in which:
        mod, p = seasonal_c2(); N, T = 10000, 1000
        mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
        y = torch.as_tensor(simulate(mat, p, N, seed=1), device=dev)
        from bayesian_dlms_amd import _lib
        for name, flags in (("reference-form backward sampling", 0), ("simulation smoother", _lib.OPT_FFBS_SIMSMOOTH)):
            dt = timeit(lambda: eng.ffbs(mat, p, y, seed=3, want_theta=False, want_stats=True, flags=flags))
            print(json.dumps({"config": f"C3 FFBS + suff-stats ({name}), d=13, N=10000, T=1000", "variant": eng.last_variant,
                              "ms": dt * 1e3, "series_steps_per_s": N * T / dt}))
    if "c4" in which:
        mod = Dlm.polynomial(2)
        for _ in range(19): mod = mod * Dlm.polynomial(2)
        N, T = 2000, 1000
        mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
        rng = np.random.default_rng(40); A = rng.standard_normal((40, 40))
        p = DlmParameters(np.eye(20), A @ A.T / 40 + 0.1 * np.eye(40), np.zeros(40), np.eye(40))
        y = torch.as_tensor(rng.standard_normal((N, T, 20)).cumsum(axis=1), device=dev)
        dt = timeit(lambda: eng.filter_smooth(mat, p, y), reps=1)
        fwd, bwd = eng.last_timing()
        print(json.dumps({"config": "C4 filter+smooth, d=40, p=20, N=2000, T=1000", "variant": eng.last_variant,
                          "ms": dt * 1e3, "forward_ms": fwd, "backward_ms": bwd, "series_steps_per_s": N * T / dt}))
    if "c4ffbs" in which:
        from bayesian_dlms_amd import _lib
        mod = Dlm.polynomial(2)
        for _ in range(19): mod = mod * Dlm.polynomial(2)
        N, T = 256, 200
        mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
        rng = np.random.default_rng(40); A = rng.standard_normal((40, 40))
        p = DlmParameters(np.eye(20), A @ A.T / 40 + 0.1 * np.eye(40), np.zeros(40), np.eye(40))
        y = torch.as_tensor(rng.standard_normal((N, T, 20)).cumsum(axis=1), device=dev)
        for name, fl in (("reference-form", 0), ("simulation smoother", _lib.OPT_FFBS_SIMSMOOTH)):
            dt = timeit(lambda: eng.ffbs(mat, p, y, seed=1, want_theta=False, flags=_lib.OPT_STATS_OUTER | fl), reps=1)
            print(json.dumps({"config": f"C4 FFBS + outer-product stats ({name}), d=40, p=20, N=256, T=200", "variant": eng.last_variant,
                              "ms": dt * 1e3, "series_steps_per_s": N * T / dt}))
    if "c4gibbs" in which:   # the named C4 configuration: 2000 series, simulation-smoother FFBS + Inverse-Wishart statistics
        from bayesian_dlms_amd import _lib
        mod = Dlm.polynomial(2)
        for _ in range(19): mod = mod * Dlm.polynomial(2)
        N, T = 2000, 1000
        mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
        rng = np.random.default_rng(40); A = rng.standard_normal((40, 40))
        p = DlmParameters(np.eye(20), A @ A.T / 40 + 0.1 * np.eye(40), np.zeros(40), np.eye(40))
        y = torch.as_tensor(rng.standard_normal((N, T, 20)).cumsum(axis=1), device=dev)
        dt = timeit(lambda: eng.ffbs(mat, p, y, seed=1, want_theta=False, flags=_lib.OPT_STATS_OUTER | _lib.OPT_FFBS_SIMSMOOTH), reps=2)
        print(json.dumps({"config": "C4 FFBS + outer-product stats (simulation smoother), d=40, p=20, N=2000, T=1000", "variant": eng.last_variant,
                          "ms": dt * 1e3, "series_steps_per_s": N * T / dt}))
    if "lane" in which:   # the smallest models over very many series: one lane per series (dlm_lane.hip)
        for name, mod, d, N in (("local level (polynomial(1)), d=1", Dlm.polynomial(1), 1, 500000),
                                ("linear growth (polynomial(2)), d=2", Dlm.polynomial(2), 2, 200000),
                                ("level + two harmonics (polynomial(1) + seasonal(12, 2)), d=5", Dlm.polynomial(1) + Dlm.seasonal(12, 2), 5, 50000)):
            T = 1000
            mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
            p = DlmParameters([[2.0]], np.eye(d) * 0.5, np.zeros(d), np.eye(d) * 10.0)
            y = torch.randn((N, T, 1), device=dev, dtype=torch.float64).cumsum(dim=1)
            out = {"filt": torch.empty((N, T + 1, d + d * d), device=dev, dtype=torch.float64),
                   "smooth": torch.empty((N, T + 1, d + d * d), device=dev, dtype=torch.float64),
                   "status": torch.zeros((N,), device=dev, dtype=torch.int32)}
            dt = timeit(lambda: eng.filter_smooth(mat, p, y, out=out))
            fwd, bwd = eng.last_timing()
            rec = (d + d * d) * 8
            print(json.dumps({"config": f"filter+smooth, {name}, N={N}, T={T}", "variant": eng.last_variant, "ms": dt * 1e3,
                              "forward_ms": fwd, "backward_ms": bwd, "series_steps_per_s": N * T / dt,
                              "GBps_algorithmic": N * T * (8 + 3 * rec) / dt / 1e9}))
    if "laneffbs" in which:   # FFBS + sufficient statistics for the smallest models (the Gibbs step of a local-level / linear-growth DLM)
        from bayesian_dlms_amd import _lib
        for name, mod, d, N in (("local level, d=1", Dlm.polynomial(1), 1, 200000), ("linear growth, d=2", Dlm.polynomial(2), 2, 100000)):
            T = 1000
            mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
            p = DlmParameters([[2.0]], np.eye(d) * 0.5, np.zeros(d), np.eye(d) * 10.0)
            y = torch.randn((N, T, 1), device=dev, dtype=torch.float64).cumsum(dim=1)
            dt = timeit(lambda: eng.ffbs(mat, p, y, seed=1, flags=_lib.OPT_FFBS_SIMSMOOTH), reps=2)
            print(json.dumps({"config": f"FFBS (simulation smoother) + statistics, {name}, N={N}, T={T}", "variant": eng.last_variant,
                              "ms": dt * 1e3, "series_steps_per_s": N * T / dt}))
    if "mv32" in which:   # d = 32, p = 16 (16 x linear growth): the two-tile instantiations of the per-wave kernels
        mod = Dlm.polynomial(2)
        for _ in range(15): mod = mod * Dlm.polynomial(2)
        N, T = 4000, 1000
        mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
        p = DlmParameters(np.eye(16), np.eye(32) * 0.2, np.zeros(32), np.eye(32))
        y = torch.randn((N, T, 16), device=dev, dtype=torch.float64).cumsum(dim=1)
        dt = timeit(lambda: eng.filter_smooth(mat, p, y), reps=2)
        fwd, bwd = eng.last_timing()
        print(json.dumps({"config": f"filter+smooth, d=32, p=16 (16 x linear growth), N={N}, T={T}", "variant": eng.last_variant,
                          "ms": dt * 1e3, "forward_ms": fwd, "backward_ms": bwd, "series_steps_per_s": N * T / dt}))
    if "mv8" in which:   # a small multivariate model (d = 8, p = 4): per-wave kernels with one tile per dimension vs the generic path
        from bayesian_dlms_amd import _lib
        mod = Dlm.polynomial(2) * Dlm.polynomial(2) * Dlm.polynomial(2) * Dlm.polynomial(2)
        N, T = 10000, 1000
        mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
        p = DlmParameters(np.eye(4), np.eye(8) * 0.2, np.zeros(8), np.eye(8))
        y = torch.randn((N, T, 4), device=dev, dtype=torch.float64).cumsum(dim=1)
        for name, fl in (("per-wave", 0), ("generic", _lib.OPT_FORCE_GENERIC)):
            dt = timeit(lambda: eng.filter_smooth(mat, p, y, flags=fl), reps=2)
            print(json.dumps({"config": f"filter+smooth, d=8, p=4 (4 x linear growth), N={N}, T={T} ({name})", "variant": eng.last_variant,
                              "ms": dt * 1e3, "series_steps_per_s": N * T / dt}))
    if "c2s" in which:
        mod, p = seasonal_c2(); N, T = 10000, 1000
        mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
        y = torch.as_tensor(simulate(mat, p, N, seed=1), device=dev)
        dt = timeit(lambda: eng.filter_smooth(mat, p, y, want_filt=False))
        fwd, bwd = eng.last_timing()
        print(json.dumps({"config": "C2 filter+smooth, smoothed moments only (filt = NULL: packed internal records), d=13, N=10000, T=1000",
                          "variant": eng.last_variant, "ms": dt * 1e3, "forward_ms": fwd, "backward_ms": bwd, "series_steps_per_s": N * T / dt}))
    if "ll" in which:
        mod, p = seasonal_c2(); N, T = 10000, 1000
        mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
        y = torch.as_tensor(simulate(mat, p, N, seed=1), device=dev)
        dt = timeit(lambda: eng.loglik(mat, p, y))
        print(json.dumps({"config": "log-likelihood (prediction-error decomposition), d=13, N=10000, T=1000, no record output",
                          "variant": eng.last_variant, "ms": dt * 1e3, "series_steps_per_s": N * T / dt}))
    if "c3gibbs" in which:
        from bayesian_dlms_amd.gibbs import InverseGamma, gibbs_dinvgamma_device
        mod, p = seasonal_c2(); N, T = 10000, 1000
        times = np.arange(1, T + 1, dtype=np.float64)
        mat = materialise(mod, times)
        y = torch.as_tensor(simulate(mat, p, N, seed=1), device=dev)
        for name, sim in (("simulation smoother", True), ("reference-form backward sampling", False)):
            iters = 10 if sim else 2
            run = lambda: gibbs_dinvgamma_device(mod, InverseGamma(5.0, 4.0), InverseGamma(17.0, 4.0), p, times, y, eng,
                                                 n_iter=iters, seed=1, simulation_smoother=sim)
            dt = timeit(run, reps=1) / iters
            print(json.dumps({"config": f"C3 d-Inverse-Gamma Gibbs, per-series V and W, device-resident ({name}), d=13, N=10000, T=1000",
                              "ms_per_iteration": dt * 1e3, "series_steps_per_s": N * T / dt}))
    if "ar1" in which:
        N, T = 100000, 1000
        rng = np.random.default_rng(3)
        y = torch.as_tensor(rng.standard_normal((N, T)).cumsum(axis=1) * 0.1, device=dev)
        v = torch.as_tensor(rng.uniform(0.2, 2.0, (N, T)), device=dev)
        sv = torch.as_tensor(np.stack([rng.uniform(0.5, 0.95, N), rng.standard_normal(N), rng.uniform(0.1, 0.5, N)], axis=1), device=dev)
        dt = timeit(lambda: eng.ar1_ffbs(y, v, sv, seed=1, want_filt=False))
        print(json.dumps({"config": "scalar AR(1) FFBS (FilterAr.ffbs), one lane per series, N=100000, T=1000, per-step v_t",
                          "variant": eng.last_variant, "ms": dt * 1e3, "series_steps_per_s": N * T / dt,
                          "GBps_algorithmic": N * T * (8 * 2 + 16 * 2 + 8) / dt / 1e9}))
    if "c5" in which:
        mod, p = seasonal_c2(); N, T = 10000, 200
        mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
        y = torch.as_tensor(simulate(mat, p, N, seed=1), device=dev)
        dt = timeit(lambda: eng.svd_filter(mat, p, y), reps=1)
        print(json.dumps({"config": "C5 SVD filter, d=13, N=10000, T=200", "variant": eng.last_variant,
                          "ms": dt * 1e3, "series_steps_per_s": N * T / dt}))

if __name__ == "__main__":
    main()
