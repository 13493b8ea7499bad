#!/usr/bin/env python3
"""Headline benchmark: Kalman filter + smoother throughput (series*timesteps / second).

Workload (BASELINE.json configs[1], SURVEY.md 8d "C2"): seasonal DLM `polynomial(1) |+|
seasonal(24, 6)` (d = 13, p = 1), 10 000 series x T = 1000, fused filter + smoother on one
MI355X through the C ABI (dlm_filter_smooth_batch), inputs and outputs resident in HBM.
A "step" is one pass of the hot path over the whole batch.

  python bench.py --gpus N --steps K --warmup W
N > 1 is launched by the driver with torch.distributed.run (one rank per GPU).  Series are
independent, so ranks shard them with no data-path collective: each rank filters+smooths its
own 10 000 series (weak scaling) and the whole-job value is the sum over ranks.

Prints ONE JSON line on rank 0 (contract in the task description) with the extra objects
  roofline     -- the dominant kernel (backward pass) against the HBM roof, timed with HIP
                  events on the engine's stream inside the timed region
  cpu_baseline -- the oracle's batched filter+smoother (a C port of the Scala operation
                  sequence; the Scala/Breeze reference cannot run: no JVM on the box)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is achievable


def measured_traffic(kernel, N, T):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/r01_hbm_traffic.json: FETCH_SIZE doubled per the gfx950 note + WRITE_SIZE), valid
    for the default workload only; None otherwise (PMC counters cannot be read from in here)."""
    path = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
    if not os.path.exists(path) or (N, T) != (10000, 1000):
        return None
    k = json.load(open(path))["kernels"].get(kernel)
    return None if k is None else k["hbm_bytes_per_launch"]


def seasonal_c2():
    from bayesian_dlms_amd.dlm import Dlm, DlmParameters, materialise
    mod = Dlm.polynomial(1) + Dlm.seasonal(24, 6)
    w = np.diag([0.01, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4])
    p = DlmParameters([[1.0]], w, np.zeros(13), np.eye(13))
    return mod, p


def simulate(mat, p, N, seed):
    """x0 ~ N(m0, C0), x_t = G x_{t-1} + w_t, y_t = F^T x_t + v_t (Dlm.scala:245-292), vectorised
    over series; Philox-keyed numpy generator."""
    rng = np.random.Generator(np.random.Philox(key=seed))
    d, T = mat.d, mat.T
    G = mat.G[: d * d].reshape(d, d).T
    F = mat.F[:d]
    sw = np.sqrt(np.diag(p.w)); sv = float(np.sqrt(p.v[0, 0]))
    x = p.m0 + rng.standard_normal((N, d)) * np.sqrt(np.diag(p.c0))
    y = np.empty((N, T, 1))
    for t in range(T):
        x = x @ G.T + rng.standard_normal((N, d)) * sw
        y[:, t, 0] = x @ F + rng.standard_normal(N) * sv
    return y


def cpu_baseline(mat, p, y_host, budget_s=12.0):
    """Oracle (C port of the reference's per-series op sequence) on the host cores, OpenMP over
    series, on a bounded sample of the same workload."""
    import oracle
    om = oracle.Model(mat.d, mat.p, mat.T, mat.F, mat.G, mat.g_index, mat.dt, mat.f_stride)
    cores = os.cpu_count() or 1
    n = min(y_host.shape[0], 2 * cores)
    t0 = time.perf_counter()
    oracle.filter_smooth_batch(n, om, p.v, p.w, p.m0, p.c0, y_host[:n], want_out=False)
    dt = time.perf_counter() - t0
    rate = n * mat.T / dt
    n2 = int(min(y_host.shape[0], max(n, rate * budget_s / mat.T)))
    n2 = max(cores, (n2 // cores) * cores)
    t0 = time.perf_counter()
    oracle.filter_smooth_batch(n2, om, p.v, p.w, p.m0, p.c0, y_host[:n2], want_out=False)
    dt = time.perf_counter() - t0
    return {"value": n2 * mat.T / dt, "unit": "series*timesteps/s", "cores": cores, "kind": "port",
            "sample": f"{n2} of the {y_host.shape[0]} series x T={mat.T} (same inputs), {dt:.1f} s, "
                      "oracle/dlm_oracle.c: Joseph-form update + LU solves, OpenMP over series"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--series", type=int, default=10000, help="series per GPU")
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--flags", type=int, default=0, help="DLM_OPT_* bits (e.g. 8 = force generic kernels)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from bayesian_dlms_amd.dlm import materialise
    from bayesian_dlms_amd.engine import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    mod, p = seasonal_c2()
    N, T = args.series, args.T
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    d = mat.d
    rec = d + d * d
    y_host = simulate(mat, p, N, seed=0xD1A5EED0 + rank)
    y = torch.as_tensor(y_host, device=dev)

    eng = Engine(local)
    out = {"filt": torch.empty((N, T + 1, rec), dtype=torch.float64, device=dev),
           "smooth": torch.empty((N, T + 1, rec), dtype=torch.float64, device=dev),
           "status": torch.empty((N,), dtype=torch.int32, device=dev)}

    def step():
        eng.filter_smooth(mat, p, y, flags=args.flags, out=out)

    for _ in range(args.warmup):
        step()
    fwd_ms, bwd_ms = [], []
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()                       # synchronous on return (engine stream is drained)
        f, b = eng.last_timing()     # HIP events recorded by the engine around its two kernels
        fwd_ms.append(f); bwd_ms.append(b)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    status_bad = int((out["status"] != 0).sum().item())
    units = world * N * T * args.steps
    value = units / elapsed

    if rank == 0:
        # dominant kernel = backward pass: reads [m|C], writes [s|S] -> 16 (d + d^2) B per series-step
        f_ms, b_ms = float(np.mean(fwd_ms)), float(np.mean(bwd_ms))
        bwd_bytes = 16.0 * rec * N * T
        fwd_bytes = (8.0 + 8.0 * rec) * N * T
        dom_is_bwd = b_ms >= f_ms
        dom_bytes, dom_ms = (bwd_bytes, b_ms) if dom_is_bwd else (fwd_bytes, f_ms)
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        kname = ("k_smoother_" if dom_is_bwd else "k_filter_") + {"sparse16": "sp16", "mfma16": "mfma16"}.get(eng.last_variant, eng.last_variant)
        line = {
            "metric": "Kalman filter+smooth series*timesteps/sec", "value": value,
            "unit": "series*timesteps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2: seasonal DLM polynomial(1)|+|seasonal(24,6), d=13, p=1, "
                                   f"{N} series/GPU x T={T}, fused filter+smooth (dlm_filter_smooth_batch)",
                       "series_per_gpu": N, "T": T, "d": d, "p": 1, "variant": eng.last_variant,
                       "parallelism": f"series-sharded x{world}, no collective"},
            "roofline": {"bound": "hbm", "kernel": kname,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(kname, N, T),
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom_ms,
                         "forward_ms": f_ms, "backward_ms": b_ms,
                         "path_GBps": (fwd_bytes + bwd_bytes) / ((f_ms + b_ms) * 1e-3) / 1e9},
            "status_nonzero_series": status_bad,
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(mat, p, y_host)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
