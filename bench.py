#!/usr/bin/env python3
"""Benchmark of the hot path through the C ABI.  Headline (default): Kalman filter + smoother throughput,
series*timesteps / second, on BASELINE.json configs[1] (SURVEY.md 8d "C2"): seasonal DLM `polynomial(1) |+|
seasonal(24, 6)` (d = 13, p = 1), 10 000 series x T = 1000, fused filter + smoother (dlm_filter_smooth_batch),
inputs and outputs resident in HBM.  A "step" is one pass of the hot path over the whole batch.

  python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4|c5] [--scaling strong|weak] ...

N > 1 is launched by the driver with torch.distributed.run (one rank per GPU).  Series are independent, so ranks
shard them with no data-path collective.  The contract case (BASELINE.json `metric`) is STRONG scaling --
the 10 000 series are split into contiguous blocks (gibbs.shard_bounds), 1250 per GPU at 8 -- and that is the default;
`--scaling weak` gives every rank the full batch instead.  `value` = all series of the job x T x steps / max-over-ranks time.

Other configs (so that the driver can time them; each prints the same one-line contract):
  c3  pooled d-Inverse-Gamma Gibbs (GibbsSampling.sample, Gibbs.scala:134-180): per iteration FFBS with on-device
      sufficient statistics, dlm_stats_pool, ONE RCCL all-reduce of 2p + d + 1 doubles (dlm_gibbs_suffstats_allreduce on a
      communicator whose id travels through the torch.distributed store), the same conjugate draw on every rank
  c4  d = 40, p = 20 (20 x polynomial(2) under |*|), 2000 series: fused filter + smoother, fp64-MFMA bound
  c4g the same model inside pooled Inverse-Wishart Gibbs (GibbsWishart.sample, GibbsWishart.scala:40-80): FFBS with
      outer-product statistics on the device, pooling, the RCCL all-reduce of 2p + d^2 + 1 doubles, one W and V draw
  c5  SVD (square-root) filter, d = 13, 10 000 series

The default run (no flags, one GPU) also times -- a few steps each, AFTER the headline's timed region -- the same workload with
the steady-state shortcut off (`value_full_recursion`) and with 5 % of the observations missing (`value_missing_0.05`), and the
other BASELINE configurations (`secondary`: c3 in both sampler forms (and with the shared factors off, and at one GPU's share of 8), c4 with and without the shortcut, c4g, c5, the literal-Q1
smoother), each with its own roofline by SURVEY 8d's units, so that the driver's record carries them.  `--no-secondary` skips that.

Prints ONE JSON line on rank 0 with the extra objects
  roofline     -- the dominant kernel against its roof (HBM bytes or fp64 MFMA flops per SURVEY 8d), timed with HIP
                  events on the engine's stream inside the timed region; `peak_measured` is a device copy on this box
  cpu_baseline -- the oracle's C port of the Scala operation sequence on the host cores (N = 1 only; the Scala/Breeze
                  reference cannot run: no JVM on the box)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); a device copy reaches 5.3-6.3 TB/s
MFMA_F64_PEAK_TFLOPS = 78.6  # fp64 vector / matrix peak per MI355X
W_C2 = [0.01, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4]   # SeasonalModel.scala:18


def seasonal_c2():
    from bayesian_dlms_amd.dlm import Dlm, DlmParameters
    mod = Dlm.polynomial(1) + Dlm.seasonal(24, 6)
    return mod, DlmParameters([[1.0]], np.diag(W_C2), np.zeros(13), np.eye(13))


def multivariate_c4():
    """|*| of 20 polynomial(2): d = 40, p = 20; V = I, W = A A^T / 40 + 0.1 I (seed 40), m0 = 0, C0 = I."""
    from bayesian_dlms_amd.dlm import Dlm, DlmParameters
    mod = Dlm.polynomial(2)
    for _ in range(19):
        mod = mod * Dlm.polynomial(2)
    A = np.random.default_rng(40).standard_normal((40, 40))
    return mod, DlmParameters(np.eye(20), A @ A.T / 40 + 0.1 * np.eye(40), np.zeros(40), np.eye(40))


def simulate(mat, p, N, seed):
    """x0 ~ N(m0, C0), x_t = G x_{t-1} + w_t, y_t = F^T x_t + v_t (Dlm.scala:245-292) on the host, vectorised over series
    (diagonal W, C0, p = 1: tools and tests; the bench itself simulates on the device, dlm_simulate_batch)."""
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


def profile_tag(args):
    """The name tools/profile_round4.sh gives this configuration (its bench arguments without blanks, dashes and dots), or None."""
    if args.T != 1000 or args.records != "dense" or args.gpus != 1:
        return None
    tag = args.config
    opts = [(args.flags != 0, f"flags{args.flags}"), (args.missing > 0.0, "missing" + str(args.missing).replace(".", "")),
            (args.semantics != "textbook", "semantics" + args.semantics.replace("-", "")), (args.series is not None, f"series{args.series}"),
            (args.config in ("c3", "c4g") and args.sampler != "reference", "sampler" + args.sampler)]
    return tag + "".join(name for cond, name in opts if cond)   # (several options: in this order, as the script lists them)


def traffic_from_profiles(kernel, args):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (2 x FETCH_SIZE per the gfx950 note + WRITE_SIZE,
    profiles/r04_hbm_traffic.json: one entry per configuration, kernel and grid size, tools/profile_round4.sh) -- PMC counters cannot be
    read from inside the run.  The figure is DROPPED (None, with the reason) when the .hip file that defines the kernel has changed
    since the counters were taken: a stale number is not reported."""
    import hashlib
    tag = profile_tag(args)
    path = os.path.join(ROOT, "profiles", "r04_hbm_traffic.json")
    if tag is None or not os.path.exists(path):
        return None, "no PMC pass for this configuration"
    ents = [e for e in json.load(open(path))["configs"].get(tag, []) if kernel in e["kernel"]]
    if not ents:
        return None, "no PMC pass for this configuration"
    e = max(ents, key=lambda x: x["grid_threads"])        # (the batch's launch, not the one-wave table launches of the same kernel)
    src = os.path.join(ROOT, "bayesian_dlms_amd", "csrc", e["source"] or "")
    if not e.get("source") or not os.path.exists(src) or hashlib.sha256(open(src, "rb").read()).hexdigest()[:16] != e["source_sha16"]:
        return None, f"PMC figure dropped: {e.get('source')} changed since profiles/r04_hbm_traffic.json was taken"
    return e["hbm_bytes_per_launch"], None


def measure_copy_peak(torch, dev, nbytes=4 << 30):
    """Device copy rate on this box, GB/s counting read + write (the ceiling a streaming kernel can reach here)."""
    a = torch.empty(nbytes // 8, dtype=torch.float64, device=dev).fill_(1.0)
    b = torch.empty_like(a)
    b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    del a, b
    return 2.0 * nbytes / (ms * 1e-3) / 1e9


def measure_write_peak(torch, dev, nbytes=8 << 30):
    """Device fill rate on this box, GB/s written: the ceiling of a kernel that mostly writes (the record streams of both passes) -- above the copy rate,
    which pays for a read stream as well (tools/micro/store_pattern.hip measures 6.0-6.2 TB/s for hand-written stores)."""
    a = torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
    a.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        a.zero_()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    del a
    return nbytes / (ms * 1e-3) / 1e9


def cpu_baseline(config, mat, p, y_host, budget_s=12.0):
    """Oracle (C port of the reference's per-series op sequence) on the host cores, on a bounded sample of the same
    workload: OpenMP over series for the filter + smoother; the sampler / SVD legs loop over series on one core."""
    import oracle
    om = oracle.Model(mat.d, mat.p, mat.T, mat.F, mat.G, mat.g_index, mat.dt, mat.f_stride)
    single = None
    if config in ("c2", "c4"):
        cores = os.cpu_count() or 1
        # one thread first (a few series), then all of them on a bounded sample: the line carries both rates
        oracle.set_threads(1)
        n1 = 2 if config == "c2" else 1
        t0 = time.perf_counter()
        oracle.filter_smooth_batch(n1, om, p.v, p.w, p.m0, p.c0, y_host[:n1], want_out=False)
        t1 = time.perf_counter() - t0
        n1 = int(max(n1, min(y_host.shape[0], n1 * 2.0 / t1)))          # about two seconds of one thread
        t0 = time.perf_counter()
        oracle.filter_smooth_batch(n1, om, p.v, p.w, p.m0, p.c0, y_host[:n1], want_out=False)
        single = n1 * mat.T / (time.perf_counter() - t0)
        # how many threads pay: the box shows every logical CPU of the host, the run owns a share of them (16 per GPU) -- try a few
        # counts on a short sample and keep the best (reported as `cores`)
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = cores
        best = (0.0, 1)
        for th in sorted({c for c in (8, 16, 32, 64, 128, avail) if c <= avail}):
            oracle.set_threads(th)
            nn = min(y_host.shape[0], 2 * th)
            t0 = time.perf_counter()
            oracle.filter_smooth_batch(nn, om, p.v, p.w, p.m0, p.c0, y_host[:nn], want_out=False)
            r = nn * mat.T / (time.perf_counter() - t0)
            if r > best[0]:
                best = (r, th)
        rate, cores = best
        cores = oracle.set_threads(cores)
        n = min(y_host.shape[0], 2 * cores)
        n2 = int(min(y_host.shape[0], max(n, rate * budget_s / mat.T)))
        n2 = max(min(cores, y_host.shape[0]), (n2 // cores) * cores)
        t0 = time.perf_counter()
        oracle.filter_smooth_batch(n2, om, p.v, p.w, p.m0, p.c0, y_host[:n2], want_out=False)
        dt = time.perf_counter() - t0
        what = "oracle/dlm_oracle.c filter + RTS smoother: Joseph-form update + LU solves, OpenMP over series (static schedule, per-thread scratch)"
    else:
        cores, n2, t0 = 1, 0, time.perf_counter()
        while n2 < y_host.shape[0] and time.perf_counter() - t0 < budget_s:
            if config in ("c3", "c4g"):
                f = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y_host[n2])
                z = oracle.normals(1, n2, mat.T + 1, mat.d)
                th = oracle.backward_sample(om, p.w, f, z, factor="chol")["theta"]
                oracle.gibbs_stats(om, y_host[n2], th, want_outer=(config == "c4g"))
            else:
                oracle.svd_filter(om, p.v, p.w, p.m0, p.c0, y_host[n2])
            n2 += 1
        dt = time.perf_counter() - t0
        what = ("oracle/dlm_oracle.c filter + backward sampler (Smoothing.step) + Gibbs sums" if config in ("c3", "c4g")
                else "oracle/dlm_oracle.c SVD filter (two one-sided Jacobi SVDs per step)") + ", one core, series after series"
    try:
        avail_all = len(os.sched_getaffinity(0))
    except AttributeError:
        avail_all = os.cpu_count() or 1
    out = {"value": n2 * mat.T / dt, "unit": "series*timesteps/s", "cores": cores, "kind": "port",
           "host_logical_cpus": os.cpu_count() or 1, "cpus_available_to_the_run": avail_all,   # `cores` = the threads actually used (the best of a short sweep up to the available count)
           "sample": f"{n2} of the {y_host.shape[0]} series x T={mat.T} (same inputs), {dt:.1f} s, {what}",
           "scala_reference": "unavailable (no JVM / Breeze jars on the box)"}
    if single is not None:
        out["single_thread_value"] = single
        out["threads"] = cores
        out["scaling_over_one_thread"] = out["value"] / single
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c4g", "c5"], default="c2")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = the series of the job are sharded over the ranks (the contract case), weak = every rank runs all of them")
    ap.add_argument("--series", type=int, default=None, help="series of the whole job (strong) / per GPU (weak); default 10000 (c4: 2000)")
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--missing", type=float, default=0.0, help="fraction of observations set missing (SURVEY 8d: a second run at 0.05)")
    ap.add_argument("--records", choices=["dense", "packed"], default="dense", help="c2: packed = DLM_OPT_PACKED_SYM records (2504 B / series-step)")
    ap.add_argument("--semantics", choices=["textbook", "literal-q1"], default="textbook",
                    help="c2/c4 smoother covariance: textbook J X J^T (default) or the reference's literal J X J (Smoothing.scala:44)")
    ap.add_argument("--sampler", choices=["reference", "simsmooth"], default="reference", help="c3: Smoothing.step backward sampler or the simulation smoother")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="default run: skip the extra configurations timed beside the headline")
    ap.add_argument("--flags", type=int, default=0, help="extra DLM_OPT_* bits (e.g. 8 = force generic kernels)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from bayesian_dlms_amd import _lib
    from bayesian_dlms_amd.dlm import materialise
    from bayesian_dlms_amd.engine import Engine
    from bayesian_dlms_amd.gibbs import GibbsSampling, GibbsWishart, InverseGamma, InverseWishart, shard_bounds

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # DLM_BENCH_BACKEND=gloo with DLM_BENCH_DEVICE=0 rehearses the multi-rank control flow with every rank on one GPU
    # (tests/test_full_size_gpu.py; timings of such a run mean nothing).  The driver's runs use nccl (= RCCL), one GPU per rank.
    backend = os.environ.get("DLM_BENCH_BACKEND", "nccl")
    if "DLM_BENCH_DEVICE" in os.environ:
        local = int(os.environ["DLM_BENCH_DEVICE"])
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    eng = Engine(local)
    ctx = dict(torch=torch, dist=dist, _lib=_lib, eng=eng, world=world, rank=rank, local=local, backend=backend, dev=dev)
    line = run_one(args, ctx)
    plain = (args.config == "c2" and world == 1 and args.flags == 0 and args.missing == 0.0 and args.records == "dense" and
             args.semantics == "textbook" and args.series is None and args.T == 1000)
    if rank == 0 and plain and not args.no_secondary:
        add_secondary(line, args, ctx)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def variant_of(args, **kw):
    import copy
    a = copy.copy(args)
    a.no_cpu_baseline = True
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def add_secondary(line, args, ctx):
    """The other configurations beside the headline: a few steps each, after the headline's timed
    region, every one through the same run_one (same contract, its own roofline by SURVEY 8d's units)."""
    torch, _lib = ctx["torch"], ctx["_lib"]

    def run(**kw):
        torch.cuda.empty_cache()
        r = run_one(variant_of(args, **kw), ctx)
        torch.cuda.empty_cache()
        return r

    def brief(r):
        keep = {k: r[k] for k in ("metric", "value", "unit", "steps", "ms_per_step", "step_ms")}
        keep["workload"] = r["config"]["workload"]
        keep["variant"] = r["config"]["variant"]
        keep["semantics"] = r["config"]["semantics"]
        keep["steady_fraction"] = r["config"].get("steady_fraction")
        keep["roofline"] = {k: r["roofline"].get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_GBps", "traffic_note", "peak_measured", "frac_of_measured", "avg_launch_ms", "forward_ms", "backward_ms", "note") if k in r["roofline"]}
        keep["status_nonzero_series"] = r["status_nonzero_series"]
        return keep

    # (10 steps after 3 warm-ups each, SURVEY 8d's protocol: with 5 steps after 1 warm-up the first timed steps still paid for the
    #  workspaces of the configuration before and the two values read 1-3 % low)
    full = run(flags=_lib.OPT_NO_STEADY, steps=10, warmup=3)
    miss = run(missing=0.05, steps=10, warmup=3)
    line["value_full_recursion"] = full["value"]        # every step recomputes the covariance recursion (DLM_OPT_NO_STEADY): the like-for-like figure, the reference recomputes every step
    line["value_missing_0.05"] = miss["value"]          # 5 % of the observations missing (SURVEY 8d's second run)
    line["config"]["value_full_recursion"] = full["value"]      # (also under `config`: the driver's record keeps that object whole)
    line["config"]["value_missing_0.05"] = miss["value"]
    line["config"]["ms_per_step_full_recursion"] = full["ms_per_step"]
    line["config"]["ms_per_step_missing_0.05"] = miss["ms_per_step"]
    # the headline shares J_t, S_t across the batch where V, W, C0 are shared; with DLM_OPT_SMOOTHER_PER_SERIES every series
    # runs its own covariance recursions in both passes (the steady-state shortcut still on): the per-series kernels' figure
    own = run(flags=_lib.OPT_SMOOTHER_PER_SERIES, steps=10, warmup=3)
    line["config"]["value_per_series_factors"] = own["value"]
    line["config"]["ms_per_step_per_series_factors"] = own["ms_per_step"]
    sec = {"c2_full_recursion": brief(full), "c2_missing_0.05": brief(miss), "c2_per_series_factors": brief(own)}
    sec["c2_literal_q1"] = brief(run(semantics="literal-q1", steps=10, warmup=3))
    sec["c2_literal_q1_per_series_factors"] = brief(run(semantics="literal-q1", flags=_lib.OPT_SMOOTHER_PER_SERIES, steps=3, warmup=1))
    sec["c2_shared_covariance_opt_in"] = brief(run(flags=_lib.OPT_SHARED_COV, steps=5, warmup=1))
    sec["c3_reference_sampler"] = brief(run(config="c3", sampler="reference", steps=3, warmup=2))
    sec["c3_reference_sampler_own_factors"] = brief(run(config="c3", sampler="reference", flags=_lib.OPT_SAMPLER_PER_SERIES, steps=2, warmup=2))   # every series its own J_t, H_t, chol(H_t)
    sec["c3_reference_sampler_1250_series"] = brief(run(config="c3", sampler="reference", series=1250, steps=3, warmup=2))   # one GPU's share of configs[2] on 8 GPUs
    sec["c3_simulation_smoother"] = brief(run(config="c3", sampler="simsmooth", steps=3, warmup=2))
    sec["c4"] = brief(run(config="c4", steps=3, warmup=1))
    sec["c4_full_recursion"] = brief(run(config="c4", flags=_lib.OPT_NO_STEADY, steps=2, warmup=1))
    sec["c4g"] = brief(run(config="c4g", steps=3, warmup=2))
    sec["c4g_own_factors"] = brief(run(config="c4g", flags=_lib.OPT_SAMPLER_PER_SERIES, steps=2, warmup=2))
    sec["c5"] = brief(run(config="c5", steps=2, warmup=1))
    line["secondary"] = sec


def run_one(args, ctx):
    """One configuration through the contract: W warm-up steps, K timed steps between barriers, one dict on rank 0."""
    torch, dist, _lib, eng = ctx["torch"], ctx["dist"], ctx["_lib"], ctx["eng"]
    world, rank, local, backend, dev = ctx["world"], ctx["rank"], ctx["local"], ctx["backend"], ctx["dev"]
    from bayesian_dlms_amd.dlm import materialise
    from bayesian_dlms_amd.gibbs import GibbsSampling, GibbsWishart, InverseGamma, InverseWishart, shard_bounds
    cfg = args.config
    steps = args.steps if args.steps is not None else {"c2": 20, "c3": 5, "c4": 5, "c4g": 3, "c5": 3}[cfg]
    warmup = args.warmup if args.warmup is not None else {"c2": 3, "c3": 2, "c4": 1, "c4g": 2, "c5": 1}[cfg]   # (Gibbs: every workspace of the loop exists after two iterations)

    mod, p = multivariate_c4() if cfg in ("c4", "c4g") else seasonal_c2()
    total = args.series if args.series is not None else (2000 if cfg in ("c4", "c4g") else 10000)
    T = args.T
    if world > 1 and args.scaling == "strong":
        lo, hi = shard_bounds(total, world, rank)
        job_series = total
    else:
        lo, hi = rank * total, (rank + 1) * total        # weak: every rank its own `total` series (distinct global indices)
        job_series = world * total
    N = hi - lo
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    d, q = mat.d, mat.p
    rec = d + d * d

    # synthetic inputs, simulated from the model itself ON the device (dlm_simulate_batch: Philox keyed by the GLOBAL
    # series index, so the data of series i does not depend on the sharding)
    y = eng.simulate(mat, p, N, seed=0xD1A5EED0, series_offset=lo, device=True, want_x=False)["y"]
    if args.missing > 0.0:
        g = torch.Generator(device=dev); g.manual_seed(0xD1A5EED1 + lo)
        y[torch.rand(y.shape, device=dev, generator=g) < args.missing] = float("nan")

    flags = args.flags
    if args.semantics == "literal-q1":
        flags |= _lib.OPT_SMOOTHER_COMPAT_Q1
    packed = cfg == "c2" and args.records == "packed"
    if packed:
        flags |= _lib.OPT_PACKED_SYM
    recw = eng.lib.dlm_packed_record_doubles(d) if packed else rec

    fwd_ms, bwd_ms = [], []
    status = torch.zeros((N,), dtype=torch.int32, device=dev)
    comm_world = None
    if cfg in ("c2", "c4"):
        out = {"filt": torch.empty((N, T + 1, recw), dtype=torch.float64, device=dev),
               "smooth": torch.empty((N, T + 1, recw), dtype=torch.float64, device=dev), "status": status}

        def step():
            eng.filter_smooth(mat, p, y, flags=flags, out=out)
    elif cfg == "c5":
        def step():
            status.copy_(eng.svd_filter(mat, p, y, flags=flags)["status"])
    else:
        # pooled-parameter Gibbs over all ranks: the communicator id goes through the torch.distributed store
        if backend == "nccl" and ctx.get("allreduce") is not None:
            allreduce = ctx["allreduce"]          # (the communicator of an earlier configuration of this process)
        elif backend == "nccl":
            uid = [eng.comm_unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(uid, src=0)
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)            # RCCL prints a version banner on stdout when a communicator is created: keep stdout to the ONE JSON line
            try:
                eng.comm_init_rank(world, rank, uid[0])
            finally:
                os.dup2(saved, 1); os.close(saved)
            allreduce = eng.allreduce_stats
            ctx["allreduce"] = allreduce
        else:                        # rehearsal on one GPU: the same sum through the torch.distributed backend given
            def allreduce(t):
                c = t.cpu()
                if world > 1:
                    dist.all_reduce(c)
                return c
        comm_world = world
        sim = args.sampler == "simsmooth"
        extra = flags
        ffbs = (lambda *a_, flags=0, **k_: eng.ffbs(*a_, flags=flags | extra, **k_)) if extra else None   # --flags: A/B runs of the FFBS call
        if cfg == "c3":
            chain = GibbsSampling.sample(mod, InverseGamma(5.0, 4.0), InverseGamma(17.0, 4.0), p, mat.times, y, eng,
                                         n_iter=steps + warmup, seed=7, pooled=True, series_offset=lo,
                                         allreduce=allreduce, simulation_smoother=sim, ffbs=ffbs)   # priors: SeasonalModel.scala:127
        else:
            chain = GibbsWishart.sample(mod, InverseGamma(5.0, 4.0), InverseWishart(d + 2.0, np.eye(d)), p, mat.times, y, eng,
                                        n_iter=steps + warmup, seed=7, pooled=True, series_offset=lo,
                                        allreduce=allreduce, simulation_smoother=sim, ffbs=ffbs)
        last = {}

        def step():
            last["state"] = next(chain)
            if last["state"].status is not None:          # the FFBS call's per-series flags, OR-ed over the iterations
                status.bitwise_or_(last["state"].status.to(status.dtype))

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    step_ms, tprev = [], t0
    for _ in range(steps):
        step()                       # synchronous on return (engine stream is drained)
        f, b = eng.last_timing()     # HIP events recorded by the engine around its forward / backward kernels
        fwd_ms.append(f); bwd_ms.append(b)
        tnow = time.perf_counter(); step_ms.append(round((tnow - tprev) * 1e3, 3)); tprev = tnow
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    variant = eng.last_variant
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    status_bad = int((status != 0).sum().item())
    value = job_series * T * steps / elapsed
    # how many steps took a short (steady-state) path: one more call with DLM_OPT_COUNT_STEPS, outside the timed region
    steady_fraction = None
    if cfg in ("c2", "c4"):
        eng.filter_smooth(mat, p, y, flags=flags | _lib.OPT_COUNT_STEPS, out=out)
    elif cfg == "c5":
        eng.svd_filter(mat, p, y, flags=flags | _lib.OPT_COUNT_STEPS)
    if cfg in ("c2", "c4", "c5"):
        cnt = eng.last_counters()
        steady_fraction = {"forward": cnt[0] / (float(N) * T), "backward": cnt[1] / (float(N) * T)}
        if cnt[2] or cnt[3]:
            steady_fraction["shared_covariance_series"] = cnt[2]
            steady_fraction["own_recursion_series"] = cnt[3]

    if rank == 0:
        moved_note = None
        route_note = None
        f_ms, b_ms = float(np.mean(fwd_ms)), float(np.mean(bwd_ms))
        nt = float(N) * T
        if cfg == "c2":
            # forward: read y, write [m|C]; backward: read [m|C], write [s|S] (8p + 24 (d + d^2) = 4376 B, packed 2504)
            fwd_u, bwd_u, unit, peak, bound = 8.0 * q + 8.0 * recw, 16.0 * recw, "GB/s", HBM_PEAK_GBS, "hbm"
            names = ("k_filter_", "k_smoother_")
            tables_served = bool(steady_fraction and steady_fraction.get("shared_covariance_series", 0) * 2 > N)
            if variant == "sparse16-rts-shared" and not tables_served:
                # the call decided on the device to make no tables (most series have a gap): the per-series kernels ran
                names = ("k_filter_", "k_smoother_")
                variant = "sparse16-rts" if args.semantics == "literal-q1" else "sparse16"
                route_note = "sparse16-rts-shared route, no tables made (decided on the device): per-series kernels"
            elif variant == "sparse16-rts-shared":
                # shared factors: J_t, S_t come from the call's tables; the backward kernel reads the head of a filtered record (16
                # doubles: the mean) and writes the smoothed record -- the bytes this algorithm has to move, not SURVEY 8d's read + write of whole records
                bwd_u = 128.0 + 8.0 * recw
                names = ("k_filter_", "k_mean_")
                moved_note = ("algorithmic bytes of the shared-factor backward pass: 128 B of each filtered record in, the smoothed record out (1584 B per series-step); "
                              "SURVEY 8d's 2912 B (whole filtered record in) would price the same launch at `achieved_by_contract_bytes`")
        elif cfg == "c3" and variant.endswith("-shared") and args.sampler == "reference":
            # shared factors and no filter records: the call does not move SURVEY 8d's 2920 B per series-step any more --
            # forward: y in, the compact means out (512 B per four series); draw: those means and the normals in.  The bytes MOVED:
            fwd_u, bwd_u, unit, peak, bound = 8.0 * q + 128.0, 256.0, "GB/s", HBM_PEAK_GBS, "hbm"
            names = ("k_mean_filter_", "k_mean_sampler_")
            moved_note = ("bytes really moved (the records-free call: 8 + 128 B forward, 256 B in the draw kernel per series-step; the contract's algorithmic figure, 2920 B, "
                          "is the traffic this path removed): the kernels are bound by their dependent chains and instruction issue, not by HBM")
        elif cfg == "c3":
            # FFBS with on-device statistics: write + re-read the filtered records, theta never written (8p + 16 (d + d^2) = 2920 B)
            fwd_u, bwd_u, unit, peak, bound = 8.0 * q + 8.0 * rec, 8.0 * rec, "GB/s", HBM_PEAK_GBS, "hbm"
            names = ("k_filter_", "k_mean_sampler_" if variant.endswith("-shared") else "k_sampler_")
        elif cfg == "c4" and steady_fraction is not None and steady_fraction["forward"] > 0.5:
            # the covariance recursion of this model settles within 30 steps: from then on both passes stream records (steady-state
            # steps) and the bound is HBM -- the contract's algorithmic bytes 8p + 24 (d + d^2) = 39 520 B per series-step
            fwd_u, bwd_u, unit, peak, bound = 8.0 * q + 8.0 * rec, 16.0 * rec, "GB/s", HBM_PEAK_GBS, "hbm"
            names = ("k_filter_", "k_smoother_")
        elif cfg == "c4":
            # algorithmic flops: filter 8 d^3 + 6 d^2 p + 2 p^3 / 3, RTS 8.67 d^3
            fwd_u, bwd_u, unit, peak, bound = 8.0 * d ** 3 + 6.0 * d * d * q + 2.0 * q ** 3 / 3.0, 8.67 * d ** 3, "TFLOP/s", MFMA_F64_PEAK_TFLOPS, "mfma"
            names = ("k_filter_", "k_smoother_")
        elif cfg == "c4g" and variant.endswith("-shared"):
            # pooled parameters: J_t, H_t and the factors are computed once per call, the covariance recursion of the
            # forward pass settles within 30 steps and the records are nobody's output (filt_ws = NULL): what moves is the observations,
            # the means and the normals
            fwd_u, bwd_u, unit, peak, bound = 8.0 * q + 8.0 * d, 16.0 * d + 8.0 * q, "GB/s", HBM_PEAK_GBS, "hbm"
            names = ("k_filter_", "k_mean_sampler_")
            moved_note = ("bytes really moved (the records-free call: observations in and means out in the forward pass, means, normals and observations in the draw "
                          "kernel; the records of the contract's algorithmic figure are no longer written): the kernels are bound by their dependent chains, not by HBM")
        elif cfg == "c4g":
            # FFBS adds about 8 d^3 (J, H) plus the Cholesky factor d^3 / 3 to the filter's flops
            fwd_u, bwd_u, unit, peak, bound = 8.0 * d ** 3 + 6.0 * d * d * q + 2.0 * q ** 3 / 3.0, 8.0 * d ** 3 + d ** 3 / 3.0, "TFLOP/s", MFMA_F64_PEAK_TFLOPS, "mfma"
            names = ("k_filter_", "k_sampler_")
        else:
            fwd_u, bwd_u, unit, peak, bound = 8.0 * q + 8.0 * (2 * d + d * d), 0.0, "GB/s", HBM_PEAK_GBS, "hbm"
            names = ("k_svd_filter", "")
        dom_is_bwd = b_ms >= f_ms and bwd_u > 0
        if cfg == "c2" and variant == "sparse16-rts-shared":
            # forward_ms of this route is three kernels and a wait (covariance-only filter 0.22 ms + k_filter_sp16 2.9 + what is left of the table run);
            # backward_ms is ONE launch, k_mean_rts16, the longest kernel of the call: that is the kernel priced here
            dom_is_bwd = True
        dom_u, dom_ms = (bwd_u, b_ms) if dom_is_bwd else (fwd_u, f_ms)
        scale = 1e9 if unit == "GB/s" else 1e12
        achieved = dom_u * nt / (dom_ms * 1e-3) / scale
        short = {"sparse16": "sp16", "mfma16": "mfma16", "wave-mfma": "w48", "tiled-mfma": "tiled", "sparse16-sampler": "sp16", "sparse16-rts": "rts16", "wave-sampler": "w48", "wave-simsmooth": "w48",
                 "sparse16-simsmooth": "sp16", "svd-jacobi": "", "sparse16-sampler-shared": "sp16", "wave-sampler-shared": "w48", "sparse16-rts-shared": "rts16"}.get(variant, variant)
        kname = (names[1] if dom_is_bwd else names[0]) + short
        workloads = {
            "c2": f"C2: seasonal DLM polynomial(1)|+|seasonal(24,6), d=13, p=1, {job_series} series x T={T}, fused filter+smooth (dlm_filter_smooth_batch)",
            "c3": f"C3: the C2 model inside pooled d-Inverse-Gamma Gibbs, {job_series} series x T={T}: FFBS + on-device statistics + dlm_stats_pool + RCCL all-reduce + conjugate draw per iteration",
            "c4": f"C4: 20 x polynomial(2) under |*|, d=40, p=20, {job_series} series x T={T}, fused filter+smooth",
            "c4g": f"C4 Gibbs: the C4 model inside pooled Inverse-Wishart Gibbs, {job_series} series x T={T}: FFBS + outer-product statistics + dlm_stats_pool + RCCL all-reduce + W, V draws per iteration",
            "c5": f"C5: SVD (square-root) filter on the C2 model, d=13, {job_series} series x T={T} (dlm_svd_filter_batch)",
        }
        line = {
            "metric": {"c2": "Kalman filter+smooth series*timesteps/sec", "c3": "Gibbs (FFBS + conjugate step) series*timesteps/sec",
                       "c4": "Kalman filter+smooth series*timesteps/sec (d=40)", "c4g": "Gibbs (FFBS + Inverse-Wishart step) series*timesteps/sec (d=40)",
                       "c5": "SVD filter series*timesteps/sec"}[cfg],
            "value": value, "unit": "series*timesteps/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "step_ms": step_ms, "higher_is_better": True,
            "scaling": "weak" if (world > 1 and args.scaling == "weak") else "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workloads[cfg], "series_total": job_series, "series_per_gpu": N, "T": T, "d": d, "p": q,
                       "variant": variant, "records": args.records if cfg == "c2" else "dense",
                       "semantics": ({"textbook": "textbook (S = C - J (R+ - S+) J^T)", "literal-q1": "literal-q1 (Smoothing.scala:44: J X J)"}[args.semantics]
                                     if cfg in ("c2", "c4") else ("Smoothing.step backward sampler" if cfg in ("c3", "c4g") and args.sampler == "reference"
                                                                  else "simulation smoother" if cfg in ("c3", "c4g") else "sqrt(W) in the time update (Q2 off)")),
                       "missing_fraction": args.missing,
                       "steady_state_steps": "off (DLM_OPT_NO_STEADY)" if flags & _lib.OPT_NO_STEADY else "on where the covariance recursion has settled within 1e-12 (settle_test, DESIGN.md 4.5)",
                       "steady_fraction": steady_fraction,
                       "parallelism": f"series-sharded x{world}" + (", one RCCL all-reduce of %d doubles per iteration (comm world %d)" % (2 * q + (d if cfg == "c3" else d * d) + 1, comm_world)
                                                                    if cfg in ("c3", "c4g") else ", no collective")},
            "roofline": {"bound": bound, "kernel": kname, "achieved": achieved, "peak": peak, "unit": unit,
                         "frac": achieved / peak, "traffic": None,
                         "algorithmic_units_per_launch": dom_u * nt, "avg_launch_ms": dom_ms,
                         "forward_ms": f_ms, "backward_ms": b_ms},
            "status_nonzero_series": status_bad,
        }
        if route_note:
            line["config"]["route_note"] = route_note
        if cfg == "c2" and variant == "sparse16-rts-shared" and dom_is_bwd:
            line["roofline"]["achieved_by_contract_bytes"] = 16.0 * recw * nt / (dom_ms * 1e-3) / scale
        tr_bytes, tr_note = traffic_from_profiles(kname, args)
        line["roofline"]["traffic"] = tr_bytes
        if tr_note:
            line["roofline"]["traffic_note"] = tr_note
        line["roofline"]["achieved_basis"] = "ALGORITHMIC units per launch (SURVEY 8d) / launch time"
        if moved_note:
            line["roofline"]["achieved_basis"] = moved_note
        if cfg == "c4" and bound == "hbm":
            line["roofline"]["note_c4"] = ("steady_fraction > 0.5: most steps skip the covariance recursion, so SURVEY 8d's flops (1.26e6 per series-step) would price "
                                           "the run above the fp64 MFMA peak; priced against HBM by algorithmic bytes instead -- `c4_full_recursion` is the MFMA-bound figure")
        skips = steady_fraction is not None and max(steady_fraction["forward"], steady_fraction["backward"]) > 0.0
        if bound == "hbm":
            pm = ctx.get("copy_peak") or measure_copy_peak(torch, dev)
            ctx["copy_peak"] = pm
            line["roofline"]["peak_measured"] = pm
            tr = line["roofline"]["traffic"]
            if tr is None and skips:
                # steady-state steps fetch a settled record as its mean alone: fewer bytes cross HBM than the algorithmic figure,
                # and no PMC figure exists for this configuration -- no fraction of the copy rate is claimed
                line["roofline"]["note"] = "steady-state steps move fewer bytes than the algorithmic figure; no PMC traffic for this configuration: frac_of_measured omitted"
            else:
                line["roofline"]["frac_of_measured"] = achieved / pm
            if tr:   # what actually crossed the HBM interface per launch (PMC, profiles/), as a rate: below `achieved` where the
                     # backward pass fetches only the mean of a record whose covariance it already holds -- and
                     # the fraction of the copy rate is that of the bytes really moved
                line["roofline"]["traffic_GBps"] = tr / (dom_ms * 1e-3) / 1e9
                line["roofline"]["frac_of_measured"] = line["roofline"]["traffic_GBps"] / pm
            if cfg == "c2":   # (the headline's kernels mostly write)
                pw = ctx.get("write_peak") or measure_write_peak(torch, dev)
                ctx["write_peak"] = pw
                line["roofline"]["peak_measured_write"] = pw
                if tr:
                    line["roofline"]["frac_of_measured_write"] = line["roofline"]["traffic_GBps"] / pw
            if line["roofline"].get("frac_of_measured", 0) and line["roofline"]["frac_of_measured"] > 1.0:
                line["roofline"]["peak_measured_note"] = ("peak_measured is a device copy (one read stream + one write stream); a kernel that mostly writes is not bound by it: "
                                                          "write-only kernels reach 5.6-6.2 TB/s on this pool (tools/micro, profiles/r04_notes.md section 9)")
            line["roofline"]["path_GBps"] = (fwd_u + bwd_u) * nt / ((f_ms + b_ms) * 1e-3) / 1e9
        if cfg in ("c3", "c4g"):
            st = last["state"]
            line["config"]["pooled_V"] = float(np.diag(st.p.v)[0])
            line["config"]["pooled_W_first"] = [float(x) for x in np.diag(st.p.w)[:3]]
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(cfg, mat, p, y[: min(N, 4096)].cpu().numpy())
        return line
    return None


if __name__ == "__main__":
    main()
