"""CPU-only tests of the host-side logic: model builders (Dlm.scala), parameter packing,
materialisation, Gibbs conjugate updates and the multi-rank (gloo, world_size 2) path."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from bayesian_dlms_amd.dlm import (Data, Dlm, DlmParameters, angle, block_diagonal, materialise,
                                   rotation_matrix, seasonal_g)
from bayesian_dlms_amd.engine import pack_params
from bayesian_dlms_amd.gibbs import (GibbsSampling, GibbsWishart, InverseGamma, InverseWishart,
                                     shard_bounds, split_stats)
from bayesian_dlms_amd.api import pack_observations

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_polynomial_and_seasonal_builders():
    g = Dlm.polynomial(3).g(1.0)
    np.testing.assert_array_equal(g, [[1, 1, 0], [0, 1, 1], [0, 0, 1]])           # Dlm.scala:146-151
    np.testing.assert_array_equal(Dlm.polynomial(3).f(5.0).ravel(), [1, 0, 0])
    s = Dlm.seasonal(24, 3)
    np.testing.assert_array_equal(s.f(1.0).ravel(), [1, 0, 1, 0, 1, 0])          # Dlm.scala:238-240
    np.testing.assert_allclose(s.g(1.0)[2:4, 2:4], rotation_matrix(2 * angle(24, 1.0)))
    assert angle(24, 25.0) == pytest.approx(2 * np.pi / 24)                       # dt % period
    np.testing.assert_allclose(seasonal_g(24, 2, 24.0), np.eye(4), atol=1e-12)


def test_compose_and_outer():
    a, b = Dlm.polynomial(1), Dlm.polynomial(2)
    c = a + b                                                                      # |+|
    assert c.f(1.0).shape == (3, 1) and c.g(1.0).shape == (3, 3)
    o = a * b                                                                      # |*|
    assert o.f(1.0).shape == (3, 2)
    np.testing.assert_array_equal(o.f(1.0), block_diagonal(a.f(1.0), b.f(1.0)))
    p1 = DlmParameters([[2.0]], [[3.0]], [0.0], [[1.0]])
    p2 = DlmParameters([[1.0]], np.eye(2), [1.0, 2.0], np.eye(2) * 4)
    po = p1 * p2
    assert po.v.shape == (2, 2) and po.w.shape == (3, 3) and list(po.m0) == [0, 1, 2]


def test_parameters_roundtrip():
    """core/src/test/scala/Parameters.scala:32-54: toList . fromList == identity for diagonal params."""
    p = DlmParameters(np.diag([1.0, 2.0]), np.diag([3.0, 4.0, 5.0]), [6.0, 7.0, 8.0], np.diag([9.0, 10.0, 11.0]))
    l = p.to_list()
    assert l == [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11]
    q = DlmParameters.from_list(2, 3, l)
    for x, y in ((p.v, q.v), (p.w, q.w), (p.m0, q.m0), (p.c0, q.c0)):
        np.testing.assert_array_equal(x, y)


def test_materialise_irregular_times_and_time_varying_f():
    x = [np.array([0.5]), np.array([1.5]), np.array([2.5])]
    mat = materialise(Dlm.regression(x), [1.0, 2.0, 3.0])
    assert mat.f_stride == 2 and mat.F.size == 6 and list(mat.F) == [1, 0.5, 1, 1.5, 1, 2.5]
    mat = materialise(Dlm.seasonal(12, 1), [1.0, 2.0, 4.0, 5.0])
    assert mat.n_g == 2 and list(mat.g_index) == [0, 0, 1, 0] and list(mat.dt) == [1, 1, 2, 1]
    with pytest.raises(ValueError):
        materialise(Dlm.polynomial(1), [])


def test_pack_params_and_observations():
    p = DlmParameters([[1.0]], np.array([[1.0, 2.0], [3.0, 4.0]]), [0.0, 1.0], np.eye(2))
    V, vs, W, ws, m0, ms, C0, cs, vts, wts = pack_params(p, 5)
    assert (vts, wts) == (0, 0)
    assert (vs, ws, ms, cs) == (0, 0, 0, 0) and list(W) == [1, 3, 2, 4]           # column-major
    V, vs, W, ws, m0, ms, C0, cs, vts, wts = pack_params([p, p, p], 3)
    assert (vs, ws, ms, cs) == (1, 4, 2, 4) and W.size == 12
    with pytest.raises(ValueError):
        pack_params([p], 2)
    ys = [Data(1.0, [1.0]), Data(2.0, [None]), Data(4.0, [3.0])]
    t, y, batched = pack_observations(ys)
    assert not batched and y.shape == (1, 3, 1) and np.isnan(y[0, 1, 0]) and list(t) == [1, 2, 4]
    t, y, batched = pack_observations([ys, ys])
    assert batched and y.shape == (2, 3, 1)
    with pytest.raises(ValueError):
        pack_observations([ys, [Data(1.0, [1.0]), Data(2.0, [1.0]), Data(3.0, [1.0])]])
    with pytest.raises(ValueError):
        pack_observations([])


def test_shard_bounds_cover_everything():
    for N, G in ((10000, 8), (10, 4), (3, 8), (1250, 1)):
        spans = [shard_bounds(N, G, g) for g in range(G)]
        assert spans[0][0] == 0 and spans[-1][1] == N
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


# ------------------------------------------------------------------------------------------
# Gibbs host logic with the oracle standing in for the engine's FFBS (no GPU needed)
# ------------------------------------------------------------------------------------------
def oracle_ffbs(mat, params, y, *, seed=0, series_offset=0, flags=0, want_theta=True, want_stats=True, **kw):
    om = oracle.Model(mat.d, mat.p, mat.T, mat.F, mat.G, mat.g_index, mat.dt, mat.f_stride)
    plist = [params] * y.shape[0] if isinstance(params, DlmParameters) else list(params)
    outer = bool(flags & 16)
    thetas, stats = [], []
    for n in range(y.shape[0]):
        p = plist[n]
        f = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y[n])
        z = oracle.normals(seed, series_offset + n, mat.T + 1, mat.d)
        th = oracle.backward_sample(om, p.w, f, z, factor="chol")["theta"]
        st = oracle.gibbs_stats(om, y[n], th, want_outer=outer)
        body = st["outer"] if outer else st["ss"]
        stats.append(np.concatenate([st["ssy"], st["n"], body, [mat.T]]))
        thetas.append(th)
    return {"theta": np.stack(thetas), "stats": np.stack(stats)}


def _toy(N=6, T=40, seed=0):
    mod = Dlm.polynomial(2)
    times = np.arange(1, T + 1, dtype=np.float64)
    rng = np.random.default_rng(seed)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1)
    y[rng.random(y.shape) < 0.1] = np.nan
    p = DlmParameters([[2.0]], np.diag([0.5, 0.2]), [0.0, 0.0], np.eye(2) * 10)
    return mod, times, y, p


def test_gibbs_per_series_is_shard_invariant():
    """Per-series Gibbs: running a shard with its series_offset reproduces the full-batch chain."""
    mod, times, y, p = _toy()
    pv, pw = InverseGamma(5.0, 4.0), InverseGamma(17.0, 4.0)
    full = list(GibbsSampling.sample(mod, pv, pw, p, times, y, None, n_iter=3, seed=7, ffbs=oracle_ffbs))
    lo, hi = shard_bounds(6, 2, 1)
    part = list(GibbsSampling.sample(mod, pv, pw, p, times, y[lo:hi], None, n_iter=3, seed=7,
                                     series_offset=lo, ffbs=oracle_ffbs))
    for a, b in zip(full, part):
        for k in range(hi - lo):
            np.testing.assert_allclose(a.p[lo + k].v, b.p[k].v, rtol=1e-12)
            np.testing.assert_allclose(a.p[lo + k].w, b.p[k].w, rtol=1e-12)
    assert all(np.all(np.diag(s.p[0].w) > 0) for s in full)


def test_gibbs_posterior_parameters():
    """The conjugate update uses shape alpha + n/2, rate beta + ssy/2 (Gibbs.scala:41-48, :72-75)."""
    stats = np.array([3.0, 7.0, 1.5, 2.5, 40.0])       # ssy, n, ss0, ss1, T   (d = 2, p = 1)
    ssy, n, ss, T = split_stats(stats, 2, 1, False)
    assert ssy[0] == 3.0 and n[0] == 7.0 and list(ss) == [1.5, 2.5] and T == 40.0
    rng = np.random.default_rng(1)
    draws = np.array([1.0 / rng.gamma(5.0 + 3.5, 1.0 / (4.0 + 1.5)) for _ in range(20000)])
    assert draws.mean() == pytest.approx((4.0 + 1.5) / (5.0 + 3.5 - 1), rel=0.03)   # InverseGamma.mean
    iw = InverseWishart(12.0, np.eye(3) * 2.0)
    m = np.mean([iw.draw(rng) for _ in range(4000)], axis=0)
    np.testing.assert_allclose(m, np.eye(3) * 2.0 / (12.0 - 3 - 1), atol=0.05)      # InverseWishart.mean


def test_gibbs_wishart_runs_and_is_spd():
    mod, times, y, p = _toy(N=2, T=30)
    out = list(GibbsWishart.sample(mod, InverseGamma(5.0, 4.0), InverseWishart(5.0, np.eye(2)), p, times, y, None,
                                   n_iter=2, seed=3, ffbs=oracle_ffbs))
    w = out[-1].p[0].w
    np.testing.assert_allclose(w, w.T, atol=1e-12)
    assert np.linalg.eigvalsh(w).min() > 0


# ------------------------------------------------------------------------------------------
# world_size 2 over gloo: pooled Gibbs, statistics all-reduced, identical draws on both ranks
# ------------------------------------------------------------------------------------------
WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["DLM_ROOT"]); sys.path.insert(0, os.path.join(os.environ["DLM_ROOT"], "tests"))
from bayesian_dlms_amd.gibbs import GibbsSampling, InverseGamma, shard_bounds
from test_host_logic import oracle_ffbs, _toy
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
mod, times, y, p = _toy()
lo, hi = shard_bounds(y.shape[0], world, rank)
def allreduce(a):
    t = torch.from_numpy(np.ascontiguousarray(a)); dist.all_reduce(t); return t.numpy()
chain = list(GibbsSampling.sample(mod, InverseGamma(5.0, 4.0), InverseGamma(17.0, 4.0), p, times, y[lo:hi], None,
                                  n_iter=3, seed=11, pooled=True, series_offset=lo, allreduce=allreduce, ffbs=oracle_ffbs))
np.save(os.path.join(os.environ["DLM_OUT"], f"rank{rank}.npy"),
        np.array([np.concatenate([np.diag(s.p.v), np.diag(s.p.w)]) for s in chain]))
dist.barrier(); dist.destroy_process_group()
'''


def test_pooled_gibbs_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, DLM_ROOT=ROOT, DLM_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", MASTER_PORT="29533",
               WORLD_SIZE="2", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    for pr in procs:
        assert pr.wait(timeout=240) == 0
    r0, r1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    np.testing.assert_array_equal(r0, r1)                     # every rank makes the same draw
    # ... and it is the chain a single process gets on the whole batch
    mod, times, y, p = _toy()
    one = list(GibbsSampling.sample(mod, InverseGamma(5.0, 4.0), InverseGamma(17.0, 4.0), p, times, y, None,
                                    n_iter=3, seed=11, pooled=True, ffbs=oracle_ffbs))
    ref = np.array([np.concatenate([np.diag(s.p.v), np.diag(s.p.w)]) for s in one])
    np.testing.assert_allclose(r0, ref, rtol=1e-9)


def test_chain_and_data_csv_round_trip(tmp_path, golden_dir):
    """streaming.py (Streaming.scala:25-78; FirstOrderDlm.scala:31-50): chains and simulated data survive the disk."""
    from bayesian_dlms_amd import streaming
    from bayesian_dlms_amd.dlm import DlmParameters
    rng = np.random.default_rng(0)
    chain = [DlmParameters(np.diag(rng.uniform(1, 2, 2)), np.diag(rng.uniform(1, 2, 3)), rng.standard_normal(3),
                           np.diag(rng.uniform(1, 2, 3))) for _ in range(7)]
    f = str(tmp_path / "chain.csv")
    assert streaming.write_chain(lambda p: p.to_list(), f, iter(chain), header=["v1", "v2", "w1", "w2", "w3", "m1", "m2", "m3", "c1", "c2", "c3"]) == 7
    rows = list(streaming.read_mcmc_chain(f))
    assert len(rows) == 7
    for p, row in zip(chain, rows):
        assert row == [float(x) for x in p.to_list()]          # repr round-trips doubles exactly
        q = DlmParameters.from_list(2, 3, row)
        assert np.array_equal(q.v, p.v) and np.array_equal(q.w, p.w) and np.array_equal(q.c0, p.c0)
    means = streaming.col_means(rows)
    assert np.allclose(means, np.mean(rows, axis=0))
    assert streaming.quantile([5.0, 1.0, 3.0, 2.0, 4.0], 0.5) == 3.0 and streaming.quantile([5.0, 1.0, 3.0, 2.0], 0.0) == 1.0
    # parseDiagonalParameters keeps the full c0, column-major
    c0 = np.array([[2.0, 0.5], [0.25, 1.0]])
    p = streaming.parse_diagonal_parameters(1, 2, [3.0, 1.0, 2.0, 0.1, 0.2] + list(c0.T.reshape(-1)))
    assert np.array_equal(p.c0, c0) and np.array_equal(p.w, np.diag([1.0, 2.0])) and p.v[0, 0] == 3.0
    # the reference's fold starts from the empty parameter set with count 1: sum / (n + 1)
    m = streaming.mean_parameters(iter(chain), 2, 3)
    assert np.allclose(m.w, sum(c.w for c in chain) / 8)
    # the examples' data file reads back as Data, and a written file has the same layout
    data = streaming.read_data(os.path.join(golden_dir, "first_order_dlm.csv"))
    assert len(data) == 1000 and data[0].time == 1.0 and data[0].observation[0] == -3.941883011167026
    g = str(tmp_path / "sim.csv")
    ys = np.array([[1.5], [np.nan], [2.5]]); xs = np.array([[0.1], [0.2], [0.3]])
    streaming.write_simulated(g, [1.0, 2.0, 3.0], xs, ys)
    assert open(g).readline().strip() == "time,observation,state"
    back = streaming.read_data(g)
    assert back[0].observation[0] == 1.5 and np.isnan(back[1].observation[0]) and back[2].time == 3.0


def test_headline_kernels_keep_their_occupancy(tmp_path):
    """The C2 kernels are tuned to a register budget: the backward kernel of the K = 2 regular instantiation at <= 96 VGPRs (five
    waves per SIMD) and the forward kernel at <= 96 (five), with no spills -- except, since round 3, ONE value of the backward kernel
    (an LDS address of the transposed read) that lives in scratch and is reloaded only in the every-8th-step symmetrisation branch:
    with the error-bounded steady-state test the allocator needs 98 registers otherwise, and five waves with that reload measured
    1.5 % faster than four without (profiles/r03_notes.md).  Since the gap-marked series take the body without the shortcut's machinery
    inside the same kernel (two inlined bodies) three values are spilled; same-box A/B on the headline: backward 6.05 / 5.96 / 5.95 ms
    with, 5.98 / 5.96 / 5.96 ms without (profiles/r03_notes.md section 3).  A change that costs a wave per SIMD costs ~3 % of the
    headline and shows up nowhere else -- so the build is checked here (device-only assembly of dlm_sparse16.hip, ~10 s)."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from bayesian_dlms_amd import build as b
    out = tmp_path / "sp16.s"
    subprocess.check_call([b.HIPCC] + b.FLAGS + ["-S", "--cuda-device-only", "-o", str(out), os.path.join(root, "bayesian_dlms_amd", "csrc", "dlm_sparse16.hip")],
                          stderr=subprocess.DEVNULL)
    meta = {}
    name = None
    for line in open(out):
        m = re.match(r"\s*\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
        m = re.match(r"\s*\.(vgpr_count|vgpr_spill_count):\s+(\d+)", line)
        if m and name:
            meta.setdefault(name, {})[m.group(1)] = int(m.group(2))
    bwd = [v for k, v in meta.items() if "k_smoother_sp16ILi2ELb0ELb0ELb0E" in k]       # <K = 2, IRR = false, PIPE = false, PLAIN = false>
    fwd = [v for k, v in meta.items() if "k_filter_sp16ILi2ELb0ELb0ELb0E" in k]
    assert len(bwd) == 1 and len(fwd) == 1, sorted(meta)
    assert bwd[0]["vgpr_count"] <= 96 and bwd[0]["vgpr_spill_count"] <= 3, bwd
    assert fwd[0]["vgpr_count"] <= 96 and fwd[0]["vgpr_spill_count"] == 0, fwd


def test_option_flags_agree_between_header_python_and_scala():
    """dlm_options.flags: every DLM_OPT_* bit of include/dlm_engine.h has the same value in bayesian_dlms_amd/_lib.py, and the
    bits the Scala shim names (integration/scala/Batched.scala) are the header's."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "dlm_engine.h")).read()
    bits = {m.group(1): int(m.group(2)) for m in re.finditer(r"DLM_OPT_([A-Z0-9_]+)\s*=\s*1u\s*<<\s*(\d+)", hdr)}
    assert len(bits) >= 20 and len(set(bits.values())) == len(bits)          # no bit used twice
    from bayesian_dlms_amd import _lib
    for name, sh in bits.items():
        assert getattr(_lib, "OPT_" + name) == 1 << sh, name
    scala = open(os.path.join(root, "integration", "scala", "Batched.scala")).read()
    camel = lambda n: "".join(w.capitalize() for w in n.lower().split("_"))
    named = {m.group(1): int(m.group(2)) for m in re.finditer(r"val (\w+) = 1 << (\d+)", scala)}
    by_camel = {camel(n).lower(): sh for n, sh in bits.items()}          # (SvdRawWQ2, FfbsSimSmooth: compared without case)
    assert len(named) >= 10
    for n, sh in named.items():
        assert by_camel.get(n.lower()) == sh, n


def test_per_wave_kernels_keep_their_registers(tmp_path):
    """The 16 <= d <= 48 kernels live at the 512-register limit of a wave and the allocator's choices for a whole kernel turn on a
    few instructions: in round 4 ONE store added inside k_filter_w48's time loop took the C4 instantiation from 474 registers and no
    scratch to 395 spilled values (every-step C4 forward pass 24 -> 48 ms, 4 x the algorithmic HBM traffic by PMC) without failing
    a single parity test.  The code object of the built dlm_wave48.o is checked here (seconds: no compilation)."""
    import re
    import subprocess
    from bayesian_dlms_amd import build as b
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    obj = os.path.join(root, "bayesian_dlms_amd", "build", "dlm_wave48.o")
    if not os.path.exists(obj):
        b.build()
    llvm = "/opt/rocm/lib/llvm/bin"
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "w48.co")
    subprocess.check_call([f"{llvm}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj])
    subprocess.check_call([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
    notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    meta = {}
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        get = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
        meta[re.search(r"\.name:\s+(\S+)", blk).group(1)] = (get("private_segment_fixed_size"), get("vgpr_spill_count"))
    def of(frag):
        hit = [v for k, v in meta.items() if frag in k]
        assert len(hit) == 1, (frag, [k for k in meta if frag.split("IL")[0] in k][:4])
        return hit[0]
    # <DT, PT, K, KF>: (3, 2, 2, 1) is C4 (d = 40, p = 20, structured F), (2, 1, 2, 1) d = 20 / p = 10
    assert of("k_filter_w48ILi3ELi2ELi2ELi1E") == (0, 0)
    assert of("k_filter_w48ILi2ELi1ELi2ELi1E") == (0, 0)
    assert of("k_smoother_w48ILi3ELi2ELi2ELi1E")[0] <= 96           # 16 values, in the full step
    assert of("k_steady_filter_w48ILi3ELi2ELi2ELi1E") == (0, 0)
    assert of("k_mean_sampler_w48ILi3ELi2ELb1E") == (0, 0) and of("k_mean_sampler_w48ILi3ELi2ELb0E") == (0, 0)


def test_the_oracle_is_used_from_tests_smoke_and_the_cpu_baseline_only():
    """oracle/ is test infrastructure: nothing under the package, tools/ or integration/ imports it (the product path must fail loudly without the HIP
    library rather than fall back on it); bench.py uses it inside cpu_baseline only, __graft_entry__.py inside smoke() (and build(), which compiles it) only."""
    import ast, glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    def oracle_imports(path):
        tree = ast.parse(open(path).read())
        hits = []
        for node in ast.walk(tree):
            if isinstance(node, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in node.names):
                hits.append(node.lineno)
            if isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle":
                hits.append(node.lineno)
        return tree, hits
    for pat in ("bayesian_dlms_amd/*.py", "tools/*.py", "integration/**/*.py"):
        for f in glob.glob(os.path.join(root, pat), recursive=True):
            assert oracle_imports(f)[1] == [], f
    for f, allowed in (("bench.py", ("cpu_baseline",)), ("__graft_entry__.py", ("smoke", "build"))):   # (build() compiles the checker: building it is not using it)
        tree, hits = oracle_imports(os.path.join(root, f))
        spans = [(n.lineno, n.end_lineno) for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name in allowed]
        assert hits and all(any(a <= h <= b for a, b in spans) for h in hits), (f, hits, spans)
