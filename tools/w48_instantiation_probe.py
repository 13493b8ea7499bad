"""Round-4 diagnostic for the round-3 `k_sampler_w48` discrepancy: which machine codes of the d = 40
backward sampler agree with which, and is each of them repeatable?  Inputs are those of
tests/test_shared_sampler_gpu.py::test_multivariate_draw_for_draw_the_per_series_kernel[20-130-6-16].

  python tools/w48_instantiation_probe.py [package_dir]

package_dir: a directory holding an alternative `bayesian_dlms_amd` package (e.g. the tree of commit eb23394, whose table is
made by a second template instantiation of the kernel); default: the repo's own.
Runs (all with the same seed): shared/outer, shared/plain, per-series/outer (A), per-series/plain (B), each twice."""
import os
import sys

pkg = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.abspath(pkg))
import numpy as np  # noqa: E402
from bayesian_dlms_amd import _lib  # noqa: E402
from bayesian_dlms_amd.dlm import Dlm, DlmParameters, materialise  # noqa: E402
from bayesian_dlms_amd.engine import Engine  # noqa: E402

print("library:", _lib.LIB_PATH, flush=True)
nblk, T, N = 20, 130, 6
mod = Dlm.polynomial(2)
for _ in range(nblk - 1):
    mod = mod * Dlm.polynomial(2)
mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
d, q = mat.d, mat.p
A_ = np.random.default_rng(nblk).standard_normal((d, d))
p = DlmParameters(np.eye(q) * 1.1, A_ @ A_.T / d + 0.1 * np.eye(d), np.zeros(d), np.eye(d))
rng = np.random.default_rng(nblk + T)
y = rng.standard_normal((N, T, q)).cumsum(axis=1) * 0.5 + rng.standard_normal((N, T, q))
y[2, T // 2, 3] = np.nan

eng = Engine(0)
OUTER, PER = _lib.OPT_STATS_OUTER, _lib.OPT_SAMPLER_PER_SERIES
runs = {}
for name, flags in (("shared_outer", OUTER), ("shared_plain", 0), ("A_perseries_outer", OUTER | PER), ("B_perseries_plain", PER)):
    for rep in (0, 1):
        out = eng.ffbs(mat, p, y, flags=flags, seed=5, series_offset=3)
        runs[(name, rep)] = (np.array(out["theta"]), np.array(out["status"]), eng.last_variant)
        print(name, rep, eng.last_variant, "status", out["status"].tolist(), flush=True)


def diff(a, b):
    ta, tb = runs[a][0], runs[b][0]
    ne = np.argwhere(ta != tb)
    if len(ne) == 0:
        return "equal"
    ts = np.unique(ne[:, 1])
    rel = np.abs(ta - tb).max() / np.abs(tb).max()
    return f"DIFFER: {len(ne)} values, series {np.unique(ne[:, 0]).tolist()}, t in [{ts.min()}, {ts.max()}] ({len(ts)} steps), max rel {rel:.3e}"


names = ["shared_outer", "shared_plain", "A_perseries_outer", "B_perseries_plain"]
for nm in names:
    print("repeat", nm, diff((nm, 0), (nm, 1)))
for i in range(len(names)):
    for j in range(i + 1, len(names)):
        print(names[i], "vs", names[j], diff((names[i], 0), (names[j], 0)))
# conditional-moment records (h_t, H_t) of the two per-series machine codes: H_t bit for bit?
ca = eng.ffbs(mat, p, y, flags=OUTER | PER, seed=5, series_offset=3, want_cond=True)
cb = eng.ffbs(mat, p, y, flags=PER, seed=5, series_offset=3, want_cond=True)
Ha, Hb = np.array(ca["cond"])[..., d:], np.array(cb["cond"])[..., d:]
ne = np.argwhere((Ha != Hb).any(axis=-1))
print("H_t of A vs B:", "equal" if len(ne) == 0 else f"DIFFER at (series, t): {ne[:12].tolist()} ... {len(ne)} records")
# the Newton-Schulz exit each full step should take, recomputed on the host from series 0's filter records
filt = np.array(ca["filt"])[0]
G = mat.G[:d * d].reshape(d, d, order="F")
W = p.w
prev = None
rows = []
for t in range(T - 1, -1, -1):
    C = filt[t, d:].reshape(d, d, order="F")
    R = G @ C @ G.T + W
    if prev is not None and (t & 31) != 31 and not (t < 64 and (t & 7) == 7):
        e = np.abs(np.eye(d) - R @ prev).max() * d
        rows.append((t, e))
    prev = np.linalg.inv(R)
far = [(t, round(e, 3)) for t, e in rows if e >= 0.25]
print("host: steps whose warm start is near or beyond the exit (n max|e| >= 0.25; exit at 0.5):", far)
eng.close()
