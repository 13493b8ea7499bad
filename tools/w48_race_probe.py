"""Round-4 diagnostic: how often, where and by how much do the outputs of a shared-factor FFBS call (batch kernels on the engine's
stream beside the table / normals kernels on its auxiliary streams) differ from those of the same call made alone?  Three shapes in
rotation on one engine, host-memory calls as in tests/test_shared_sampler_gpu.py; the references are computed once per shape.
  python tools/w48_race_probe.py [rounds] [extra flags for the probed call, e.g. 67108864 = DLM_OPT_SAMPLER_PER_SERIES as a control]"""
import sys
sys.path.insert(0, ".")
import numpy as np
from bayesian_dlms_amd import _lib
from bayesian_dlms_amd.dlm import Dlm, DlmParameters, materialise
from bayesian_dlms_amd.engine import Engine

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
extra = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def blocks(nblk, T, seed):
    mod = Dlm.polynomial(2)
    for _ in range(nblk - 1):
        mod = mod * Dlm.polynomial(2)
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    d, q = mat.d, mat.p
    A = np.random.default_rng(seed).standard_normal((d, d))
    return mat, DlmParameters(np.eye(q) * 1.1, A @ A.T / d + 0.1 * np.eye(d), np.zeros(d), np.eye(d))


eng = Engine(0)
shapes = []
for nblk, T, N, flags in ((20, 130, 6, _lib.OPT_STATS_OUTER), (20, 70, 5, 0), (10, 65, 9, _lib.OPT_STATS_OUTER)):
    mat, p = blocks(nblk, T, nblk)
    rng = np.random.default_rng(nblk + T)
    y = rng.standard_normal((N, T, mat.p)).cumsum(axis=1) * 0.5 + rng.standard_normal((N, T, mat.p))
    y[2, T // 2, 3] = np.nan
    ref = eng.ffbs(mat, p, y, flags=flags | _lib.OPT_SAMPLER_PER_SERIES, seed=5, series_offset=3)
    reff = eng.filter(mat, p, y)["filt"]
    assert np.array_equal(ref["filt"], reff, equal_nan=True)
    shapes.append((mat, p, y, flags, ref))
bad = 0
for r in range(rounds):
    for i, (mat, p, y, flags, ref) in enumerate(shapes):
        out = eng.ffbs(mat, p, y, flags=flags | extra, seed=5, series_offset=3)
        for key in ("filt", "theta", "stats", "status"):
            a, b = np.asarray(out[key]), np.asarray(ref[key])
            ne = np.argwhere(~((a == b) | ((a != a) & (b != b))))
            if len(ne):
                bad += 1
                d = mat.d
                msg = f"round {r} shape {i} (d={d}, T={mat.T}) {eng.last_variant} {key}: {len(ne)} of {a.size} values differ"
                if a.ndim == 3:
                    comp = ne[:, 2]
                    msg += (f"; series {np.unique(ne[:, 0]).tolist()}; t {np.unique(ne[:, 1]).tolist()[:16]}{'...' if len(np.unique(ne[:, 1])) > 16 else ''}; "
                            f"mean part {int((comp < d).sum())}, covariance part {int((comp >= d).sum())}" if key == "filt" else f"; series {np.unique(ne[:, 0]).tolist()}; t range [{ne[:, 1].min()}, {ne[:, 1].max()}]")
                with np.errstate(invalid="ignore"):
                    msg += f"; max |diff| {np.nanmax(np.abs(a.astype(float) - b.astype(float))):.3e}; first {ne[0].tolist()}: {a[tuple(ne[0])]!r} vs {b[tuple(ne[0])]!r}"
                print(msg, flush=True)
    if r % 25 == 24:
        print(f"... {r + 1} rounds, {bad} mismatching outputs so far", flush=True)
print(f"DONE: {rounds} rounds x {len(shapes)} shapes, {bad} mismatching outputs (extra flags {extra})")
eng.close()
