"""Round-4 diagnostic: does the result of k_filter_w48 depend on what its OUTPUT buffer held before the call?  Its convergence test
re-reads the record it stored one step earlier (dlm_wave48.hip, "has the covariance stopped moving?") -- a store -> load round trip
through memory inside one wave.  If that load can return the buffer's OLD content, the step at which a series starts its steady
steps depends on the history of the buffer: invisible when a call is repeated on the same buffer, visible after a call of another
shape (tests/test_shared_sampler_gpu.py [10-65-9-16], filt mismatch, 1 run of 3)."""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from bayesian_dlms_amd import _lib
from bayesian_dlms_amd.dlm import Dlm, DlmParameters, materialise
from bayesian_dlms_amd.engine import Engine

eng = Engine(0)
for nblk, T, N in ((10, 65, 9), (10, 400, 512), (20, 1000, 2000)):
    mod = Dlm.polynomial(2)
    for _ in range(nblk - 1):
        mod = mod * Dlm.polynomial(2)
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    d, q = mat.d, mat.p
    A_ = np.random.default_rng(nblk).standard_normal((d, d))
    p = DlmParameters(np.eye(q) * 1.1, A_ @ A_.T / d + 0.1 * np.eye(d), np.zeros(d), np.eye(d))
    y = torch.as_tensor(np.random.default_rng(nblk + T).standard_normal((N, T, q)).cumsum(axis=1), device="cuda:0")
    buf = torch.empty((N, T + 1, d + d * d), dtype=torch.float64, device="cuda:0")
    res = {}
    for name, fill in (("zeros", 0.0), ("nan", float("nan")), ("1e300", 1e300), ("previous result", None), ("0.37", 0.37)):
        if fill is not None:
            buf.fill_(fill)
        torch.cuda.synchronize()
        out = eng.filter(mat, p, y, out=buf, flags=_lib.OPT_COUNT_STEPS)
        res[name] = (out["filt"].clone(), eng.last_counters()[0], int((out["status"] != 0).sum().item()))
        print(f"d={d} T={T} N={N} buffer pre-filled with {name:16s}: variant {eng.last_variant}, steady steps {res[name][1]}, status!=0 {res[name][2]}", flush=True)
    ref = res["zeros"][0]
    for name, (r, cnt, _) in res.items():
        ne = (r != ref) & ~(torch.isnan(r) & torch.isnan(ref))
        k = int(ne.sum().item())
        msg = "equal" if k == 0 else f"DIFFER in {k} values, series {torch.unique(ne.nonzero()[:, 0])[:8].tolist()}, first t {int(ne.nonzero()[:, 1].min())}, max |diff| {float((r - ref).abs().nan_to_num().max()):.3e}"
        print(f"   vs zeros-prefilled: {name:16s} {msg}", flush=True)
eng.close()
