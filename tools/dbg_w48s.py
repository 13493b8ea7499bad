import numpy as np, sys, ctypes, os
os.environ["DLM_ENGINE_LIB"] = os.path.abspath("bayesian_dlms_amd/libdlm_engine_dbg.so")
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from bayesian_dlms_amd import _lib
from bayesian_dlms_amd.engine import Engine
from test_shared_sampler_gpu import blocks
eng = Engine(0)
lib = eng.lib
nblk, T = 20, 8
mat, p = blocks(nblk, T, seed=20)
rng = np.random.default_rng(20 + T)
N = 2
y = rng.standard_normal((N, T, mat.p)).cumsum(axis=1)
z = np.zeros((N, T + 1, mat.d))
dbg = np.zeros((2, 4096), dtype=np.int32)
sh = eng.ffbs(mat, p, y, z=z)
print(eng.last_variant)
lib.dlm_debug_w48_steps(dbg.ctypes.data_as(ctypes.c_void_p))
print("EXP   ", [hex(v) for v in dbg[1, :T]])

ps = eng.ffbs(mat, p, y, z=z, flags=_lib.OPT_SAMPLER_PER_SERIES)
lib.dlm_debug_w48_steps(dbg.ctypes.data_as(ctypes.c_void_p))
print("series", [hex(v) for v in dbg[0, :T]])

print([bool(np.array_equal(sh["theta"][0, t], ps["theta"][0, t])) for t in range(T + 1)])
