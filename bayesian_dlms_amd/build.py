"""Builds the HIP engine in-tree: bayesian_dlms_amd/libdlm_engine.so (gfx950 only)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdlm_engine.so")
SOURCES = ["dlm_engine.hip", "dlm_generic.hip", "dlm_mfma16.hip", "dlm_sparse16.hip", "dlm_sampler16.hip", "dlm_tiled.hip", "dlm_wave48.hip", "dlm_svd.hip", "dlm_ar1.hip", "dlm_lane.hip", "dlm_gibbs.hip", "dlm_loglik.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
FLAGS = BASE_FLAGS + ["-mllvm", "-amdgpu-mfma-vgpr-form"]
# dlm_wave48.hip keeps whole matrices in registers (up to the 512-register budget of a wave): its accumulators may live
# in AGPRs, and the VGPR-form rewrite pass of this compiler crashes on it
FILE_FLAGS = {"dlm_wave48.hip": BASE_FLAGS}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name, defines):
    """Experimental build with extra -D flags -> bayesian_dlms_amd/libdlm_engine_<name>.so."""
    out = os.path.join(HERE, f"libdlm_engine_{name}.so")
    objs = []
    os.makedirs(os.path.join(HERE, "build", name), exist_ok=True)
    for src in SOURCES:
        o = os.path.join(HERE, "build", name, src.replace(".hip", ".o"))
        subprocess.check_call([HIPCC] + FILE_FLAGS.get(src, FLAGS) + [f"-D{d}" for d in defines] + ["-c", os.path.join(CSRC, src), "-o", o])
        objs.append(o)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-L/opt/rocm/lib", "-lrccl"])
    return out


def build_tu_variant(name, src, defines):
    """Experimental build in which only ONE translation unit gets extra -D flags (the other objects come from the regular
    build) -> bayesian_dlms_amd/libdlm_engine_<name>.so.  Select it with DLM_ENGINE_LIB."""
    build()
    out = os.path.join(HERE, f"libdlm_engine_{name}.so")
    o = os.path.join(HERE, "build", f"{name}_{src.replace('.hip', '.o')}")
    subprocess.check_call([HIPCC] + FILE_FLAGS.get(src, FLAGS) + [f"-D{d}" for d in defines] + ["-c", os.path.join(CSRC, src), "-o", o])
    objs = [o if s == src else os.path.join(HERE, "build", s.replace(".hip", ".o")) for s in SOURCES]
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-L/opt/rocm/lib", "-lrccl"])
    return out


def build(force=False, verbose=False):
    hdrs = [os.path.join(CSRC, "dlm_internal.h"), os.path.join(HERE, "..", "include", "dlm_engine.h")]
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    todo = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            todo.append([HIPCC] + FILE_FLAGS.get(src, FLAGS) + ["-c", s, "-o", o])
    if todo:   # the translation units are independent: compile them side by side (the largest takes about two minutes)
        from concurrent.futures import ThreadPoolExecutor
        def run(cmd):
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(len(todo), max(1, (os.cpu_count() or 2) // 2))) as pool:
            list(pool.map(run, todo))
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-lrccl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
