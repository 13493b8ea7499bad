"""The reference-side binding (integration/): every export of include/dlm_engine.h has a JNI function in
integration/jni/dlm_jni.cpp and an @native declaration in integration/scala/Batched.scala with the same arity; the JNI
glue compiles (against tests/cpp/jni_stub/jni.h -- there is no JDK in this image) and, on the GPU, reproduces the
reference's golden CSVs through every entry point (tests/cpp/jni_glue_check.cpp)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JNI = os.path.join(ROOT, "integration", "jni", "dlm_jni.cpp")
SCALA = os.path.join(ROOT, "integration", "scala", "Batched.scala")
SRC = os.path.join(ROOT, "tests", "cpp", "jni_glue_check.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "jni_glue_check")
PREFIX = "Java_com_github_jonnylaw_dlm_gpu_Native_"


def _strip_comments(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def _exports():
    src = _strip_comments(open(os.path.join(ROOT, "include", "dlm_engine.h")).read())
    return sorted(set(re.findall(r"\b(dlm_[a-z0-9_]+)\s*\(", src)))


def _jni_functions():
    """name -> number of Java-visible parameters (everything after JNIEnv*, jobject)"""
    src = _strip_comments(open(JNI).read())
    out = {}
    for m in re.finditer(PREFIX + r"(\w+)\s*\(([^)]*)\)", src):
        params = [p.strip() for p in m.group(2).split(",") if p.strip()]
        assert params[0].startswith("JNIEnv") and params[1].startswith("jobject"), m.group(0)
        out[m.group(1)] = len(params) - 2
    return out


def _native_declarations():
    src = _strip_comments(open(SCALA).read())
    out = {}
    for m in re.finditer(r"@native\s+def\s+(\w+)\s*\(([^)]*)\)", src):
        out[m.group(1)] = len([p for p in m.group(2).split(",") if p.strip()])
    return out


def test_every_export_is_bound_in_the_jni_glue():
    body = _strip_comments(open(JNI).read())
    exports = _exports()
    assert len(exports) >= 35
    missing = [e for e in exports if not re.search(r"\b" + e + r"\s*\(", body)]
    assert not missing, f"exports without a JNI binding: {missing}"


def test_jni_functions_and_native_declarations_agree():
    jni, nat = _jni_functions(), _native_declarations()
    assert len(jni) >= 35
    assert set(jni) == set(nat), f"only in JNI: {set(jni) - set(nat)}, only in Scala: {set(nat) - set(jni)}"
    for name in jni:
        assert jni[name] == nat[name], f"{name}: {jni[name]} JNI parameters, {nat[name]} in the @native declaration"


def test_scala_shim_offers_the_seven_reference_calls_and_bounded_buffers():
    src = _strip_comments(open(SCALA).read())
    for call in ("def filterDlm(", "def backwardsSmoother(", "def ffbsDlm(", "def svdFilterDlm(", "def svdFfbsDlm(",
                 "def gibbsSample(", "def gibbsSampleSvd(", "def gibbsWishartSample(", "def filterSmooth(", "def logLikelihood("):
        assert call in src, call
    # one direct buffer only (the bounded staging buffer); byte counts never narrowed to Int
    assert src.count("allocateDirect(") == 1 and "allocateDirect(stagingBytes)" in src
    assert not re.search(r"\*\s*8\s*\)\.toInt", src)
    # the quirk switches are reachable
    for flag in ("SmootherCompatQ1", "SvdRawWQ2", "SvdSamplerQ9"):
        assert src.count(flag) >= 2, flag


def _build():
    lib = os.path.join(ROOT, "bayesian_dlms_amd")
    cmd = ["g++", "-std=c++17", "-O2", "-pthread", "-Wall", "-Wno-comment", "-I" + os.path.join(ROOT, "tests", "cpp", "jni_stub"),
           "-I" + os.path.join(ROOT, "include"), SRC, "-L" + lib, "-ldlm_engine", "-Wl,-rpath," + lib,
           "-Wl,-rpath,/opt/rocm/lib", "-o", EXE]
    subprocess.check_call(cmd)


def test_jni_glue_compiles_and_links():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_jni_glue_runs_every_entry_point_on_the_gpu():
    _build()
    # (the binary's own watchdog names the native call in flight every 20 s and ends the process after 150 s in one call; should even
    #  that fail, what the child wrote so far is printed before the test fails)
    try:
        out = subprocess.run([EXE, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, timeout=300)
    except subprocess.TimeoutExpired as exc:
        def txt(b):
            return b.decode(errors="replace") if isinstance(b, bytes) else (b or "")
        pytest.fail("jni_glue_check hung; its output so far:\n" + txt(exc.stdout) + "\n" + txt(exc.stderr))
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "JNI GLUE OK" in out.stdout, out.stdout + out.stderr
