"""CPU-only: the C-ABI library loads and exports every symbol include/dlm_engine.h declares."""
import os
import re

import pytest

from bayesian_dlms_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "dlm_engine.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dlm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 19
    bound = {name for name, _, _ in _lib.SYMBOLS}
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in dlm_engine.h but not exported"
        assert name in bound, f"{name} has no ctypes prototype in _lib.SYMBOLS"
    assert b"gfx950" in lib.dlm_version()


def test_stats_len_matches_layout():
    lib = _lib.load()
    assert lib.dlm_stats_len(13, 1, 0) == 2 + 13 + 1
    assert lib.dlm_stats_len(40, 20, _lib.OPT_STATS_OUTER) == 40 + 1600 + 1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.EngineLibraryMissing):
        _lib.load()


def test_no_device_is_an_error_not_a_fallback():
    """Without a GPU the engine refuses to exist; nothing silently computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bayesian_dlms_amd.engine import Engine, EngineError
    with pytest.raises(EngineError):
        Engine(0)
