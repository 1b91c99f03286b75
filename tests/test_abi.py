"""CPU: the shared library loads and exports every entry point the header declares
(no compute call is made: there is no GPU in the CPU test tier)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "spmm_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(spmm_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from spmm_amd import capi
    assert os.path.exists(capi.LIB_PATH), "build first: make -C spmm_amd/csrc"
    lib = ctypes.CDLL(capi.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_no_gpu_is_reported_not_hidden():
    """Without a GPU the context constructor must fail loudly (no CPU fallback exists)."""
    import os
    from spmm_amd import capi
    if os.path.exists("/dev/kfd"):
        return
    try:
        capi.Context(0)
    except capi.SpmmError as e:
        assert e.code in (-5, -2)
    else:
        raise AssertionError("Context() succeeded without a GPU")
