"""CPU tests of the drop-in boundary: libbohip.so loads and exports exactly what include/bohip.h declares;
without a GPU every compute entry point fails loudly (there is no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, has_gpu


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "bohip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bohip_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from bohip import _lib

    lib = C.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/bohip.h but not exported by libbohip.so"
    assert set(syms) == set(_lib.SIGNATURES), set(syms) ^ set(_lib.SIGNATURES)


def test_header_cites_reference_call_sites():
    txt = open(os.path.join(ROOT, "include", "bohip.h")).read()
    for cite in ("src/models/gp.jl:11", "src/models/gp.jl:8", "src/acquisition.jl:54-68", "src/acquisitionfunctions.jl:4-9",
                 "src/acquisition.jl:11-17", "src/acquisitionfunctions.jl:107-108"):
        assert cite in txt


def test_version_and_error_channel():
    from bohip import _lib

    lib = _lib.load()
    assert b"gfx950" in lib.bohip_version()
    h = C.c_void_p()
    assert lib.bohip_gp_create(0, 10, 0, 0, C.byref(h)) == _lib.E_ARG      # d < 1
    assert b"d must be" in lib.bohip_last_error()
    assert lib.bohip_gp_create(2, 10, 9, 0, C.byref(h)) == _lib.E_ARG      # unknown kernel
    assert lib.bohip_gp_dims(None, None, None) == _lib.E_ARG               # null handle never crashes


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu():
    import bohip
    from bohip import _lib

    assert _lib.load().bohip_device_count() == 0
    with pytest.raises(bohip.BohipError) as e:
        bohip.ElasticGPE(2)
    assert e.value.code == _lib.E_NODEVICE
    assert "no CPU fallback" in str(e.value)


def test_thompson_generator_is_a_pure_function():
    from bohip import _lib

    lib = _lib.load()
    a = lib.bohip_thompson_normal(7, 3, 11)
    assert a == lib.bohip_thompson_normal(7, 3, 11)
    assert a != lib.bohip_thompson_normal(7, 4, 11) and a != lib.bohip_thompson_normal(8, 3, 11)
    import numpy as np
    z = np.array([lib.bohip_thompson_normal(1, s, j) for s in range(40) for j in range(250)])
    assert abs(z.mean()) < 0.05 and abs(z.std() - 1) < 0.05


def test_product_path_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package may import, link or dlopen it."""
    pkg = os.path.join(ROOT, "bayesianoptimization.jl_amd")
    pat = re.compile(r"import\s+oracle|from\s+oracle|liboracle|gp_oracle|oracle/|oracle\.oracle|COracle")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                assert not pat.search(open(os.path.join(dp, f)).read()), f
