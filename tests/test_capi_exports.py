"""CPU-side checks of the drop-in boundary: the shared library builds for gfx950, loads, and exports
exactly the entry points include/somar_amd.h declares.  No compute call is made (no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from somar_amd import build
    path = build.build()
    assert os.path.exists(path)
    return path


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "somar_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(somar_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree(built_lib):
    from somar_amd import api
    assert _header_functions() == api.EXPORTS


def test_library_exports_every_declared_symbol(built_lib):
    lib = C.CDLL(built_lib)
    for name in _header_functions():
        assert hasattr(lib, name), "missing export " + name
    assert lib.somar_abi_version() == 11


def test_no_cpu_fallback_without_gpu(built_lib):
    """On a box without a GPU the product must fail loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from somar_amd import AMRPressureSolver, SomarError
    s = AMRPressureSolver()
    with pytest.raises(SomarError):
        s.define((0, 0, 0), (7, 7, 7), (0, 0, 0), (1.0, 1.0, 1.0), [((0, 0, 0), (7, 7, 7))])


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "somar_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "liboracle" not in src, f


def test_header_is_plain_c(tmp_path):
    """the boundary is a C ABI: include/somar_amd.h must compile as C99 on its own (no C++-isms, no torch/HIP types)"""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "somar_amd.h"\nint main(void) { int (*f)(void) = somar_abi_version; return f == 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only",
                           "-I", os.path.join(ROOT, "include"), str(src)])
