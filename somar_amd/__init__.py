"""somar_amd -- MI355X-native (gfx950 / HIP / RCCL) drop-in for the pressure-projection hot path of
UNC-CFD/somar: the semicoarsening multigrid behind AMRPressureSolver / MappedAMRPoissonOp.

The product is the C-ABI shared library ``somar_amd/libsomar_amd.so`` (include/somar_amd.h), built
from somar_amd/csrc by ``python -m somar_amd.build``.  This package is only the thin ctypes binding
used by tests and bench.py; it never computes anything itself and has NO CPU fallback: importing
``somar_amd.api`` raises if the library is missing, and every call fails loudly without a GPU.
"""
from .api import (  # noqa: F401
    AMRPressureSolver,
    LevelLepticSolver,
    SomarError,
    lib,
    lib_path,
)

__all__ = ["AMRPressureSolver", "LevelLepticSolver", "SomarError", "lib", "lib_path"]
