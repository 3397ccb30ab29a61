"""Builds somar_amd/libsomar_amd.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m somar_amd.build [--force]

-ffp-contract=off: the kernels reproduce the reference's Fortran operation order bit for bit
(no FMA contraction); they are HBM-bound, so the extra multiply-add issue slots are free.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libsomar_amd.so")
SOURCES = ["kernels.hip", "box_bicgstab.hip", "gsrb_fused.hip", "resid_march.hip", "full19.hip", "full19_march.hip", "full19_fused.hip", "projection.hip", "line_gsrb.hip", "amr_kernels.hip", "leptic_kernels.hip", "maps.hip", "level.cpp", "solver.cpp", "solver_full.cpp", "amr.cpp", "leptic.cpp", "comm_rccl.cpp", "comm_shm.cpp", "capi.cpp"]
HEADERS = ["common.h", "kernels.h", "level.h", "solver.h", "amr.h", "leptic.h", os.path.join("..", "..", "include", "somar_amd.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-result", "-Wno-unused-value", "-I/opt/rocm/include"]


EXTRA = os.environ.get("SOMAR_EXTRA_FLAGS", "").split()   # A/B experiments, e.g. -DSOMAR_NT_LOADS


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not (force or _stale()):
        return OUT
    objs = []
    for f in SOURCES:
        src = os.path.join(CSRC, f)
        obj = os.path.join(CSRC, os.path.splitext(f)[0] + ".o")
        cmd = [HIPCC] + FLAGS + EXTRA + (["-x", "hip"] if f.endswith(".cpp") else []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl", "-lrt"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
