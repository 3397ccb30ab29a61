"""TEST INFRASTRUCTURE, NOT PRODUCT CODE: ctypes wrapper of oracle/cpu_vcycle.c, the C/OpenMP-orchestrated
single-box V-cycle that bench.py times as `cpu_baseline` (the reference's CPU path cannot be built here)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}
_PD = C.POINTER(C.c_double)


def _host_tag():
    """identifies the CPU the -march=native build is for (model name + ISA flags)"""
    import hashlib
    try:
        txt = open("/proc/cpuinfo").read()
        keep = [ln for ln in txt.splitlines() if ln.startswith(("model name", "flags"))][:2]
    except OSError:
        keep = ["unknown"]
    return hashlib.sha1("\n".join(keep).encode()).hexdigest()[:10]


def lib(fast=True):
    """fast: -O3 -march=native (timing build, one per host CPU); otherwise -O2 -ffp-contract=off (parity build, as liboracle.so)"""
    name = ("liboracle_cpu_%s.so" % _host_tag()) if fast else "liboracle_cpu_exact.so"
    if name not in _LIBS:
        subprocess.check_call(["make", "-s", "-C", _HERE, name])
        L = C.CDLL(os.path.join(_HERE, name))
        L.cpuvc_create.restype = C.c_void_p
        L.cpuvc_create.argtypes = [C.POINTER(C.c_int), _PD, _PD, _PD, _PD, _PD, C.c_int, C.c_int, C.c_int, C.c_int]
        L.cpuvc_vcycle.argtypes = [C.c_void_p, _PD, _PD]
        L.cpuvc_relax.argtypes = [C.c_void_p, _PD, _PD, C.c_int]
        L.cpuvc_residual.argtypes = [C.c_void_p, _PD, _PD, _PD]
        L.cpuvc_depth.argtypes = [C.c_void_p]
        L.cpuvc_zero_avg.argtypes = [C.c_void_p, C.c_int]
        L.cpuvc_bottom_iters.argtypes = [C.c_void_p]
        L.cpuvc_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.cpuvc_set_bottom_metric.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.cpuvc_destroy.argtypes = [C.c_void_p]
        L.cpuvc_triad.restype = C.c_double
        L.cpuvc_triad.argtypes = [C.c_long, C.c_int, C.c_int]
        _LIBS[name] = L
    return _LIBS[name]


def _p(a):
    assert a.dtype == np.float64 and a.flags.f_contiguous
    return a.ctypes.data_as(_PD)


class CpuVCycle:
    """One box [0, n) with a diagonal metric (Fortran-ordered arrays: jg[d] over faces(valid, d), jinv over valid),
    homogeneous Neumann on every side."""

    def __init__(self, n, dx, jg, jinv, pre=2, post=2, bottom=2, nthreads=1, fast=True):
        self.L = lib(fast)
        self.n = tuple(int(x) for x in n)
        self._keep = (jg, jinv)   # the finest level borrows the caller's arrays
        self.h = self.L.cpuvc_create((C.c_int * 3)(*self.n), (C.c_double * 3)(*dx), _p(jg[0]), _p(jg[1]), _p(jg[2]),
                                     _p(jinv), pre, post, bottom, nthreads)

    def depth(self):
        return self.L.cpuvc_depth(self.h)

    def zero_avg(self, d):
        return bool(self.L.cpuvc_zero_avg(self.h, d))

    def set_threads(self, n):
        self.L.cpuvc_set_threads(self.h, n)

    def vcycle(self, corr, res):
        """corr: (n+2)^3 (overwritten, starts from zero), res: n^3"""
        assert corr.shape == tuple(a + 2 for a in self.n) and res.shape == self.n
        self.L.cpuvc_vcycle(self.h, _p(corr), _p(res))

    def relax(self, phi, rhs, iters=1):
        self.L.cpuvc_relax(self.h, _p(phi), _p(rhs), iters)

    def residual(self, out, phi, rhs):
        self.L.cpuvc_residual(self.h, _p(out), _p(phi), _p(rhs))

    def close(self):
        if self.h:
            self.L.cpuvc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def triad_gbs(nthreads, n=1 << 26, reps=5):
    return lib(True).cpuvc_triad(n, nthreads, reps)
