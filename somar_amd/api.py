"""ctypes binding of libsomar_amd.so + a Python mirror of the reference's AMRPressureSolver interface
(projection/AMRPressureSolver.H:42-172: setAMRMGParameters, setBottomParameters, define, solve, undefine)
so the parity tests read like calls into the reference.  No computation happens here and there is no
CPU fallback: a missing library or a missing GPU raises.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BC_NONE, BC_NEUM, BC_DIRI = -1, 0, 1
F_PHI, F_RHS, F_RES, F_CORR, F_BEST, F_SCRATCH, F_AMR_CORR, F_AMR_RES, F_HEAT_OLD, F_HEAT_SRC = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
MAP_CYLINDRICAL, MAP_BATHYMETRIC, MAP_TWISTED, MAP_TWISTED1 = 1, 2, 3, 4   # SOMAR_MAP_* of include/somar_amd.h
MAX_HISTORY = 64
COMM_ID_BYTES = 128


def FIELD(depth, which):
    return (depth << 8) | which


class SomarError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [("imin", C.c_int), ("imax", C.c_int), ("eps", C.c_double), ("hang", C.c_double),
                ("norm_thresh", C.c_double), ("num_smooth_down", C.c_int), ("num_smooth_up", C.c_int),
                ("num_smooth_bottom", C.c_int), ("num_smooth_precond", C.c_int), ("num_mg", C.c_int),
                ("max_depth", C.c_int), ("precond_mode", C.c_int), ("relax_mode", C.c_int), ("verbosity", C.c_int),
                ("bottom_imax", C.c_int), ("bottom_num_restarts", C.c_int), ("bottom_norm_type", C.c_int),
                ("bottom_verbosity", C.c_int), ("bottom_eps", C.c_double), ("bottom_reps", C.c_double),
                ("bottom_hang", C.c_double), ("bottom_small", C.c_double), ("space_dim", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("iters", C.c_int), ("exit_status", C.c_int), ("status", C.c_int), ("bottom_iters", C.c_int),
                ("bottom_exit", C.c_int), ("nhistory", C.c_int), ("initial_rnorm", C.c_double),
                ("final_rnorm", C.c_double), ("history", C.c_double * MAX_HISTORY)]


class LepticParams(C.Structure):
    _fields_ = [("max_order", C.c_int), ("norm_type", C.c_int), ("hang", C.c_double), ("horiz_rhs_tol", C.c_double),
                ("domain_height", C.c_double), ("horiz", Params), ("full", Params)]


class LepticStats(C.Structure):
    _fields_ = [("exit_status", C.c_int), ("orders", C.c_int), ("horiz_solves", C.c_int),
                ("used_full_solver", C.c_int), ("nres", C.c_int), ("res_norms", C.c_double * MAX_HISTORY),
                ("horiz", Stats), ("full", Stats)]


def lib_path():
    return os.path.join(_HERE, "libsomar_amd.so")


# name -> argtypes; every function returns int (0 ok) except somar_last_error
_PD, _PI = C.POINTER(C.c_double), C.POINTER(C.c_int)
_H = C.c_void_p
_SIGS = {
    "somar_abi_version": [],
    "somar_device_count": [_PI],
    "somar_params_default": [C.POINTER(Params)],
    "somar_solver_create": [C.POINTER(_H), _PI, _PI, _PI, _PD, _PI, C.c_int, _PI, _PI, C.c_double, C.c_double,
                            C.POINTER(Params), _H],
    "somar_solver_destroy": [_H],
    "somar_solver_num_local_patches": [_H, _PI],
    "somar_solver_patch_box": [_H, C.c_int, C.c_int, _PI, _PI],
    "somar_solver_set_metric_ortho": [_H, C.c_int, _PD, _PD, _PD, _PD],
    "somar_solver_set_metric_full": [_H, C.c_int, _PD, _PD, _PD, _PD],
    "somar_solver_set_bc_values": [_H, _PD],
    "somar_solver_finalize": [_H],
    "somar_solver_depth": [_H, _PI],
    "somar_solver_mg_ref_ratio": [_H, C.c_int, _PI],
    "somar_solver_zero_avg": [_H, C.c_int, _PI],
    "somar_solver_metric_uniform": [_H, C.c_int, _PI, _PD],
    "somar_solver_level_info": [_H, C.c_int, _PI, _PD, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)],
    "somar_field_upload": [_H, C.c_int, C.c_int, _PD, _PI],
    "somar_field_download": [_H, C.c_int, C.c_int, _PD, _PI],
    "somar_field_set": [_H, C.c_int, C.c_double],
    "somar_field_fill_hash": [_H, C.c_int, C.c_ulonglong],
    "somar_field_remove_mean": [_H, C.c_int],
    "somar_field_norm": [_H, C.c_int, C.c_int, _PD],
    "somar_field_dot": [_H, C.c_int, C.c_int, _PD],
    "somar_solver_solve": [_H, C.c_int, C.c_int, C.POINTER(Stats)],
    "somar_solver_solve_host": [_H, C.POINTER(_PD), _PI, C.POINTER(_PD), _PI, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.POINTER(Stats)],
    "somar_level_relax": [_H, C.c_int, C.c_int, C.c_int, C.c_int],
    "somar_level_residual": [_H, C.c_int, C.c_int, C.c_int, C.c_int],
    "somar_level_apply_op": [_H, C.c_int, C.c_int, C.c_int],
    "somar_level_apply_op_bc": [_H, C.c_int, C.c_int, C.c_int],
    "somar_level_residual_bc": [_H, C.c_int, C.c_int, C.c_int, C.c_int],
    "somar_level_restrict_residual": [_H, C.c_int, C.c_int, C.c_int, C.c_int],
    "somar_level_prolong_increment": [_H, C.c_int, C.c_int, C.c_int],
    "somar_level_precond": [_H, C.c_int, C.c_int, C.c_int],
    "somar_vcycle": [_H, C.c_int, C.c_int],
    "somar_vcycle_from_zero": [_H, C.c_int, C.c_int],
    "somar_mini_vcycle": [_H, C.c_int, C.c_int],
    "somar_bottom_solve": [_H, C.c_int, C.c_int, _PI, _PI],
    "somar_bottom_kind": [_H, _PI],
    "somar_solver_counters": [_H, C.POINTER(C.c_longlong)],
    "somar_solver_fused19_sweeps": [_H, C.POINTER(C.c_longlong)],
    "somar_last_history": [_PD, C.c_int, _PI],
    "somar_host_fill_mt19937_64": [_PD, C.c_longlong, C.c_ulonglong, C.c_double, C.c_double],
    "somar_bathymetry_ledge": [_PD, C.c_longlong, _PD, _PD, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double],
    "somar_bathymetry_beam_generator": [_PD, C.c_longlong, _PD, C.c_double, C.c_double],
    "somar_dem_cubic_spline": [_PD, C.c_longlong, _PD, C.c_int, _PD, _PD],
    "somar_dem_bilinear": [_PD, C.c_longlong, _PD, _PD, C.c_int, C.c_int, _PD, _PD, _PD],
    "somar_dem_hermite": [_PD, C.c_longlong, _PD, _PD, C.c_int, C.c_int, _PD, _PD, _PD],
    "somar_vel_upload": [_H, C.c_int, C.c_int, _PD],
    "somar_vel_download": [_H, C.c_int, C.c_int, _PD],
    "somar_vel_wall_bc": [_H],
    "somar_level_divergence_mac": [_H, C.c_int, C.c_double],
    "somar_level_mac_correct": [_H, C.c_int, C.c_double],
    "somar_mac_project": [_H, C.c_double, C.c_int, C.c_int, C.POINTER(Stats)],
    "somar_mac_project_host": [_H, C.POINTER(_PD), C.POINTER(_PD), C.POINTER(_PD), C.c_double, C.c_int, C.c_int,
                               C.POINTER(Stats)],
    "somar_solver_set_cc_j": [_H, C.c_int, _PD, _PD, _PI],
    "somar_solver_set_face_j": [_H, C.c_int, C.c_int, _PD, _PD],
    "somar_vel_mult_by_j": [_H, C.c_int],
    "somar_vel_div_by_j": [_H, C.c_int],
    "somar_solver_set_alpha_beta": [_H, C.c_double, C.c_double],
    "somar_heat_step": [_H, C.c_int, C.c_double, C.c_int, C.POINTER(Stats)],
    "somar_ccvel_upload": [_H, C.c_int, _PD, _PI],
    "somar_ccvel_download": [_H, C.c_int, _PD, _PI],
    "somar_level_divergence_cc": [_H, C.c_int, C.c_double, C.c_int],
    "somar_level_cc_correct": [_H, C.c_int, C.c_double],
    "somar_cc_project": [_H, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(Stats)],
    "somar_cc_project_host": [_H, C.POINTER(_PD), _PI, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(Stats)],
    # FORT_PROTO shapes: FRA = data, 3 lo, 3 hi, ncomp; FRA1 = data, 3 lo, 3 hi; BOX = 3 lo, 3 hi; then dx, alpha, beta, redBlack
    "somar_k_gsrbiter3dortho": ([_PD] + [_PI] * 7 + [_PD] + [_PI] * 7 + ([_PD] + [_PI] * 6) * 5 + [_PI] * 6
                                + [_PD, _PD, _PD, _PI]),
    # FILLMAPPEDLAPDIAG3D: FRA1 lapDiag, 3 x FRA Jg, FRA1 Jinv, BOX region, REALVECT dx
    "somar_k_fillmappedlapdiag3d": ([_PD] + [_PI] * 6 + ([_PD] + [_PI] * 7) * 3 + [_PD] + [_PI] * 6 + [_PI] * 6 + [_PD]),
    # MAPPEDAVERAGE2: FRA coarse, FRA fine, FRA1 fineCCJinv, BOX box, INTVECT refRatio, BOX bref
    "somar_k_mappedaverage2": ([_PD] + [_PI] * 7 + [_PD] + [_PI] * 7 + [_PD] + [_PI] * 6 + [_PI] * 6 + [_PI] + [_PI] * 6),
    "somar_sync": [_H],
    "somar_timer_start": [_H],
    "somar_timer_stop": [_H, _PD],
    "somar_profile_enable": [_H, C.c_int],
    "somar_profile_get": [_H, C.c_int, _PI, _PD],
    "somar_plan_exchange": [_PI, _PI, _PI, C.c_int, _PI, _PI, C.c_int, C.c_int, C.c_int, _PI, _PI, _PI, _PI, _PI, _PI],
    "somar_amr_create": [C.POINTER(_H), C.c_int, _PI, _PI, _PI, _PD, _PI, _PI, _PI, _PI, _PI, C.c_double, C.c_double,
                         C.POINTER(Params), _H],
    "somar_amr_destroy": [_H],
    "somar_amr_level": [_H, C.c_int, C.POINTER(_H)],
    "somar_amr_finalize": [_H],
    "somar_amr_solve": [_H, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Stats)],
    "somar_amr_solve_host": [_H, C.POINTER(C.POINTER(_PD)), _PI, C.POINTER(C.POINTER(_PD)), _PI, C.c_int, C.c_int, C.c_int,
                             C.c_int, C.POINTER(Stats)],
    "somar_amr_interp_cf": [_H, C.c_int, C.c_int, C.c_int],
    "somar_amr_level_project": [_H, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(Stats)],
    "somar_amr_set_inspector": [_H, C.c_void_p, C.c_void_p],
    "somar_amr_cc_project": [_H, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(Stats)],
    "somar_amr_comp_divergence_cc": [_H, C.c_int, C.c_int, C.c_int, C.c_int],
    "somar_amr_comp_grad_correct_cc": [_H, C.c_int, C.c_int, C.c_int, C.c_double],
    "somar_amr_average_down_ccvel": [_H, C.c_int],
    "somar_amr_residual_level": [_H, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int],
    "somar_amr_zero_covered": [_H, C.c_int, C.c_int],
    "somar_amr_vcycle": [_H, C.c_int, C.c_int],
    "somar_altered_jgup": [C.c_longlong, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, _PD, C.c_double, C.c_double],
    "somar_leptic_params_default": [C.POINTER(LepticParams)],
    "somar_leptic_create": [C.POINTER(_H), _PI, _PI, _PI, _PD, _PI, C.c_int, _PI, _PI, C.c_double, C.c_double,
                            C.POINTER(Params), C.POINTER(LepticParams), _H],
    "somar_leptic_destroy": [_H],
    "somar_leptic_level": [_H, C.POINTER(_H)],
    "somar_leptic_part": [_H, C.c_int, C.POINTER(_H)],
    "somar_leptic_finalize": [_H],
    "somar_leptic_solve": [_H, C.c_int, C.POINTER(LepticStats)],
    "somar_amr_set_alpha_beta": [_H, C.c_double, C.c_double],
    "somar_amr_heat_step": [_H, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.POINTER(Stats)],
    "somar_solver_set_metric_map": [_H, C.c_int, _PD, _PD, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "somar_solver_set_vel_bc": [_H, C.POINTER(C.c_int), _PD],
    "somar_amr_tga_step": [_H, C.c_int, C.c_int, C.c_double, C.POINTER(Stats)],
    "somar_heat_flux_download": [_H, C.c_int, C.c_int, _PD],
    "somar_amr_enable_leptic": [_H, C.POINTER(LepticParams), C.c_int],
    "somar_amr_solve_leptic": [_H, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Stats)],
    "somar_amr_leptic_stats": [_H, C.c_int, C.POINTER(LepticStats)],
    "somar_diag_stream_probe": [C.c_int, C.c_longlong, C.c_int, _PD],
    "somar_metric_jgup_from_dxdxi": [C.c_longlong, C.c_int, _PD, _PD, C.c_double, _PD],
    "somar_solver_set_metric_uniform": [_H, _PD],
    "somar_comm_unique_id": [C.POINTER(C.c_ubyte)],
    "somar_comm_create": [C.POINTER(_H), C.POINTER(C.c_ubyte), C.c_int, C.c_int, C.c_int],
    "somar_comm_create_shm": [C.POINTER(_H), C.c_char_p, C.c_int, C.c_int, C.c_longlong],
    "somar_comm_selftest": [_H],
    "somar_comm_destroy": [_H],
}
EXPORTS = sorted(list(_SIGS) + ["somar_last_error"])


def lib():
    """Load libsomar_amd.so (fails loudly if it has not been built: there is no fallback)."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise SomarError("%s is missing: run `python -m somar_amd.build` (hipcc, gfx950). "
                             "somar_amd has no CPU fallback." % p)
        # One HIP runtime per process: PyTorch-ROCm carries its own copies of libamdhip64 / libhsa-runtime64 /
        # librccl and loads them by path.  If this library pulled in /opt/rocm's copies first, a later `import torch`
        # (bench.py's torch.distributed control plane, a test) would bring a SECOND runtime that finds no GPU, and
        # RCCL would fail with "unhandled cuda error".  Loading torch's first makes the dynamic loader satisfy this
        # library's DT_NEEDED entries with the same objects (matching SONAMEs).  A process without PyTorch installed
        # (a SOMAR build linking the C ABI) has only one runtime anyway.
        if "torch" not in sys.modules and os.environ.get("SOMAR_NO_TORCH_PRELOAD") is None:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(p)
        for name, args in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = C.c_int
        L.somar_last_error.restype = C.c_char_p
        L.somar_last_error.argtypes = []
        _LIB = L
    return _LIB


def _ck(rc):
    if rc != 0:
        raise SomarError("libsomar_amd error %d: %s" % (rc, lib().somar_last_error().decode()))


def _ia(v):
    return (C.c_int * len(v))(*[int(x) for x in v])


def _da(v):
    return (C.c_double * len(v))(*[float(x) for x in v])


def _dp(a):
    assert a.dtype == np.float64 and a.flags["F_CONTIGUOUS"]
    return a.ctypes.data_as(_PD)


class AMRPressureSolver:
    """Mirror of projection/AMRPressureSolver.H.  Boxes are (lo, hi) integer 3-tuples; host FABs are
    Fortran-ordered numpy arrays over valid.grow(ghost)."""

    def __init__(self):
        self._p = Params()
        _ck(lib().somar_params_default(C.byref(self._p)))
        self._h = None
        self._comm = None
        self.exitStatus = 0
        self.stats = None

    # -- AMRPressureSolver.H:53-77 ----------------------------------------------------------------
    def setAMRMGParameters(self, imin, imax, eps, maxDepth, num_precond_iters, num_smooth_down, num_smooth_up,
                           num_smooth_bottom, precondMode, relaxMode, numMG, hang, norm_thresh, verbosity):
        assert self._h is None, "setAMRMGParameters can only be called before define"
        p = self._p
        p.imin, p.imax, p.eps, p.max_depth = imin, imax, eps, maxDepth
        p.num_smooth_precond, p.num_smooth_down, p.num_smooth_up = num_precond_iters, num_smooth_down, num_smooth_up
        p.num_smooth_bottom, p.precond_mode, p.relax_mode, p.num_mg = num_smooth_bottom, precondMode, relaxMode, numMG
        p.hang, p.norm_thresh, p.verbosity = hang, norm_thresh, verbosity

    def setSpaceDim(self, n):
        """CH_SPACEDIM of the reference build this solver stands in for (3, or 2 with boxes one cell thick in z)"""
        assert self._h is None and n in (2, 3)
        self._p.space_dim = n

    def setBottomParameters(self, imax, numRestarts, eps, reps, hang, small, normType, verbosity):
        assert self._h is None, "setBottomParameters can only be called before define"
        p = self._p
        p.bottom_imax, p.bottom_num_restarts, p.bottom_eps, p.bottom_reps = int(imax), numRestarts, eps, reps
        p.bottom_hang, p.bottom_small, p.bottom_norm_type, p.bottom_verbosity = hang, small, normType, verbosity

    # -- define: what MappedAMRPoissonOpFactory::define receives (Factory.cpp:95-108), single level ---
    def define(self, domain_lo, domain_hi, periodic, dx, boxes, bc_type=None, owner=None, alpha=0.0, beta=1.0,
               comm=None):
        assert self._h is None, "already defined (call undefine first)"
        bc = bc_type if bc_type is not None else [BC_NEUM] * 6
        flat = []
        for lo, hi in boxes:
            flat += list(lo) + list(hi)
        h = _H()
        _ck(lib().somar_solver_create(C.byref(h), _ia(domain_lo), _ia(domain_hi), _ia([int(bool(x)) for x in periodic]),
                                      _da(dx), _ia(bc), len(boxes), _ia(flat), _ia(owner) if owner is not None else None,
                                      float(alpha), float(beta), C.byref(self._p), comm))
        self._h = h
        self._comm = comm
        n = C.c_int()
        _ck(lib().somar_solver_num_local_patches(h, C.byref(n)))
        self.num_local_patches = n.value

    # -- define on several AMR levels: AMRPressureSolver::define(levGeos, boxes, ...) (AMRPressureSolver.cpp:272-491)
    def defineAMR(self, domain_lo, domain_hi, periodic, dx0, ref_ratios, boxes_per_level, bc_type=None,
                  owners_per_level=None, alpha=0.0, beta=1.0, comm=None):
        """boxes_per_level[l]: (lo, hi) boxes of level l in level-l index space.  After this call self.levels[l]
        is a per-level view (metric upload, field I/O, single-level pieces); finalize() and solve*() act on the
        whole hierarchy."""
        assert self._h is None and getattr(self, "_amr", None) is None, "already defined"
        bc = bc_type if bc_type is not None else [BC_NEUM] * 6
        flat, nb, own = [], [], []
        for l, boxes in enumerate(boxes_per_level):
            nb.append(len(boxes))
            for lo, hi in boxes:
                flat += list(lo) + list(hi)
            own += list(owners_per_level[l]) if owners_per_level is not None else [0] * len(boxes)
        rr = [int(x) for r in ref_ratios for x in r] or [1, 1, 1]
        h = _H()
        _ck(lib().somar_amr_create(C.byref(h), len(boxes_per_level), _ia(domain_lo), _ia(domain_hi),
                                   _ia([int(bool(x)) for x in periodic]), _da(dx0), _ia(bc), _ia(rr), _ia(nb), _ia(flat),
                                   _ia(own), float(alpha), float(beta), C.byref(self._p), comm))
        self._amr = h
        self.levels = []
        for l in range(len(boxes_per_level)):
            v = AMRPressureSolver()
            lh = _H()
            _ck(lib().somar_amr_level(h, l, C.byref(lh)))
            v._h, v._borrowed = lh, True
            n = C.c_int()
            _ck(lib().somar_solver_num_local_patches(lh, C.byref(n)))
            v.num_local_patches = n.value
            self.levels.append(v)

    def solveAMR(self, lmax, lbase, zeroPhi=True, forceHomogeneous=False):
        """MappedAMRMultiGrid::solve on the levels' resident PHI / RHS."""
        st = Stats()
        _ck(lib().somar_amr_solve(self._amr, lmax, lbase, int(zeroPhi), int(forceHomogeneous), C.byref(st)))
        return self._stats(st)

    # -- level heat integrators on a level of the hierarchy (AMRParabolic) -------------------------------------------
    def setAlphaAndBetaAMR(self, a, b):
        _ck(lib().somar_amr_set_alpha_beta(self._amr, a, b))

    def heatStepAMR(self, level, scheme, dt, zeroPhi=True, oldTime=0.0, crseOldTime=0.0, crseNewTime=0.0):
        st = Stats()
        _ck(lib().somar_amr_heat_step(self._amr, level, scheme, dt, int(zeroPhi), oldTime, crseOldTime, crseNewTime,
                                      C.byref(st)))
        return self._stats(st)

    def tgaStepAMR(self, lmax, lbase, dt):
        """MappedAMRTGA::oneStep over levels lbase..lmax (PHI = phiNew, HEAT_OLD = phiOld, HEAT_SRC = source)"""
        st = Stats()
        _ck(lib().somar_amr_tga_step(self._amr, lmax, lbase, dt, C.byref(st)))
        return self._stats(st)

    # -- AMRLepticSolver (AMRPressureSolver::s_useAMRLepticSolver) ------------------------------------------------
    def enableLeptic(self, params=None, baseFromRestricted=False):
        """One leptic level solver per level (AMRLepticSolver::init); params: LepticParams or None for the defaults.
        baseFromRestricted=True is NOT the reference (see include/somar_amd.h)."""
        if params is None:
            params = LepticParams()
            _ck(lib().somar_leptic_params_default(C.byref(params)))
        _ck(lib().somar_amr_enable_leptic(self._amr, C.byref(params), int(baseFromRestricted)))

    def solveAMRLeptic(self, lmax, lbase, zeroPhi=True, forceHomogeneous=False):
        st = Stats()
        _ck(lib().somar_amr_solve_leptic(self._amr, lmax, lbase, int(zeroPhi), int(forceHomogeneous), C.byref(st)))
        return self._stats(st)

    def lepticStats(self, level):
        st = LepticStats()
        _ck(lib().somar_amr_leptic_stats(self._amr, level, C.byref(st)))
        return {"exitStatus": st.exit_status, "orders": st.orders, "horizSolves": st.horiz_solves,
                "usedFullSolver": bool(st.used_full_solver), "resNorms": [st.res_norms[i] for i in range(st.nres)]}

    def solveAMRHost(self, phi, rhs, lmin, lmax, zeroPhi=True, forceHomogeneous=False, phi_ghost=(1, 1, 1),
                     rhs_ghost=(0, 0, 0)):
        """AMRPressureSolver::solve(Vector<LevelData*> phi, rhs, lmin, lmax) (AMRPressureSolver.cpp:494-561) on host data:
        phi[l] / rhs[l] = list (one per local patch of level l) of Fortran-ordered arrays, or None for levels outside
        [lmin-1, lmax]; phi is updated in place."""
        nl = len(self.levels)
        keep = []

        def table(v):
            rows = (C.POINTER(_PD) * nl)()
            for l in range(nl):
                if v[l] is None:
                    rows[l] = None
                    continue
                assert len(v[l]) == self.levels[l].num_local_patches
                row = (_PD * max(len(v[l]), 1))(*[_dp(a) for a in v[l]])
                keep.append(row)
                rows[l] = row
            return rows

        st = Stats()
        # the reference's call order: solve(phi, rhs, a_lmax, a_lmin, ...) (AMRPressureSolver.cpp:529-534)
        _ck(lib().somar_amr_solve_host(self._amr, table(phi), _ia(phi_ghost), table(rhs), _ia(rhs_ghost), lmax, lmin,
                                       int(zeroPhi), int(forceHomogeneous), C.byref(st)))
        return self._stats(st)

    def levelProjectAMR(self, level, centring, dt, zeroPressure=True, forceHomogeneous=False, wall=True):
        """centring 0: the level's uploadVel'ed MAC velocity, 1: its uploadCCVel'ed cell-centred velocity (in place)"""
        st = Stats()
        _ck(lib().somar_amr_level_project(self._amr, level, int(centring), float(dt), int(zeroPressure),
                                          int(forceHomogeneous), int(wall), C.byref(st)))
        return self._stats(st)

    def setInspector(self, fn):
        """fn(kind, iter, lmin, lmax) -- kind 0: every level's F_RES holds uberResidual (before V-cycle iter), kind 1: F_CORR
        holds uberCorrection (after it).  MappedAMRMultiGridInspector, MappedAMRMultiGrid.H:260-298.  None removes it."""
        if fn is None:
            self._inspector = None
            _ck(lib().somar_amr_set_inspector(self._amr, None, None))
            return
        proto = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int)
        self._inspector = proto(lambda user, kind, it, l0, l1: fn(kind, it, l0, l1))   # keep the thunk alive
        _ck(lib().somar_amr_set_inspector(self._amr, C.cast(self._inspector, C.c_void_p), None))

    def projectAMRCC(self, lmin, lmax, dt, zeroPressure=False, forceHomogeneous=False, wall=True):
        """AMRCCProjector / BaseProjector<FArrayBox>::project(lmin, lmax) on the levels' uploadCCVel'ed velocities (in place)"""
        st = Stats()
        _ck(lib().somar_amr_cc_project(self._amr, lmin, lmax, float(dt), int(zeroPressure), int(forceHomogeneous), int(wall),
                                       C.byref(st)))
        return self._stats(st)

    def compDivergenceCC(self, level, lmax, out_field=F_RHS, wall=True):
        _ck(lib().somar_amr_comp_divergence_cc(self._amr, level, lmax, out_field, int(wall)))

    def compGradCorrectCC(self, level, lmax, phi_field, dt):
        _ck(lib().somar_amr_comp_grad_correct_cc(self._amr, level, lmax, phi_field, float(dt)))

    def averageDownCCVel(self, level):
        _ck(lib().somar_amr_average_down_ccvel(self._amr, level))

    def interpCF(self, level, fine_field=F_PHI, coarse_field=F_PHI):
        _ck(lib().somar_amr_interp_cf(self._amr, level, fine_field, coarse_field))

    def residualLevel(self, lmax, lbase, ilev, res_field=F_RES, phi_field=F_PHI, rhs_field=F_RHS):
        _ck(lib().somar_amr_residual_level(self._amr, lmax, lbase, ilev, res_field, phi_field, rhs_field))

    def zeroCovered(self, level, field):
        _ck(lib().somar_amr_zero_covered(self._amr, level, field))

    def vcycleAMR(self, lmax, lbase):
        _ck(lib().somar_amr_vcycle(self._amr, lmax, lbase))

    def patch_box(self, patch, depth=0):
        b = (C.c_int * 6)()
        g = C.c_int()
        _ck(lib().somar_solver_patch_box(self._h, depth, patch, b, C.byref(g)))
        return tuple(b[0:3]), tuple(b[3:6]), g.value

    def setMetricOrtho(self, patch, jg0, jg1, jg2, jinv):
        _ck(lib().somar_solver_set_metric_ortho(self._h, patch, _dp(jg0), _dp(jg1), _dp(jg2) if jg2 is not None else None,
                                                _dp(jinv)))

    def setBCValues(self, values6):
        """values of the Dirichlet sides {loX,hiX,loY,hiY,loZ,hiZ} (before finalize)"""
        _ck(lib().somar_solver_set_bc_values(self._h, _da(values6)))

    def heatFlux(self, dir, patch):
        """a_flux of the last heat step(s) on faces(valid, dir) of one local patch"""
        lo, hi, _ = self.patch_box(patch)
        shp = [h - l + 1 + (1 if d == dir else 0) for d, (l, h) in enumerate(zip(lo, hi))]
        out = np.zeros(shp, order="F")
        _ck(lib().somar_heat_flux_download(self._h, dir, patch, _dp(out)))
        return out

    def setMetricUniform(self, jgxx, jgyy, jgzz, jinv):
        """CartesianMap's constants into every local patch, on the device (no host arrays)"""
        _ck(lib().somar_solver_set_metric_uniform(self._h, _da([jgxx, jgyy, jgzz, jinv])))

    def setMetricMap(self, kind, L=(0.0, 0.0, 0.0), depth=None, depth_lo=(0, 0)):
        """SOMAR_MAP_CYLINDRICAL (1) / SOMAR_MAP_BATHYMETRIC (2) evaluated on the device into every local patch; depth:
        2-D array of NODAL depths, depth[i - depth_lo[0], j - depth_lo[1]]"""
        if depth is None:
            _ck(lib().somar_solver_set_metric_map(self._h, int(kind), _da(list(L)), None, None, None))
            return
        d = np.asfortranarray(depth, dtype=np.float64)
        lo = (C.c_int * 2)(int(depth_lo[0]), int(depth_lo[1]))
        n = (C.c_int * 2)(int(d.shape[0]), int(d.shape[1]))
        _ck(lib().somar_solver_set_metric_map(self._h, int(kind), _da(list(L)), _dp(d), lo, n))

    def setMetricFull(self, patch, jg0, jg1, jg2, jinv):
        """jgD: array (faces(valid, D) shape + (SpaceDim,)), Fortran order = component slowest; jg2 = None in 2-D"""
        _ck(lib().somar_solver_set_metric_full(self._h, patch, _dp(jg0), _dp(jg1), _dp(jg2) if jg2 is not None else None,
                                               _dp(jinv)))

    def finalize(self):
        if getattr(self, "_amr", None) is not None:
            _ck(lib().somar_amr_finalize(self._amr))
        else:
            _ck(lib().somar_solver_finalize(self._h))

    def isDefined(self):
        return self._h is not None or getattr(self, "_amr", None) is not None

    def undefine(self):
        if getattr(self, "_amr", None) is not None:
            for v in self.levels:
                v._h = None
            _ck(lib().somar_amr_destroy(self._amr))
            self._amr, self.levels = None, []
        if self._h is not None:
            if not getattr(self, "_borrowed", False):
                _ck(lib().somar_solver_destroy(self._h))
            self._h = None

    def __del__(self):
        try:
            self.undefine()
        except Exception:
            pass

    # -- hierarchy inspection ---------------------------------------------------------------------
    def depth(self):
        d = C.c_int()
        _ck(lib().somar_solver_depth(self._h, C.byref(d)))
        return d.value

    def metricUniform(self, depth=0):
        """None, or (Jg^xx, Jg^yy, Jg^zz, Jinv) when this depth's metric was found constant at finalize"""
        f, c = C.c_int(), (C.c_double * 4)()
        _ck(lib().somar_solver_metric_uniform(self._h, depth, C.byref(f), c))
        return tuple(c) if f.value else None

    def mgRefRatios(self):
        out = []
        for d in range(self.depth() - 1):
            r = (C.c_int * 3)()
            _ck(lib().somar_solver_mg_ref_ratio(self._h, d, r))
            out.append(tuple(r))
        return out

    def zeroAvg(self, depth):
        f = C.c_int()
        _ck(lib().somar_solver_zero_avg(self._h, depth, C.byref(f)))
        return bool(f.value)

    def levelInfo(self, depth):
        dom, dx = (C.c_int * 6)(), (C.c_double * 3)()
        cells, elems = C.c_longlong(), C.c_longlong()
        _ck(lib().somar_solver_level_info(self._h, depth, dom, dx, C.byref(cells), C.byref(elems)))
        return {"domain": (tuple(dom[0:3]), tuple(dom[3:6])), "dx": tuple(dx), "cells": cells.value,
                "field_elems": elems.value}

    # -- fields -------------------------------------------------------------------------------------
    def upload(self, field, patch, host, ghost):
        _ck(lib().somar_field_upload(self._h, field, patch, _dp(host), _ia(ghost)))

    def download(self, field, patch, ghost, depth=None):
        depth = field >> 8 if depth is None else depth
        lo, hi, _ = self.patch_box(patch, depth)
        shape = tuple(h - l + 1 + 2 * g for l, h, g in zip(lo, hi, ghost))
        out = np.zeros(shape, dtype=np.float64, order="F")
        _ck(lib().somar_field_download(self._h, field, patch, _dp(out), _ia(ghost)))
        return out

    def setVal(self, field, value):
        _ck(lib().somar_field_set(self._h, field, float(value)))

    def fillHash(self, field, seed):
        _ck(lib().somar_field_fill_hash(self._h, field, int(seed)))

    def removeMean(self, field):
        _ck(lib().somar_field_remove_mean(self._h, field))

    def norm(self, field, order):
        v = C.c_double()
        _ck(lib().somar_field_norm(self._h, field, order, C.byref(v)))
        return v.value

    def dotProduct(self, a, b):
        v = C.c_double()
        _ck(lib().somar_field_dot(self._h, a, b, C.byref(v)))
        return v.value

    # -- solve (AMRPressureSolver::solve, projection/AMRPressureSolver.cpp:494-561) -----------------------
    def solve(self, phi, rhs, lmin=0, lmax=0, zeroPhi=True, forceHomogeneous=False, phi_ghost=(1, 1, 1),
              rhs_ghost=(0, 0, 0)):
        """phi, rhs: lists (one per local patch) of Fortran-ordered float64 arrays; phi is updated in place."""
        n = self.num_local_patches
        assert len(phi) == n and len(rhs) == n
        P = (_PD * n)(*[_dp(a) for a in phi])
        R = (_PD * n)(*[_dp(a) for a in rhs])
        st = Stats()
        # note the reference's call order solve(phi, rhs, a_lmax, a_lmin, ...) (AMRPressureSolver.cpp:529-534)
        _ck(lib().somar_solver_solve_host(self._h, P, _ia(phi_ghost), R, _ia(rhs_ghost), lmax, lmin, int(zeroPhi),
                                          int(forceHomogeneous), C.byref(st)))
        return self._stats(st)

    def solveResident(self, zeroPhi=True, forceHomogeneous=False):
        st = Stats()
        _ck(lib().somar_solver_solve(self._h, int(zeroPhi), int(forceHomogeneous), C.byref(st)))
        return self._stats(st)

    def _stats(self, st):
        self.exitStatus = st.exit_status
        self.stats = {"iters": st.iters, "exitStatus": st.exit_status, "status": st.status,
                      "bottom_iters": st.bottom_iters, "bottom_exit": st.bottom_exit,
                      "initial_rnorm": st.initial_rnorm, "final_rnorm": st.final_rnorm,
                      "history": [st.history[i] for i in range(st.nhistory)]}
        if st.status == 1:
            raise SomarError("kaboom: solver seems to have blown up (MappedAMRMultiGrid.H:1134-1137)")
        if st.status == 2:
            raise SomarError("MappedAMRMultiGrid solver blew up (MappedAMRMultiGrid.H:1141-1145)")
        return self.stats

    # -- level-operator pieces on resident fields ---------------------------------------------------
    def relax(self, depth, phi_field, rhs_field, iters):
        _ck(lib().somar_level_relax(self._h, depth, phi_field, rhs_field, iters))

    def residual(self, depth, out_field, phi_field, rhs_field):
        _ck(lib().somar_level_residual(self._h, depth, out_field, phi_field, rhs_field))

    def applyOp(self, depth, out_field, phi_field):
        _ck(lib().somar_level_apply_op(self._h, depth, out_field, phi_field))

    def applyOpBC(self, out_field, phi_field, homogeneous):
        _ck(lib().somar_level_apply_op_bc(self._h, out_field, phi_field, int(homogeneous)))

    def residualBC(self, out_field, phi_field, rhs_field, homogeneous):
        _ck(lib().somar_level_residual_bc(self._h, out_field, phi_field, rhs_field, int(homogeneous)))

    def restrictResidual(self, depth, coarse_res_field, phi_field, rhs_field):
        _ck(lib().somar_level_restrict_residual(self._h, depth, coarse_res_field, phi_field, rhs_field))

    def prolongIncrement(self, depth, phi_field, coarse_corr_field):
        _ck(lib().somar_level_prolong_increment(self._h, depth, phi_field, coarse_corr_field))

    def preCond(self, depth, phi_field, rhs_field):
        _ck(lib().somar_level_precond(self._h, depth, phi_field, rhs_field))

    def vcycle(self, corr_field=F_CORR, res_field=F_RES):
        _ck(lib().somar_vcycle(self._h, corr_field, res_field))

    def miniVCycle(self, corr_field=F_CORR, res_field=F_RES):
        _ck(lib().somar_mini_vcycle(self._h, corr_field, res_field))

    def vcycleFromZero(self, corr_field=F_CORR, res_field=F_RES):
        """oneCycle on a correction taken to be zero (contents of corr_field are ignored and overwritten)"""
        _ck(lib().somar_vcycle_from_zero(self._h, corr_field, res_field))

    def bottomSolve(self, phi_field, rhs_field):
        it, ex = C.c_int(), C.c_int()
        _ck(lib().somar_bottom_solve(self._h, phi_field, rhs_field, C.byref(it), C.byref(ex)))
        return it.value, ex.value

    def counters(self):
        """{overlapped_sweeps, ghost_programs_one_launch, ghost_programs_staged, bottom_solves} so far"""
        c = (C.c_longlong * 4)()
        _ck(lib().somar_solver_counters(self._h, c))
        return {"overlapped_sweeps": c[0], "ghost_programs_one_launch": c[1], "ghost_programs_staged": c[2], "bottom_solves": c[3]}

    def fused19Sweeps(self):
        """19-point LevelGSRB sweeps that ran as one red+black launch plus a shell pass (levels of large boxes)"""
        n = C.c_longlong()
        _ck(lib().somar_solver_fused19_sweeps(self._h, C.byref(n)))
        return n.value

    def bottomKind(self):
        """how the last bottom solve ran: 0 launch by launch, 1 one single-workgroup launch, 2 one persistent launch, a workgroup per box"""
        k = C.c_int()
        _ck(lib().somar_bottom_kind(self._h, C.byref(k)))
        return k.value

    # -- MAC level projection (LevelMACProjector / BaseProjector::project, velocity in flux form) -------
    def uploadVel(self, d, patch, host):
        _ck(lib().somar_vel_upload(self._h, d, patch, _dp(host)))

    def downloadVel(self, d, patch):
        lo, hi, _ = self.patch_box(patch)
        shape = [h - l + 1 for l, h in zip(lo, hi)]
        shape[d] += 1
        out = np.zeros(shape, dtype=np.float64, order="F")
        _ck(lib().somar_vel_download(self._h, d, patch, _dp(out)))
        return out

    def velWallBC(self):
        """zero wall-normal faces of the resident MAC velocity (uStarFuncBC, solid walls; or what setVelBC installed)"""
        _ck(lib().somar_vel_wall_bc(self._h))

    def setVelBC(self, kind, value):
        """BasicVelocityBCGhostClass's inflow / outflow sides: kind[2*dir+side] 0 wall, 1 prescribed value, 2 outflow"""
        k = (C.c_int * 6)(*[int(x) for x in kind])
        v = (C.c_double * 6)(*[float(x) for x in value])
        _ck(lib().somar_solver_set_vel_bc(self._h, k, v))

    def divergenceMAC(self, out_field, dt):
        _ck(lib().somar_level_divergence_mac(self._h, out_field, float(dt)))

    def macCorrect(self, phi_field, dt):
        _ck(lib().somar_level_mac_correct(self._h, phi_field, float(dt)))

    def levelProject(self, vel, dt, zeroPressure=True, forceHomogeneous=False):
        """vel: [u0, u1, u2], each a list (per local patch) of F-ordered face arrays; projected in place."""
        n = self.num_local_patches
        U = [( _PD * n)(*[_dp(a) for a in vel[d]]) for d in range(3)]
        st = Stats()
        _ck(lib().somar_mac_project_host(self._h, U[0], U[1], U[2], float(dt), int(zeroPressure),
                                         int(forceHomogeneous), C.byref(st)))
        return self._stats(st)

    # -- viscous / diffusive Helmholtz solves (MappedBaseLevelHeatSolver and its BE / CN integrators) --
    # -- LevelGeometry::multByJ / divByJ on the resident velocities (a_velIsFlux = false, BaseProjectorI.H:235-241, 291-297)
    def setCCJ(self, patch, J, Jinv, ghost):
        _ck(lib().somar_solver_set_cc_j(self._h, patch, _dp(J), _dp(Jinv), _ia(ghost)))

    def setFaceJ(self, d, patch, J, Jinv):
        _ck(lib().somar_solver_set_face_j(self._h, d, patch, _dp(J), _dp(Jinv)))

    def multByJ(self, centring):
        _ck(lib().somar_vel_mult_by_j(self._h, centring))

    def divByJ(self, centring):
        _ck(lib().somar_vel_div_by_j(self._h, centring))

    def setAlphaAndBeta(self, a, b):
        _ck(lib().somar_solver_set_alpha_beta(self._h, float(a), float(b)))

    def heatStep(self, scheme, dt, zeroPhi=True):
        """scheme 0 backward Euler, 1 Crank-Nicolson; phiOld in F_HEAT_OLD, src in F_HEAT_SRC, result in F_PHI"""
        st = Stats()
        _ck(lib().somar_heat_step(self._h, int(scheme), float(dt), int(zeroPhi), C.byref(st)))
        return self._stats(st)

    # -- cell-centred level projection (LevelCCProjector, velocity in flux form, SpaceDim comps + ghosts) --
    def uploadCCVel(self, patch, host, ghost):
        _ck(lib().somar_ccvel_upload(self._h, patch, _dp(host), _ia(ghost)))

    def downloadCCVel(self, patch, host, ghost):
        """writes the valid cells of `host` (F-ordered (nx+2g, ny+2g, nz+2g, SpaceDim) array) in place"""
        _ck(lib().somar_ccvel_download(self._h, patch, _dp(host), _ia(ghost)))

    def divergenceCC(self, out_field, dt, wall=True):
        _ck(lib().somar_level_divergence_cc(self._h, out_field, float(dt), int(wall)))

    def ccCorrect(self, phi_field, dt):
        _ck(lib().somar_level_cc_correct(self._h, phi_field, float(dt)))

    def levelProjectCC(self, vel, ghost, dt, zeroPressure=True, forceHomogeneous=False, wall=True):
        """vel: list (per local patch) of F-ordered arrays (valid + ghost, SpaceDim comps last); projected in place."""
        n = self.num_local_patches
        U = (_PD * n)(*[_dp(a) for a in vel])
        st = Stats()
        _ck(lib().somar_cc_project_host(self._h, U, _ia(ghost), float(dt), int(zeroPressure), int(forceHomogeneous),
                                        int(wall), C.byref(st)))
        return self._stats(st)

    def sync(self):
        _ck(lib().somar_sync(self._h))

    def profileEnable(self, on=True):
        _ck(lib().somar_profile_enable(self._h, int(on)))

    def profileGet(self, kernel):
        n, ms = C.c_int(), C.c_double()
        _ck(lib().somar_profile_get(self._h, kernel, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def timerStart(self):
        _ck(lib().somar_timer_start(self._h))

    def timerStop(self):
        ms = C.c_double()
        _ck(lib().somar_timer_stop(self._h, C.byref(ms)))
        return ms.value


def k_gsrbiter3dortho(phi, phi_lo, rhs, rhs_lo, jg, jg_lo, jinv, jinv_lo, lapdiag, lapdiag_lo, region, dx, alpha, beta,
                      redBlack):
    """GSRBITER3DORTHO through the Fortran-shaped C entry (RelaxationMethods/GSRBF_F.H:216-232): arrays are Fortran-ordered
    numpy arrays (phi / rhs may carry a trailing component axis), *_lo their low corners, region = (lo, hi); phi in place."""
    def ip(v):
        return [C.byref(C.c_int(int(x))) for x in v]

    def fab(a, lo, ncomp):
        shp = a.shape[:3]
        hi = [l + n - 1 for l, n in zip(lo, shp)]
        out = [_dp(a)] + ip(lo) + ip(hi)
        if ncomp:
            out.append(C.byref(C.c_int(a.shape[3] if a.ndim == 4 else 1)))
        return out

    args = fab(phi, phi_lo, True) + fab(rhs, rhs_lo, True)
    for d in range(3):
        args += fab(jg[d], jg_lo[d], False)
    args += fab(jinv, jinv_lo, False) + fab(lapdiag, lapdiag_lo, False)
    args += ip(region[0]) + ip(region[1])
    args += [_da(dx), C.byref(C.c_double(alpha)), C.byref(C.c_double(beta)), C.byref(C.c_int(redBlack))]
    _ck(lib().somar_k_gsrbiter3dortho(*args))


def _fort_ip(v):
    return [C.byref(C.c_int(int(x))) for x in v]


def _fort_fab(a, lo, ncomp):
    """CHFp_FRA / CHFp_FRA1 argument group of a Fortran-ordered numpy array placed at lo"""
    shp = a.shape[:3]
    hi = [l + n - 1 for l, n in zip(lo, shp)]
    out = [_dp(a)] + _fort_ip(lo) + _fort_ip(hi)
    if ncomp:
        out.append(C.byref(C.c_int(a.shape[3] if a.ndim == 4 else 1)))
    return out


def k_fillmappedlapdiag3d(lapdiag, lap_lo, jg, jg_lo, jinv, jinv_lo, region, dx):
    """FILLMAPPEDLAPDIAG3D through the Fortran-shaped C entry; jg[d]: (faces..., 3) Fortran-ordered FluxBox FABs"""
    args = _fort_fab(lapdiag, lap_lo, False)
    for d in range(3):
        args += _fort_fab(jg[d], jg_lo[d], True)
    args += _fort_fab(jinv, jinv_lo, False) + _fort_ip(region[0]) + _fort_ip(region[1]) + [_da(dx)]
    _ck(lib().somar_k_fillmappedlapdiag3d(*args))


def k_mappedaverage2(coarse, coarse_lo, fine, fine_lo, fjinv, fjinv_lo, box, refRatio):
    """MAPPEDAVERAGE2 through the Fortran-shaped C entry; coarse is written on `box`"""
    args = _fort_fab(coarse, coarse_lo, True) + _fort_fab(fine, fine_lo, True) + _fort_fab(fjinv, fjinv_lo, False)
    args += _fort_ip(box[0]) + _fort_ip(box[1]) + [(C.c_int * 3)(*[int(x) for x in refRatio])]
    args += _fort_ip((0, 0, 0)) + _fort_ip([r - 1 for r in refRatio])
    _ck(lib().somar_k_mappedaverage2(*args))


def altered_jgup(nsq_fc, dximu_dz, dxinu_dz, gup, J, dt_theta, coriolis_f, hjac=None):
    """AlteredMetric::fill_Jgup on flat float64 arrays; hjac = (ix, jy, iy, jx) when mu != nu."""
    arrs = [np.ascontiguousarray(a, dtype=np.float64).ravel() for a in (nsq_fc, dximu_dz, dxinu_dz, gup, J)]
    n = arrs[0].size
    assert all(a.size == n for a in arrs)
    h = [None] * 4
    if hjac is not None:
        h = [np.ascontiguousarray(a, dtype=np.float64).ravel() for a in hjac]
        assert all(a.size == n for a in h)
    out = np.empty(n)
    p = lambda a: a.ctypes.data_as(_PD) if a is not None else None   # noqa: E731
    _ck(lib().somar_altered_jgup(n, p(out), p(arrs[0]), p(arrs[1]), p(arrs[2]), p(h[0]), p(h[1]), p(h[2]), p(h[3]),
                                 p(arrs[3]), p(arrs[4]), float(dt_theta), float(coriolis_f)))
    return out


class LevelLepticSolver:
    """Mirror of calculus/LepticSolver/LevelLepticSolver.H: define(op) + solve(phi, rhs).  `level` is the level's own
    operator (an AMRPressureSolver view: metric upload, field I/O); `params` the somar_leptic_params_t block."""

    EXIT_NAMES = {-1: "NONE", 0: "CONVERGE", 1: "ITER", 2: "HANG", 3: "DIVERGE", 4: "KABOOM"}

    def __init__(self):
        self.params = LepticParams()
        _ck(lib().somar_leptic_params_default(C.byref(self.params)))
        self.level_params = Params()
        _ck(lib().somar_params_default(C.byref(self.level_params)))
        self._h = None
        self.level = None

    def define(self, domain_lo, domain_hi, periodic, dx, boxes, bc_type=None, owner=None, alpha=0.0, beta=1.0,
               comm=None):
        assert self._h is None, "already defined"
        bc = bc_type if bc_type is not None else [BC_NEUM] * 6
        flat = [int(x) for lo, hi in boxes for x in list(lo) + list(hi)]
        own = _ia(owner) if owner is not None else None
        h = _H()
        _ck(lib().somar_leptic_create(C.byref(h), _ia(domain_lo), _ia(domain_hi), _ia([int(bool(x)) for x in periodic]),
                                      _da(dx), _ia(bc), len(boxes), _ia(flat), own, float(alpha), float(beta),
                                      C.byref(self.level_params), C.byref(self.params), comm))
        self._h = h
        v = AMRPressureSolver()
        lh = _H()
        _ck(lib().somar_leptic_level(h, C.byref(lh)))
        v._h, v._borrowed = lh, True
        n = C.c_int()
        _ck(lib().somar_solver_num_local_patches(lh, C.byref(n)))
        v.num_local_patches = n.value
        self.level = v
        self.vert = self._view(1)
        try:
            self.horiz = self._view(2)
        except SomarError:
            self.horiz = None   # no column is Neumann-Neumann: no flat problem (gatherVerticalBCTypes)

    def _view(self, which):
        v = AMRPressureSolver()
        lh = _H()
        _ck(lib().somar_leptic_part(self._h, which, C.byref(lh)))
        v._h, v._borrowed = lh, True
        n = C.c_int()
        _ck(lib().somar_solver_num_local_patches(lh, C.byref(n)))
        v.num_local_patches = n.value
        return v

    def finalize(self):
        _ck(lib().somar_leptic_finalize(self._h))

    def solve(self, homogeneous=False):
        """phi += leptic correction on the level's resident PHI / RHS"""
        st = LepticStats()
        _ck(lib().somar_leptic_solve(self._h, int(homogeneous), C.byref(st)))
        self.exitStatus = st.exit_status
        self.stats = {"exitStatus": st.exit_status, "orders": st.orders, "horizSolves": st.horiz_solves,
                      "usedFullSolver": bool(st.used_full_solver),
                      "resNorms": [st.res_norms[i] for i in range(st.nres)],
                      "horiz": {"iters": st.horiz.iters, "exitStatus": st.horiz.exit_status,
                                "history": [st.horiz.history[i] for i in range(st.horiz.nhistory)]},
                      "full": {"iters": st.full.iters, "exitStatus": st.full.exit_status,
                               "history": [st.full.history[i] for i in range(st.full.nhistory)]}}
        return self.stats

    def undefine(self):
        if self._h is not None:
            for v in (self.level, getattr(self, "vert", None), getattr(self, "horiz", None)):
                if v is not None:
                    v._h = None
            _ck(lib().somar_leptic_destroy(self._h))
            self._h, self.level = None, None

    def __del__(self):
        try:
            self.undefine()
        except Exception:
            pass


def plan_exchange(domain_lo, domain_hi, periodic, boxes, owner, rank, ghost=2, max_items=4096):
    """-> (local, send, recv): lists of dicts {src, dst, src_lo, dst_lo, n, peer} (pure host call, no GPU)."""
    flat = []
    for lo, hi in boxes:
        flat += list(lo) + list(hi)
    bufs = [(C.c_int * (12 * max_items))() for _ in range(3)]
    ns = [C.c_int() for _ in range(3)]
    _ck(lib().somar_plan_exchange(_ia(domain_lo), _ia(domain_hi), _ia([int(bool(p)) for p in periodic]), len(boxes),
                                  _ia(flat), _ia(owner), rank, ghost, max_items, C.byref(ns[0]), bufs[0],
                                  C.byref(ns[1]), bufs[1], C.byref(ns[2]), bufs[2]))
    out = []
    for n, b in zip(ns, bufs):
        items = []
        for i in range(n.value):
            o = b[12 * i:12 * i + 12]
            items.append({"src": o[0], "dst": o[1], "src_lo": tuple(o[2:5]), "dst_lo": tuple(o[5:8]), "n": tuple(o[8:11]),
                          "peer": o[11]})
        out.append(items)
    return tuple(out)


def ledge_bathymetry(x, y=None, order=3, hl=1.0, hr=0.5, xl=0.0, xr=1.0):
    """LedgeMap::fill_bathymetry at the nodes' Cartesian coordinates (y given: the 3-D build's Gaussian bump)"""
    xa = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(xa)
    ya = None if y is None else np.ascontiguousarray(np.broadcast_to(y, xa.shape), dtype=np.float64)
    _ck(lib().somar_bathymetry_ledge(out.ctypes.data_as(_PD), xa.size, xa.ctypes.data_as(_PD),
                                     None if ya is None else ya.ctypes.data_as(_PD), int(order), hl, hr, xl, xr))
    return out


def beam_generator_bathymetry(x, Lx, angle):
    """BeamGeneratorMap::fill_bathymetry (elevation of the smoothed ridge of critical slope `angle`, radians) at the nodes' x"""
    xa = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(xa)
    _ck(lib().somar_bathymetry_beam_generator(out.ctypes.data_as(_PD), xa.size, xa.ctypes.data_as(_PD), float(Lx), float(angle)))
    return out


def dem_cubic_spline(x, xd, fd):
    """DEMMap's 2-D path: the natural cubic spline through (xd, fd) (CubicSpline::solve / interp) at x"""
    xa = np.ascontiguousarray(x, dtype=np.float64)
    xda, fda = np.ascontiguousarray(xd, dtype=np.float64), np.ascontiguousarray(fd, dtype=np.float64)
    out = np.empty_like(xa)
    _ck(lib().somar_dem_cubic_spline(out.ctypes.data_as(_PD), xa.size, xa.ctypes.data_as(_PD), xda.size, xda.ctypes.data_as(_PD),
                                     fda.ctypes.data_as(_PD)))
    return out


def dem_bilinear(x, y, xd, yd, fd, hermite=False):
    """DEMMap's 3-D path: BilinearInterp2D (interpOrder 0) or, hermite=True, Create_Level_DEM_3D's difference tables + HermiteInterp2D
    (interpOrder > 0) of fd[i, j] on the tensor grid xd x yd at the points (x, y)"""
    xa = np.ascontiguousarray(x, dtype=np.float64)
    ya = np.ascontiguousarray(np.broadcast_to(y, xa.shape), dtype=np.float64)
    xda, yda = np.ascontiguousarray(xd, dtype=np.float64), np.ascontiguousarray(yd, dtype=np.float64)
    fda = np.asfortranarray(fd, dtype=np.float64)          # i fastest
    assert fda.shape == (xda.size, yda.size)
    out = np.empty_like(xa)
    fn = lib().somar_dem_hermite if hermite else lib().somar_dem_bilinear
    _ck(fn(out.ctypes.data_as(_PD), xa.size, xa.ctypes.data_as(_PD), ya.ctypes.data_as(_PD), xda.size,
           yda.size, xda.ctypes.data_as(_PD), yda.ctypes.data_as(_PD), fda.ctypes.data_as(_PD)))
    return out


def host_random_field(shape, seed, lo=-1.0, hi=1.0):
    """std::mt19937_64(seed) + uniform_real_distribution(lo, hi), drawn in Fortran order over `shape`: the
    benchmark configurations' random fields, generated on the host"""
    a = np.empty(shape, order="F")
    _ck(lib().somar_host_fill_mt19937_64(a.ctypes.data_as(_PD), a.size, seed, lo, hi))
    return a


def last_history():
    """the complete residual history of this thread's last solve (somar_stats_t.history holds at most 64 entries)"""
    n = C.c_int()
    _ck(lib().somar_last_history(None, 0, C.byref(n)))
    out = np.zeros(max(n.value, 1))
    _ck(lib().somar_last_history(out.ctypes.data_as(_PD), n.value, C.byref(n)))
    return out[:n.value]


def device_count():
    n = C.c_int()
    _ck(lib().somar_device_count(C.byref(n)))
    return n.value


def comm_unique_id():
    buf = (C.c_ubyte * COMM_ID_BYTES)()
    _ck(lib().somar_comm_unique_id(buf))
    return bytes(buf)


def comm_create(id_bytes, rank, nranks, device):
    h = _H()
    buf = (C.c_ubyte * COMM_ID_BYTES)(*id_bytes)
    _ck(lib().somar_comm_create(C.byref(h), buf, rank, nranks, device))
    return h


def comm_create_shm(name, rank, nranks, outbox_bytes=64 << 20):
    h = _H()
    _ck(lib().somar_comm_create_shm(C.byref(h), name.encode(), rank, nranks, outbox_bytes))
    return h


def jgup_from_dxdxi(dxdxi, detJ, mu, scale=1.0):
    """GeoSourceInterface::fill_Jgup's generic algebra on the device.  dxdxi: array (n, 3, 3) with [i, rho, sigma] =
    dx^rho/dXi^sigma; detJ: (n,).  Returns (n, 3): scale * J g^{mu nu}, nu = 0..2."""
    d = np.ascontiguousarray(np.asarray(dxdxi, dtype=np.float64).reshape(-1, 9).T)   # component slowest
    J = np.ascontiguousarray(np.asarray(detJ, dtype=np.float64).ravel())
    n = J.size
    out = np.zeros((3, n))
    _ck(lib().somar_metric_jgup_from_dxdxi(n, mu, d.ctypes.data_as(_PD), J.ctypes.data_as(_PD), float(scale),
                                           out.ctypes.data_as(_PD)))
    return out.T.copy()


def stream_probe(kind, cells=512 ** 3, reps=10):
    """GB/s this device streams for a stream mix (0 copy, 1 read, 2 six reads + one write); diagnostics"""
    g = C.c_double()
    _ck(lib().somar_diag_stream_probe(kind, cells, reps, C.byref(g)))
    return g.value


def comm_selftest(h):
    _ck(lib().somar_comm_selftest(h))


def comm_destroy(h):
    _ck(lib().somar_comm_destroy(h))
