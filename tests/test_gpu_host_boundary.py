"""The drop-in boundary on caller-owned host data:
  * somar_amr_solve_host = AMREllipticSolver::solve(Vector<LevelData*>& phi, rhs, l_max, l_base, zeroPhi, forceHomogeneous)
    (calculus/AMRElliptic/AMREllipticSolver.H:33-48) for hierarchies: composite solve (l_base 0) and a level solve with
    coarse CF values (l_base = l_max = 1), against oracle/somar_amr.py -- same iterations, exit status, history to 1e-12,
    and the phi written back to the host arrays;
  * somar_k_gsrbiter3dortho = the Fortran kernel's own argument shapes (RelaxationMethods/GSRBF_F.H:216-232), box by box,
    bit for bit against the C restatement on the same host buffers."""
import ctypes as C

import numpy as np
import pytest

from helpers import make_amr_levels, make_gpu_amr, max_rel_diff, valid_of
from test_gpu_amr import LAYOUTS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def am(oracle):
    from oracle import somar_amr
    return somar_amr


def _host(ld, v):
    """the caller's FABs of one level: a list (one per LOCAL patch) of Fortran-ordered copies"""
    out = []
    for p in range(v.num_local_patches):
        _, _, gi = v.patch_box(p)
        out.append(np.asfortranarray(ld[gi].a[..., 0]).copy(order="F"))
    return out


@pytest.mark.parametrize("layout", [LAYOUTS[1], LAYOUTS[3]])
def test_composite_solve_on_host_data(oracle, am, layout):
    so = oracle
    periodic, ratios, boxes = layout
    fb = [[so.Box(lo, hi) for lo, hi in lev] for lev in boxes]
    levels = make_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.5), periodic, ratios, fb)
    comp = am.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab())
    gpu = make_gpu_amr(levels, ratios)
    try:
        lmax = len(levels) - 1
        phi = [so.random_field(L.grids, 5 + l, (1, 1, 1), L.domain.box) for l, L in enumerate(levels)]
        zero = [so.LevelData(L.grids, 1) for L in levels]
        rhs = [so.LevelData(L.grids, 1) for L in levels]
        comp.init(phi, zero, lmax, 0)
        comp.compute_amr_residual(rhs, phi, zero, lmax, 0, True)
        for r in rhs:
            so.ld_scale(r, -1.0)
        sol = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
        comp.solve(sol, rhs, lmax, 0)
        hphi = [[np.full(a.shape, 7.0, order="F") for a in _host(sol[l], v)] for l, v in enumerate(gpu.levels)]
        hrhs = [_host(rhs[l], v) for l, v in enumerate(gpu.levels)]
        st = gpu.solveAMRHost(hphi, hrhs, 0, lmax, zeroPhi=True)
        assert st["iters"] == comp.iters and st["exitStatus"] == comp.exitStatus
        np.testing.assert_allclose(st["history"], comp.history, rtol=1e-12, atol=0.0)
        for l in range(lmax + 1):
            got = [a[1:-1, 1:-1, 1:-1] for a in hphi[l]]
            assert max_rel_diff(got, valid_of(sol[l])) < 1e-8
    finally:
        gpu.undefine()


def test_level_solve_on_host_data_takes_cf_values_from_the_coarser_phi(oracle, am):
    so = oracle
    periodic, ratios, boxes = LAYOUTS[0]
    fb = [[so.Box(lo, hi) for lo, hi in lev] for lev in boxes]
    levels = make_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.5), periodic, ratios, fb)
    comp = am.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab())
    gpu = make_gpu_amr(levels, ratios)
    try:
        coarse = so.random_field(levels[0].grids, 11, (1, 1, 1), levels[0].domain.box)
        rhs1 = so.random_field(levels[1].grids, 12, (0, 0, 0), levels[1].domain.box)
        phi1 = so.LevelData(levels[1].grids, 1, (1, 1, 1))
        hphi = [_host(coarse, gpu.levels[0]), _host(phi1, gpu.levels[1])]
        hrhs = [None, _host(rhs1, gpu.levels[1])]
        comp.solve([coarse, phi1], [None, rhs1], 1, 1)
        st = gpu.solveAMRHost(hphi, hrhs, 1, 1, zeroPhi=True)
        assert st["iters"] == comp.iters and st["exitStatus"] == comp.exitStatus
        np.testing.assert_allclose(st["history"], comp.history, rtol=1e-12, atol=0.0)
        got = [a[1:-1, 1:-1, 1:-1] for a in hphi[1]]
        assert max_rel_diff(got, valid_of(phi1)) < 1e-8
        # a missing coarse phi is an error, not a silent zero
        from somar_amd import SomarError
        with pytest.raises(SomarError):
            gpu.solveAMRHost([None, hphi[1]], hrhs, 1, 1)
    finally:
        gpu.undefine()


@pytest.mark.parametrize("redBlack", [0, 1])
def test_fortran_shaped_gsrb_kernel_hook_bit_exact(oracle, redBlack):
    so = oracle
    from somar_amd import api as F
    rng = np.random.default_rng(3)
    lo, n = (3, -2, 5), (20, 13, 9)              # an arbitrary FAB placement; phi has one ghost layer
    plo = tuple(a - 1 for a in lo)
    phi = np.asfortranarray(rng.uniform(-1, 1, tuple(a + 2 for a in n)))
    rhs = np.asfortranarray(rng.uniform(-1, 1, n))
    jg = [np.asfortranarray(rng.uniform(0.5, 1.5, tuple(a + (d == q) for q, a in enumerate(n)))) for d in range(3)]
    jinv = np.asfortranarray(rng.uniform(0.5, 1.5, n))
    dx = (0.1, 0.07, 0.2)
    hi = tuple(a + b - 1 for a, b in zip(lo, n))
    phi_hi = tuple(a + 1 for a in hi)
    lapd = np.zeros(n, order="F")
    L = so.lib()
    iv = lambda v: (C.c_int * 3)(*v)      # noqa: E731
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))   # noqa: E731
    fhi = [tuple(h + (d == q) for q, h in enumerate(hi)) for d in range(3)]
    # lapDiag as the factory fills it (component a of FAB a): build 3-comp FABs for the oracle's FILLMAPPEDLAPDIAG3D
    jg3 = [np.zeros(jg[d].shape + (3,), order="F") for d in range(3)]
    for d in range(3):
        jg3[d][..., d] = jg[d]
    L.orc_fillmappedlapdiag3d(dp(lapd), iv(lo), iv(hi), dp(jg3[0]), iv(lo), iv(fhi[0]), dp(jg3[1]), iv(lo), iv(fhi[1]),
                              dp(jg3[2]), iv(lo), iv(fhi[2]), dp(jinv), iv(lo), iv(hi), iv(lo), iv(hi), (C.c_double * 3)(*dx))
    region = ((lo[0] + 2, lo[1], lo[2] + 1), (hi[0], hi[1] - 3, hi[2]))   # a sub-box, as boundary/interior splits produce
    want = phi.copy(order="F")
    L.orc_gsrbiter3dortho(dp(want), iv(plo), iv(phi_hi), 1, dp(rhs), iv(lo), iv(hi), dp(jg[0]), iv(lo), iv(fhi[0]),
                          dp(jg[1]), iv(lo), iv(fhi[1]), dp(jg[2]), iv(lo), iv(fhi[2]), dp(jinv), iv(lo), iv(hi),
                          dp(lapd), iv(lo), iv(hi), iv(region[0]), iv(region[1]), (C.c_double * 3)(*dx),
                          C.c_double(0.3), C.c_double(1.7), redBlack)
    got = phi.copy(order="F")
    F.k_gsrbiter3dortho(got, plo, rhs, lo, jg, [lo, lo, lo], jinv, lo, lapd, lo, region, dx, 0.3, 1.7, redBlack)
    assert not np.array_equal(got, phi)
    np.testing.assert_array_equal(got, want)


def test_fortran_shaped_lapdiag_and_average_hooks_bit_exact(oracle):
    """FILLMAPPEDLAPDIAG3D and MAPPEDAVERAGE2 with the Fortran exports' own argument shapes (MappedAMRPoissonOpF_F.H:139-146,
    MappedCoarseAverageF_F.H:111-117), on host FABs placed anywhere, against the C restatement on the same buffers"""
    so = oracle
    from somar_amd import api as F
    rng = np.random.default_rng(5)
    L = so.lib()
    iv = lambda v: (C.c_int * 3)(*v)      # noqa: E731
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))   # noqa: E731
    # lapDiag
    lo, n = (3, -2, 5), (20, 13, 9)
    hi = tuple(a + b - 1 for a, b in zip(lo, n))
    fhi = [tuple(h + (d == q) for q, h in enumerate(hi)) for d in range(3)]
    jg3 = [np.asfortranarray(rng.uniform(0.5, 1.5, tuple(a + (d == q) for q, a in enumerate(n)) + (3,))) for d in range(3)]
    jinv = np.asfortranarray(rng.uniform(0.5, 1.5, n))
    dx = (0.1, 0.07, 0.2)
    region = ((lo[0] + 2, lo[1], lo[2] + 1), (hi[0], hi[1] - 3, hi[2]))
    want = np.full(n, -7.0, order="F")
    L.orc_fillmappedlapdiag3d(dp(want), iv(lo), iv(hi), dp(jg3[0]), iv(lo), iv(fhi[0]), dp(jg3[1]), iv(lo), iv(fhi[1]),
                              dp(jg3[2]), iv(lo), iv(fhi[2]), dp(jinv), iv(lo), iv(hi), iv(region[0]), iv(region[1]),
                              (C.c_double * 3)(*dx))
    got = np.full(n, -7.0, order="F")
    F.k_fillmappedlapdiag3d(got, lo, jg3, [lo, lo, lo], jinv, lo, region, dx)
    np.testing.assert_array_equal(got, want)
    assert np.any(got != -7.0) and np.any(got == -7.0)          # written on the region only
    # J-weighted average, two components, semicoarsening ratio (2, 1, 2)
    r = (2, 1, 2)
    cbox = ((2, -1, 3), (9, 6, 6))
    flo = tuple(a * q for a, q in zip(cbox[0], r))
    fn = tuple((h - a + 1) * q for a, h, q in zip(cbox[0], cbox[1], r))
    fhi_ = tuple(a + b - 1 for a, b in zip(flo, fn))
    fine = np.asfortranarray(rng.uniform(-1, 1, fn + (2,)))
    fjinv = np.asfortranarray(rng.uniform(0.5, 1.5, fn))
    cn = tuple(h - a + 1 for a, h in zip(cbox[0], cbox[1]))
    want = np.zeros(cn + (2,), order="F")
    L.orc_mappedaverage2(dp(want), iv(cbox[0]), iv(cbox[1]), 2, dp(fine), iv(flo), iv(fhi_), dp(fjinv), iv(flo), iv(fhi_),
                         iv(cbox[0]), iv(cbox[1]), iv(r))
    got = np.zeros(cn + (2,), order="F")
    F.k_mappedaverage2(got, cbox[0], fine, flo, fjinv, flo, cbox, r)
    np.testing.assert_array_equal(got, want)


def test_history_longer_than_the_stats_block(oracle):
    """somar_stats_t.history holds SOMAR_MAX_HISTORY = 64 entries; an adapter whose AMRMG.imax is larger reads the whole history
    of its last solve through somar_last_history (imin = imax = 80 forces 80 V-cycles whatever the residual does)"""
    from helpers import make_problem
    from somar_amd import AMRPressureSolver, api
    so = oracle
    dom, grids, dx, Jgup, Jinv = make_problem(so, (16, 16, 16), 8, "cartesian", (False, False, False), (1.0, 1.0, 1.0))
    s = AMRPressureSolver()
    p = s._p
    s.setAMRMGParameters(80, 80, 1e-300, -1, p.num_smooth_precond, 2, 2, 2, p.precond_mode, 1, p.num_mg, p.hang, 0.0, 0)
    s.define(dom.box.lo, dom.box.hi, dom.periodic, dx, [(g.lo, g.hi) for g in grids])
    for q in range(s.num_local_patches):
        _, _, gi = s.patch_box(q)
        s.setMetricOrtho(q, *[np.asfortranarray(Jgup[gi][d].a[..., d]) for d in range(3)], np.asfortranarray(Jinv[gi].a[..., 0]))
    s.finalize()
    s.fillHash(api.F_RHS, 5)
    s.removeMean(api.F_RHS)
    st = s.solveResident(True, False)
    assert st["iters"] == 80 and len(st["history"]) == 64
    h = api.last_history()
    assert len(h) == 81
    np.testing.assert_array_equal(h[:64], st["history"])
    assert h[-1] == st["final_rnorm"] or st["final_rnorm"] <= h[-1]     # (best-phi rollback may report an earlier, smaller norm)
    s.undefine()
