"""Dirichlet sides on the GPU (GHOST_DIRI ops + two-pass kernels) vs the oracle's setSideDiriBC / ELLIPTICCONSTDIRIBCGHOST:
level operator pieces bit for bit, homogeneous and inhomogeneous solves, the viscous-type Helmholtz operator."""
import numpy as np
import pytest

from helpers import download_valid, make_gpu_solver, make_problem, max_rel_diff, upload, valid_of

pytestmark = pytest.mark.gpu
D, N = 1, 0

CASES = [
    # n, box, periodic, bc types (x, y, z) lo/hi, alpha, beta
    ((16, 16, 8), 8, (False, False, False), [(D, D), (N, N), (N, D)], 0.0, 1.0),
    ((16, 16, 8), 8, (False, True, False), [(D, N), (N, N), (D, D)], 0.0, 1.0),
    ((32, 16, 16), (16, 8, 8), (False, False, False), [(D, D), (D, D), (D, D)], 1.0, -1e-3),
]


@pytest.fixture(autouse=True, params=["direct", "march", "fused"])
def _kernel_path(request, monkeypatch):
    """every test three times: levels this small run the direct-load residual and the two-pass smoother;
    SOMAR_MARCH_MIN_CELLS = 0 sends them through the LDS-marching residual and the fused residual + restriction,
    SOMAR_FUSED_MIN_CELLS = 0 also through the fused red+black sweep (Dirichlet ghosts synthesized in the kernel) --
    the kernels large Dirichlet levels (viscous solves) use"""
    if request.param == "march":
        monkeypatch.setenv("SOMAR_MARCH_MIN_CELLS", "0")
    elif request.param == "fused":
        monkeypatch.setenv("SOMAR_FUSED_MIN_CELLS", "0")


def _setup(so, case, values=None, **kw):
    n, bs, per, types, alpha, beta = case
    dom, grids, dx, Jgup, Jinv = make_problem(so, n, bs, "stretched", per, (1.0, 1.0, 0.5))
    bc = so.BCHolder([list(t) for t in types], [list(v) for v in values] if values else None)
    fac = so.Factory(dom, grids, dx, bc, Jgup, Jinv, alpha=alpha, beta=beta)
    amr = so.AMRMultiGrid(fac, so.BiCGStab())
    flat_types = [t for pair in types for t in pair]
    flat_vals = [v for pair in values for v in pair] if values else None
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv, alpha=alpha, beta=beta, bc_type=flat_types, bc_values=flat_vals, **kw)
    return dom, grids, amr, gpu


@pytest.mark.parametrize("case", CASES)
def test_dirichlet_pieces_bit_exact(oracle, case):
    from somar_amd import api as F
    so = oracle
    dom, grids, amr, gpu = _setup(so, case)
    try:
        assert gpu.depth() == amr.mg.depth
        assert [gpu.zeroAvg(d) for d in range(gpu.depth())] == [bool(op.zeroAvg) for op in amr.mg.ops]
        for d in range(min(amr.mg.depth, 2)):
            op = amr.mg.ops[d]
            g = op.grids
            phi = so.random_field(g, 7 + d, (1, 1, 1), op.domain.box)
            rhs = so.random_field(g, 8 + d, (0, 0, 0), op.domain.box)
            fc, fr, fs = ((F.F_PHI, F.F_RHS, F.F_RES) if d == 0 else
                          (F.FIELD(d, F.F_CORR), F.FIELD(d, F.F_RES), F.FIELD(d, F.F_SCRATCH)))
            upload(gpu, fc, phi, depth=d)
            upload(gpu, fr, rhs, depth=d)
            res = so.LevelData(g, 1)
            op.residual(res, phi, rhs, True)
            gpu.residual(d, fs, fc, fr)
            for a, b in zip(download_valid(gpu, fs, g, d), valid_of(res)):
                np.testing.assert_array_equal(a, b)
            op.relax(phi, rhs, 2)
            gpu.relax(d, fc, fr, 2)
            for a, b in zip(download_valid(gpu, fc, g, d), valid_of(phi)):
                np.testing.assert_array_equal(a, b)
            if d + 1 < amr.mg.depth:
                cg = [b.coarsen(op.mgCrseRefRatio) for b in g]
                crse = op.create_coarser(rhs)
                op.restrict_residual(crse, phi, rhs)
                gpu.restrictResidual(d, F.FIELD(d + 1, F.F_RES), fc, fr)
                for a, b in zip(download_valid(gpu, F.FIELD(d + 1, F.F_RES), cg, d + 1), valid_of(crse)):
                    np.testing.assert_array_equal(a, b)
    finally:
        gpu.undefine()


@pytest.mark.parametrize("case", CASES)
def test_dirichlet_homogeneous_solve_history(oracle, case):
    so = oracle
    dom, grids, amr, gpu = _setup(so, case)
    try:
        phi0 = so.random_field(grids, 3, (1, 1, 1), dom.box)
        b = so.LevelData(grids, 1)
        amr.op.apply_op(b, phi0, True)
        x = so.LevelData(grids, 1, (1, 1, 1))
        amr.solve(x, b, forceHomogeneous=True)
        gx = [np.zeros(f.a.shape[:3], order="F") for f in x.fabs]
        gb = [np.asfortranarray(f.a[..., 0]) for f in b.fabs]
        st = gpu.solve(gx, gb, 0, 0, True, True)
        assert st["iters"] == amr.iters and st["exitStatus"] == amr.exitStatus
        np.testing.assert_allclose(st["history"], amr.history, rtol=1e-10, atol=1e-13 * amr.history[0])
        assert st["history"][-1] <= 1e-6 * st["history"][0]
    finally:
        gpu.undefine()


def test_inhomogeneous_dirichlet_values(oracle):
    """Values 1 and 3 on the x sides and -2 on the z-high side enter the outer residuals (force_homogeneous = false)."""
    so = oracle
    vals = [(1.0, 3.0), (0.0, 0.0), (0.0, -2.0)]
    dom, grids, amr, gpu = _setup(so, CASES[0], values=vals)
    try:
        rhs = so.random_field(grids, 9, (0, 0, 0), dom.box)
        x = so.LevelData(grids, 1, (1, 1, 1))
        amr.solve(x, rhs, zeroPhi=True, forceHomogeneous=False)
        gx = [np.zeros(f.a.shape[:3], order="F") for f in x.fabs]
        gb = [np.asfortranarray(f.a[..., 0]) for f in rhs.fabs]
        st = gpu.solve(gx, gb, 0, 0, True, False)
        assert st["iters"] == amr.iters and st["exitStatus"] == amr.exitStatus
        np.testing.assert_allclose(st["history"], amr.history, rtol=1e-10, atol=1e-13 * amr.history[0])
        got = [a[1:-1, 1:-1, 1:-1] for a in gx]
        assert max_rel_diff(got, valid_of(x)) < 1e-8
    finally:
        gpu.undefine()


def test_dirichlet_sides_with_a_nondiagonal_metric(oracle):
    """Viscous-type solve on a sheared map: Dirichlet ghosts (order 1) sit in the ghost programs next to the cross-term
    Neumann ghosts, after fillExtrap as in applyOpI / fillGhostsAndExtrapolate."""
    from somar_amd import AMRPressureSolver
    from somar_amd import api as F
    so = oracle
    n, bs, per, L = (16, 16, 8), 8, (False, False, False), (2.0, 1.0, 0.5)
    types = [(D, D), (N, D), (D, N)]
    vals = [(0.5, -1.0), (0.0, 2.0), (0.25, 0.0)]
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), per)
    grids = so.split_domain(dom.box, bs)
    dx = tuple(L[d] / n[d] for d in range(3))
    Jgup, Jinv = so.make_full_metric(grids, dx, L, dom)
    bc = so.BCHolder([list(t) for t in types], [list(v) for v in vals])
    fac = so.Factory(dom, grids, dx, bc, Jgup, Jinv, alpha=1.0, beta=-0.02, isDiagonal=False)
    amr = so.AMRMultiGrid(fac, so.BiCGStab())
    s = AMRPressureSolver()
    s.define(dom.box.lo, dom.box.hi, dom.periodic, dx, [(g.lo, g.hi) for g in grids], alpha=1.0, beta=-0.02,
             bc_type=[t for pair in types for t in pair])
    s.setBCValues([v for pair in vals for v in pair])
    for q in range(s.num_local_patches):
        _, _, gi = s.patch_box(q)
        s.setMetricFull(q, *[np.asfortranarray(Jgup[gi][d].a) for d in range(3)], np.asfortranarray(Jinv[gi].a[..., 0]))
    s.finalize()
    try:
        op = amr.mg.ops[0]
        phi = so.random_field(grids, 7, (1, 1, 1), dom.box)
        rhs = so.random_field(grids, 8, (0, 0, 0), dom.box)
        upload(s, F.F_PHI, phi)
        upload(s, F.F_RHS, rhs)
        res = so.LevelData(grids, 1)
        op.residual(res, phi, rhs, True)
        s.residual(0, F.F_RES, F.F_PHI, F.F_RHS)
        for a, b in zip(download_valid(s, F.F_RES, grids), valid_of(res)):
            np.testing.assert_array_equal(a, b)
        op.relax(phi, rhs, 2)
        s.relax(0, F.F_PHI, F.F_RHS, 2)
        for a, b in zip(download_valid(s, F.F_PHI, grids), valid_of(phi)):
            np.testing.assert_array_equal(a, b)
        # inhomogeneous solve
        x = so.LevelData(grids, 1, (1, 1, 1))
        amr.solve(x, rhs, zeroPhi=True, forceHomogeneous=False)
        gx = [np.zeros(f.a.shape[:3], order="F") for f in x.fabs]
        gb = [np.asfortranarray(f.a[..., 0]) for f in rhs.fabs]
        st = s.solve(gx, gb, 0, 0, True, False)
        assert st["iters"] == amr.iters and st["exitStatus"] == amr.exitStatus
        np.testing.assert_allclose(st["history"], amr.history, rtol=1e-9, atol=1e-13 * amr.history[0])
    finally:
        s.undefine()


def test_amr_composite_solve_with_inhomogeneous_dirichlet_values(oracle):
    """Two levels (refinement (2,2,1)), Dirichlet values on the x sides and the z-high side: they enter the composite
    outer residuals (force_homogeneous = false), every correction sees zero."""
    from oracle import somar_amr as am
    from somar_amd import AMRPressureSolver
    from somar_amd import api as F
    from helpers import make_amr_levels
    so = oracle
    types = [(D, D), (N, N), (N, D)]
    vals = [(1.0, -0.5), (0.0, 0.0), (0.0, 2.0)]
    ratios = [(2, 2, 1)]
    fb = [[so.Box((8, 8, 0), (23, 23, 7))]]
    levels = make_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.5), (False, False, False), ratios, fb)
    bc = so.BCHolder([list(t) for t in types], [list(v) for v in vals])
    comp = am.AMRComposite(levels, ratios, bc, so.BiCGStab())
    s = AMRPressureSolver()
    L0 = levels[0]
    s.defineAMR(L0.domain.box.lo, L0.domain.box.hi, L0.domain.periodic, L0.dx, ratios,
                [[(g.lo, g.hi) for g in L.grids] for L in levels], bc_type=[t for pair in types for t in pair])
    for L, v in zip(levels, s.levels):
        v.setBCValues([x for pair in vals for x in pair])
        for p_ in range(v.num_local_patches):
            _, _, gi = v.patch_box(p_)
            jg = [np.asfortranarray(L.Jgup[gi][d].a[..., d]) for d in range(3)]
            v.setMetricOrtho(p_, jg[0], jg[1], jg[2], np.asfortranarray(L.Jinv[gi].a[..., 0]))
    s.finalize()
    try:
        rhs = [so.random_field(L.grids, 50 + l, (0, 0, 0), L.domain.box) for l, L in enumerate(levels)]
        comp.zero_covered(0, rhs[0])
        sol = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
        comp.solve(sol, rhs, 1, 0, forceHomogeneous=False)
        for l, v in enumerate(s.levels):
            upload(v, F.F_RHS, rhs[l])
        st = s.solveAMR(1, 0, zeroPhi=True, forceHomogeneous=False)
        assert st["iters"] == comp.iters and st["exitStatus"] == comp.exitStatus
        np.testing.assert_allclose(st["history"], comp.history, rtol=1e-10, atol=0.0)
        for l in (0, 1):
            assert max_rel_diff(download_valid(s.levels[l], F.F_PHI, levels[l].grids), valid_of(sol[l])) < 1e-8
    finally:
        s.undefine()
