"""Vertical-line GSRB (relax_mode 3) on the GPU vs the oracle's LineGSRBIter3D + dgtsv restatement."""
import numpy as np
import pytest

from helpers import download_valid, make_gpu_solver, make_oracle_solver, make_problem, max_rel_diff, upload, valid_of

pytestmark = pytest.mark.gpu

CASES = [
    # columns are never split in z (GSRB.H:89-91)
    ((32, 32, 32), (16, 16, 32), "cartesian", (False, True, False), (8.0, 8.0, 1.0)),
    ((32, 16, 24), (16, 8, 24), "stretched", (False, False, False), (4.0, 2.0, 0.5)),
    ((36, 20, 12), (12, 20, 12), "stretched", (True, False, False), (2.0, 1.0, 0.25)),
    ((16, 16, 1), (8, 16, 1), "stretched", (False, False, False), (1.0, 1.0, 0.1)),   # degenerate 1-cell columns
]


def test_dgtsv_restatement_matches_lapack(oracle):
    """The no-interchange dgtsv used by oracle and GPU alike equals SciPy's LAPACK dgtsv bit for bit on
    diagonally dominant systems of the shape LineGSRB assembles."""
    import ctypes as C
    from scipy.linalg import lapack
    rng = np.random.default_rng(0)
    P = C.POINTER(C.c_double)
    for n in (2, 3, 17, 128):
        dl = rng.uniform(0.5, 1.0, n - 1)
        d = -(2.5 + rng.uniform(0.0, 1.0, n))
        b = rng.uniform(-1, 1, n)
        dl2, du2, d2, b2 = np.append(dl, 0.0), np.append(dl, 0.0), d.copy(), b.copy()
        info = oracle.lib().orc_dgtsv_nopivot(n, dl2.ctypes.data_as(P), d2.ctypes.data_as(P), du2.ctypes.data_as(P),
                                              b2.ctypes.data_as(P))
        _, _, _, x, linfo = lapack.dgtsv(dl, d, dl, b)
        assert info == 0 and linfo == 0
        np.testing.assert_array_equal(b2, x)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("alpha_beta", [(0.0, 1.0), (1.0, -0.05)])
def test_line_gsrb_sweep_bit_exact(oracle, case, alpha_beta):
    from somar_amd import api as F
    so = oracle
    n, boxsz, variant, periodic, L = case
    a, b = alpha_beta
    dom, grids, dx, Jgup, Jinv = make_problem(so, n, boxsz, variant, periodic, L)
    fac = so.Factory(dom, grids, dx, so.BCHolder(), Jgup, Jinv, alpha=a, beta=b, relaxMode=so.RELAX_LINE_GSRB, maxDepth=0)
    op = fac.mg_new_op(0, None)
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv, alpha=a, beta=b, relaxMode=3, maxDepth=0)
    phi = so.random_field(grids, 41, (1, 1, 1), dom.box)
    rhs = so.random_field(grids, 42, (0, 0, 0), dom.box)
    upload(gpu, F.F_PHI, phi)
    upload(gpu, F.F_RHS, rhs)
    op.relax(phi, rhs, 2)
    gpu.relax(0, F.F_PHI, F.F_RHS, 2)
    for g, w in zip(download_valid(gpu, F.F_PHI, grids), valid_of(phi)):
        np.testing.assert_array_equal(g, w)
    gpu.undefine()


def test_line_relaxation_solve_matches_oracle_and_beats_point_gsrb(oracle):
    so = oracle
    n, boxsz, variant, periodic, L = CASES[0]
    dom, grids, dx, Jgup, Jinv = make_problem(so, n, boxsz, variant, periodic, L)
    amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv, relaxMode=so.RELAX_LINE_GSRB)
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv, relaxMode=3)
    rhs = so.random_field(grids, 12345, (0, 0, 0), dom.box)
    so.remove_weighted_mean(rhs, amr.op.Jinv)
    phi = so.LevelData(grids, 1, (1, 1, 1))
    amr.solve(phi, rhs)
    gphi = [np.zeros(f.a.shape[:3], order="F") for f in phi.fabs]
    grhs = [np.asfortranarray(f.a[..., 0]) for f in rhs.fabs]
    st = gpu.solve(gphi, grhs, 0, 0, True, False)
    assert st["iters"] == amr.iters and st["exitStatus"] == amr.exitStatus
    np.testing.assert_allclose(st["history"], amr.history, rtol=1e-10, atol=1e-10 * amr.history[0])
    got = [a[1:-1, 1:-1, 1:-1] for a in gphi]
    assert max_rel_diff(got, valid_of(phi)) < 1e-8
    gpu.undefine()
    # dz << dx: the point smoother needs far more V-cycles than the line smoother on the same problem
    gp = make_gpu_solver(dom, grids, dx, Jgup, Jinv, relaxMode=1)
    gphi2 = [np.zeros(f.a.shape[:3], order="F") for f in phi.fabs]
    try:
        st2 = gp.solve(gphi2, grhs, 0, 0, True, False)
        assert st2["iters"] > st["iters"]
    finally:
        gp.undefine()


LOOSE_CASES = [
    ((32, 32, 16), 16, "stretched", (False, True, False), (2.0, 1.0, 1.0)),
    ((36, 20, 12), (12, 20, 4), "stretched", (True, False, False), (2.0, 1.0, 0.25)),
    ((24, 24, 8), (8, 8, 8), "cartesian", (True, True, True), (1.0, 1.0, 1.0)),
]


@pytest.mark.parametrize("case", LOOSE_CASES)
def test_loose_gsrb_sweep_bit_exact_and_solve(oracle, case):
    """LooseGSRB (relax_mode 2, GSRB.cpp:104-141): one exchange per sweep, interior cells then box shells."""
    from somar_amd import api as F
    so = oracle
    n, boxsz, variant, periodic, L = case
    dom, grids, dx, Jgup, Jinv = make_problem(so, n, boxsz, variant, periodic, L)
    fac = so.Factory(dom, grids, dx, so.BCHolder(), Jgup, Jinv, relaxMode=so.RELAX_LOOSE_GSRB, maxDepth=0)
    op = fac.mg_new_op(0, None)
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv, relaxMode=2, maxDepth=0)
    phi = so.random_field(grids, 41, (1, 1, 1), dom.box)
    rhs = so.random_field(grids, 42, (0, 0, 0), dom.box)
    upload(gpu, F.F_PHI, phi)
    upload(gpu, F.F_RHS, rhs)
    op.relax(phi, rhs, 2)
    gpu.relax(0, F.F_PHI, F.F_RHS, 2)
    for g, w in zip(download_valid(gpu, F.F_PHI, grids), valid_of(phi)):
        np.testing.assert_array_equal(g, w)
    gpu.undefine()
    # full solve with the loose smoother
    amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv, relaxMode=so.RELAX_LOOSE_GSRB)
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv, relaxMode=2)
    b = so.random_field(grids, 12345, (0, 0, 0), dom.box)
    so.remove_weighted_mean(b, amr.op.Jinv)
    x = so.LevelData(grids, 1, (1, 1, 1))
    amr.solve(x, b)
    upload(gpu, F.F_RHS, b)
    depth = gpu.depth()
    st = gpu.solveResident(True, False)
    assert st["iters"] == amr.iters and st["exitStatus"] == amr.exitStatus
    if depth > 1:
        np.testing.assert_allclose(st["history"], amr.history, rtol=1e-10, atol=1e-12 * amr.history[0])
    else:
        # boxes too thin to coarsen: the "V-cycle" is one BiCGStab solve to 1e-6, whose tree-ordered sums (level above
        # the ordered-reduction threshold) wander within that tolerance
        assert st["history"][-1] <= 1e-6 * st["history"][0]
    gpu.undefine()


@pytest.mark.parametrize("types", [[(1, 1), (0, 0), (0, 1)], [(0, 1), (1, 0), (1, 1)]])
def test_line_gsrb_with_dirichlet_sides_bit_exact(oracle, types):
    """Dirichlet vertical ends are folded into the tridiagonal system (coefficient 2 on the end face), lateral Dirichlet
    sides arrive through their ghost cells."""
    from somar_amd import api as F
    so = oracle
    n, boxsz, variant, periodic, L = (16, 16, 8), 8, "stretched", (False, False, False), (1.0, 1.0, 0.5)
    dom, grids, dx, Jgup, Jinv = make_problem(so, n, boxsz, variant, periodic, L)
    bc = so.BCHolder([list(t) for t in types])
    fac = so.Factory(dom, grids, dx, bc, Jgup, Jinv, alpha=1.0, beta=-0.05, relaxMode=so.RELAX_LINE_GSRB, maxDepth=0)
    op = fac.mg_new_op(0, None)
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv, alpha=1.0, beta=-0.05, relaxMode=3, maxDepth=0,
                          bc_type=[t for pair in types for t in pair])
    try:
        phi = so.random_field(grids, 41, (1, 1, 1), dom.box)
        rhs = so.random_field(grids, 42, (0, 0, 0), dom.box)
        upload(gpu, F.F_PHI, phi)
        upload(gpu, F.F_RHS, rhs)
        op.relax(phi, rhs, 2)
        gpu.relax(0, F.F_PHI, F.F_RHS, 2)
        for g, w in zip(download_valid(gpu, F.F_PHI, grids), valid_of(phi)):
            np.testing.assert_array_equal(g, w)
    finally:
        gpu.undefine()


@pytest.mark.parametrize("per", [(False, True, False), (False, False, False)])
def test_line_gsrb_with_a_nondiagonal_metric_bit_exact_and_solve(oracle, per):
    """LineGSRBIter3D's explicit cross terms (from the extrapolated copy) next to the implicit column solve."""
    from somar_amd import AMRPressureSolver
    from somar_amd import api as F
    so = oracle
    n, bs, L = (16, 16, 8), 8, (2.0, 1.0, 0.5)
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), per)
    grids = so.split_domain(dom.box, bs)
    dx = tuple(L[d] / n[d] for d in range(3))
    Jgup, Jinv = so.make_full_metric(grids, dx, L, dom)
    fac = so.Factory(dom, grids, dx, so.BCHolder(), Jgup, Jinv, isDiagonal=False, relaxMode=so.RELAX_LINE_GSRB)
    amr = so.AMRMultiGrid(fac, so.BiCGStab())
    s = AMRPressureSolver()
    p = s._p
    s.setAMRMGParameters(p.imin, p.imax, p.eps, -1, p.num_smooth_precond, 2, 2, 2, p.precond_mode, 3, p.num_mg, p.hang,
                         p.norm_thresh, 0)
    s.define(dom.box.lo, dom.box.hi, dom.periodic, dx, [(g.lo, g.hi) for g in grids])
    for q in range(s.num_local_patches):
        _, _, gi = s.patch_box(q)
        s.setMetricFull(q, *[np.asfortranarray(Jgup[gi][d].a) for d in range(3)], np.asfortranarray(Jinv[gi].a[..., 0]))
    s.finalize()
    try:
        op = amr.mg.ops[0]
        phi = so.random_field(grids, 41, (1, 1, 1), dom.box)
        rhs = so.random_field(grids, 42, (0, 0, 0), dom.box)
        upload(s, F.F_PHI, phi)
        upload(s, F.F_RHS, rhs)
        op.relax(phi, rhs, 2)
        s.relax(0, F.F_PHI, F.F_RHS, 2)
        for g, w in zip(download_valid(s, F.F_PHI, grids), valid_of(phi)):
            np.testing.assert_array_equal(g, w)
        phi0 = so.random_field(grids, 3, (1, 1, 1), dom.box)
        b = so.LevelData(grids, 1)
        amr.op.apply_op(b, phi0, True)
        x = so.LevelData(grids, 1, (1, 1, 1))
        amr.solve(x, b)
        gx = [np.zeros(f.a.shape[:3], order="F") for f in x.fabs]
        gb = [np.asfortranarray(f.a[..., 0]) for f in b.fabs]
        st = s.solve(gx, gb, 0, 0, True, False)
        assert st["iters"] == amr.iters and st["exitStatus"] == amr.exitStatus
        np.testing.assert_allclose(st["history"], amr.history, rtol=1e-9, atol=1e-12 * amr.history[0])
    finally:
        s.undefine()


def test_line_relaxation_on_a_fine_level_whose_columns_end_at_coarse_fine_faces(oracle):
    """(2,2,2) refinement, fine box strictly inside the domain: the columns of the fine level end at coarse-fine faces.
    LineGSRB::relax passes BCDescriptor::stencil's codes, None for such ends (BCDescriptor.H:218-229), so the Fortran's CF row
    is never reached: coeff1 = 0, the CF ghost is not read.  Level relaxation and a whole AMR V-cycle, bit for bit."""
    from oracle import somar_amr as am
    from somar_amd import api as F
    from helpers import make_amr_levels, make_gpu_amr
    so = oracle
    ratios = [(2, 2, 2)]
    fb = [[so.Box((8, 8, 4), (23, 23, 11))]]
    levels = make_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.25), (False, False, False), ratios, fb)
    comp = am.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab(), relaxMode=so.RELAX_LINE_GSRB)
    gpu = make_gpu_amr(levels, ratios, relaxMode=3)
    try:
        L, v = levels[1], gpu.levels[1]
        res = so.random_field(L.grids, 70, (0, 0, 0), L.domain.box)
        corr = so.random_field(L.grids, 71, (1, 1, 1), L.domain.box)
        upload(v, F.F_RES, res)
        upload(v, F.F_CORR, corr)
        comp.ops[1].relax(corr, res, 2)
        v.relax(0, F.F_CORR, F.F_RES, 2)
        for g, w in zip(download_valid(v, F.F_CORR, L.grids), valid_of(corr)):
            np.testing.assert_array_equal(g, w)
        # one AMR V-cycle on a compatible composite residual
        phi = [so.random_field(X.grids, 5 + l, (1, 1, 1), X.domain.box) for l, X in enumerate(levels)]
        zero = [so.LevelData(X.grids, 1) for X in levels]
        rhs = [so.LevelData(X.grids, 1) for X in levels]
        comp.init(phi, zero, 1, 0)
        comp.compute_amr_residual(rhs, phi, zero, 1, 0, True)
        for r in rhs:
            so.ld_scale(r, -1.0)
        sol = [so.LevelData(X.grids, 1, (1, 1, 1)) for X in levels]
        comp.solve(sol, rhs, 1, 0)
        for l, vv in enumerate(gpu.levels):
            upload(vv, F.F_RHS, rhs[l])
        st = gpu.solveAMR(1, 0)
        assert st["iters"] == comp.iters and st["exitStatus"] == comp.exitStatus
        np.testing.assert_allclose(st["history"], comp.history, rtol=0, atol=1e-10 * comp.history[0])
    finally:
        gpu.undefine()
