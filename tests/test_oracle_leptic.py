"""Known answers for the oracle's restatement of the leptic level solver (oracle/somar_leptic.py).  The reference
ships no tests for this path: the checks are analytic properties of the method."""
import ctypes as C

import numpy as np
import pytest

from oracle import somar_leptic as sl
from oracle import somar_oracle as so
from tests.helpers import make_oracle_solver


def test_nn_tridiagonal_solves_consistent_columns():
    rng = np.random.default_rng(5)
    n = (3, 2, 11)
    box = so.Box((0, 0, 0), tuple(a - 1 for a in n))
    phi = so.Fab(box.grow((1, 1, 1)))
    rhs = so.Fab(box)
    sig = so.Fab(box.faces(2), 3)
    sig.a[...] = rng.uniform(0.5, 2.0, sig.a.shape)
    r = rng.standard_normal(n)
    r -= r.mean(axis=2, keepdims=True)   # solvable Neumann-Neumann columns
    rhs.a[..., 0] = r
    dz = 0.37
    blo, bhi = so._b(so.Box((0, 0, 0), (n[0] - 1, n[1] - 1, 0)))
    so.lib().orc_tridiagpoissonnn1dfab(*phi.fra1(0), *rhs.fra1(0), *sig.fra1(2), blo, bhi, n[2], C.c_double(dz), 2)
    p = phi.view(box)[..., 0]
    s = sig.a[..., 2]
    flux = np.zeros((n[0], n[1], n[2] + 1))
    flux[:, :, 1:-1] = s[:, :, 1:-1] * (p[:, :, 1:] - p[:, :, :-1]) / dz
    lap = (flux[:, :, 1:] - flux[:, :, :-1]) / dz
    np.testing.assert_allclose(lap, r, rtol=0, atol=1e-11)
    np.testing.assert_allclose(p.mean(axis=2), 0.0, atol=1e-14)


def _thin_problem(n=(32, 32, 8), box=(16, 16, 8), L=(1.0, 1.0, 0.02), seed=3, variant="stretched"):
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (False, False, False))
    grids = so.split_domain(dom.box, box)
    dx = tuple(L[d] / n[d] for d in range(3))
    Jgup, Jinv = so.make_diagonal_metric(grids, dx, L, 3, variant, domain=dom)
    rhs = so.random_field(grids, seed, domainBox=dom.box)
    so.remove_weighted_mean(rhs, Jinv)
    return dom, grids, dx, Jgup, Jinv, rhs


def _run(H, maxOrder, variant="stretched", **kw):
    L = (1.0, 1.0, H)
    dom, grids, dx, Jgup, Jinv, rhs = _thin_problem(L=L, variant=variant)
    amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv)
    lep = sl.LevelLepticSolver(amr.op, maxOrder=maxOrder, domainHeight=H, **kw)
    phi = so.LevelData(grids, 1, (1, 1, 1))
    status = lep.solve(phi, rhs)
    return amr, lep, phi, rhs, status, (dom, grids, dx, Jgup, Jinv)


def test_leptic_orders_gain_eps_squared():
    """O(1): vertical solves + ONE horizontal solve (diagonal metric) leave a residual concentrated in the top
    cells (the excess parked in the upper BC); O(eps) removes it; what is left is O(eps^2), eps = H / dx_horizontal."""
    out = {}
    for H in (0.005, 0.001):
        amr, lep, phi, rhs, status, (dom, grids, dx, Jgup, Jinv) = _run(H, 4)
        h = lep.resNorms
        assert len(h) == 6 and status == sl.EXIT_ITER
        assert lep.horizSolves == 1 and not lep.usedFullSolver
        assert h[2] < 0.02 * h[0]
        # the accumulated correction solves the original equation to the reported residual
        res = so.LevelData(grids, 1, (0, 0, 0))
        amr.op.residual(res, phi, rhs, False)
        jres = max(float(np.max(np.abs(res[i].view(g) / Jinv[i].view(g)))) for i, g in enumerate(grids))
        assert abs(jres - h[-1]) <= 1e-9 * h[0]
        out[H] = h[2] / h[0]
    assert 20.0 < out[0.005] / out[0.001] < 30.0   # (0.005 / 0.001)^2 = 25


def test_leptic_falls_back_to_full_multigrid_when_hanging_at_the_last_order():
    """maxOrder = 0: the O(1) residual is larger than the initial one by construction, so the full 3-D multigrid
    (LINE_GSRB, 4/4/4) takes over (LevelLepticSolver.cpp:851-875) and its correction is the one that is used."""
    amr, lep, phi, rhs, status, (dom, grids, dx, Jgup, Jinv) = _run(0.02, 0)
    assert lep.usedFullSolver and status == sl.EXIT_ITER
    assert lep.resNorms[-1] < 1e-5 * lep.resNorms[0]
    res = so.LevelData(grids, 1, (0, 0, 0))
    amr.op.residual(res, phi, rhs, False)
    jres = max(float(np.max(np.abs(res[i].view(g) / Jinv[i].view(g)))) for i, g in enumerate(grids))
    assert abs(jres - lep.resNorms[-1]) <= 1e-9 * lep.resNorms[0]


def test_leptic_agrees_with_multigrid_solution():
    """Cartesian metric: the horizontal operator commutes with the vertical average, so the single horizontal solve
    of the diagonal-metric path is all that is needed and the orders converge geometrically (eps^2 each)."""
    H = 0.005
    amr, lep, phi, rhs, status, (dom, grids, dx, Jgup, Jinv) = _run(H, 5, variant="cartesian")
    h = lep.resNorms
    assert h[-1] < 1e-8 * h[0]
    amr.eps = 1e-12
    amr.iterMax = 40
    phi_mg = so.LevelData(grids, 1, (1, 1, 1))
    amr.solve(phi_mg, rhs, zeroPhi=True)

    def demean(ld):
        v = np.concatenate([f.view(g).ravel() for g, f in zip(ld.grids, ld.fabs)])
        return v - v.mean()
    a, b = demean(phi), demean(phi_mg)
    # point-GSRB multigrid stalls near 1e-4 on this thin domain (exit status 4) -- the reason the leptic solver
    # exists -- so the comparison can only be as good as the multigrid answer
    assert np.max(np.abs(a - b)) < 1e-3 * np.max(np.abs(b))


def _terrain_problem(n=(32, 32, 8), box=(16, 16, 8), L=(64.0, 64.0, 1.0), seed=3, metric="terrain"):
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (False, False, False))
    grids = so.split_domain(dom.box, box)
    dx = tuple(L[d] / n[d] for d in range(3))
    if metric == "terrain":
        Jgup, Jinv = so.make_terrain_metric(grids, dx, L, dom)
    else:
        Jgup, Jinv = so.make_full_metric(grids, dx, L, dom, amp=(0.05, 0.04, 0.03))
    rhs = so.random_field(grids, seed, domainBox=dom.box)
    so.remove_weighted_mean(rhs, Jinv)
    return dom, grids, dx, Jgup, Jinv, rhs


def test_leptic_nondiagonal_metric_converges_with_the_cross_terms():
    """Terrain-following map (J g^{xz}, J g^{yz} != 0): every order runs a horizontal solve
    (LevelLepticSolver.cpp:820-826 keeps m_doHorizSolve for a non-diagonal metric), the vertical boundary data of
    order k is -J g^{z m} d_m phi_{k-1} (levelVertHorizGradient, :1107-1176), and the orders converge on the FULL
    19-point operator: the reported J-scaled residual is the one of the level's own operator."""
    dom, grids, dx, Jgup, Jinv, rhs = _terrain_problem()
    amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv, isDiagonal=False)
    assert not amr.op.isDiagonal
    lep = sl.LevelLepticSolver(amr.op, maxOrder=4, domainHeight=1.0)
    phi = so.LevelData(grids, 1, (1, 1, 1))
    status = lep.solve(phi, rhs)
    h = lep.resNorms
    assert status in (sl.EXIT_ITER, sl.EXIT_CONVERGE) and not lep.usedFullSolver
    assert lep.horizSolves >= 4
    assert h[-1] < 1e-4 * h[0]   # eps = H / dx = 0.5: about eps^2 ... eps^3 per order
    assert all(h[k + 1] < 0.2 * h[k] for k in range(2, len(h) - 1))
    res = so.LevelData(grids, 1, (0, 0, 0))
    amr.op.residual(res, phi, rhs, False)
    jres = max(float(np.max(np.abs(res[i].view(g) / Jinv[i].view(g)))) for i, g in enumerate(grids))
    assert abs(jres - h[-1]) <= 1e-9 * h[0]


def test_leptic_nondiagonal_reduces_to_diagonal_when_cross_terms_vanish():
    """The non-diagonal code path fed a diagonal metric stored in full form must give the diagonal path's
    vertical solves: bcLo / bcHi from levelVertHorizGradient are exactly zero, so order-k vertical problems agree;
    only the extra horizontal solves (useHorizPhi stays on) differ, and they act on a right-hand side that is
    zero up to round-off after order 0."""
    L = (1.0, 1.0, 0.005)
    dom, grids, dx, Jgup, Jinv, rhs = _thin_problem(L=L)
    amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv)
    lepD = sl.LevelLepticSolver(amr.op, maxOrder=3, domainHeight=L[2])
    phiD = so.LevelData(grids, 1, (1, 1, 1))
    lepD.solve(phiD, rhs)
    amrF = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv, isDiagonal=False)
    lepF = sl.LevelLepticSolver(amrF.op, maxOrder=3, domainHeight=L[2])
    phiF = so.LevelData(grids, 1, (1, 1, 1))
    lepF.solve(phiF, rhs)
    assert lepF.horizSolves > lepD.horizSolves
    # order 0 is the same computation (zero vertical boundary data, one horizontal solve); later orders differ by the
    # extra horizontal solves, which only help on a stretched metric (the average does not commute with the operator)
    np.testing.assert_allclose(lepF.resNorms[:2], lepD.resNorms[:2], rtol=1e-12)
    assert lepF.resNorms[3] < 0.1 * lepD.resNorms[3]
    assert not lepF.usedFullSolver and lepF.resNorms[-1] < 1e-5 * lepF.resNorms[0]


# ---- lateral coarse-fine boundaries and the composite leptic solver ------------------------------------------------
def _two_level(variant="cartesian", L=(1.0, 1.0, 0.005)):
    from oracle import somar_amr as sa
    from tests.helpers import make_amr_levels
    n, ratios = (32, 32, 8), [(2, 2, 1)]
    fine = [[so.Box((16, 16, 0), (31, 47, 7)), so.Box((32, 16, 0), (47, 47, 7))]]   # two column boxes, CF on all lateral sides
    levels = make_amr_levels(so, sa, n, L, (False, False, False), ratios, fine, variant=variant, cbox=(16, 16, 8))
    return levels, ratios


def _compatible_rhs(amr, levels, lmax):
    phi = [so.random_field(Lv.grids, 5 + l, (1, 1, 1), Lv.domain.box) for l, Lv in enumerate(levels)]
    zero = [so.LevelData(Lv.grids, 1) for Lv in levels]
    rhs = [so.LevelData(Lv.grids, 1) for Lv in levels]
    amr.init(phi, zero, lmax, 0)
    amr.compute_amr_residual(rhs, phi, zero, lmax, 0, True)
    for r in rhs:
        so.ld_scale(r, -1.0)
    return rhs


def test_level_leptic_with_lateral_cf_converges_on_the_fine_level():
    """A refined level alone (l_base = l_max = 1, homogeneous CF values): columns span the domain, the flat problem is a
    Dirichlet-like one (no mean removal), every order gains about eps^2."""
    from oracle import somar_amr as sa
    levels, ratios = _two_level()
    comp = sa.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab())
    op = comp.ops[1]
    assert op.cf is not None and op.cf.has_cf()
    lep = sl.LevelLepticSolver(op, maxOrder=4, domainHeight=0.005)
    assert not lep.horizRemoveAvg
    rhs = so.random_field(levels[1].grids, 9, domainBox=levels[1].domain.box)
    phi = so.LevelData(levels[1].grids, 1, (1, 1, 1))
    status = lep.solve(phi, rhs, True)
    h = lep.resNorms
    assert status == sl.EXIT_ITER and not lep.usedFullSolver
    assert h[-1] < 1e-4 * h[0] and all(b < 0.5 * a for a, b in zip(h[1:], h[2:]))
    res = so.LevelData(levels[1].grids, 1, (0, 0, 0))
    op.residual(res, phi, rhs, True)
    jres = max(float(np.max(np.abs(res[i].view(g) / op.Jinv[i].view(g)))) for i, g in enumerate(levels[1].grids))
    assert abs(jres - h[-1]) <= 1e-9 * h[0]


def test_amr_leptic_vcycle_as_written_and_with_the_base_level_fed_the_restricted_residual():
    """AMRLepticSolver::AMRVCycle as written solves the base level from a_uberResidual and never prolongs its correction
    (AMRLepticSolver.cpp:444-449): the first cycle still removes the bulk of the residual, later ones drift.  Fed the
    restricted residual (not the reference) the same pieces converge by five orders per cycle."""
    levels, ratios = _two_level()
    amr = sl.AMRLepticSolver(levels, ratios, so.BCHolder(), leptic=dict(maxOrder=3, domainHeight=0.005))
    rhs = _compatible_rhs(amr, levels, 1)
    sol = [so.LevelData(Lv.grids, 1, (1, 1, 1)) for Lv in levels]
    amr.iterMax = 3
    amr.solve(sol, rhs, 1, 0)
    h = amr.history
    assert amr.iters == 3 and h[1] < 1e-3 * h[0] and h[3] > h[2]
    fixed = sl.AMRLepticSolver(levels, ratios, so.BCHolder(), leptic=dict(maxOrder=3, domainHeight=0.005),
                               baseFromRestricted=True)
    sol2 = [so.LevelData(Lv.grids, 1, (1, 1, 1)) for Lv in levels]
    fixed.iterMax = 3
    fixed.solve(sol2, rhs, 1, 0)
    g = fixed.history
    assert g[1] < 1e-4 * g[0] and g[2] < 1e-2 * g[1]


# ---- columns that END at Dirichlet walls or coarse-fine interfaces: LepticLapackVerticalSolver + dptsv ---------------------
def test_dptsv_restatement_equals_scipys_lapack_bit_for_bit():
    """the symmetric tridiagonal systems LepticLapackVerticalSolver assembles, through the restated
    dpttrf + dptts2 loops and through SciPy's LAPACK dptsv"""
    from scipy.linalg import lapack
    rng = np.random.default_rng(11)
    for n in (2, 3, 8, 17, 64):
        J = rng.uniform(0.5, 2.0, (5, n + 1))
        invdzsq = 1.0 / (0.37 * 0.37)
        D = (J[:, :-1] + J[:, 1:]) * invdzsq
        D[:, 0] = (2.0 * J[:, 0] + J[:, 1]) * invdzsq          # Dirichlet below
        D[:, -1] = J[:, n - 1] * invdzsq                        # Neumann above
        E = -J[:, 1:n] * invdzsq
        B = rng.uniform(-1, 1, (5, n))
        d2, e2, b2 = D.copy(), E.copy(), B.copy()
        sl.dptsv(d2, e2, b2)
        for c in range(5):
            _, _, x, info = lapack.dptsv(D[c], E[c], B[c])
            assert info == 0
            np.testing.assert_array_equal(b2[c], x)


def _dirichlet_top_problem(n=(16, 16, 8), box=(8, 8, 8), L=(1.0, 1.0, 0.02), variant="stretched"):
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (False, False, False))
    grids = so.split_domain(dom.box, box)
    dx = tuple(L[d] / n[d] for d in range(3))
    Jgup, Jinv = so.make_diagonal_metric(grids, dx, L, 3, variant=variant, domain=dom)
    bc = so.BCHolder([[so.BC_NEUM, so.BC_NEUM], [so.BC_NEUM, so.BC_NEUM], [so.BC_NEUM, so.BC_DIRI]])
    fac = so.Factory(dom, grids, dx, bc, Jgup, Jinv)
    return dom, grids, dx, Jgup, Jinv, fac


def test_dirichlet_topped_columns_need_no_horizontal_solve():
    """Neumann below, Dirichlet above (a free surface): no column is Neumann-Neumann, so gatherVerticalBCTypes switches the
    horizontal problem off and every order is ONE dptsv per column.  Known answers: (1) with a right-hand side that depends
    on z only (Cartesian map) the vertical solve IS the solution -- order 0 leaves round-off; (2) on a thin stretched
    domain every order gains about eps^2 and the reported norm is the true J-weighted residual."""
    dom, grids, dx, Jgup, Jinv, fac = _dirichlet_top_problem(variant="cartesian")
    op = so.AMRMultiGrid(fac, so.BiCGStab()).op
    lep = sl.LevelLepticSolver(op, maxOrder=1)
    assert not lep.doHorizSolve and lep.vertBCTypes[0] == (sl.VBC_NEUM, sl.VBC_DIRI)
    rhs = so.LevelData(grids, 1)
    for g, f in zip(grids, rhs.fabs):
        k = np.arange(g.lo[2], g.hi[2] + 1)
        f.view(g)[..., 0] = np.cos(0.7 * k)[None, None, :]
    phi = so.LevelData(grids, 1, (1, 1, 1))
    lep.solve(phi, rhs, True)
    assert lep.resNorms[1] < 1e-11 * lep.resNorms[0]
    # thin stretched domain, random data
    dom, grids, dx, Jgup, Jinv, fac = _dirichlet_top_problem(L=(1.0, 1.0, 0.005))
    op = so.AMRMultiGrid(fac, so.BiCGStab()).op
    lep = sl.LevelLepticSolver(op, maxOrder=3)
    rhs = so.random_field(grids, 9, domainBox=dom.box)
    phi = so.LevelData(grids, 1, (1, 1, 1))
    status = lep.solve(phi, rhs, True)
    h = lep.resNorms
    assert status == sl.EXIT_ITER and not lep.usedFullSolver and lep.horizSolves == 0
    assert all(b < 0.2 * a for a, b in zip(h, h[1:])) and h[-1] < 1e-4 * h[0]
    res = so.LevelData(grids, 1, (0, 0, 0))
    op.residual(res, phi, rhs, True)
    jres = max(float(np.max(np.abs(res[i].view(g) / Jinv[i].view(g)))) for i, g in enumerate(grids))
    assert abs(jres - h[-1]) <= 1e-9 * h[0]


def _bottom_half_refined(L=(1.0, 1.0, 0.005)):
    """level 1 = the lower half of the water column of the central block, refined by (2, 2, 2): its columns start at the
    Neumann bottom and END at a coarse-fine interface"""
    from oracle import somar_amr as sa
    from tests.helpers import make_amr_levels
    n, ratios = (16, 16, 8), [(2, 2, 2)]
    fine = [[so.Box((8, 8, 0), (15, 23, 7)), so.Box((16, 8, 0), (23, 23, 7))]]
    levels = make_amr_levels(so, sa, n, L, (False, False, False), ratios, fine, cbox=(8, 8, 8))
    return levels, ratios


def test_columns_ending_at_a_coarse_fine_interface():
    """LepticLapackVerticalSolver's BCType_CF row (linear interpolation against a zero coarse value, alpha =
    1 - 2 dz / (dzCrse + dz)): on a refined level whose columns end under the coarse level the orders converge (the lateral
    coarse-fine faces and the column ends both carry homogeneous values) and the reported norm is the true residual"""
    from oracle import somar_amr as sa
    levels, ratios = _bottom_half_refined()
    comp = sa.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab())
    op = comp.ops[1]
    lep = sl.LevelLepticSolver(op, maxOrder=4, domainHeight=0.005)
    assert not lep.doHorizSolve and all(t == (sl.VBC_NEUM, sl.VBC_CF) for t in lep.vertBCTypes)
    rhs = so.random_field(levels[1].grids, 9, domainBox=levels[1].domain.box)
    phi = so.LevelData(levels[1].grids, 1, (1, 1, 1))
    status = lep.solve(phi, rhs, True)
    h = lep.resNorms
    assert status in (sl.EXIT_ITER, sl.EXIT_CONVERGE) and h[-1] < 1e-3 * h[0] and all(b < 0.5 * a for a, b in zip(h, h[1:]))
    res = so.LevelData(levels[1].grids, 1, (0, 0, 0))
    op.residual(res, phi, rhs, True)
    jres = max(float(np.max(np.abs(res[i].view(g) / op.Jinv[i].view(g)))) for i, g in enumerate(levels[1].grids))
    assert abs(jres - h[-1]) <= 1e-9 * h[0]


def _mixed_levels():
    """level 1, refined by (2, 2, 2): one box spanning the whole depth (Neumann-Neumann, part of the horizontal problem) next
    to one that covers the lower half only (Neumann bottom, coarse-fine top: LAPACK line solves, no horizontal part)"""
    from oracle import somar_amr as sa
    from tests.helpers import make_amr_levels
    n, ratios = (16, 16, 8), [(2, 2, 2)]
    fine = [[so.Box((8, 8, 0), (15, 23, 15)), so.Box((16, 8, 0), (23, 23, 7))]]
    levels = make_amr_levels(so, sa, n, (1.0, 1.0, 0.005), (False, False, False), ratios, fine, cbox=(8, 8, 8))
    return levels, ratios


def test_mixed_column_kinds_share_one_level_and_stall():
    """m_flatDI / m_flatDIComplement (LevelLepticSolver.cpp:318-333): the spanning box alone carries the excess, the flat
    problem and the extrusion; the half-depth box is solved column by column with dptsv.  As restated, the horizontal
    correction is added to the spanning columns only (addHorizontalCorrection loops m_flatDI, :1504), which leaves a jump
    across the fine-fine face to the non-spanning neighbour: the first order RAISES the residual and the later, purely
    vertical orders cannot remove it -- the solve stalls near a quarter of the initial residual on the spanning box while
    the half-depth box converges.  (Unpinned like everything here; the GPU path refuses such layouts.)  What must hold:
    the reported norm is the true J-weighted residual of the level operator."""
    from oracle import somar_amr as sa
    levels, ratios = _mixed_levels()
    comp = sa.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab())
    op = comp.ops[1]
    lep = sl.LevelLepticSolver(op, maxOrder=4, domainHeight=0.005)
    assert lep.doHorizSolve and lep.flatDI == [0]
    assert lep.vertBCTypes == [(sl.VBC_NEUM, sl.VBC_NEUM), (sl.VBC_NEUM, sl.VBC_CF)]
    assert len(lep.horizGrids) == 1 and not lep.horizRemoveAvg
    rhs = so.random_field(levels[1].grids, 9, domainBox=levels[1].domain.box)
    phi = so.LevelData(levels[1].grids, 1, (1, 1, 1))
    lep.solve(phi, rhs, True)
    h = lep.resNorms
    assert lep.horizSolves == 1 and h[1] > h[0] and 0.1 * h[0] < h[-1] < h[0]
    res = so.LevelData(levels[1].grids, 1, (0, 0, 0))
    op.residual(res, phi, rhs, True)
    per_box = [float(np.max(np.abs(res[i].view(g) / op.Jinv[i].view(g)))) for i, g in enumerate(levels[1].grids)]
    assert abs(max(per_box) - h[-1]) <= 1e-9 * h[0]
    assert per_box[1] < 1e-3 * h[0] < per_box[0]          # the dptsv columns converge, the spanning box does not
