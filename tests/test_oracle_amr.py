"""The AMR part of the oracle (oracle/somar_amr.py).  The reference ships no fixtures for these paths
("parity unpinned"), so the restatement is pinned by properties the scheme must have:
  * the quadratic coarse-fine interpolation reproduces quadratics to rounding,
  * the homogeneous CF interpolation is the quadratic through a zero coarse value,
  * the refluxed composite operator is conservative (its volume integral over a closed/periodic domain is 0),
  * composite and level solves converge with the reference's stopping logic."""
import numpy as np
import pytest

from helpers import make_amr_levels


@pytest.fixture(scope="module")
def am(oracle):
    from oracle import somar_amr
    return somar_amr


def _quad_error(so, am, periodic, cross, fboxes, r, n=(16, 8, 8)):
    cdom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), periodic)
    fdom = cdom.refine(r)
    cgr = so.split_domain(cdom.box, 8)
    dxc = (1 / 16, 1 / 8, 1 / 8)
    dxf = tuple(a / b for a, b in zip(dxc, r))
    q = am.QuadCFInterp(fboxes, cgr, dxf, r, fdom)

    def fun(x, y, z):
        return 1.0 + 0.3 * x + 0.2 * y - 0.4 * z + 0.5 * x * x - 0.25 * y * y + 0.125 * z * z + cross * y * z + 0.2 * cross * x * z

    def fill(ld, dx):
        for f in ld.fabs:
            b = f.box
            X, Y, Z = np.meshgrid(*[(np.arange(b.lo[d], b.hi[d] + 1) + 0.5) * dx[d] for d in range(3)], indexing="ij")
            f.a[..., 0] = fun(X, Y, Z)

    phic = so.LevelData(cgr, 1, (1, 1, 1))
    phif = so.LevelData(fboxes, 1, (1, 1, 1))
    fill(phic, dxc)
    fill(phif, dxf)
    exact = [f.a.copy() for f in phif.fabs]
    cf = am.CFRegion(fboxes, fdom)
    ncf = 0
    for (i, d, s), (gb, m) in cf.ivs.items():
        if m is not None:
            v = phif[i].view(gb)[..., 0]
            v[m] = 1e30
            ncf += int(m.sum())
    assert ncf > 0
    q.coarse_fine_interp(phif, phic)
    worst = 0.0
    for (i, d, s), (gb, m) in cf.ivs.items():
        if m is not None:
            got = phif[i].view(gb)[..., 0]
            want = exact[i][gb.slices(phif[i].box.lo) + (0,)]
            worst = max(worst, float((np.abs(got - want) * m).max()))
    return worst


def test_quadratic_cf_interpolation_is_exact_on_quadratics(oracle, am):
    so = oracle
    slab = so.split_domain(so.Box((8, 0, 0), (23, 15, 7)), (8, 8, 8))          # fine slab, walls in y and z
    island = [so.Box((8, 4, 2), (15, 11, 5)), so.Box((16, 4, 2), (23, 11, 5))]  # interior island
    island222 = [so.Box((8, 4, 4), (15, 11, 11)), so.Box((16, 4, 4), (23, 11, 11))]
    nope = (False, False, False)
    assert _quad_error(so, am, nope, 0.0, slab, (2, 2, 1)) < 5e-15
    assert _quad_error(so, am, nope, 0.3, slab, (2, 2, 1)) < 5e-15
    assert _quad_error(so, am, nope, 0.3, island, (2, 2, 1)) < 5e-15
    assert _quad_error(so, am, nope, 0.3, island222, (2, 2, 2)) < 5e-15
    assert _quad_error(so, am, (False, True, True), 0.3, island222, (2, 2, 2)) < 5e-15
    # ratio 4 (the reference's LockExchange inputs refine by (4,1,1)): same stencils, other fine-cell offsets
    slab411 = so.split_domain(so.Box((16, 0, 0), (47, 7, 7)), (16, 8, 8))
    island441 = [so.Box((16, 8, 2), (31, 23, 5)), so.Box((32, 8, 2), (47, 23, 5))]
    assert _quad_error(so, am, nope, 0.3, slab411, (4, 1, 1)) < 1e-14
    assert _quad_error(so, am, nope, 0.3, island441, (4, 4, 1)) < 1e-14


def test_one_sided_mixed_derivative_keeps_the_reference_sign(oracle, am):
    """MappedQuadCFStencil::buildStencils (MappedCFStencil.cpp:1180-1232) weights the one-sided mixed
    stencil -1,+1,+1,-1, the opposite sign of the centred one.  Restated as is: exact without cross terms,
    O(dx^2) off where a wall forces the one-sided stencil and the field has yz / xz cross terms."""
    so = oracle
    slab222 = so.split_domain(so.Box((8, 0, 0), (23, 15, 15)), (8, 8, 8))
    nope = (False, False, False)
    assert _quad_error(so, am, nope, 0.0, slab222, (2, 2, 2)) < 5e-15
    e = _quad_error(so, am, nope, 0.3, slab222, (2, 2, 2))
    assert 1e-5 < e < 1e-3


def test_homogeneous_cf_interp_is_quadratic_through_zero(oracle, am):
    so = oracle
    fdom = so.Domain(so.Box((0, 0, 0), (31, 15, 15)))
    grids = [so.Box((8, 0, 0), (23, 15, 15))]
    cf = am.CFRegion(grids, fdom)
    phi = so.random_field(grids, 3, (1, 1, 1), fdom.box)
    Df, Dc = 0.1, 0.2
    cf.homogeneous_cf_interp(phi, (Df, 0.3, 0.3), (Dc, 0.6, 0.6), (1, 1, 1))
    f = phi[0].a[..., 0]
    # hi side in x: interface at 0, fine centres -Df/2 (b), -3Df/2 (a), coarse centre +Dc/2 with value 0
    xs = np.array([-1.5 * Df, -0.5 * Df, 0.5 * Dc])
    pa, pb = f[-3, 1:-1, 1:-1], f[-2, 1:-1, 1:-1]
    want = np.empty_like(pa)
    for idx in np.ndindex(pa.shape):
        c = np.polyfit(xs, [pa[idx], pb[idx], 0.0], 2)
        want[idx] = np.polyval(c, 0.5 * Df)
    np.testing.assert_allclose(f[-1, 1:-1, 1:-1], want, rtol=1e-12, atol=1e-13)


LAYOUTS = [
    ((False, False, False), [(2, 2, 2)], [[((8, 8, 4), (23, 23, 11))]]),
    ((True, False, False), [(2, 2, 1)], [[((0, 8, 0), (15, 23, 7)), ((24, 8, 0), (31, 23, 7))]]),
    ((False, True, False), [(2, 2, 2)], [[((8, 0, 0), (23, 15, 7)), ((8, 16, 0), (15, 31, 7))]]),
    ((True, False, False), [(2, 2, 2), (2, 2, 1)], [[((8, 8, 4), (23, 23, 11))], [((24, 24, 6), (39, 39, 9))]]),
    # refinement by 4: mini V-cycles through one forced (2,1,1) / (2,2,1) depth
    ((False, True, False), [(4, 1, 1)], [[((16, 0, 0), (31, 15, 7)), ((32, 0, 0), (47, 15, 7))]]),
    ((False, False, False), [(4, 4, 1)], [[((16, 16, 0), (47, 47, 7))]]),
    ((True, False, False), [(2, 2, 1), (4, 1, 1)], [[((8, 8, 0), (23, 23, 7))], [((40, 12, 0), (71, 19, 7))]]),
]


def _composite(so, am, layout):
    periodic, ratios, boxes = layout
    fb = [[so.Box(lo, hi) for lo, hi in lev] for lev in boxes]
    levels = make_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.5), periodic, ratios, fb)
    return levels, am.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab())


@pytest.mark.parametrize("layout", LAYOUTS)
def test_refluxed_composite_operator_is_conservative_and_solve_converges(oracle, am, layout):
    so = oracle
    levels, comp = _composite(so, am, layout)
    lmax = len(levels) - 1
    phi = [so.random_field(L.grids, 5 + l, (1, 1, 1), L.domain.box) for l, L in enumerate(levels)]
    res = [so.LevelData(L.grids, 1) for L in levels]
    zero = [so.LevelData(L.grids, 1) for L in levels]
    comp.init(phi, zero, lmax, 0)
    comp.compute_amr_residual(res, phi, zero, lmax, 0, True)  # res = -L[phi], covered cells zeroed
    tot = 0.0
    for L, r in zip(levels, res):
        for i, g in enumerate(L.grids):
            tot += float((r[i].view(g)[..., 0] / L.Jinv[i].view(g)[..., 0]).sum()) * float(np.prod(L.dx))
    mag = max(so.ld_norm(r, 0) for r in res)
    assert abs(tot) < 1e-13 * mag
    # solve L[x] = L[phi]: compatible by construction
    rhs = [so.ld_create(r) for r in res]
    for a, b in zip(rhs, res):
        so.ld_assign(a, b)
        so.ld_scale(a, -1.0)
    sol = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
    comp.solve(sol, rhs, lmax, 0)
    h = comp.history
    if layout[1] == [(4, 4, 1)]:
        # 16-fold refinement of a 16 x 16 x 8 base behind piecewise-constant interpolation: contracts by only 0.75
        # per cycle and runs into iterMax (exit status 2) -- monotone, five orders in 20 cycles
        assert comp.exitStatus == 2 and h[-1] <= 2e-5 * h[0]
    else:
        assert comp.exitStatus == 1 and h[-1] <= 1e-6 * h[0]
    assert all(b < a for a, b in zip(h, h[1:]))


def test_level_solve_takes_its_cf_values_from_the_coarser_level(oracle, am):
    so = oracle
    levels, comp = _composite(so, am, LAYOUTS[0])
    coarse = so.random_field(levels[0].grids, 11, (1, 1, 1), levels[0].domain.box)
    rhs1 = so.random_field(levels[1].grids, 12, (0, 0, 0), levels[1].domain.box)
    phi1 = so.LevelData(levels[1].grids, 1, (1, 1, 1))
    comp.solve([coarse, phi1], [None, rhs1], 1, 1)
    assert comp.exitStatus == 1
    # the converged fine solution satisfies the inhomogeneous-CF residual equation
    res = so.LevelData(levels[1].grids, 1)
    comp.amr_residual_nf(1, res, phi1, coarse, rhs1, False)
    assert so.ld_norm(res, 0) <= 1e-6 * comp.history[0]
    # and it depends on the coarse data: a different coarse field gives a different answer
    phi2 = so.LevelData(levels[1].grids, 1, (1, 1, 1))
    so.ld_scale(coarse, 2.0)
    comp.solve([coarse, phi2], [None, rhs1], 1, 1)
    assert max(float(np.max(np.abs(a.view(g) - b.view(g)))) for g, a, b in zip(phi1.grids, phi1.fabs, phi2.fabs)) > 1e-3


# ---- CH_SPACEDIM = 2 (the reference's 2-D lock-exchange decks refine by (4,1) and (4,2)) ---------------------------
def test_2d_quadratic_cf_interpolation_is_exact_on_quadratics(oracle, am):
    so = oracle
    for r, fboxes in (((2, 2, 1), [so.Box((16, 8, 0), (47, 23, 0))]),
                      ((4, 1, 1), [so.Box((32, 0, 0), (63, 15, 0)), so.Box((64, 0, 0), (95, 15, 0))]),
                      ((4, 2, 1), [so.Box((32, 8, 0), (95, 23, 0))])):
        cdom = so.Domain(so.Box((0, 0, 0), (31, 15, 0)), (False, False, False))
        fdom = cdom.refine(r)
        cgr = so.split_domain(cdom.box, (8, 8, 1))
        dxc = (1 / 16, 1 / 8, 1.0)
        dxf = tuple(a / b for a, b in zip(dxc, r))
        q = am.QuadCFInterp(fboxes, cgr, dxf, r, fdom, ndim=2)

        def fill(ld, dx):
            for f in ld.fabs:
                b = f.box
                X, Y = np.meshgrid(*[(np.arange(b.lo[d], b.hi[d] + 1) + 0.5) * dx[d] for d in range(2)], indexing="ij")
                f.a[:, :, 0, 0] = 1.0 + 0.3 * X + 0.2 * Y + 0.5 * X * X - 0.25 * Y * Y + 0.3 * X * Y

        phic = so.LevelData(cgr, 1, (1, 1, 0))
        phif = so.LevelData(fboxes, 1, (1, 1, 0))
        fill(phic, dxc)
        fill(phif, dxf)
        exact = [f.a.copy() for f in phif.fabs]
        cf = am.CFRegion(fboxes, fdom)
        ncf = 0
        for (i, d, s), (gb, m) in cf.ivs.items():
            if m is not None:
                phif[i].view(gb)[..., 0][m] = 1e30
                ncf += int(m.sum())
        assert ncf > 0
        q.coarse_fine_interp(phif, phic)
        for (i, d, s), (gb, m) in cf.ivs.items():
            if m is not None:
                got = phif[i].view(gb)[..., 0]
                want = exact[i][gb.slices(phif[i].box.lo) + (0,)]
                assert float((np.abs(got - want) * m).max()) < 1e-14, (r, i, d, s)


LAYOUTS_2D = [
    ((False, False, False), [(2, 2, 1)], [[((16, 8, 0), (47, 23, 0))]]),
    ((False, False, False), [(4, 1, 1)], [[((32, 0, 0), (95, 15, 0))]]),
    ((False, True, False), [(4, 1, 1)], [[((32, 0, 0), (63, 15, 0)), ((64, 0, 0), (95, 15, 0))]]),
]


@pytest.mark.parametrize("layout", LAYOUTS_2D)
def test_2d_composite_operator_is_conservative_and_solve_converges(oracle, am, layout):
    so = oracle
    periodic, ratios, boxes = layout
    fb = [[so.Box(lo, hi) for lo, hi in lev] for lev in boxes]
    levels = make_amr_levels(so, am, (32, 16, 1), (2.0, 1.0, 1.0), periodic, ratios, fb, cbox=(8, 8, 1), ndim=2)
    comp = am.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab(), ndim=2)
    g = (1, 1, 0)
    phi = [so.random_field(L.grids, 5 + l, g, L.domain.box) for l, L in enumerate(levels)]
    res = [so.LevelData(L.grids, 1) for L in levels]
    zero = [so.LevelData(L.grids, 1) for L in levels]
    comp.init(phi, zero, 1, 0)
    comp.compute_amr_residual(res, phi, zero, 1, 0, True)
    tot = 0.0
    for L, r in zip(levels, res):
        for i, gg in enumerate(L.grids):
            tot += float((r[i].view(gg)[..., 0] / L.Jinv[i].view(gg)[..., 0]).sum()) * float(np.prod(L.dx[:2]))
    assert abs(tot) < 1e-13 * max(so.ld_norm(r, 0) for r in res)
    rhs = [so.ld_create(r) for r in res]
    for a, b in zip(rhs, res):
        so.ld_assign(a, b)
        so.ld_scale(a, -1.0)
    sol = [so.LevelData(L.grids, 1, g) for L in levels]
    comp.solve(sol, rhs, 1, 0)
    h = comp.history
    assert comp.exitStatus == 1 and h[-1] <= 1e-6 * h[0]
    assert all(b < a for a, b in zip(h, h[1:]))


def test_nondiagonal_amr_path_reduces_to_the_diagonal_one(oracle, am):
    """AMRComposite(isDiagonal=False) runs fillExtrap / ExtrapolateCFEV / MAPPEDGETFLUX through interpolation, operator and
    refluxing; fed a DIAGONAL metric it must reproduce the 7-point composite residual bit for bit.  (On a sheared map
    the refluxed composite operator is conservative to round-off on a one-box coarse level; on multi-box levels it
    inherits the layout quirk of the non-diagonal Neumann ghost, tests/test_oracle_full.py.)"""
    so = oracle
    fb = [[so.Box((8, 8, 4), (23, 23, 11))]]
    lv = make_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.5), (False, False, False), [(2, 2, 2)], fb)
    out = []
    for diag in (True, False):
        comp = am.AMRComposite(lv, [(2, 2, 2)], so.BCHolder(), so.BiCGStab(), isDiagonal=diag)
        phi = [so.random_field(L.grids, 5 + l, (1, 1, 1), L.domain.box) for l, L in enumerate(lv)]
        res = [so.LevelData(L.grids, 1) for L in lv]
        zero = [so.LevelData(L.grids, 1) for L in lv]
        comp.init(phi, zero, 1, 0)
        comp.compute_amr_residual(res, phi, zero, 1, 0, True)
        out.append(res)
    for a, b, L in zip(out[0], out[1], lv):
        for i, g in enumerate(L.grids):
            np.testing.assert_array_equal(a[i].view(g), b[i].view(g))


def test_nondiagonal_composite_operator_is_conservative_on_a_one_box_coarse_level(oracle, am):
    """Interior fine regions only: where a coarse-fine face ends on a physical wall the Neumann ghost's own extrapolation
    and ExtrapolateCFEV disagree about the corner cell, and the balance is off by ~3e-5 of the operator's magnitude
    (same family as the layout quirk; reproduced, not fixed)."""
    from helpers import make_full_amr_levels
    so = oracle
    for ndim, n, L, cbox, ratios, fb in (
            (2, (32, 16, 1), (2.0, 1.0, 1.0), (32, 16, 1), [(2, 2, 1)], [[so.Box((16, 8, 0), (47, 23, 0))]]),
            (2, (32, 16, 1), (2.0, 1.0, 1.0), (32, 16, 1), [(4, 1, 1)], [[so.Box((32, 4, 0), (95, 11, 0))]]),
            (3, (16, 16, 8), (2.0, 1.0, 0.5), (16, 16, 8), [(2, 2, 2)], [[so.Box((8, 8, 4), (23, 23, 11))]])):
        levels = make_full_amr_levels(so, am, n, L, (False, False, False), ratios, fb, cbox=cbox, ndim=ndim)
        comp = am.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab(), ndim=ndim, isDiagonal=False)
        G = (1, 1, 1) if ndim == 3 else (1, 1, 0)
        phi = [so.random_field(Lv.grids, 5 + l, G, Lv.domain.box) for l, Lv in enumerate(levels)]
        res = [so.LevelData(Lv.grids, 1) for Lv in levels]
        zero = [so.LevelData(Lv.grids, 1) for Lv in levels]
        comp.init(phi, zero, 1, 0)
        comp.compute_amr_residual(res, phi, zero, 1, 0, True)
        tot = 0.0
        for Lv, r in zip(levels, res):
            for i, gg in enumerate(Lv.grids):
                tot += float((r[i].view(gg)[..., 0] / Lv.Jinv[i].view(gg)[..., 0]).sum()) * float(np.prod(Lv.dx[:ndim]))
        assert abs(tot) < 1e-12 * max(so.ld_norm(r, 0) for r in res), (ndim, ratios, tot)
