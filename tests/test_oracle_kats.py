"""Known-answer tests that pin the CPU oracle.

The reference has no tests of its own for this path; these are analytic facts
that any faithful restatement of the reference's kernels must satisfy.
"""
import ctypes as C

import numpy as np
import pytest


def _setup(so, n, boxsz, variant, periodic=(False, False, False), L=(1.0, 1.0, 1.0), ndim=3):
    n = so._iv(n)
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), periodic)
    grids = so.split_domain(dom.box, boxsz)
    dx = tuple(L[d] / n[d] for d in range(3))
    Jgup, Jinv = so.make_diagonal_metric(grids, dx, L, ndim, variant)
    return dom, grids, dx, Jgup, Jinv


def _op(so, dom, grids, dx, Jgup, Jinv, alpha=0.0, beta=1.0, ndim=3, relax=1):
    fac = so.Factory(dom, grids, dx, so.BCHolder(), Jgup, Jinv, alpha=alpha, beta=beta, ndim=ndim, relaxMode=relax)
    return fac, fac.mg_new_op(0, None)


@pytest.mark.parametrize("variant", ["cartesian", "stretched"])
@pytest.mark.parametrize("periodic", [(False, False, False), (False, True, False), (True, True, True)])
def test_k1_constant_in_null_space(oracle, variant, periodic):
    so = oracle
    dom, grids, dx, Jgup, Jinv = _setup(so, (16, 12, 8), 4, variant, periodic, (2.0, 1.0, 0.5))
    fac, op = _op(so, dom, grids, dx, Jgup, Jinv)
    phi = so.LevelData(grids, 1, (1, 1, 1), 3.75)
    lhs = so.LevelData(grids, 1)
    op.apply_op(lhs, phi, True)
    assert so.ld_norm(lhs, 0) < 1e-10 * 3.75 / min(dx) ** 2 * 1e-3
    assert op.zeroAvg  # factory probe, MappedAMRPoissonOpFactory.cpp:659-693


@pytest.mark.parametrize("periodic", [(False, False, False), (True, False, True)])
def test_k2_cosine_modes_are_eigenvectors(oracle, periodic):
    so = oracle
    n = (16, 8, 12)
    L = (1.0, 2.0, 0.75)
    dom, grids, dx, Jgup, Jinv = _setup(so, n, (8, 4, 6), "cartesian", periodic, L)
    fac, op = _op(so, dom, grids, dx, Jgup, Jinv)
    kmode = (3, 1, 2)
    idx = [np.arange(n[d]) for d in range(3)]
    f, lam = [], 0.0
    for d in range(3):
        if periodic[d]:
            f.append(np.cos(2 * np.pi * kmode[d] * idx[d] / n[d]))
            lam += (2 * np.cos(2 * np.pi * kmode[d] / n[d]) - 2) / dx[d] ** 2
        else:
            f.append(np.cos(np.pi * kmode[d] * (idx[d] + 0.5) / n[d]))
            lam += (2 * np.cos(np.pi * kmode[d] / n[d]) - 2) / dx[d] ** 2
    full = f[0][:, None, None] * f[1][None, :, None] * f[2][None, None, :]
    phi = so.LevelData(grids, 1, (1, 1, 1))
    for g, fab in zip(grids, phi.fabs):
        fab.view(g)[..., 0] = full[g.slices((0, 0, 0))]
    lhs = so.LevelData(grids, 1)
    op.apply_op(lhs, phi, True)
    for g, fab in zip(grids, lhs.fabs):
        np.testing.assert_allclose(fab.view(g)[..., 0], lam * full[g.slices((0, 0, 0))], rtol=0, atol=1e-10 * abs(lam))


def test_k3_lapdiag_is_unit_impulse_coefficient(oracle):
    so = oracle
    dom, grids, dx, Jgup, Jinv = _setup(so, (8, 8, 8), 8, "stretched", (True, True, True))
    fac, op = _op(so, dom, grids, dx, Jgup, Jinv)
    phi = so.LevelData(grids, 1, (1, 1, 1))
    lhs = so.LevelData(grids, 1)
    for cell in [(3, 4, 5), (0, 0, 0), (7, 2, 7)]:
        so.ld_set(phi, 0.0)
        phi[0].view(so.Box(cell, cell))[...] = 1.0
        op.apply_op(lhs, phi, True)
        got = lhs[0].view(so.Box(cell, cell))[0, 0, 0, 0]
        want = op.lapDiag[0].view(so.Box(cell, cell))[0, 0, 0, 0]
        assert abs(got - want) <= 1e-12 * abs(want)


@pytest.mark.parametrize("variant", ["cartesian", "stretched"])
def test_k4_exact_solution_is_gsrb_fixed_point(oracle, variant):
    so = oracle
    dom, grids, dx, Jgup, Jinv = _setup(so, (12, 8, 8), 4, variant, (False, True, False))
    fac, op = _op(so, dom, grids, dx, Jgup, Jinv)
    phi = so.random_field(grids, 5, (1, 1, 1), dom.box)
    rhs = so.LevelData(grids, 1)
    op.apply_op(rhs, phi, True)           # rhs := L[phi] exactly (discretely)
    before = [f.view(g).copy() for g, f in zip(grids, phi.fabs)]
    op.relax(phi, rhs, 2)
    for g, f, b in zip(grids, phi.fabs, before):
        np.testing.assert_allclose(f.view(g), b, rtol=0, atol=5e-12)


def test_k5_restrict_prolong(oracle):
    so = oracle
    dom, grids, dx, Jgup, Jinv = _setup(so, (16, 16, 8), 8, "cartesian")
    fac = so.Factory(dom, grids, dx, so.BCHolder(), Jgup, Jinv)
    mg = so.MultiGrid(fac, so.BiCGStab())
    op = mg.ops[0]
    r = op.mgCrseRefRatio
    assert r == (2, 2, 1)   # dx = (1/16, 1/16, 1/8): only x,y satisfy dx <= max(dx)/2
    # restriction with Jinv == 1 and phi == 0: coarse = block mean of rhs
    rhs = so.random_field(grids, 11, (0, 0, 0), dom.box)
    phi = so.LevelData(grids, 1, (1, 1, 1))
    crse = op.create_coarser(rhs)
    op.restrict_residual(crse, phi, rhs)
    for g, cf, ff in zip(grids, crse.fabs, rhs.fabs):
        a = ff.view(g)[..., 0]
        m = a.reshape(a.shape[0] // r[0], r[0], a.shape[1] // r[1], r[1], a.shape[2] // r[2], r[2]).mean(axis=(1, 3, 5))
        np.testing.assert_allclose(cf.view(g.coarsen(r))[..., 0], m, rtol=0, atol=1e-14)
    # prolongation adds a constant per block; zero-avg variant leaves zero J-weighted mean
    corr = so.random_field(crse.grids, 12, (1, 1, 1), dom.box.coarsen(r))
    fine = so.LevelData(grids, 1, (1, 1, 1))
    op.zeroAvg = False
    op.prolong_increment(fine, corr)
    for g, ff, cf in zip(grids, fine.fabs, corr.fabs):
        want = np.repeat(np.repeat(np.repeat(cf.view(g.coarsen(r))[..., 0], r[0], 0), r[1], 1), r[2], 2)
        np.testing.assert_array_equal(ff.view(g)[..., 0], want)
    dom2, grids2, dx2, Jgup2, Jinv2 = _setup(so, (16, 16, 8), 8, "stretched")
    fac2 = so.Factory(dom2, grids2, dx2, so.BCHolder(), Jgup2, Jinv2)
    mg2 = so.MultiGrid(fac2, so.BiCGStab())
    op2 = mg2.ops[0]
    assert op2.zeroAvg
    fine2 = so.random_field(grids2, 13, (1, 1, 1), dom2.box)
    op2.prolong_increment(fine2, corr)
    num = sum(float(np.sum(f.view(g) / j.view(g))) for g, f, j in zip(grids2, fine2.fabs, Jinv2.fabs))
    den = sum(float(np.sum(1.0 / j.view(g))) for g, j in zip(grids2, Jinv2.fabs))
    assert abs(num / den) < 1e-14


def test_k6_tridiag_vs_lapack(oracle):
    so = oracle
    from scipy.linalg import lapack
    rng = np.random.default_rng(3)
    for n in (4, 17, 64, 128):
        a = -rng.uniform(0.5, 1.0, n - 1)
        c = -rng.uniform(0.5, 1.0, n - 1)
        b = 2.5 + rng.uniform(0.0, 1.0, n)
        d = rng.uniform(-1, 1, n)
        x = np.zeros(n)
        P = C.POINTER(C.c_double)
        so.lib().orc_solve_tridiag(a.ctypes.data_as(P), b.ctypes.data_as(P), c.ctypes.data_as(P),
                                   d.ctypes.data_as(P), x.ctypes.data_as(P), n)
        _, _, _, xl, info = lapack.dgtsv(a, b, c, d)
        assert info == 0
        np.testing.assert_allclose(x, xl, rtol=1e-12, atol=1e-14)


def test_k7_operator_is_J_symmetric(oracle):
    so = oracle
    dom, grids, dx, Jgup, Jinv = _setup(so, (8, 12, 8), 4, "stretched", (False, False, True))
    fac, op = _op(so, dom, grids, dx, Jgup, Jinv)
    u = so.random_field(grids, 21, (1, 1, 1), dom.box)
    v = so.random_field(grids, 22, (1, 1, 1), dom.box)
    Lu, Lv = so.LevelData(grids, 1), so.LevelData(grids, 1)
    op.apply_op(Lu, u, True)
    op.apply_op(Lv, v, True)
    a = sum(float(np.sum(fu.view(g) * fl.view(g) / j.view(g))) for g, fu, fl, j in zip(grids, u.fabs, Lv.fabs, Jinv.fabs))
    b = sum(float(np.sum(fv.view(g) * fl.view(g) / j.view(g))) for g, fv, fl, j in zip(grids, v.fabs, Lu.fabs, Jinv.fabs))
    assert abs(a - b) <= 1e-11 * max(abs(a), abs(b))


def test_k8_neumann_ghost_kills_boundary_flux(oracle):
    so = oracle
    dom, grids, dx, Jgup, Jinv = _setup(so, (8, 8, 8), 8, "stretched")
    fac, op = _op(so, dom, grids, dx, Jgup, Jinv)
    phi = so.random_field(grids, 31, (1, 1, 1), dom.box)
    so.bc_set_ghosts(op.bc, phi[0], None, grids[0], dom, dx, Jgup[0], True, True, 3)
    for d in range(3):
        flux = so.Fab(grids[0].faces(d), 1)
        op.get_flux_complete(flux, phi[0], None, grids[0].faces(d), 0, d)
        sl = [slice(None)] * 3
        sl[d] = 0
        assert np.max(np.abs(flux.a[tuple(sl)])) == 0.0
        sl[d] = -1
        assert np.max(np.abs(flux.a[tuple(sl)])) == 0.0


@pytest.mark.parametrize("periodic", [(False, False, False), (False, True, False)])
@pytest.mark.parametrize("boxsz", [4, 8, (16, 4, 8)])
def test_k9_interior_plus_boundary_boxes_tile_each_box_once(oracle, periodic, boxsz):
    so = oracle
    dom = so.Domain(so.Box((0, 0, 0), (15, 7, 7)), periodic)
    grids = so.split_domain(dom.box, boxsz)
    bc = so.BCHolder()
    act = (1, 1, 1)
    count = so.LevelData(grids, 1)
    domInt = dom.box.grow((-1, -1, -1))
    for g, f in zip(grids, count.fabs):
        r = g & domInt
        if not r.isEmpty():
            f.view(r)[...] += 1
    for e in so.collect_boundary_data(grids, dom, bc, act, False):
        count[e.index].view(e.validBdry)[...] += 1
        # a face flagged Neumann must really sit on a non-periodic domain face
        for d in range(3):
            for s in (0, 1):
                if e.stencil[d][s] == so.BC_NEUM:
                    assert not periodic[d]
                    assert (e.validBdry.lo[d] if s == 0 else e.validBdry.hi[d]) == (dom.box.lo[d] if s == 0 else dom.box.hi[d])
    for f in count.fabs:
        assert np.all(f.a == 1.0)
    # simple (all box boundaries) variant used by LooseGSRB
    count2 = so.LevelData(grids, 1)
    for g, f in zip(grids, count2.fabs):
        r = g.grow((-1, -1, -1))
        if not r.isEmpty():
            f.view(r)[...] += 1
    for e in so.collect_boundary_data(grids, dom, bc, act, True):
        count2[e.index].view(e.validBdry)[...] += 1
    for f in count2.fabs:
        assert np.all(f.a == 1.0)


def test_gsrb_is_layout_independent(oracle):
    """LevelGSRB exchanges before each colour, so one sweep on 1 box == on 8 boxes, bitwise."""
    so = oracle
    outs = []
    for boxsz in (16, 8, (4, 8, 16)):
        dom, grids, dx, Jgup, Jinv = _setup(so, (16, 16, 16), boxsz, "stretched", (False, True, False))
        fac, op = _op(so, dom, grids, dx, Jgup, Jinv)
        phi = so.random_field(grids, 41, (1, 1, 1), dom.box)
        rhs = so.random_field(grids, 42, (0, 0, 0), dom.box)
        op.relax(phi, rhs, 2)
        full = np.zeros(dom.box.size())
        for g, f in zip(grids, phi.fabs):
            full[g.slices((0, 0, 0))] = f.view(g)[..., 0]
        outs.append(full)
    np.testing.assert_array_equal(outs[0], outs[1])
    np.testing.assert_array_equal(outs[0], outs[2])


def test_vcycle_contraction_and_stopping(oracle):
    """Convergence monitor of MappedAMRMultiGrid.H:1104-1145 as a regression: Cartesian
    Neumann 32^3 contracts by > 10x per V-cycle and stops on goRedu (exitStatus 1)."""
    so = oracle
    dom, grids, dx, Jgup, Jinv = _setup(so, 32, 32, "cartesian")
    fac = so.Factory(dom, grids, dx, so.BCHolder(), Jgup, Jinv)
    amr = so.AMRMultiGrid(fac, so.BiCGStab())
    assert amr.mg.mgRefRatios == [(2, 2, 2)] * 3
    rhs = so.random_field(grids, 12345, domainBox=dom.box)
    so.remove_weighted_mean(rhs, Jinv)
    phi = so.LevelData(grids, 1, (1, 1, 1))
    amr.solve(phi, rhs)
    h = amr.history
    assert amr.exitStatus == 1 and amr.iters == 5
    assert all(h[i + 1] < 0.1 * h[i] for i in range(len(h) - 1))


def test_semicoarsening_rule(oracle):
    """MappedAMRPoissonOpFactory.cpp:476-495 on the lock-exchange aspect ratio (L=15x3x2,
    n=64x96x64 shipped deck: dx = .234,.03125,.03125) coarsens y,z first."""
    so = oracle
    assert so.choose_mg_ref_ratio((15 / 64, 3 / 96, 2 / 64), 3) == (1, 2, 2)
    assert so.choose_mg_ref_ratio((0.1, 0.1, 0.1), 3) == (2, 2, 2)
    assert so.choose_mg_ref_ratio((0.1, 0.06, 0.1), 3) == (2, 2, 2)   # nothing <= max/2 -> isotropic
    assert so.choose_mg_ref_ratio((0.1, 0.05, 0.1), 3) == (1, 2, 1)
    assert so.choose_mg_ref_ratio((0.1, 0.05, 1.0), 2) == (1, 2, 1)


# ---- Dirichlet sides (setSideDiriBC + ELLIPTICCONSTDIRIBCGHOST, order 1): the viscous / diffusive Helmholtz solves -------
def _diri_setup(so, types, values=None, n=(16, 16, 8), alpha=0.0, beta=1.0, variant="cartesian"):
    from helpers import make_problem
    dom, grids, dx, Jgup, Jinv = make_problem(so, n, 8, variant, (False, False, False), (1.0, 1.0, 0.5))
    bc = so.BCHolder([list(t) for t in types], [list(v) for v in values] if values else None)
    fac = so.Factory(dom, grids, dx, bc, Jgup, Jinv, alpha=alpha, beta=beta)
    return dom, grids, dx, Jinv, so.AMRMultiGrid(fac, so.BiCGStab())


def test_dirichlet_sides_remove_the_null_space_and_the_solve_converges(oracle):
    so = oracle
    D, N = so.BC_DIRI, so.BC_NEUM
    dom, grids, dx, Jinv, amr = _diri_setup(so, [(D, D), (N, N), (N, D)], variant="stretched")
    assert not any(op.zeroAvg for op in amr.mg.ops)
    phi0 = so.random_field(grids, 3, (1, 1, 1), dom.box)
    b = so.LevelData(grids, 1)
    amr.op.apply_op(b, phi0, True)
    x = so.LevelData(grids, 1, (1, 1, 1))
    amr.solve(x, b, forceHomogeneous=True)
    assert amr.exitStatus == 1 and amr.history[-1] <= 1e-6 * amr.history[0]
    # unique solution: x == phi0 up to the solver tolerance
    err = max(float(np.max(np.abs(a.view(g) - c.view(g)))) for g, a, c in zip(grids, x.fabs, phi0.fabs))
    assert err < 1e-4


def test_inhomogeneous_dirichlet_values_reproduce_a_linear_profile(oracle):
    """Cartesian metric, Dirichlet values 1 and 3 on the x sides, Neumann elsewhere, rhs = 0: the discrete solution
    is the linear profile (the order-1 ghost 2 bcval - phi is exact for it)."""
    so = oracle
    D, N = so.BC_DIRI, so.BC_NEUM
    dom, grids, dx, Jinv, amr = _diri_setup(so, [(D, D), (N, N), (N, N)], [(1.0, 3.0), (0, 0), (0, 0)])
    amr.eps = 1e-10
    rhs = so.LevelData(grids, 1)
    x = so.LevelData(grids, 1, (1, 1, 1))
    amr.solve(x, rhs, zeroPhi=True, forceHomogeneous=False)
    for g, f in zip(grids, x.fabs):
        X = (np.arange(g.lo[0], g.hi[0] + 1) + 0.5) * dx[0]
        want = 1.0 + 2.0 * X[:, None, None]
        np.testing.assert_allclose(f.view(g)[..., 0], np.broadcast_to(want, f.view(g)[..., 0].shape), atol=1e-8)


def test_helmholtz_with_dirichlet_walls_converges_fast(oracle):
    """alpha = 1, beta = -nu dt (a viscous backward-Euler step): strongly diagonally dominant."""
    so = oracle
    D = so.BC_DIRI
    dom, grids, dx, Jinv, amr = _diri_setup(so, [(D, D), (D, D), (D, D)], alpha=1.0, beta=-1e-3, variant="stretched")
    b = so.random_field(grids, 8, (0, 0, 0), dom.box)
    x = so.LevelData(grids, 1, (1, 1, 1))
    amr.solve(x, b, forceHomogeneous=True)
    assert amr.exitStatus == 1 and amr.iters <= 6
