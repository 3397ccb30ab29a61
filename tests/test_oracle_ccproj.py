"""Known answers for the cell-centred level projection's restatement (oracle/somar_oracle.py: cell_to_edge,
set_wall_normal_flux, edge_to_cell, level_divergence_cc, cc_level_project).  The reference holds no fixtures for it
 and CellToEdge / EdgeToCell are Chombo's (EXTERNAL): parity unpinned w.r.t. reference tests, pinned here
by exact arithmetic identities and the symbol of the approximate projection."""
import numpy as np
import pytest

from helpers import make_oracle_solver, make_problem


def _index_field(so, grids, ghost, ndim=3):
    """comp d holds 2*i_d + 1 (the cell centre in half-cell units), ghosts included"""
    u = so.LevelData(grids, ndim, ghost)
    for f in u.fabs:
        idx = np.meshgrid(*[np.arange(f.box.lo[a], f.box.hi[a] + 1) for a in range(3)], indexing="ij")
        for d in range(ndim):
            f.a[..., d] = 2.0 * idx[d] + 1.0
    return u


def test_cell_to_edge_and_back_are_exact_on_linear_fields(oracle):
    so = oracle
    dom, grids, dx, Jgup, Jinv = make_problem(so, (16, 8, 8), (8, 4, 8), "cartesian")
    u = _index_field(so, grids, (1, 1, 1))
    e = so.FluxData(grids, 1, 3, fill=np.nan)
    so.cell_to_edge(u, e)
    for i, g in enumerate(grids):
        for d in range(3):
            fb = g.faces(d)
            idx = np.meshgrid(*[np.arange(fb.lo[a], fb.hi[a] + 1) for a in range(3)], indexing="ij")
            np.testing.assert_array_equal(e[i][d].a[..., 0], 2.0 * idx[d])     # the face position, exactly
    back = so.LevelData(grids, 3, (0, 0, 0), np.nan)
    so.edge_to_cell(e, back)
    for i, g in enumerate(grids):
        np.testing.assert_array_equal(back[i].a, u[i].view(g))


def test_cell_to_edge_needs_the_ghost_layer(oracle):
    """with no ghosts the faces on the box boundary have only one cell in the FAB and are left alone (Chombo's edgeBox)"""
    so = oracle
    dom, grids, dx, Jgup, Jinv = make_problem(so, (8, 8, 8), 8, "cartesian")
    u = _index_field(so, grids, (0, 0, 0))
    e = so.FluxData(grids, 1, 3, fill=-7.0)
    so.cell_to_edge(u, e)
    f = e[0][0].a[..., 0]
    assert np.all(f[0] == -7.0) and np.all(f[-1] == -7.0) and np.all(f[1:-1] != -7.0)


@pytest.mark.parametrize("periodic", [(False, False, False), (True, False, True)])
def test_wall_bc_zeroes_exactly_the_physical_boundary_faces(oracle, periodic):
    so = oracle
    dom, grids, dx, Jgup, Jinv = make_problem(so, (16, 8, 8), (8, 4, 8), "cartesian", periodic)
    e = so.FluxData(grids, 1, 3, fill=1.0)
    so.set_wall_normal_flux(e, grids, dom)
    for i, g in enumerate(grids):
        for d in range(3):
            fb = g.faces(d)
            idx = np.meshgrid(*[np.arange(fb.lo[a], fb.hi[a] + 1) for a in range(3)], indexing="ij")[d]
            wall = (idx == dom.box.lo[d]) | (idx == dom.box.hi[d] + 1)
            want = np.where(wall & (not periodic[d]), 0.0, 1.0)
            np.testing.assert_array_equal(e[i][d].a[..., 0], want)


def test_inflow_outflow_sides_channel_flow_is_divergence_free_and_carries_the_inflow(oracle):
    """BasicVelocityBCGhostClass with an inflow side (x lo, value U) and an outflow side (x hi, order-0 extrapolation):
    a uniform x-velocity U then has zero divergence in EVERY cell (solid walls would leave a source / sink layer at both
    ends), the inflow faces hold U, the outflow faces the value of the next face inside, the other walls stay solid"""
    so = oracle
    dom, grids, dx, Jgup, Jinv = make_problem(so, (16, 8, 8), (8, 4, 8), "cartesian", (False, False, False))
    U = 0.75
    u = so.LevelData(grids, 3, (1, 1, 1), 0.0)
    for f in u.fabs:
        f.a[..., 0] = U
    kind, value = [1, 2, 0, 0, 0, 0], [U, 0.0, 0.0, 0.0, 0.0, 0.0]
    div = so.LevelData(grids, 1, (0, 0, 0), np.nan)
    edge = so.level_divergence_cc(div, u, Jinv, grids, dom, dx, velbc=(kind, value))
    assert all(np.all(f.a == 0.0) for f in div.fabs)
    for i, g in enumerate(grids):
        assert np.all(edge[i][0].a == U)
        assert np.all(edge[i][1].a == 0.0) and np.all(edge[i][2].a == 0.0)
    so.level_divergence_cc(div, u, Jinv, grids, dom, dx)            # solid walls at both ends: a sink and a source layer
    tot = sum(float(f.a.sum()) for f in div.fabs)
    assert abs(tot) < 1e-12 and any(np.any(f.a != 0.0) for f in div.fabs)
    # outflow copies the next face inside, not the cell average
    e = so.FluxData(grids, 1, 3)
    rng = np.random.default_rng(1)
    for i in range(len(grids)):
        for d in range(3):
            e[i][d].a[...] = rng.uniform(1.0, 2.0, e[i][d].a.shape)
    ref = [[e[i][d].a.copy() for d in range(3)] for i in range(len(grids))]
    so.set_normal_flux_bc(e, grids, dom, kind, value)
    for i, g in enumerate(grids):
        a, r = e[i][0].a[..., 0], ref[i][0][..., 0]
        if g.lo[0] == dom.box.lo[0]:
            assert np.all(a[0] == U)
        else:
            np.testing.assert_array_equal(a[0], r[0])
        if g.hi[0] == dom.box.hi[0]:
            np.testing.assert_array_equal(a[-1], r[-2])
        else:
            np.testing.assert_array_equal(a[-1], r[-1])
        np.testing.assert_array_equal(a[1:-1], r[1:-1])


def test_constant_flux_is_divergence_free_except_next_to_walls(oracle):
    so = oracle
    dom, grids, dx, Jgup, Jinv = make_problem(so, (16, 8, 8), (8, 4, 8), "stretched", (True, False, True))
    u = so.LevelData(grids, 3, (1, 1, 1), 1.5)
    div = so.LevelData(grids, 1, (0, 0, 0), np.nan)
    so.level_divergence_cc(div, u, Jinv, grids, dom, dx)
    for g, f in zip(grids, div.fabs):
        j = np.arange(g.lo[1], g.hi[1] + 1)
        inner = (j != dom.box.lo[1]) & (j != dom.box.hi[1])
        assert np.all(f.a[:, inner, :, 0] == 0.0)
        assert np.all(f.a[:, ~inner, :, 0] != 0.0)
    so.level_divergence_cc(div, u, Jinv, grids, dom, dx, wall=False)   # a_fluxBC = NULL: nothing special at walls
    assert all(np.all(f.a == 0.0) for f in div.fabs)


def test_approximate_projection_removes_smooth_divergence_by_its_symbol(oracle):
    """Cartesian, periodic: for the mode exp(i k x) the cell-centred D.G has symbol -(sin(kh)/h)^2, the solved operator
    -(2 sin(kh/2)/h)^2, so one projection leaves the fraction 1 - cos^2(kh/2) = sin^2(kh/2) of the divergence."""
    so = oracle
    n = 32
    dom, grids, dx, Jgup, Jinv = make_problem(so, n, 16, "cartesian", (True, True, True))
    amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv)
    amr.eps = 1e-12
    u = so.LevelData(grids, 3, (1, 1, 1))
    for f in u.fabs:
        I = np.arange(f.box.lo[0], f.box.hi[0] + 1)[:, None, None]
        f.a[..., 0] = np.sin(2 * np.pi * (I + 0.5) / n) * np.ones(f.box.size())
    div0 = so.LevelData(grids, 1)
    so.level_divergence_cc(div0, u, Jinv, grids, dom, dx)
    phi = so.LevelData(grids, 1, (1, 1, 1))
    so.cc_level_project(amr, u, phi, 1.0)
    so.exchange(u, dom, (1, 1, 1))
    div1 = so.LevelData(grids, 1)
    so.level_divergence_cc(div1, u, Jinv, grids, dom, dx)
    before = max(float(np.max(np.abs(f.a))) for f in div0.fabs)
    after = max(float(np.max(np.abs(f.a))) for f in div1.fabs)
    assert after / before == pytest.approx(np.sin(np.pi / n) ** 2, rel=1e-3)
