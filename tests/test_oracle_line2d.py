"""Known answers for the oracle's LineGSRBIter2D restatement (oracle/kernels.c::orc_linegsrbiter2d, GSRBF.ChF:1529-1724)."""
import numpy as np
import pytest

from oracle import somar_oracle as so


def _op(metric, n=(32, 16), bx=16, L=(1.0, 0.2)):
    dom = so.Domain(so.Box((0, 0, 0), (n[0] - 1, n[1] - 1, 0)), (False, False, False))
    grids = so.split_domain(dom.box, (bx, n[1], 1))
    dx = (L[0] / n[0], L[1] / n[1], 1.0)
    full = metric == "sheared"
    if full:
        Jgup, Jinv = so.make_full_metric_2d(grids, dx, L, dom, amp=(0.05, 0.04))
    else:
        Jgup, Jinv = so.make_diagonal_metric(grids, dx, (L[0], L[1], 1.0), 2, metric, domain=dom)
    fac = so.Factory(dom, grids, dx, so.BCHolder(), Jgup, Jinv, isDiagonal=not full, ndim=2, relaxMode=so.RELAX_LINE_GSRB)
    return dom, grids, Jinv, fac


@pytest.mark.parametrize("metric", ["stretched"])
def test_exact_solution_is_a_fixed_point_of_the_line_sweep(metric):
    """rhs := L[phi] => every column solve returns the column it started from (the lagged terms are consistent).
    Diagonal metric only: with cross terms LineGSRBIter2D keeps J g^{10} d_x(phi) on the Neumann END faces of a column
    (it differences the extrapolated copy there, GSRBF.ChF:1617-1626, 1688-1697) while the operator's boundary flux is zero,
    so the exact solution is NOT its fixed point -- the reference's line smoother is inconsistent at those faces, which is
    why its non-diagonal solves hang near 1e-3 (oracle and GPU alike, tests/test_gpu_line2d.py)."""
    dom, grids, Jinv, fac = _op(metric)
    op = fac.mg_new_op(0, None)
    phi = so.random_field(grids, 3, (1, 1, 0), dom.box)
    rhs = so.LevelData(grids, 1)
    op.apply_op(rhs, phi, True)
    before = [f.view(g).copy() for g, f in zip(grids, phi.fabs)]
    op.relax(phi, rhs, 1)
    for g, f, b in zip(grids, phi.fabs, before):
        np.testing.assert_allclose(f.view(g), b, rtol=0, atol=2e-11 * np.max(np.abs(b)))


def test_line_sweeps_beat_point_sweeps_on_a_thin_domain():
    """dy << dx: one V-cycle's worth of vertical-line sweeps reduces the residual far more than point GSRB"""
    out = {}
    for mode in (so.RELAX_LINE_GSRB, so.RELAX_LEVEL_GSRB):
        dom = so.Domain(so.Box((0, 0, 0), (31, 15, 0)), (False, False, False))
        grids = so.split_domain(dom.box, (16, 16, 1))
        dx = (1.0 / 32, 0.02 / 16, 1.0)
        Jgup, Jinv = so.make_diagonal_metric(grids, dx, (1.0, 0.02, 1.0), 2, "cartesian", domain=dom)
        fac = so.Factory(dom, grids, dx, so.BCHolder(), Jgup, Jinv, ndim=2, relaxMode=mode)
        amr = so.AMRMultiGrid(fac, so.BiCGStab())
        rhs = so.random_field(grids, 5, (0, 0, 0), dom.box)
        so.remove_weighted_mean(rhs, Jinv)
        phi = so.LevelData(grids, 1, (1, 1, 0))
        amr.solve(phi, rhs)
        out[mode] = amr.iters
    assert out[so.RELAX_LINE_GSRB] < out[so.RELAX_LEVEL_GSRB]


def test_region_must_start_at_the_bottom():
    dom = so.Domain(so.Box((0, 0, 0), (15, 15, 0)), (False, False, False))
    grids = [so.Box((0, 0, 0), (15, 7, 0)), so.Box((0, 8, 0), (15, 15, 0))]
    dx = (1.0 / 16, 1.0 / 16, 1.0)
    Jgup, Jinv = so.make_diagonal_metric(grids, dx, (1.0, 1.0, 1.0), 2, "cartesian", domain=dom)
    fac = so.Factory(dom, grids, dx, so.BCHolder(), Jgup, Jinv, ndim=2, relaxMode=so.RELAX_LINE_GSRB, maxDepth=0)
    op = fac.mg_new_op(0, None)
    phi = so.LevelData(grids, 1, (1, 1, 0))
    rhs = so.LevelData(grids, 1)
    with pytest.raises(AssertionError, match="INFO = -2"):
        op.relax(phi, rhs, 1)
