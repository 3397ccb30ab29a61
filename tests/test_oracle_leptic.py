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


def _thin_problem(n=(32, 32, 8), box=(16, 16, 8), L=(1.0, 1.0, 0.02), seed=3):
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (False, False, False))
    grids = so.split_domain(dom.box, box)
    dx = tuple(L[d] / n[d] for d in range(3))
    Jgup, Jinv = so.make_diagonal_metric(grids, dx, L, 3, "stretched", domain=dom)
    rhs = so.random_field(grids, seed, domainBox=dom.box)
    so.remove_weighted_mean(rhs, Jinv)
    return dom, grids, dx, Jgup, Jinv, rhs


@pytest.mark.parametrize("L", [(1.0, 1.0, 0.02), (1.0, 1.0, 0.1)])
def test_leptic_orders_contract_on_a_thin_domain(L):
    dom, grids, dx, Jgup, Jinv, rhs = _thin_problem(L=L)
    amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv)
    lep = sl.LevelLepticSolver(amr.op, maxOrder=4, domainHeight=L[2])
    phi = so.LevelData(grids, 1, (1, 1, 1))
    status = lep.solve(phi, rhs)
    h = lep.resNorms
    assert status in (sl.EXIT_ITER, sl.EXIT_CONVERGE, sl.EXIT_HANG)
    assert lep.horizSolves == 1            # diagonal metric: the horizontal problem is solved at O(1) only
    # every order gains about eps^2 = (H/L * aspect of the cells)^2; the thinner, the faster
    assert h[1] < 0.2 * h[0] and h[2] < 0.5 * h[1]
    # the accumulated correction solves the original equation to the final residual
    res = so.LevelData(grids, 1, (0, 0, 0))
    amr.op.residual(res, phi, rhs, False)
    jres = max(float(np.max(np.abs(res[i].view(g) / Jinv[i].view(g)))) for i, g in enumerate(grids))
    assert jres <= 1.0000001 * h[-1] + 1e-30
    assert h[-1] < 1e-6 * h[0]


def test_leptic_agrees_with_multigrid_solution():
    dom, grids, dx, Jgup, Jinv, rhs = _thin_problem()
    amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv)
    amr.eps = 1e-12
    amr.iterMax = 40
    phi_mg = so.LevelData(grids, 1, (1, 1, 1))
    amr.solve(phi_mg, rhs, zeroPhi=True)
    lep = sl.LevelLepticSolver(amr.op, maxOrder=6, domainHeight=0.02)
    phi = so.LevelData(grids, 1, (1, 1, 1))
    lep.solve(phi, rhs)

    def demean(ld):
        v = np.concatenate([f.view(g).ravel() for g, f in zip(ld.grids, ld.fabs)])
        return v - v.mean()
    a, b = demean(phi), demean(phi_mg)
    assert np.max(np.abs(a - b)) < 1e-6 * np.max(np.abs(b))
