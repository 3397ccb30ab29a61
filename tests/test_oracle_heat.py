"""Known answers for the viscous / diffusive Helmholtz restatement (oracle/somar_oracle.py: reset_solver_alpha_and_beta,
level_backward_euler, level_crank_nicolson).  The reference holds no fixtures: parity unpinned w.r.t.
reference tests; pinned here by the exact amplification factors of discrete eigenmodes and a steady state."""
import numpy as np
import pytest

from helpers import make_problem

D, N = 1, 0


def _cosine_mode(so, dom, grids, k):
    """Neumann eigenmode prod cos(pi k_a (i_a + 1/2) / n_a) and its eigenvalue magnitude on the unit-spacing-free grid"""
    n = dom.box.size()
    phi = so.LevelData(grids, 1, (1, 1, 1))
    for f in phi.fabs:
        idx = np.meshgrid(*[np.arange(f.box.lo[a], f.box.hi[a] + 1) for a in range(3)], indexing="ij")
        v = np.ones(f.box.size())
        for a in range(3):
            v = v * np.cos(np.pi * k[a] * (idx[a] + 0.5) / n[a])
        f.a[..., 0] = v
    return phi


def _solver(so, n, bs, nu, bc=None, L=(1.0, 1.0, 1.0)):
    dom, grids, dx, Jgup, Jinv = make_problem(so, n, bs, "cartesian", (False, False, False), L)
    fac = so.Factory(dom, grids, dx, bc if bc is not None else so.BCHolder(), Jgup, Jinv, alpha=1.0, beta=nu)
    amr = so.AMRMultiGrid(fac, so.BiCGStab())
    amr.eps = 1e-12
    amr.iterMax = 40
    return dom, grids, dx, amr


@pytest.mark.parametrize("scheme", ["be", "cn", "tga"])
def test_eigenmode_amplification_factor(oracle, scheme):
    so = oracle
    n, nu, dt, k = (16, 16, 8), 0.05, 0.3, (1, 2, 1)
    dom, grids, dx, amr = _solver(so, n, 8, nu)
    lam = sum((2.0 - 2.0 * np.cos(np.pi * k[a] / n[a])) / dx[a] ** 2 for a in range(3))
    old = _cosine_mode(so, dom, grids, k)
    src = so.LevelData(grids, 1)
    new = so.LevelData(grids, 1, (1, 1, 1))
    if scheme == "be":
        so.level_backward_euler(amr, new, old, src, dt)
        amp = 1.0 / (1.0 + dt * nu * lam)
    elif scheme == "cn":
        so.level_crank_nicolson(amr, new, old, src, dt)
        amp = (1.0 - 0.5 * dt * nu * lam) / (1.0 + 0.5 * dt * nu * lam)
    else:
        so.level_tga(amr, new, old, src, dt)
        mu1, mu2, mu3, mu4, _ = so.tga_coefficients()
        z = dt * nu * lam
        amp = (1.0 - mu3 * z) / ((1.0 + mu1 * z) * (1.0 + mu2 * z))
    for g, fn, fo in zip(grids, new.fabs, old.fabs):
        np.testing.assert_allclose(fn.view(g), amp * fo.view(g), rtol=0, atol=1e-9)


def test_crank_nicolson_source_enters_with_dt_and_backward_euler_ignores_it(oracle):
    so = oracle
    dom, grids, dx, amr = _solver(so, (16, 16, 8), 8, 0.05)
    old = so.LevelData(grids, 1, (1, 1, 1))            # phiOld = 0, constant source: L[const] = 0 under Neumann BCs
    src = so.LevelData(grids, 1, (0, 0, 0), 2.0)
    new = so.LevelData(grids, 1, (1, 1, 1))
    so.level_crank_nicolson(amr, new, old, src, 0.25)
    for g, f in zip(grids, new.fabs):
        np.testing.assert_allclose(f.view(g), 0.5, rtol=0, atol=1e-10)       # dt * src
    so.level_backward_euler(amr, new, old, src, 0.25)
    for g, f in zip(grids, new.fabs):
        np.testing.assert_allclose(f.view(g), 0.0, rtol=0, atol=1e-12)


def test_tga_coefficients_and_its_source_half(oracle):
    """mu1 + mu2 = a, mu1 mu2 = a - 1/2 (the TGA factorisation), mu3 = 1 - a, mu4 = 1/2 - a; a source that is an
    eigenmode comes back scaled by dt (1 - mu4 z) / ((1 + mu1 z)(1 + mu2 z))"""
    so = oracle
    mu1, mu2, mu3, mu4, r1 = so.tga_coefficients()
    a = 2.0 - np.sqrt(2.0) - 1e-12
    assert mu1 + mu2 == pytest.approx(a, abs=1e-15) and mu1 * mu2 == pytest.approx(a - 0.5, abs=1e-15)
    assert mu3 == 1.0 - a and mu4 == 0.5 - a
    n, nu, dt, k = (16, 16, 8), 0.05, 0.3, (2, 1, 0)
    dom, grids, dx, amr = _solver(so, n, 8, nu)
    lam = sum((2.0 - 2.0 * np.cos(np.pi * k[i] / n[i])) / dx[i] ** 2 for i in range(3))
    mode = _cosine_mode(so, dom, grids, k)
    src = so.LevelData(grids, 1)
    for g, f, m in zip(grids, src.fabs, mode.fabs):
        f.a[...] = m.view(g)
    old = so.LevelData(grids, 1, (1, 1, 1))
    new = so.LevelData(grids, 1, (1, 1, 1))
    so.level_tga(amr, new, old, src, dt)
    z = dt * nu * lam
    amp = dt * (1.0 - mu4 * z) / ((1.0 + mu1 * z) * (1.0 + mu2 * z))
    for g, fn, m in zip(grids, new.fabs, mode.fabs):
        np.testing.assert_allclose(fn.view(g), amp * m.view(g), rtol=0, atol=1e-9)


def test_alpha_and_beta_are_products_with_the_factory_coefficients(oracle):
    so = oracle
    dom, grids, dx, amr = _solver(so, (16, 16, 8), 8, 0.05)
    for _ in range(2):                                  # not compounding
        so.reset_solver_alpha_and_beta(amr, 1.0, -0.3)
        assert all(op.alpha == 1.0 and op.beta == -0.3 * 0.05 for op in amr.mg.ops)
    so.reset_solver_alpha_and_beta(amr, 2.0, 0.5)
    assert all(op.alpha == 2.0 and op.beta == 0.5 * 0.05 for op in amr.mg.ops)


@pytest.mark.parametrize("scheme", ["be", "cn", "tga"])
def test_linear_profile_between_dirichlet_walls_is_steady(oracle, scheme):
    """phi = x on [0,1] with phi(0) = 0, phi(1) = 1 (setSideDiriBC order 1 reproduces a linear profile's ghost exactly):
    L[phi] = 0, so both integrators return phiOld."""
    so = oracle
    bc = so.BCHolder([[D, D], [N, N], [N, N]], [[0.0, 1.0], [0.0, 0.0], [0.0, 0.0]])
    dom, grids, dx, amr = _solver(so, (16, 8, 8), 8, 0.1, bc)
    old = so.LevelData(grids, 1, (1, 1, 1))
    for f in old.fabs:
        i = np.arange(f.box.lo[0], f.box.hi[0] + 1)[:, None, None]
        f.a[..., 0] = (i + 0.5) * dx[0] * np.ones(f.box.size())
    src = so.LevelData(grids, 1)
    new = so.LevelData(grids, 1, (1, 1, 1))
    {"be": so.level_backward_euler, "cn": so.level_crank_nicolson, "tga": so.level_tga}[scheme](amr, new, old, src, 0.2)
    for g, fn, fo in zip(grids, new.fabs, old.fabs):
        np.testing.assert_allclose(fn.view(g), fo.view(g), rtol=0, atol=1e-9)
