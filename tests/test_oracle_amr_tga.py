"""Known answers for the composite TGA step (oracle/somar_amr.py::amr_tga_one_step, MappedAMRTGA<T>::oneStep,
AMRElliptic/MappedAMRTGA.H:417-497).  The reference holds no fixtures and no driver of it calls the class: parity unpinned
w.r.t. reference tests; pinned here by (1) the exact amplification factor of a discrete eigenmode on one level, where the
composite step and MappedLevelTGA must agree, (2) constants: L[const] = 0, so phiNew = phiOld + dt S exactly, (3) the
composite heat content: the refluxed composite operator integrates to zero over a closed domain, so the integral of phi
over the composite grid grows by exactly dt x the integral of the source -- this only holds if the flux-register scale
follows beta through every resetAlphaAndBeta (MappedAMRPoissonOp.cpp:1661, 1693)."""
import numpy as np
import pytest

from helpers import make_amr_levels, make_problem

N = 0


@pytest.fixture(scope="module")
def am(oracle):
    from oracle import somar_amr
    return somar_amr


def _two_levels(so, am, nu, eps=1e-12):
    ratios = [(2, 2, 2)]
    fine = [[so.Box((8, 8, 4), (23, 15, 11)), so.Box((8, 16, 4), (23, 23, 11))]]
    levels = make_amr_levels(so, am, (16, 16, 8), (1.0, 1.0, 0.5), (False, False, False), ratios, fine, cbox=8)
    comp = am.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab(), alpha=1.0, beta=nu)
    comp.eps, comp.iterMax = eps, 40
    return levels, comp


def _composite_integral(so, am, levels, comp, fields):
    """sum over the composite grid (coarse cells under the fine level excluded) of phi dV, dV = prod(dx) / Jinv"""
    tot = 0.0
    nl = len(levels)
    for l, (L, f) in enumerate(zip(levels, fields)):
        w = so.ld_create(f)
        so.ld_assign(w, f)
        if l + 1 < nl:
            comp.zero_covered(l, w)
        for i, g in enumerate(L.grids):
            tot += float((w[i].view(g)[..., 0] / L.Jinv[i].view(g)[..., 0]).sum()) * float(np.prod(L.dx))
    return tot


def test_one_level_composite_step_is_the_level_tga_step_on_an_eigenmode(oracle, am):
    so = oracle
    n, nu, dt, k = (16, 16, 8), 0.05, 0.3, (1, 2, 1)
    dom, grids, dx, Jgup, Jinv = make_problem(so, n, 8, "cartesian", (False, False, False), (1.0, 1.0, 1.0))
    levels = [am.AMRLevel(dom, grids, dx, Jgup, Jinv)]
    comp = am.AMRComposite(levels, [], so.BCHolder(), so.BiCGStab(), alpha=1.0, beta=nu)
    comp.eps, comp.iterMax = 1e-12, 40
    old = so.LevelData(grids, 1, (1, 1, 1))
    for f in old.fabs:
        idx = np.meshgrid(*[np.arange(f.box.lo[a], f.box.hi[a] + 1) for a in range(3)], indexing="ij")
        v = np.ones(f.box.size())
        for a in range(3):
            v = v * np.cos(np.pi * k[a] * (idx[a] + 0.5) / n[a])
        f.a[..., 0] = v
    src = so.LevelData(grids, 1, (1, 1, 1))
    new = so.LevelData(grids, 1, (1, 1, 1))
    am.amr_tga_one_step(comp, [new], [old], [src], dt, 0, 0)
    lam = sum((2.0 - 2.0 * np.cos(np.pi * k[a] / n[a])) / dx[a] ** 2 for a in range(3))
    mu1, mu2, mu3, mu4, _ = so.tga_coefficients()
    z = dt * nu * lam
    amp = (1.0 - mu3 * z) / ((1.0 + mu1 * z) * (1.0 + mu2 * z))
    for g, fn, fo in zip(grids, new.fabs, old.fabs):
        np.testing.assert_allclose(fn.view(g), amp * fo.view(g), rtol=0, atol=1e-9)


def test_constants_advance_by_dt_times_the_source(oracle, am):
    so = oracle
    levels, comp = _two_levels(so, am, 0.05)
    old = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
    src = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
    new = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
    for o, s in zip(old, src):
        so.ld_set(o, 1.5)
        so.ld_set(s, -0.25)
    am.amr_tga_one_step(comp, new, old, src, 0.2, 0, 1)
    # coarse cells under the fine level carry no meaning after a composite solve (the caller averages down)
    for f in new[0].fabs:
        f.a[...] -= 1.5 - 0.25 * 0.2
    comp.zero_covered(0, new[0])
    for g, fab in zip(levels[0].grids, new[0].fabs):
        np.testing.assert_allclose(fab.view(g), 0.0, rtol=0, atol=1e-11)
    for g, fab in zip(levels[1].grids, new[1].fabs):
        np.testing.assert_allclose(fab.view(g), 1.5 - 0.25 * 0.2, rtol=0, atol=1e-11)


def test_composite_heat_content_grows_by_dt_times_the_integrated_source(oracle, am):
    so = oracle
    levels, comp = _two_levels(so, am, 0.05)
    old = [so.random_field(L.grids, 21 + l, (1, 1, 1), L.domain.box) for l, L in enumerate(levels)]
    src = [so.random_field(L.grids, 31 + l, (1, 1, 1), L.domain.box) for l, L in enumerate(levels)]
    new = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
    dt = 0.2
    am.amr_tga_one_step(comp, new, old, src, dt, 0, 1)
    assert comp.exitStatus == 1
    i_old = _composite_integral(so, am, levels, comp, old)
    i_src = _composite_integral(so, am, levels, comp, src)
    i_new = _composite_integral(so, am, levels, comp, new)
    scale = sum(float(np.prod(L.dx)) * sum(g.numPts() for g in L.grids) for L in levels)
    # the solves stop at eps = 1e-12 of their initial residual
    assert abs(i_new - (i_old + dt * i_src)) < 1e-9 * scale
    # and the step is not the identity
    assert abs(i_new - i_old) > 1e-4 * scale or abs(dt * i_src) < 1e-4 * scale
    d = max(float(np.max(np.abs(a.view(g) - b.view(g)))) for L, x, y in zip(levels, new, old)
            for g, a, b in zip(L.grids, x.fabs, y.fabs))
    assert d > 1e-3
