"""Level heat integrators on a level of a hierarchy (somar_amr_heat_step) against the oracle's restatement of
MappedLevelBackwardEuler / CrankNicolson / TGA with coarse data interpolated in time (oracle/somar_amr.py::amr_level_heat)."""
import numpy as np
import pytest

from oracle import somar_amr as sa
from oracle import somar_oracle as so
from tests.helpers import download_valid, make_amr_levels, make_full_amr_levels, make_gpu_amr, max_rel_diff, upload, valid_of

pytestmark = pytest.mark.gpu
D, N = 1, 0
RATIOS = [(2, 2, 2)]
FINE = [[so.Box((8, 8, 4), (23, 15, 11)), so.Box((8, 16, 4), (23, 23, 11))]]
TYPES = [(D, D), (D, N), (D, D)]
VALUES = [(0.1, 0.0), (0.0, 0.0), (0.0, -0.2)]
NU = 0.05


def _setup(full):
    n, L = (16, 16, 8), (1.0, 1.0, 0.5)
    mk = make_full_amr_levels if full else make_amr_levels
    levels = mk(so, sa, n, L, (False, False, False), RATIOS, FINE, cbox=8)
    bc = so.BCHolder([list(t) for t in TYPES], [list(v) for v in VALUES])
    comp = sa.AMRComposite(levels, RATIOS, bc, so.BiCGStab(), alpha=1.0, beta=NU, isDiagonal=not full)
    return levels, comp


def _gpu(levels, full):
    from somar_amd import AMRPressureSolver
    s = AMRPressureSolver()
    p = s._p
    s.setAMRMGParameters(p.imin, p.imax, p.eps, -1, p.num_smooth_precond, 2, 2, 2, p.precond_mode, 1, p.num_mg, p.hang,
                         p.norm_thresh, 0)
    L0 = levels[0]
    s.defineAMR(L0.domain.box.lo, L0.domain.box.hi, L0.domain.periodic, L0.dx, RATIOS,
                [[(g.lo, g.hi) for g in L.grids] for L in levels], alpha=1.0, beta=NU, bc_type=[t for q in TYPES for t in q])
    for v in s.levels:
        v.setBCValues([x for q in VALUES for x in q])
    for L, v in zip(levels, s.levels):
        for p_ in range(v.num_local_patches):
            _, _, gi = v.patch_box(p_)
            if full:
                v.setMetricFull(p_, *[np.asfortranarray(L.Jgup[gi][d].a) for d in range(3)], np.asfortranarray(L.Jinv[gi].a[..., 0]))
            else:
                jg = [np.asfortranarray(L.Jgup[gi][d].a[..., d]) for d in range(3)]
                v.setMetricOrtho(p_, jg[0], jg[1], jg[2], np.asfortranarray(L.Jinv[gi].a[..., 0]))
    s.finalize()
    return s


@pytest.mark.parametrize("full", [False, True])
@pytest.mark.parametrize("scheme", [0, 1, 2])
def test_fine_level_step_with_time_interpolated_coarse_data(full, scheme):
    from somar_amd import api as F
    levels, comp = _setup(full)
    gpu = _gpu(levels, full)
    try:
        l, dt = 1, 0.2
        g1, g0 = levels[1].grids, levels[0].grids
        old = so.random_field(g1, 3, (1, 1, 1), levels[1].domain.box)
        src = so.random_field(g1, 4, (0, 0, 0), levels[1].domain.box)
        cold = so.random_field(g0, 5, (1, 1, 1), levels[0].domain.box)
        cnew = so.random_field(g0, 6, (1, 1, 1), levels[0].domain.box)
        new = so.LevelData(g1, 1, (1, 1, 1))
        flux = so.FluxData(g1, 1)
        times = dict(oldTime=0.25, crseOldTime=0.0, crseNewTime=1.0)
        sa.amr_level_heat(comp, l, scheme, new, old, src, cold, cnew, dt=dt, zeroPhi=True, flux=flux, **times)
        upload(gpu.levels[1], F.F_HEAT_OLD, old)
        upload(gpu.levels[1], F.F_HEAT_SRC, src)
        upload(gpu.levels[0], F.F_HEAT_OLD, cold)
        upload(gpu.levels[0], F.F_PHI, cnew)
        st = gpu.heatStepAMR(l, scheme, dt, True, **times)
        assert st["iters"] == comp.iters and st["exitStatus"] == comp.exitStatus
        np.testing.assert_allclose(st["history"], comp.history, rtol=1e-10, atol=1e-13 * comp.history[0])
        assert max_rel_diff(download_valid(gpu.levels[1], F.F_PHI, g1), valid_of(new)) < 1e-9
        # a_flux on the faces inside the domain (no register reads a domain-boundary face)
        v = gpu.levels[1]
        dom = levels[1].domain.box
        for q in range(v.num_local_patches):
            lo, hi, gi = v.patch_box(q)
            for d in range(3):
                got = v.heatFlux(d, q)
                want = flux[gi][d].a[..., 0]
                sl = [slice(None)] * 3
                a = 1 if lo[d] == dom.lo[d] else 0
                b = got.shape[d] - (1 if hi[d] == dom.hi[d] else 0)
                sl[d] = slice(a, b)
                scale = float(np.max(np.abs(want))) or 1.0
                np.testing.assert_allclose(got[tuple(sl)], want[tuple(sl)], rtol=0, atol=1e-9 * scale)
    finally:
        gpu.undefine()


def test_subcycled_sequence_coarse_then_fine_then_a_composite_solve():
    """level 0 step, then two fine steps between its old and new state; afterwards a composite solve runs with whatever
    coefficients are installed (the register scales follow beta; tests/test_gpu_amr_tga.py checks the values)"""
    from somar_amd import api as F
    levels, comp = _setup(False)
    gpu = _gpu(levels, False)
    try:
        g1, g0 = levels[1].grids, levels[0].grids
        old0 = so.random_field(g0, 7, (1, 1, 1), levels[0].domain.box)
        src0 = so.random_field(g0, 8, (0, 0, 0), levels[0].domain.box)
        old1 = so.random_field(g1, 9, (1, 1, 1), levels[1].domain.box)
        src1 = so.random_field(g1, 10, (0, 0, 0), levels[1].domain.box)
        dt = 0.1
        new0 = so.LevelData(g0, 1, (1, 1, 1))
        sa.amr_level_heat(comp, 0, 1, new0, old0, src0, dt=dt)
        upload(gpu.levels[0], F.F_HEAT_OLD, old0)
        upload(gpu.levels[0], F.F_HEAT_SRC, src0)
        st = gpu.heatStepAMR(0, 1, dt)
        np.testing.assert_allclose(st["history"], comp.history, rtol=1e-10, atol=1e-13 * comp.history[0])
        assert max_rel_diff(download_valid(gpu.levels[0], F.F_PHI, g0), valid_of(new0)) < 1e-9
        cur = old1
        upload(gpu.levels[1], F.F_HEAT_SRC, src1)
        for k in range(2):
            new1 = so.LevelData(g1, 1, (1, 1, 1))
            t = dict(oldTime=0.5 * dt * k, crseOldTime=0.0, crseNewTime=dt)
            sa.amr_level_heat(comp, 1, 2, new1, cur, src1, old0, new0, dt=0.5 * dt, **t)
            upload(gpu.levels[1], F.F_HEAT_OLD, cur)
            st = gpu.heatStepAMR(1, 2, 0.5 * dt, True, **t)
            np.testing.assert_allclose(st["history"], comp.history, rtol=1e-10, atol=1e-13 * comp.history[0])
            assert max_rel_diff(download_valid(gpu.levels[1], F.F_PHI, g1), valid_of(new1)) < 1e-9
            cur = new1
        gpu.solveAMR(1, 0)
        gpu.setAlphaAndBetaAMR(1.0, 1.0)
        gpu.solveAMR(1, 0)
    finally:
        gpu.undefine()
