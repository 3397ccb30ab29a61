"""The composite TGA step (somar_amr_tga_step = MappedAMRTGA<T>::oneStep, AMRElliptic/MappedAMRTGA.H:417-497) against the
oracle's restatement (oracle/somar_amr.py::amr_tga_one_step): Dirichlet / Neumann walls with values, both metrics, level
ranges starting at 0 and above 0.  The composite operator refluxes with the coarse operator's CURRENT beta (four different
values per step), which the hierarchy's register tables have to follow."""
import numpy as np
import pytest

from oracle import somar_amr as sa
from oracle import somar_oracle as so
from tests.helpers import download_valid, make_amr_levels, make_full_amr_levels, max_rel_diff, upload, valid_of

pytestmark = pytest.mark.gpu
D, N = 1, 0
TYPES = [(D, D), (D, N), (D, D)]
VALUES = [(0.1, 0.0), (0.0, 0.0), (0.0, -0.2)]
NU = 0.05
TWO = ([(2, 2, 2)], [[so.Box((8, 8, 4), (23, 15, 11)), so.Box((8, 16, 4), (23, 23, 11))]])
THREE = ([(2, 2, 2), (2, 2, 1)], [[so.Box((8, 8, 4), (23, 23, 11))], [so.Box((24, 24, 6), (39, 39, 9))]])


def _setup(full, layout):
    ratios, fine = layout
    n, L = (16, 16, 8), (1.0, 1.0, 0.5)
    mk = make_full_amr_levels if full else make_amr_levels
    levels = mk(so, sa, n, L, (False, False, False), ratios, fine, cbox=8)
    bc = so.BCHolder([list(t) for t in TYPES], [list(v) for v in VALUES])
    comp = sa.AMRComposite(levels, ratios, bc, so.BiCGStab(), alpha=1.0, beta=NU, isDiagonal=not full)
    return levels, comp


def _gpu(levels, full, ratios):
    from somar_amd import AMRPressureSolver
    s = AMRPressureSolver()
    p = s._p
    s.setAMRMGParameters(p.imin, p.imax, p.eps, -1, p.num_smooth_precond, 2, 2, 2, p.precond_mode, 1, p.num_mg, p.hang,
                         p.norm_thresh, 0)
    L0 = levels[0]
    s.defineAMR(L0.domain.box.lo, L0.domain.box.hi, L0.domain.periodic, L0.dx, ratios,
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


def _uncovered(comp, levels, fields):
    """valid data with the coarse cells under a finer level zeroed: a composite solve leaves nothing meaningful there"""
    out = []
    for l, f in enumerate(fields):
        if f is None:
            continue
        w = so.ld_create(f)
        so.ld_assign(w, f)
        if l + 1 < len(levels) and fields[l + 1] is not None:
            comp.zero_covered(l, w)
        out += valid_of(w)
    return out


@pytest.mark.parametrize("full,layout,lbase", [(False, TWO, 0), (True, TWO, 0), (False, THREE, 0)])
def test_composite_tga_step_matches_the_oracle(full, layout, lbase):
    from somar_amd import api as F
    levels, comp = _setup(full, layout)
    gpu = _gpu(levels, full, layout[0])
    try:
        nl, dt = len(levels), 0.2
        lmax = nl - 1
        old = [so.random_field(L.grids, 3 + l, (1, 1, 1), L.domain.box) for l, L in enumerate(levels)]
        src = [so.random_field(L.grids, 13 + l, (1, 1, 1), L.domain.box) for l, L in enumerate(levels)]
        new = [so.random_field(L.grids, 23 + l, (1, 1, 1), L.domain.box) for l, L in enumerate(levels)]
        for l in range(nl):
            upload(gpu.levels[l], F.F_HEAT_OLD, old[l])
            upload(gpu.levels[l], F.F_HEAT_SRC, src[l])
            upload(gpu.levels[l], F.F_PHI, new[l])
        sa.amr_tga_one_step(comp, new, old, src, dt, lbase, lmax)
        st = gpu.tgaStepAMR(lmax, lbase, dt)
        assert st["iters"] == comp.iters and st["exitStatus"] == comp.exitStatus
        np.testing.assert_allclose(st["history"], comp.history, rtol=1e-10, atol=1e-13 * comp.history[0])
        want = [new[l] if l >= lbase else None for l in range(nl)]
        got = []
        for l in range(lbase, nl):
            w = so.LevelData(levels[l].grids, 1, (1, 1, 1))
            for g, fab, a in zip(levels[l].grids, w.fabs, download_valid(gpu.levels[l], F.F_PHI, levels[l].grids)):
                fab.view(g)[..., 0] = a
            got.append(w)
        got = [None] * lbase + got
        assert max_rel_diff(_uncovered(comp, levels, got), _uncovered(comp, levels, want)) < 1e-9
    finally:
        gpu.undefine()


def test_level_ranges_above_the_base_level_are_refused():
    """MappedAMRTGA::oneStep reads *m_srct[l_base - 1], which createData never allocates (MappedAMRTGA.H:388-403): undefined
    in the reference, refused here"""
    from somar_amd import SomarError
    levels, comp = _setup(False, THREE)
    gpu = _gpu(levels, False, THREE[0])
    try:
        with pytest.raises(SomarError, match="l_base > 0 is undefined in the reference"):
            gpu.tgaStepAMR(2, 1, 0.1)
    finally:
        gpu.undefine()


def test_composite_solve_with_heat_coefficients_installed():
    """a composite solve right after level heat steps: the registers take the Helmholtz beta (round 1 raised here)"""
    from somar_amd import api as F
    levels, comp = _setup(False, TWO)
    gpu = _gpu(levels, False, TWO[0])
    try:
        g = [L.grids for L in levels]
        rhs = [so.random_field(g[l], 40 + l, (0, 0, 0), levels[l].domain.box) for l in range(2)]
        phi = [so.LevelData(g[l], 1, (1, 1, 1)) for l in range(2)]
        sa.amr_reset_alpha_beta(comp, 1.0, -0.37)
        comp.solve(phi, rhs, 1, 0)
        gpu.setAlphaAndBetaAMR(1.0, -0.37)
        for l in range(2):
            upload(gpu.levels[l], F.F_RHS, rhs[l])
        st = gpu.solveAMR(1, 0)
        assert st["iters"] == comp.iters and st["exitStatus"] == comp.exitStatus
        np.testing.assert_allclose(st["history"], comp.history, rtol=1e-10, atol=1e-13 * comp.history[0])
        got = [download_valid(gpu.levels[l], F.F_PHI, g[l]) for l in range(2)]
        assert max_rel_diff(got[1], valid_of(phi[1])) < 1e-9
    finally:
        gpu.undefine()
