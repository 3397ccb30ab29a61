"""Several AMR levels with CH_SPACEDIM = 2 on the GPU vs the oracle (the reference's 2-D lock-exchange decks refine by
(4,1) and (4,2)): quadratic CF interpolation with ONE tangential direction and no mixed derivative, refluxed composite
residual, AMR V-cycle with the mini V-cycle, composite solve."""
import numpy as np
import pytest

from helpers import download_valid, make_amr_levels, make_gpu_amr, max_rel_diff, upload, valid_of

pytestmark = pytest.mark.gpu

LAYOUTS_2D = [
    ((False, False, False), [(2, 2, 1)], [[((16, 8, 0), (47, 23, 0))]]),
    ((False, False, False), [(4, 1, 1)], [[((32, 0, 0), (95, 15, 0))]]),
    ((False, True, False), [(4, 1, 1)], [[((32, 0, 0), (63, 15, 0)), ((64, 0, 0), (95, 15, 0))]]),
    ((True, False, False), [(2, 2, 1)], [[((0, 8, 0), (31, 23, 0)), ((48, 8, 0), (63, 23, 0))]]),
    ((False, False, False), [(4, 1, 1), (4, 2, 1)], [[((32, 0, 0), (95, 15, 0))], [((160, 8, 0), (287, 23, 0))]]),
]
G = (1, 1, 0)


@pytest.fixture(scope="module")
def am(oracle):
    from oracle import somar_amr
    return somar_amr


def _setup(so, am, layout):
    periodic, ratios, boxes = layout
    fb = [[so.Box(lo, hi) for lo, hi in lev] for lev in boxes]
    levels = make_amr_levels(so, am, (32, 16, 1), (2.0, 1.0, 1.0), periodic, ratios, fb, cbox=(8, 8, 1), ndim=2)
    comp = am.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab(), ndim=2)
    gpu = make_gpu_amr(levels, ratios, ndim=2)
    return levels, comp, gpu


@pytest.mark.parametrize("layout", LAYOUTS_2D)
def test_2d_cf_interpolation_and_composite_residual_bit_exact(oracle, am, layout):
    from somar_amd import api as F
    so = oracle
    levels, comp, gpu = _setup(so, am, layout)
    try:
        lmax = len(levels) - 1
        phi = [so.random_field(L.grids, 5 + l, G, L.domain.box) for l, L in enumerate(levels)]
        rhs = [so.random_field(L.grids, 50 + l, (0, 0, 0), L.domain.box) for l, L in enumerate(levels)]
        res = [so.LevelData(L.grids, 1) for L in levels]
        for l, v in enumerate(gpu.levels):
            upload(v, F.F_PHI, phi[l])
            upload(v, F.F_RHS, rhs[l])
        comp.init(phi, rhs, lmax, 0)
        comp.compute_amr_residual(res, phi, rhs, lmax, 0, True)   # fills the CF ghosts of phi on the way
        for ilev in range(lmax + 1):
            gpu.residualLevel(lmax, 0, ilev)
            if ilev < lmax:
                gpu.zeroCovered(ilev, F.F_RES)
            for g_, w_ in zip(download_valid(gpu.levels[ilev], F.F_RES, levels[ilev].grids), valid_of(res[ilev])):
                np.testing.assert_array_equal(g_, w_, err_msg="composite residual level %d" % ilev)
        # the CF ghost values themselves
        for l in range(1, lmax + 1):
            v = gpu.levels[l]
            for p_ in range(v.num_local_patches):
                _, _, gi = v.patch_box(p_)
                got = v.download(F.F_PHI, p_, G)
                want = phi[l][gi].a[..., 0]
                cf = comp.ops[l].cf
                for d in range(2):
                    for s in (0, 1):
                        gb, m = cf.ivs[(gi, d, s)]
                        if m is None:
                            continue
                        sl = gb.slices(phi[l][gi].box.lo)
                        np.testing.assert_array_equal(got[sl][m], want[sl][m])
    finally:
        gpu.undefine()


@pytest.mark.parametrize("layout", LAYOUTS_2D[:4])
def test_2d_amr_vcycle_bit_exact_and_solve_history(oracle, am, layout):
    from somar_amd import api as F
    so = oracle
    levels, comp, gpu = _setup(so, am, layout)
    try:
        zero = [so.LevelData(L.grids, 1, G) for L in levels]
        res = [so.random_field(L.grids, 70 + l, (0, 0, 0), L.domain.box) for l, L in enumerate(levels)]
        comp.zero_covered(0, res[0])
        comp.init(zero, res, 1, 0)
        comp.set_bottom_solver(1, 0)
        corr = [so.LevelData(L.grids, 1, G) for L in levels]
        for l, v in enumerate(gpu.levels):
            upload(v, F.F_RES, res[l])
            v.setVal(F.F_CORR, 0.0)
        comp.amr_vcycle(corr, res, 1, 1, 0)
        gpu.vcycleAMR(1, 0)
        for l in (0, 1):
            for g_, w_ in zip(download_valid(gpu.levels[l], F.F_CORR, levels[l].grids), valid_of(corr[l])):
                np.testing.assert_array_equal(g_, w_)
        # composite solve of L[x] = L[random]
        phi = [so.random_field(L.grids, 5 + l, G, L.domain.box) for l, L in enumerate(levels)]
        z0 = [so.LevelData(L.grids, 1) for L in levels]
        rhs = [so.LevelData(L.grids, 1) for L in levels]
        comp.init(phi, z0, 1, 0)
        comp.compute_amr_residual(rhs, phi, z0, 1, 0, True)
        for r in rhs:
            so.ld_scale(r, -1.0)
        sol = [so.LevelData(L.grids, 1, G) for L in levels]
        comp.solve(sol, rhs, 1, 0)
        for l, v in enumerate(gpu.levels):
            upload(v, F.F_RHS, rhs[l])
        st = gpu.solveAMR(1, 0)
        assert st["iters"] == comp.iters and st["exitStatus"] == comp.exitStatus
        np.testing.assert_allclose(st["history"], comp.history, rtol=1e-10, atol=0.0)
        for l in (0, 1):
            assert max_rel_diff(download_valid(gpu.levels[l], F.F_PHI, levels[l].grids), valid_of(sol[l])) < 1e-8
    finally:
        gpu.undefine()
