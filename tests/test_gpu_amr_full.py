"""Non-diagonal metric on AMR levels (19-point in 3-D, 9-point in 2-D) on the GPU vs the oracle: quadratic CF
interpolation followed by ExtrapolateCFEV (edge / vertex ghosts), the refluxed composite residual with MAPPEDGETFLUX
fluxes in the register, AMR V-cycles and composite solves.  Multi-box coarse levels included: the layout quirk of the
non-diagonal Neumann ghost (tests/test_oracle_full.py) is carried by both sides alike."""
import numpy as np
import pytest

from helpers import download_valid, make_full_amr_levels, make_gpu_amr, max_rel_diff, upload, valid_of

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["direct", "march", "fused"])
def kernel_path(request, monkeypatch):
    """direct: k_op_full / k_gsrb_full (small levels); march: the k-marching LDS kernels of large levels
    (full19_march.hip, psi kept in the boxes' frames only), forced onto these small cases.  Same bits either way."""
    monkeypatch.setenv("SOMAR_MARCH_MIN_CELLS", "0" if request.param != "direct" else "1000000000000")
    # fused: red + black in one marching launch three layers inside every box, then a shell pass (full19_fused.hip), forced onto
    # every level whose boxes are at least 8 cells wide (the default takes it on boxes of 192 and more)
    monkeypatch.setenv("SOMAR_FUSED19_MIN_BOX", "0" if request.param == "fused" else "-1")
    return request.param

LAYOUTS = [
    # ndim, n, L, cbox, periodic, ratios, fine boxes
    (2, (32, 16, 1), (2.0, 1.0, 1.0), (32, 16, 1), (False, False, False), [(2, 2, 1)], [[((16, 8, 0), (47, 23, 0))]]),
    (2, (32, 16, 1), (2.0, 1.0, 1.0), (8, 8, 1), (False, False, False), [(2, 2, 1)], [[((16, 8, 0), (47, 23, 0))]]),
    (2, (32, 16, 1), (2.0, 1.0, 1.0), (8, 8, 1), (False, False, False), [(4, 1, 1)], [[((32, 0, 0), (63, 15, 0)), ((64, 0, 0), (95, 15, 0))]]),
    (2, (32, 16, 1), (2.0, 1.0, 1.0), (8, 8, 1), (False, False, False), [(4, 1, 1), (4, 2, 1)],
     [[((32, 0, 0), (95, 15, 0))], [((160, 8, 0), (287, 23, 0))]]),
    (3, (16, 16, 8), (2.0, 1.0, 0.5), 8, (False, False, False), [(2, 2, 2)], [[((8, 8, 4), (23, 23, 11))]]),
    (3, (16, 16, 8), (2.0, 1.0, 0.5), 8, (True, False, False), [(2, 2, 1)], [[((0, 8, 0), (15, 23, 7)), ((24, 8, 0), (31, 23, 7))]]),
]


@pytest.fixture(scope="module")
def am(oracle):
    from oracle import somar_amr
    return somar_amr


def _setup(so, am, layout):
    ndim, n, L, cbox, periodic, ratios, boxes = layout
    fb = [[so.Box(lo, hi) for lo, hi in lev] for lev in boxes]
    levels = make_full_amr_levels(so, am, n, L, periodic, ratios, fb, cbox=cbox, ndim=ndim)
    comp = am.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab(), ndim=ndim, isDiagonal=False)
    gpu = make_gpu_amr(levels, ratios, ndim=ndim, full=True)
    return ndim, levels, comp, gpu


@pytest.mark.parametrize("layout", LAYOUTS)
def test_full_metric_composite_residual_bit_exact(oracle, am, layout):
    from somar_amd import api as F
    so = oracle
    ndim, levels, comp, gpu = _setup(so, am, layout)
    G = (1, 1, 1) if ndim == 3 else (1, 1, 0)
    try:
        lmax = len(levels) - 1
        phi = [so.random_field(L.grids, 5 + l, G, L.domain.box) for l, L in enumerate(levels)]
        rhs = [so.random_field(L.grids, 50 + l, (0, 0, 0), L.domain.box) for l, L in enumerate(levels)]
        res = [so.LevelData(L.grids, 1) for L in levels]
        for l, v in enumerate(gpu.levels):
            upload(v, F.F_PHI, phi[l])
            upload(v, F.F_RHS, rhs[l])
        comp.init(phi, rhs, lmax, 0)
        comp.compute_amr_residual(res, phi, rhs, lmax, 0, True)
        for ilev in range(lmax + 1):
            gpu.residualLevel(lmax, 0, ilev)
            if ilev < lmax:
                gpu.zeroCovered(ilev, F.F_RES)
            for g_, w_ in zip(download_valid(gpu.levels[ilev], F.F_RES, levels[ilev].grids), valid_of(res[ilev])):
                np.testing.assert_array_equal(g_, w_, err_msg="composite residual level %d" % ilev)
    finally:
        gpu.undefine()


_ORACLE_CYCLES = {}   # layout -> what the oracle computed: shared by the two kernel paths (the numpy oracle is the slow half)


@pytest.mark.parametrize("layout", [LAYOUTS[0], LAYOUTS[1], LAYOUTS[2], LAYOUTS[4], LAYOUTS[5]])
def test_full_metric_amr_vcycle_and_solve(oracle, am, layout):
    from somar_amd import api as F
    so = oracle
    ndim, n, L_, cbox, periodic, ratios, boxes = layout
    fb = [[so.Box(lo, hi) for lo, hi in lev] for lev in boxes]
    levels = make_full_amr_levels(so, am, n, L_, periodic, ratios, fb, cbox=cbox, ndim=ndim)
    G = (1, 1, 1) if ndim == 3 else (1, 1, 0)
    key = repr(layout)
    if key not in _ORACLE_CYCLES:
        comp = am.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab(), ndim=ndim, isDiagonal=False)
        zero = [so.LevelData(L.grids, 1, G) for L in levels]
        res = [so.random_field(L.grids, 70 + l, (0, 0, 0), L.domain.box) for l, L in enumerate(levels)]
        comp.zero_covered(0, res[0])
        comp.init(zero, res, 1, 0)
        comp.set_bottom_solver(1, 0)
        corr = [so.LevelData(L.grids, 1, G) for L in levels]
        comp.amr_vcycle(corr, res, 1, 1, 0)
        # composite solve: same iteration count / exit status / history whatever the layout quirk does to convergence
        phi = [so.random_field(L.grids, 5 + l, G, L.domain.box) for l, L in enumerate(levels)]
        z0 = [so.LevelData(L.grids, 1) for L in levels]
        rhs = [so.LevelData(L.grids, 1) for L in levels]
        comp.init(phi, z0, 1, 0)
        comp.compute_amr_residual(rhs, phi, z0, 1, 0, True)
        for r in rhs:
            so.ld_scale(r, -1.0)
        sol = [so.LevelData(L.grids, 1, G) for L in levels]
        comp.solve(sol, rhs, 1, 0)
        _ORACLE_CYCLES[key] = {"res": res, "corr": [[np.array(x) for x in valid_of(c)] for c in corr], "rhs": rhs,
                               "iters": comp.iters, "exitStatus": comp.exitStatus, "history": list(comp.history)}
    o = _ORACLE_CYCLES[key]
    gpu = make_gpu_amr(levels, ratios, ndim=ndim, full=True)
    try:
        for l, v in enumerate(gpu.levels):
            upload(v, F.F_RES, o["res"][l])
            v.setVal(F.F_CORR, 0.0)
        gpu.vcycleAMR(1, 0)
        for l in (0, 1):
            for g_, w_ in zip(download_valid(gpu.levels[l], F.F_CORR, levels[l].grids), o["corr"][l]):
                np.testing.assert_array_equal(g_, w_)
        for l, v in enumerate(gpu.levels):
            upload(v, F.F_RHS, o["rhs"][l])
        try:
            st = gpu.solveAMR(1, 0)
        except Exception:
            st = gpu.stats
        assert st["iters"] == o["iters"] and st["exitStatus"] == o["exitStatus"]
        np.testing.assert_allclose(st["history"], o["history"], rtol=1e-8, atol=0.0)
    finally:
        gpu.undefine()
