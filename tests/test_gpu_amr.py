"""Several AMR levels on the GPU vs oracle/somar_amr.py: quadratic coarse-fine
interpolation, refluxed composite residual, AMR V-cycle, composite and level solves.

Bit-exact: CF interpolation, composite residual (every level), whole AMR V-cycles.  Residual histories of full
solves are compared to 1e-12 (north star: 1e-10): the levels of these cases are small enough for the library
to use reference-ordered sums everywhere, see test_amr_vcycle_bit_exact."""
import numpy as np
import pytest

from helpers import download_valid, make_amr_levels, make_gpu_amr, max_rel_diff, upload, valid_of

pytestmark = pytest.mark.gpu

LAYOUTS = [
    ((False, False, False), [(2, 2, 2)], [[((8, 8, 4), (23, 23, 11))]]),
    ((True, False, False), [(2, 2, 1)], [[((0, 8, 0), (15, 23, 7)), ((24, 8, 0), (31, 23, 7))]]),
    ((False, True, False), [(2, 2, 2)], [[((8, 0, 0), (23, 15, 7)), ((8, 16, 0), (15, 31, 7))]]),
    ((True, False, False), [(2, 2, 2), (2, 2, 1)], [[((8, 8, 4), (23, 23, 11))], [((24, 24, 6), (39, 39, 9))]]),
    # a fine slab against the walls: one-sided / dropped CF stencils
    ((False, False, False), [(2, 2, 2)], [[((8, 0, 0), (23, 15, 15)), ((8, 16, 0), (23, 31, 15))]]),
    # refinement by 4 (the reference's LockExchange inputs use (4,1,1)): forced MG depths + mini V-cycles
    ((False, True, False), [(4, 1, 1)], [[((16, 0, 0), (31, 15, 7)), ((32, 0, 0), (47, 15, 7))]]),
    ((False, False, False), [(4, 4, 1)], [[((16, 16, 0), (47, 47, 7))]]),
    ((True, False, False), [(2, 2, 1), (4, 1, 1)], [[((8, 8, 0), (23, 23, 7))], [((40, 12, 0), (71, 19, 7))]]),
]
RATIO4 = LAYOUTS[5:]


@pytest.fixture(scope="module")
def am(oracle):
    from oracle import somar_amr
    return somar_amr


def _setup(so, am, layout):
    periodic, ratios, boxes = layout
    fb = [[so.Box(lo, hi) for lo, hi in lev] for lev in boxes]
    levels = make_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.5), periodic, ratios, fb)
    comp = am.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab())
    gpu = make_gpu_amr(levels, ratios)
    return levels, comp, gpu


def _download_ghosted(v, field, ld):
    out = [None] * len(ld.grids)
    for p in range(v.num_local_patches):
        _, _, gi = v.patch_box(p)
        out[gi] = v.download(field, p, ld.ghost)
    return out


@pytest.mark.parametrize("layout", LAYOUTS)
def test_quadratic_cf_interpolation_bit_exact(oracle, am, layout):
    from somar_amd import api as F
    so = oracle
    levels, comp, gpu = _setup(so, am, layout)
    try:
        phi = [so.random_field(L.grids, 5 + l, (1, 1, 1), L.domain.box) for l, L in enumerate(levels)]
        for l, v in enumerate(gpu.levels):
            upload(v, F.F_PHI, phi[l])
        for l in range(1, len(levels)):
            comp.interp_cf_ghosts(l, phi[l], phi[l - 1])
            gpu.interpCF(l)
            got = _download_ghosted(gpu.levels[l], F.F_PHI, phi[l])
            cf = comp.ops[l].cf
            ncf = 0
            for (i, d, s), (gb, m) in cf.ivs.items():
                if m is None:
                    continue
                want = phi[l][i].view(gb)[..., 0]
                have = got[i][gb.slices(phi[l][i].box.lo)]
                np.testing.assert_array_equal(have[m], want[m])
                ncf += int(m.sum())
            assert ncf > 0
    finally:
        gpu.undefine()


@pytest.mark.parametrize("layout", LAYOUTS)
def test_composite_residual_bit_exact(oracle, am, layout):
    from somar_amd import api as F
    so = oracle
    levels, comp, gpu = _setup(so, am, layout)
    try:
        lmax = len(levels) - 1
        phi = [so.random_field(L.grids, 5 + l, (1, 1, 1), L.domain.box) for l, L in enumerate(levels)]
        rhs = [so.random_field(L.grids, 50 + l, (0, 0, 0), L.domain.box) for l, L in enumerate(levels)]
        res = [so.LevelData(L.grids, 1) for L in levels]
        comp.init(phi, rhs, lmax, 0)
        for l, v in enumerate(gpu.levels):
            upload(v, F.F_PHI, phi[l])
            upload(v, F.F_RHS, rhs[l])
        comp.compute_amr_residual(res, phi, rhs, lmax, 0, True)
        for ilev in range(lmax + 1):
            gpu.residualLevel(lmax, 0, ilev)
            if ilev != lmax:
                gpu.zeroCovered(ilev, F.F_RES)
            for g, w in zip(download_valid(gpu.levels[ilev], F.F_RES, levels[ilev].grids), valid_of(res[ilev])):
                np.testing.assert_array_equal(g, w)
    finally:
        gpu.undefine()


VCYCLES = [(LAYOUTS[0], 1, 0), (LAYOUTS[1], 1, 0), (LAYOUTS[3], 2, 0), (LAYOUTS[3], 2, 1), (LAYOUTS[3], 1, 1)]


@pytest.fixture(params=["twopass", "fused"])
def sweep_kernel(request, monkeypatch):
    """fine levels smooth either with the two-pass colour kernel or with the CF-aware fused red+black sweep (layouts
    whose box faces are entirely coarse-fine or not at all; LAYOUTS[2] has a partly-CF face and falls back)"""
    monkeypatch.setenv("SOMAR_FUSED_MIN_CELLS", "0" if request.param == "fused" else "1000000000000")
    return request.param


@pytest.mark.parametrize("case", VCYCLES + [(LAYOUTS[2], 1, 0), (LAYOUTS[4], 1, 0), (LAYOUTS[5], 1, 0), (LAYOUTS[6], 1, 0),
                                            (LAYOUTS[7], 2, 0), (LAYOUTS[7], 2, 1)])
def test_amr_vcycle_bit_exact(oracle, am, case, sweep_kernel):
    """One AMRVCycle from identical inputs.  Every level here is small enough (<= 4096 cells) for the library
    to sum BiCGStab's scalars and the zero-average mean in the reference's serial order (k_reduce_ordered), so
    the whole cycle -- smoothing, CF interpolation, refluxed residual, restriction, bottom solve,
    prolongation -- reproduces the oracle bit for bit."""
    from somar_amd import api as F
    so = oracle
    layout, lmax, lbase = case
    levels, comp, gpu = _setup(so, am, layout)
    try:
        phi = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
        res = [so.random_field(L.grids, 70 + l, (0, 0, 0), L.domain.box) for l, L in enumerate(levels)]
        for l in range(lbase, lmax):
            comp.zero_covered(l, res[l])
        comp.init(phi, res, lmax, lbase)
        comp.set_bottom_solver(lmax, lbase)
        corr = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
        for l, v in enumerate(gpu.levels):
            upload(v, F.F_RES, res[l])
            v.setVal(F.F_CORR, 0.0)
        comp.amr_vcycle(corr, res, lmax, lmax, lbase)
        gpu.vcycleAMR(lmax, lbase)
        for l in range(lbase, lmax + 1):
            for g, w in zip(download_valid(gpu.levels[l], F.F_CORR, levels[l].grids), valid_of(corr[l])):
                np.testing.assert_array_equal(g, w)
    finally:
        gpu.undefine()


@pytest.mark.parametrize("layout", [LAYOUTS[5], LAYOUTS[6]])
def test_mini_vcycle_bit_exact(oracle, am, layout):
    """MappedAMRMultiGrid::relax on a level refined by 4: a V-cycle over the forced (2,.,.) depth whose bottom is
    smoothed and then ZEROED by the AMR solver's NoOpSolver (Chombo 3.1: solve() = setToZero) -- reproduced as is."""
    from somar_amd import api as F
    so = oracle
    levels, comp, gpu = _setup(so, am, layout)
    try:
        L, v = levels[1], gpu.levels[1]
        assert v.mgRefRatios() == [tuple(r) for r in comp.mg[1].mgRefRatios]
        assert comp.mg[1].maxForcedDepth == 1 and max(comp.mg[1].mgRefRatios[0]) == 2
        res = so.random_field(L.grids, 70, (0, 0, 0), L.domain.box)
        corr = so.random_field(L.grids, 71, (1, 1, 1), L.domain.box)
        zero = [so.LevelData(X.grids, 1, (1, 1, 1)) for X in levels]
        zres = [so.LevelData(X.grids, 1, (0, 0, 0)) for X in levels]
        comp.init(zero, zres, 1, 0)
        comp.set_bottom_solver(1, 0)
        upload(v, F.F_RES, res)
        upload(v, F.F_CORR, corr)
        comp.relax(1, corr, res, 2)
        v.miniVCycle(F.F_CORR, F.F_RES)
        for g, w in zip(download_valid(v, F.F_CORR, L.grids), valid_of(corr)):
            np.testing.assert_array_equal(g, w)
    finally:
        gpu.undefine()


_ORACLE_SOLVES = {}   # layout -> the oracle's composite solve, shared by the two sweep-kernel variants


@pytest.mark.parametrize("layout", LAYOUTS[:4] + RATIO4)
def test_composite_solve_history_matches(oracle, am, layout, sweep_kernel):
    from somar_amd import api as F
    so = oracle
    periodic, ratios, boxes = layout
    fb = [[so.Box(lo, hi) for lo, hi in lev] for lev in boxes]
    levels = make_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.5), periodic, ratios, fb)
    lmax = len(levels) - 1
    key = repr(layout)
    if key not in _ORACLE_SOLVES:
        comp = am.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab())
        # a compatible right-hand side: rhs = L_composite[random phi]
        phi = [so.random_field(L.grids, 5 + l, (1, 1, 1), L.domain.box) for l, L in enumerate(levels)]
        zero = [so.LevelData(L.grids, 1) for L in levels]
        rhs = [so.LevelData(L.grids, 1) for L in levels]
        comp.init(phi, zero, lmax, 0)
        comp.compute_amr_residual(rhs, phi, zero, lmax, 0, True)
        for r in rhs:
            so.ld_scale(r, -1.0)
        sol = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
        comp.solve(sol, rhs, lmax, 0)
        _ORACLE_SOLVES[key] = {"rhs": rhs, "sol": [[np.array(x) for x in valid_of(s_)] for s_ in sol], "iters": comp.iters,
                               "exitStatus": comp.exitStatus, "history": list(comp.history)}
    o = _ORACLE_SOLVES[key]
    gpu = make_gpu_amr(levels, ratios)
    try:
        for l, v in enumerate(gpu.levels):
            upload(v, F.F_RHS, o["rhs"][l])
        st = gpu.solveAMR(lmax, 0)
        assert st["iters"] == o["iters"] and st["exitStatus"] == o["exitStatus"]
        np.testing.assert_allclose(st["history"], o["history"], rtol=1e-12, atol=0.0)
        for l in range(lmax + 1):
            got = download_valid(gpu.levels[l], F.F_PHI, levels[l].grids)
            assert max_rel_diff(got, o["sol"][l]) < 1e-8
    finally:
        gpu.undefine()


def test_level_solve_with_coarse_cf_values(oracle, am):
    from somar_amd import api as F
    so = oracle
    levels, comp, gpu = _setup(so, am, LAYOUTS[0])
    try:
        coarse = so.random_field(levels[0].grids, 11, (1, 1, 1), levels[0].domain.box)
        rhs1 = so.random_field(levels[1].grids, 12, (0, 0, 0), levels[1].domain.box)
        phi1 = so.LevelData(levels[1].grids, 1, (1, 1, 1))
        comp.solve([coarse, phi1], [None, rhs1], 1, 1)
        upload(gpu.levels[0], F.F_PHI, coarse)
        upload(gpu.levels[1], F.F_RHS, rhs1)
        st = gpu.solveAMR(1, 1)
        assert st["iters"] == comp.iters and st["exitStatus"] == comp.exitStatus
        np.testing.assert_allclose(st["history"], comp.history, rtol=1e-12, atol=0.0)
        got = download_valid(gpu.levels[1], F.F_PHI, levels[1].grids)
        assert max_rel_diff(got, valid_of(phi1)) < 1e-8
    finally:
        gpu.undefine()


@pytest.mark.parametrize("layout", [LAYOUTS[1], LAYOUTS[5]])
def test_amr_vcycle_with_line_relaxation(oracle, am, layout):
    """relax_mode 3 (vertical-line GSRB) on a fine level whose lateral faces are coarse-fine boundaries and whose
    columns span the domain: homogeneous CF values in the lateral ghosts, then the column solves."""
    from somar_amd import api as F
    so = oracle
    periodic, ratios, boxes = layout
    fb = [[so.Box(lo, hi) for lo, hi in lev] for lev in boxes]
    levels = make_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.5), periodic, ratios, fb)
    comp = am.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab(), relaxMode=so.RELAX_LINE_GSRB)
    gpu = make_gpu_amr(levels, ratios, relaxMode=3)
    try:
        phi = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
        res = [so.random_field(L.grids, 70 + l, (0, 0, 0), L.domain.box) for l, L in enumerate(levels)]
        comp.zero_covered(0, res[0])
        comp.init(phi, res, 1, 0)
        comp.set_bottom_solver(1, 0)
        corr = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
        for l, v in enumerate(gpu.levels):
            upload(v, F.F_RES, res[l])
            v.setVal(F.F_CORR, 0.0)
        comp.amr_vcycle(corr, res, 1, 1, 0)
        gpu.vcycleAMR(1, 0)
        for l in (0, 1):
            for g, w in zip(download_valid(gpu.levels[l], F.F_CORR, levels[l].grids), valid_of(corr[l])):
                np.testing.assert_allclose(g, w, rtol=0, atol=1e-12 * float(np.max(np.abs(w))))
    finally:
        gpu.undefine()
