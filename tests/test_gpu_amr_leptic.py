"""AMRLepticSolver on the GPU (AMRSolver::solve_leptic, LepticSolver::attach) against the oracle's restatement
(oracle/somar_leptic.py::AMRLepticSolver): lateral coarse-fine boundaries in the level leptic solver, the composite
V-cycle as the reference wrote it and with the base level fed the restricted residual."""
import numpy as np
import pytest

from oracle import somar_amr as sa
from oracle import somar_leptic as sl
from oracle import somar_oracle as so
from tests.helpers import download_valid, make_amr_levels, make_full_amr_levels, make_gpu_amr, upload, valid_of

pytestmark = pytest.mark.gpu

H = 0.005
N, RATIOS = (32, 32, 8), [(2, 2, 1)]
FINE = [[so.Box((16, 16, 0), (31, 47, 7)), so.Box((32, 16, 0), (47, 47, 7))]]   # column boxes; CF on every lateral side


def _levels(metric):
    L = (1.0, 1.0, H)
    if metric == "sheared":
        return make_full_amr_levels(so, sa, N, (64.0, 64.0, 1.0), (False, False, False), RATIOS, FINE, cbox=(16, 16, 8)), 1.0
    return make_amr_levels(so, sa, N, L, (False, False, False), RATIOS, FINE, variant=metric, cbox=(16, 16, 8)), H


def _oracle(levels, height, maxOrder, fixed, full, iters):
    amr = sl.AMRLepticSolver(levels, RATIOS, so.BCHolder(), leptic=dict(maxOrder=maxOrder, domainHeight=height),
                             baseFromRestricted=fixed, isDiagonal=not full)
    amr.iterMax = iters
    return amr


def _gpu(levels, height, maxOrder, fixed, full, iters):
    from somar_amd import api as F
    gpu = make_gpu_amr(levels, RATIOS, full=full, imax=iters)
    lp = F.LepticParams()
    F._ck(F.lib().somar_leptic_params_default(lp))
    lp.max_order, lp.domain_height = maxOrder, height
    gpu.enableLeptic(lp, baseFromRestricted=fixed)
    return gpu


def _compatible_rhs(amr, levels, lmax):
    phi = [so.random_field(Lv.grids, 5 + l, (1, 1, 1), Lv.domain.box) for l, Lv in enumerate(levels)]
    zero = [so.LevelData(Lv.grids, 1) for Lv in levels]
    rhs = [so.LevelData(Lv.grids, 1) for Lv in levels]
    amr.init(phi, zero, lmax, 0)
    amr.compute_amr_residual(rhs, phi, zero, lmax, 0, True)
    for r in rhs:
        so.ld_scale(r, -1.0)
    return rhs


@pytest.mark.parametrize("metric", ["cartesian", "stretched", "sheared"])
def test_fine_level_alone_lateral_cf(metric):
    """l_base = l_max = 1 with a zero coarse phi: one LevelLepticSolver::solve per iteration on a level whose columns
    have coarse-fine boundaries on every lateral side (homogeneousCFInterp / ExtrapolateCFEV in computeHorizRHS, the flat
    problem with homogeneous CF values and no mean removal)."""
    from somar_amd import api as F
    full = metric == "sheared"
    levels, height = _levels(metric)
    amr = _oracle(levels, height, 3, False, full, 2)
    rhs1 = so.random_field(levels[1].grids, 9, domainBox=levels[1].domain.box)
    phi = [so.LevelData(Lv.grids, 1, (1, 1, 1)) for Lv in levels]
    amr.solve(phi, [None, rhs1], 1, 1)
    gpu = _gpu(levels, height, 3, False, full, 2)
    try:
        upload(gpu.levels[1], F.F_RHS, rhs1)
        gpu.levels[0].setVal(F.F_PHI, 0.0)
        st = gpu.solveAMRLeptic(1, 1)
        ls = gpu.lepticStats(1)
        lep = amr.leptic[1]
        assert st["iters"] == amr.iters and st["exitStatus"] == amr.exitStatus
        assert ls["exitStatus"] == lep.exitStatus and ls["horizSolves"] == lep.horizSolves
        assert ls["usedFullSolver"] == lep.usedFullSolver
        # a hanging last order hands over to the full 3-D multigrid (stretched metric), whose BiCGStab bottom solver
        # amplifies the summation-order difference of the 8192-cell depth; otherwise round-off only
        tol = 1e-6 if lep.usedFullSolver else 1e-10
        np.testing.assert_allclose(st["history"], amr.history, rtol=0, atol=tol * amr.history[0])
        np.testing.assert_allclose(ls["resNorms"], lep.resNorms, rtol=0, atol=tol * lep.resNorms[0])
        want = valid_of(phi[1])
        scale = max(float(np.max(np.abs(w))) for w in want)
        for g_, w_ in zip(download_valid(gpu.levels[1], F.F_PHI, levels[1].grids), want):
            np.testing.assert_allclose(g_, w_, rtol=0, atol=tol * scale)
    finally:
        gpu.undefine()


@pytest.mark.parametrize("metric,fixed", [("cartesian", False), ("cartesian", True), ("stretched", True), ("sheared", True)])
def test_composite_leptic_solve_matches_oracle(metric, fixed):
    from somar_amd import api as F
    full = metric == "sheared"
    levels, height = _levels(metric)
    iters = 2 if metric == "sheared" else 3   # (the sheared case's oracle run is the slowest test of the suite; its checks need h[0..2])
    # as written (fixed False) the base level is handed the composite residual with the covered cells zeroed, which is not a
    # compatible right-hand side; if its last order hangs, the full 3-D multigrid is turned loose on that singular,
    # inconsistent problem and its BiCGStab bottom solver amplifies round-off chaotically (both sides take that branch at
    # max_order 3 and then differ by 10 %).  max_order 2 keeps the comparison on the deterministic part.
    maxOrder = 3 if fixed else 2
    amr = _oracle(levels, height, maxOrder, fixed, full, iters)
    rhs = _compatible_rhs(amr, levels, 1)
    sol = [so.LevelData(Lv.grids, 1, (1, 1, 1)) for Lv in levels]
    amr.solve(sol, rhs, 1, 0)
    gpu = _gpu(levels, height, maxOrder, fixed, full, iters)
    try:
        for l, v in enumerate(gpu.levels):
            upload(v, F.F_RHS, rhs[l])
        st = gpu.solveAMRLeptic(1, 0)
        assert st["iters"] == amr.iters and st["exitStatus"] == amr.exitStatus
        h = np.array(amr.history)
        np.testing.assert_allclose(st["history"], h, rtol=0, atol=1e-10 * h[0])
        if fixed and metric == "sheared":
            assert h[1] < 0.05 * h[0] and h[2] < h[1]                # lepticity 0.5 and cross terms: slower, still converging
        elif fixed:
            assert h[1] < 1e-3 * h[0] and h[2] < 0.1 * h[1]          # converges
        else:
            assert h[1] < 1e-3 * h[0] and h[3] > h[2]                # the reference's base-level branch: drifts
        for l in (0, 1):
            ls, lep = gpu.lepticStats(l), amr.leptic[l]
            assert ls["exitStatus"] == lep.exitStatus and ls["horizSolves"] == lep.horizSolves
            assert ls["usedFullSolver"] == lep.usedFullSolver and (fixed or not lep.usedFullSolver)
            want = valid_of(sol[l])
            scale = max(float(np.max(np.abs(w))) for w in want)
            for g_, w_ in zip(download_valid(gpu.levels[l], F.F_PHI, levels[l].grids), want):
                np.testing.assert_allclose(g_, w_, rtol=0, atol=1e-8 * scale)
    finally:
        gpu.undefine()


def test_leptic_cycle_needs_enable_and_column_boxes():
    from somar_amd import SomarError
    levels, height = _levels("cartesian")
    gpu = make_gpu_amr(levels, RATIOS)
    try:
        with pytest.raises(SomarError, match="enable_leptic"):
            gpu.solveAMRLeptic(1, 0)
    finally:
        gpu.undefine()
    # boxes split in the vertical: not the vertical-solver layout
    lv = make_amr_levels(so, sa, (16, 16, 8), (1.0, 1.0, H), (False, False, False), [(2, 2, 1)],
                         [[so.Box((8, 8, 0), (23, 23, 7))]], variant="cartesian", cbox=(8, 8, 4))
    gpu = make_gpu_amr(lv, [(2, 2, 1)])
    try:
        with pytest.raises(SomarError, match="Vertical grids are ill-formed"):
            gpu.enableLeptic()
    finally:
        gpu.undefine()


def _terrain_three_levels():
    """C5's shape in small: terrain-following NON-diagonal metric, three levels nested by (2,2,1) around the bump, every level
    made of whole columns; leptic aspect ratio (lepticity dx/H = 2 on the base level)"""
    n, L = (16, 16, 8), (32.0, 32.0, 1.0)
    ratios = [(2, 2, 1), (2, 2, 1)]
    fine = [[so.Box((8, 8, 0), (23, 23, 7))], [so.Box((24, 24, 0), (39, 39, 7))]]
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (False, False, False))
    dx = tuple(L[d] / n[d] for d in range(3))
    grids = [so.split_domain(dom.box, (8, 8, 8))] + [list(b) for b in fine]
    levels = []
    for l, g in enumerate(grids):
        if l > 0:
            dom = dom.refine(ratios[l - 1])
            dx = tuple(a / b for a, b in zip(dx, ratios[l - 1]))
        Jgup, Jinv = so.make_terrain_metric(g, dx, L, dom)
        levels.append(sa.AMRLevel(dom, g, dx, Jgup, Jinv))
    return levels, ratios


def test_c5_shaped_three_level_leptic_solve_matches_oracle():
    from somar_amd import api as F
    levels, ratios = _terrain_three_levels()
    iters, maxOrder = 2, 3
    amr = sl.AMRLepticSolver(levels, ratios, so.BCHolder(), leptic=dict(maxOrder=maxOrder, domainHeight=1.0),
                             baseFromRestricted=True, isDiagonal=False)
    amr.iterMax = iters
    rhs = _compatible_rhs(amr, levels, 2)
    sol = [so.LevelData(Lv.grids, 1, (1, 1, 1)) for Lv in levels]
    amr.solve(sol, rhs, 2, 0)
    gpu = make_gpu_amr(levels, ratios, full=True, imax=iters)
    lp = F.LepticParams()
    F._ck(F.lib().somar_leptic_params_default(lp))
    lp.max_order, lp.domain_height = maxOrder, 1.0
    gpu.enableLeptic(lp, baseFromRestricted=True)
    try:
        for l, v in enumerate(gpu.levels):
            upload(v, F.F_RHS, rhs[l])
        st = gpu.solveAMRLeptic(2, 0)
        h = np.array(amr.history)
        assert st["iters"] == amr.iters and st["exitStatus"] == amr.exitStatus
        np.testing.assert_allclose(st["history"], h, rtol=0, atol=1e-9 * h[0])
        assert h[1] < 0.2 * h[0] and h[2] < h[1]
        for l in range(3):
            ls, lep = gpu.lepticStats(l), amr.leptic[l]
            assert ls["exitStatus"] == lep.exitStatus and ls["horizSolves"] == lep.horizSolves
            assert ls["usedFullSolver"] == lep.usedFullSolver
            want = valid_of(sol[l])
            scale = max(float(np.max(np.abs(w))) for w in want)
            for g_, w_ in zip(download_valid(gpu.levels[l], F.F_PHI, levels[l].grids), want):
                np.testing.assert_allclose(g_, w_, rtol=0, atol=1e-7 * scale)
    finally:
        gpu.undefine()


# ---- columns that end at a coarse-fine interface (LepticLapackVerticalSolver's BCType_CF row) ------------------------------
def test_fine_level_whose_columns_end_under_the_coarse_level():
    """level 1 = the lower half of the water column of the central block, refined by (2, 2, 2): Neumann bottom, coarse-fine
    top and lateral sides; no column is Neumann-Neumann, so every order is one dptsv per column and there is no flat
    problem.  l_base = l_max = 1 through the composite leptic solver, as AMRLepticSolver runs its level solver."""
    from somar_amd import api as F
    n, ratios = (16, 16, 8), [(2, 2, 2)]
    fine = [[so.Box((8, 8, 0), (15, 23, 7)), so.Box((16, 8, 0), (23, 23, 7))]]
    levels = make_amr_levels(so, sa, n, (1.0, 1.0, H), (False, False, False), ratios, fine, cbox=(8, 8, 8))
    amr = sl.AMRLepticSolver(levels, ratios, so.BCHolder(), leptic=dict(maxOrder=3, domainHeight=H))
    amr.iterMax = 2
    rhs1 = so.random_field(levels[1].grids, 9, domainBox=levels[1].domain.box)
    phi = [so.LevelData(Lv.grids, 1, (1, 1, 1)) for Lv in levels]
    amr.solve(phi, [None, rhs1], 1, 1)
    lep = amr.leptic[1]
    assert not lep.doHorizSolve and all(t == (sl.VBC_NEUM, sl.VBC_CF) for t in lep.vertBCTypes)
    gpu = make_gpu_amr(levels, ratios, imax=2)
    lp = F.LepticParams()
    F._ck(F.lib().somar_leptic_params_default(lp))
    lp.max_order, lp.domain_height = 3, H
    gpu.enableLeptic(lp)
    try:
        upload(gpu.levels[1], F.F_RHS, rhs1)
        gpu.levels[0].setVal(F.F_PHI, 0.0)
        st = gpu.solveAMRLeptic(1, 1)
        ls = gpu.lepticStats(1)
        assert st["iters"] == amr.iters and st["exitStatus"] == amr.exitStatus
        assert ls["exitStatus"] == lep.exitStatus and ls["horizSolves"] == 0 and ls["usedFullSolver"] == lep.usedFullSolver
        tol = 1e-6 if lep.usedFullSolver else 1e-10
        np.testing.assert_allclose(st["history"], amr.history, rtol=0, atol=tol * amr.history[0])
        np.testing.assert_allclose(ls["resNorms"], lep.resNorms, rtol=0, atol=tol * lep.resNorms[0])
        got = download_valid(gpu.levels[1], F.F_PHI, levels[1].grids)
        want = valid_of(phi[1])
        scale = max(float(np.max(np.abs(w))) for w in want)
        for g_, w_ in zip(got, want):
            np.testing.assert_allclose(g_, w_, rtol=0, atol=max(tol, 1e-9) * scale)
    finally:
        gpu.undefine()
