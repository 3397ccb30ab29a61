"""GPU parity tests proper: HIP path (through the C ABI) vs the CPU oracle on identical seeded inputs.

Tolerances.  Every stencil kernel reproduces the reference's operation order with FMA contraction
off, so per-kernel results are required to be BIT-EXACT.  The only non-exact steps are reductions
(tree sums on the GPU vs sequential sums on the CPU): the mean removed by ZeroAvgConstInterpPS and the
BiCGStab scalars; those are held to 1e-12 relative per operation, and whole solves to the north
star's 1e-10 relative on the residual max-norm history with identical iteration counts/exit status.
"""
import numpy as np
import pytest

from helpers import (download_valid, make_gpu_solver, make_oracle_solver, make_problem, max_rel_diff, upload,
                     valid_of)

pytestmark = pytest.mark.gpu

CASES = [
    # (n, boxsz, variant, periodic, L)
    ((32, 32, 32), 32, "cartesian", (False, False, False), (1.0, 1.0, 1.0)),
    ((32, 32, 32), 16, "stretched", (False, False, False), (1.0, 1.0, 1.0)),
    ((64, 32, 16), 16, "stretched", (False, True, False), (4.0, 1.0, 0.5)),
    ((48, 24, 24), (24, 12, 8), "stretched", (True, True, True), (1.0, 2.0, 1.0)),
    ((36, 20, 12), (12, 20, 4), "stretched", (False, False, True), (1.0, 1.0, 3.0)),   # ragged / odd box counts
    # tile columns of every lane class of the marching kernels (Level::build_march_tiles; the cases above give 32- / 24-wide
    # class-1 and 4-wide class-4 columns): 64 = 60 (two region rows per wavefront) + 4 (sixteen), 128 = 124 (one) + 4
    ((64, 16, 8), (64, 8, 8), "stretched", (False, True, False), (2.0, 1.0, 0.5)),
    ((128, 8, 8), (128, 8, 8), "stretched", (True, False, False), (4.0, 1.0, 1.0)),
]


@pytest.fixture(scope="module")
def F():
    from somar_amd import api
    return api


@pytest.fixture(params=["twopass", "fused", "fused-narrow"])
def gsrb_mode(request, monkeypatch):
    """LevelGSRB runs either as two colour launches (k_gsrb_ortho) or as one fused red+black launch
    (k_gsrb_fused); SOMAR_FUSED_MIN_CELLS picks per level.  fused-narrow: the marching kernels' tile tables use the narrow
    lane classes for remainder columns (the default only where the metric is uniform, SOMAR_NARROW_7PT=1 forces them onto
    these stretched cases).  All must be bit-identical to the oracle."""
    monkeypatch.setenv("SOMAR_FUSED_MIN_CELLS", "0" if request.param != "twopass" else "1000000000000")
    if request.param == "fused-narrow":
        monkeypatch.setenv("SOMAR_NARROW_7PT", "1")
    return request.param


def _both(oracle, case, **kw):
    n, boxsz, variant, periodic, L = case
    dom, grids, dx, Jgup, Jinv = make_problem(oracle, n, boxsz, variant, periodic, L)
    amr = make_oracle_solver(oracle, dom, grids, dx, Jgup, Jinv, **kw)
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv, **kw)
    return dom, grids, amr, gpu


@pytest.mark.parametrize("case", CASES)
def test_hierarchy_matches(oracle, case):
    dom, grids, amr, gpu = _both(oracle, case)
    assert gpu.depth() == amr.mg.depth
    assert gpu.mgRefRatios() == [tuple(r) for r in amr.mg.mgRefRatios]
    for d in range(gpu.depth()):
        assert gpu.zeroAvg(d) == amr.mg.ops[d].zeroAvg
        info = gpu.levelInfo(d)
        assert info["domain"] == (amr.mg.ops[d].domain.box.lo, amr.mg.ops[d].domain.box.hi)
        assert info["dx"] == tuple(amr.mg.ops[d].dx)
    gpu.undefine()


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("sweeps", [1, 2, 3])
def test_gsrb_sweep_bit_exact(oracle, case, sweeps, gsrb_mode, F):
    so = oracle
    dom, grids, amr, gpu = _both(oracle, case)
    op = amr.mg.ops[0]
    phi = so.random_field(grids, 41, (1, 1, 1), dom.box)
    rhs = so.random_field(grids, 42, (0, 0, 0), dom.box)
    upload(gpu, F.F_PHI, phi)
    upload(gpu, F.F_RHS, rhs)
    op.relax(phi, rhs, sweeps)
    gpu.relax(0, F.F_PHI, F.F_RHS, sweeps)
    got = download_valid(gpu, F.F_PHI, grids)
    for g, w in zip(got, valid_of(phi)):
        np.testing.assert_array_equal(g, w)
    gpu.undefine()


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("alpha_beta", [(0.0, 1.0), (1.0, -0.05)])
def test_residual_and_applyop_bit_exact(oracle, case, alpha_beta, gsrb_mode, F):
    so = oracle
    a, b = alpha_beta
    dom, grids, amr, gpu = _both(oracle, case, alpha=a, beta=b)
    op = amr.mg.ops[0]
    phi = so.random_field(grids, 51, (1, 1, 1), dom.box)
    rhs = so.random_field(grids, 52, (0, 0, 0), dom.box)
    upload(gpu, F.F_PHI, phi)
    upload(gpu, F.F_RHS, rhs)
    res = so.LevelData(grids, 1)
    op.residual(res, phi, rhs, True)
    gpu.residual(0, F.F_RES, F.F_PHI, F.F_RHS)
    for g, w in zip(download_valid(gpu, F.F_RES, grids), valid_of(res)):
        np.testing.assert_array_equal(g, w)
    op.apply_op(res, phi, True)
    gpu.applyOp(0, F.F_RES, F.F_PHI)
    for g, w in zip(download_valid(gpu, F.F_RES, grids), valid_of(res)):
        np.testing.assert_array_equal(g, w)
    gpu.undefine()


@pytest.mark.parametrize("case", CASES[1:4])
def test_helmholtz_gsrb_bit_exact(oracle, case, gsrb_mode, F):
    """alpha != 0 (the viscous-solve use of the same operator)."""
    so = oracle
    dom, grids, amr, gpu = _both(oracle, case, alpha=1.0, beta=-0.05)
    phi = so.random_field(grids, 43, (1, 1, 1), dom.box)
    rhs = so.random_field(grids, 44, (0, 0, 0), dom.box)
    upload(gpu, F.F_PHI, phi)
    upload(gpu, F.F_RHS, rhs)
    amr.mg.ops[0].relax(phi, rhs, 1)
    gpu.relax(0, F.F_PHI, F.F_RHS, 1)
    for g, w in zip(download_valid(gpu, F.F_PHI, grids), valid_of(phi)):
        np.testing.assert_array_equal(g, w)
    assert not gpu.zeroAvg(0)
    gpu.undefine()


@pytest.mark.parametrize("case", CASES)
def test_restrict_and_prolong(oracle, case, F):
    so = oracle
    dom, grids, amr, gpu = _both(oracle, case)
    if amr.mg.depth < 2:
        pytest.skip("no coarser depth")
    op = amr.mg.ops[0]
    r = op.mgCrseRefRatio
    cgrids = [g.coarsen(r) for g in grids]
    phi = so.random_field(grids, 61, (1, 1, 1), dom.box)
    rhs = so.random_field(grids, 62, (0, 0, 0), dom.box)
    upload(gpu, F.F_PHI, phi)
    upload(gpu, F.F_RHS, rhs)
    crse = op.create_coarser(rhs)
    op.restrict_residual(crse, phi, rhs)
    gpu.restrictResidual(0, F.FIELD(1, F.F_RES), F.F_PHI, F.F_RHS)
    for g, w in zip(download_valid(gpu, F.FIELD(1, F.F_RES), cgrids, 1), valid_of(crse)):
        np.testing.assert_array_equal(g, w)
    # prolongation of a random coarse correction
    corr = so.random_field(cgrids, 63, (1, 1, 1), dom.box.coarsen(r))
    upload(gpu, F.FIELD(1, F.F_CORR), corr, depth=1)
    op.prolong_increment(phi, corr)
    gpu.prolongIncrement(0, F.F_PHI, F.FIELD(1, F.F_CORR))
    got, want = download_valid(gpu, F.F_PHI, grids), valid_of(phi)
    if op.zeroAvg:
        assert max_rel_diff(got, want) < 1e-13     # tree-sum vs sequential-sum mean
    else:
        for g, w in zip(got, want):
            np.testing.assert_array_equal(g, w)
    gpu.undefine()


@pytest.mark.parametrize("case", CASES[1:3])
def test_coarse_depth_operator_bit_exact(oracle, case, gsrb_mode, F):
    """Coarse metrics (face-arithmetic Jgup, harmonic Jinv, regenerated lapDiag) are exercised by
    running GSRB + residual on depth 1."""
    so = oracle
    dom, grids, amr, gpu = _both(oracle, case)
    op1 = amr.mg.ops[1]
    cg = op1.grids
    phi = so.random_field(cg, 71, (1, 1, 1), op1.domain.box)
    rhs = so.random_field(cg, 72, (0, 0, 0), op1.domain.box)
    fc, fr, fs = F.FIELD(1, F.F_CORR), F.FIELD(1, F.F_RES), F.FIELD(1, F.F_SCRATCH)
    upload(gpu, fc, phi, depth=1)
    upload(gpu, fr, rhs, depth=1)
    op1.relax(phi, rhs, 1)
    gpu.relax(1, fc, fr, 1)
    for g, w in zip(download_valid(gpu, fc, cg, 1), valid_of(phi)):
        np.testing.assert_array_equal(g, w)
    res = so.LevelData(cg, 1)
    op1.residual(res, phi, rhs, True)
    gpu.residual(1, fs, fc, fr)
    for g, w in zip(download_valid(gpu, fs, cg, 1), valid_of(res)):
        np.testing.assert_array_equal(g, w)
    gpu.undefine()


def test_jacobi_and_precond(oracle, F):
    so = oracle
    case = CASES[2]
    dom, grids, amr, gpu = _both(oracle, case)
    op = amr.mg.ops[0]
    phi = so.LevelData(grids, 1, (1, 1, 1))
    rhs = so.random_field(grids, 82, (0, 0, 0), dom.box)
    upload(gpu, F.F_RHS, rhs)
    op.pre_cond(phi, rhs)
    gpu.preCond(0, F.F_PHI, F.F_RHS)
    for g, w in zip(download_valid(gpu, F.F_PHI, grids), valid_of(phi)):
        np.testing.assert_array_equal(g, w)
    gpu.undefine()
    n, boxsz, variant, periodic, L = case
    dom, grids, dx, Jgup, Jinv = make_problem(so, n, boxsz, variant, periodic, L)
    fac = so.Factory(dom, grids, dx, so.BCHolder(), Jgup, Jinv, relaxMode=so.RELAX_JACOBI)
    opj = fac.mg_new_op(0, None)
    gj = make_gpu_solver(dom, grids, dx, Jgup, Jinv, relaxMode=0)
    phi = so.random_field(grids, 83, (1, 1, 1), dom.box)
    upload(gj, F.F_PHI, phi)
    upload(gj, F.F_RHS, rhs)
    opj.relax(phi, rhs, 3)
    gj.relax(0, F.F_PHI, F.F_RHS, 3)
    for g, w in zip(download_valid(gj, F.F_PHI, grids), valid_of(phi)):
        np.testing.assert_array_equal(g, w)
    gj.undefine()


@pytest.mark.parametrize("case", CASES[:4])
def test_bottom_solver_and_vcycle(oracle, case, F):
    so = oracle
    dom, grids, amr, gpu = _both(oracle, case)
    D = amr.mg.depth
    opb = amr.mg.ops[-1]
    rhs = so.random_field(opb.grids, 91, (0, 0, 0), opb.domain.box)
    so.remove_weighted_mean(rhs, opb.Jinv)
    phi = so.LevelData(opb.grids, 1, (1, 1, 1))
    b = so.BiCGStab()
    b.define(opb, True)
    b.solve(phi, rhs)
    if D > 1:
        fp, fr = F.FIELD(D - 1, F.F_CORR), F.FIELD(D - 1, F.F_RES)
    else:
        fp, fr = F.F_CORR, F.F_RES
    upload(gpu, fr, rhs, depth=D - 1)
    gpu.setVal(fp, 0.0)
    it, ex = gpu.bottomSolve(fp, fr)
    assert (it, ex) == (b.iters, b.exitStatus)
    assert max_rel_diff(download_valid(gpu, fp, opb.grids, D - 1), valid_of(phi)) < 1e-9
    # one V-cycle from zero correction
    res = so.random_field(grids, 92, (0, 0, 0), dom.box)
    so.remove_weighted_mean(res, amr.op.Jinv)
    corr = so.LevelData(grids, 1, (1, 1, 1))
    amr.mg.init(corr, res)
    amr.mg.bottomSolver = so.BiCGStab()          # fresh: no convergence metric set, like the GPU handle
    amr.mg.bottomSolver.define(opb, True)
    amr.mg.one_cycle(corr, res)
    upload(gpu, F.F_RES, res)
    gpu.setVal(F.F_CORR, 0.0)
    gpu.vcycle(F.F_CORR, F.F_RES)
    assert max_rel_diff(download_valid(gpu, F.F_CORR, grids), valid_of(corr)) < 1e-9
    gpu.undefine()


@pytest.mark.parametrize("case", CASES[:4])
def test_one_launch_bottom_solver_equals_the_launch_by_launch_one(oracle, case, F, monkeypatch):
    """k_tiny_bicgstab (the whole BiCGStab bottom solve in one single-workgroup launch, default on bottom levels of at most
    512 cells) against PressureSolver::bottom_solve's launch-by-launch path (SOMAR_FUSED_BOTTOM_MAX_CELLS=0, read when the
    solver is created): same iteration count, same exit code, same bits -- bottom solves and whole V-cycles."""
    so = oracle
    n, boxsz, variant, periodic, L = case
    dom, grids, dx, Jgup, Jinv = make_problem(so, n, boxsz, variant, periodic, L)
    amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv)
    D = amr.mg.depth
    opb = amr.mg.ops[-1]
    rhs = so.random_field(opb.grids, 91, (0, 0, 0), opb.domain.box)
    so.remove_weighted_mean(rhs, opb.Jinv)
    res = so.random_field(grids, 92, (0, 0, 0), dom.box)
    so.remove_weighted_mean(res, amr.op.Jinv)
    fp, fr = (F.FIELD(D - 1, F.F_CORR), F.FIELD(D - 1, F.F_RES)) if D > 1 else (F.F_CORR, F.F_RES)
    out, cyc = {}, {}
    for mode in ("0", "100000"):
        monkeypatch.setenv("SOMAR_FUSED_BOTTOM_MAX_CELLS", mode)
        monkeypatch.setenv("SOMAR_BOX_BOTTOM", "0")
        gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv)
        upload(gpu, fr, rhs, depth=D - 1)
        gpu.setVal(fp, 0.0)
        it, ex = gpu.bottomSolve(fp, fr)
        assert gpu.bottomKind() == 0 if mode == "0" else gpu.bottomKind() in (0, 1)   # (case3's tiles do not fit the kernel: launch path)
        out[mode] = (it, ex, download_valid(gpu, fp, opb.grids, D - 1))
        upload(gpu, F.F_RES, res)
        gpu.setVal(F.F_CORR, 0.0)
        gpu.vcycle(F.F_CORR, F.F_RES)
        cyc[mode] = download_valid(gpu, F.F_CORR, grids)
        gpu.undefine()
    assert out["0"][:2] == out["100000"][:2]
    for a, b in zip(out["0"][2], out["100000"][2]):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(cyc["0"], cyc["100000"]):
        np.testing.assert_array_equal(a, b)


_ORACLE_SOLVES = {}   # (case, smooth) -> the oracle's solve: shared by the two sweep-kernel variants of the test below


@pytest.mark.parametrize("case", CASES[:6])   # (the 128 x 8 x 8 bar: its kernels are covered above)
@pytest.mark.parametrize("smooth", [(2, 2, 2), (4, 4, 2)])
def test_full_solve_history_matches(oracle, case, smooth, gsrb_mode):
    if gsrb_mode == "fused-narrow" and case not in (CASES[1], CASES[5]):
        pytest.skip("the narrow lane classes are covered kernel by kernel above; whole solves run them on two of the cases")
    so = oracle
    pre, post, bottom = smooth
    n, boxsz, variant, periodic, L = case
    dom, grids, dx, Jgup, Jinv = make_problem(so, n, boxsz, variant, periodic, L)
    key = (case, smooth)
    if key not in _ORACLE_SOLVES:   # the numpy oracle is the slow half of this test: once per problem
        amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv, pre=pre, post=post, bottom=bottom)
        rhs = so.random_field(grids, 12345, (0, 0, 0), dom.box)
        so.remove_weighted_mean(rhs, amr.op.Jinv)
        phi = so.LevelData(grids, 1, (1, 1, 1))
        amr.solve(phi, rhs)
        _ORACLE_SOLVES[key] = {"iters": amr.iters, "exitStatus": amr.exitStatus, "history": list(amr.history),
                               "depth": amr.mg.depth, "phi": [np.array(v) for v in valid_of(phi)],
                               "rhs": [np.asfortranarray(f.a[..., 0]) for f in rhs.fabs],
                               "shape": [f.a.shape[:3] for f in phi.fabs]}
    o = _ORACLE_SOLVES[key]
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv, pre=pre, post=post, bottom=bottom)
    gphi = [np.zeros(shp, order="F") for shp in o["shape"]]
    st = gpu.solve(gphi, o["rhs"], 0, 0, True, False)
    assert st["iters"] == o["iters"]
    assert st["exitStatus"] == o["exitStatus"]
    h_g, h_o = np.array(st["history"]), np.array(o["history"])
    if o["depth"] == 1:
        # boxes too thin to coarsen: the "V-cycle" is ONE long BiCGStab solve on the full grid, whose
        # iterates amplify the summation-order difference of the dot products (tree vs sequential).
        # Both must converge below eps; the histories are only comparable at that level.
        assert h_g[0] == pytest.approx(h_o[0], rel=1e-12)
        assert h_g[-1] <= 1e-6 * h_g[0] and h_o[-1] <= 1e-6 * h_o[0]
        gpu.undefine()
        return
    np.testing.assert_allclose(h_g, h_o, rtol=1e-10, atol=1e-10 * h_o[0])
    # solution agrees (up to the same tolerance scaled by the condition of the last V-cycles)
    got = [a[1:-1, 1:-1, 1:-1] for a in gphi]
    assert max_rel_diff(got, o["phi"]) < 1e-8
    gpu.undefine()


def test_solve_with_initial_guess_and_best_phi(oracle):
    so = oracle
    dom, grids, amr, gpu = _both(oracle, CASES[2])
    rhs = so.random_field(grids, 7, (0, 0, 0), dom.box)
    so.remove_weighted_mean(rhs, amr.op.Jinv)
    phi = so.random_field(grids, 8, (1, 1, 1), dom.box)
    gphi = [np.asfortranarray(f.a[..., 0]).copy(order="F") for f in phi.fabs]
    grhs = [np.asfortranarray(f.a[..., 0]) for f in rhs.fabs]
    amr.solve(phi, rhs, zeroPhi=False)
    st = gpu.solve(gphi, grhs, 0, 0, False, False)
    assert st["iters"] == amr.iters and st["exitStatus"] == amr.exitStatus
    np.testing.assert_allclose(st["history"], amr.history, rtol=1e-10, atol=1e-10 * amr.history[0])
    gpu.undefine()


def test_remove_mean_and_norms(oracle, F):
    so = oracle
    dom, grids, amr, gpu = _both(oracle, CASES[3])
    f = so.random_field(grids, 99, (0, 0, 0), dom.box)
    upload(gpu, F.F_RHS, f)
    g = so.random_field(grids, 98, (0, 0, 0), dom.box)
    upload(gpu, F.F_RES, g)
    for order in (0, 1, 2):
        assert gpu.norm(F.F_RHS, order) == pytest.approx(so.ld_norm(f, order), rel=1e-13)
    assert gpu.dotProduct(F.F_RHS, F.F_RES) == pytest.approx(so.ld_dot(f, g), rel=1e-11, abs=1e-11)
    so.remove_weighted_mean(f, amr.op.Jinv)
    gpu.removeMean(F.F_RHS)
    assert max_rel_diff(download_valid(gpu, F.F_RHS, grids), valid_of(f)) < 1e-14
    gpu.undefine()


def test_error_paths(F):
    from somar_amd import AMRPressureSolver, SomarError
    s = AMRPressureSolver()
    with pytest.raises(SomarError):   # overlapping boxes
        s.define((0, 0, 0), (7, 7, 7), (0, 0, 0), (1, 1, 1), [((0, 0, 0), (7, 7, 7)), ((4, 4, 4), (7, 7, 7))])
    s = AMRPressureSolver()
    with pytest.raises(SomarError):   # an unknown BC code -> loud failure, no silent fallback
        s.define((0, 0, 0), (7, 7, 7), (0, 0, 0), (1, 1, 1), [((0, 0, 0), (7, 7, 7))], bc_type=[7, 0, 0, 0, 0, 0])
    s = AMRPressureSolver()
    s.define((0, 0, 0), (7, 7, 7), (0, 0, 0), (1, 1, 1), [((0, 0, 0), (7, 7, 7))])
    with pytest.raises(SomarError):   # field access before finalize
        s.setVal(F.F_PHI, 0.0)
    s.undefine()


def test_full_size_properties_512(oracle, F):
    """BASELINE config C2 size (512^3, one box, stretched diagonal metric): size-independent
    properties instead of a CPU comparison."""
    so = oracle
    n = 512
    dom = so.Domain(so.Box((0, 0, 0), (n - 1,) * 3))
    grids = [dom.box]
    dx = (1.0 / n,) * 3
    Jgup, Jinv = so.make_diagonal_metric(grids, dx, (1.0, 1.0, 1.0), 3, "stretched")
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv)
    del Jgup
    assert gpu.depth() == 8 and gpu.mgRefRatios() == [(2, 2, 2)] * 7
    assert all(gpu.zeroAvg(d) for d in range(8))
    # k1: constants are in the null space
    gpu.setVal(F.F_PHI, 2.5)
    gpu.applyOp(0, F.F_RES, F.F_PHI)
    assert gpu.norm(F.F_RES, 0) < 1e-6          # |L[c]| ~ c * eps / dx^2 = 2.5 * 2.2e-16 * 2.6e5 * O(10)
    # k4: a discretely exact solution is a fixed point of GSRB
    gpu.fillHash(F.F_PHI, 5)
    gpu.applyOp(0, F.F_RHS, F.F_PHI)
    before = gpu.norm(F.F_PHI, 2)
    gpu.relax(0, F.F_PHI, F.F_RHS, 1)
    gpu.residual(0, F.F_RES, F.F_PHI, F.F_RHS)
    assert gpu.norm(F.F_RES, 0) < 1e-9 * gpu.norm(F.F_RHS, 0)
    assert abs(gpu.norm(F.F_PHI, 2) - before) < 1e-9 * before
    # V-cycle contraction on the standard synthetic rhs: uniform(-1,1) minus its J-weighted mean
    gpu.fillHash(F.F_RHS, 12345)
    gpu.removeMean(F.F_RHS)
    # Point GSRB + piecewise-constant transfer (m_P + m_R = 2, not > 2) lose h-independence on this
    # strongly stretched metric: the CPU oracle stalls the same way (256^3: 1.0, .49, .22, .19, .18,
    # .182 -> "hang", exitStatus 4).  Reference behaviour, so only the first cycles are asserted.
    st = gpu.solveResident(True, False)
    h = st["history"]
    assert st["status"] == 0 and len(h) >= 4, st
    assert h[1] < 0.6 * h[0] and h[2] < 0.7 * h[1], h
    assert st["final_rnorm"] == min(h), st      # best-phi rollback (MappedAMRMultiGrid.H:1091-1131)
    gpu.undefine()
    # Cartesian metric at the same size: textbook multigrid contraction and a clean goRedu exit
    Jgup, Jinv = so.make_diagonal_metric(grids, dx, (1.0, 1.0, 1.0), 3, "cartesian")
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv)
    del Jgup
    gpu.fillHash(F.F_RHS, 12345)
    gpu.removeMean(F.F_RHS)
    st = gpu.solveResident(True, False)
    h = st["history"]
    assert st["exitStatus"] == 1 and st["iters"] <= 7, st
    assert all(h[i + 1] < 0.2 * h[i] for i in range(len(h) - 1)), h
    gpu.undefine()
