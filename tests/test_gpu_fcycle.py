"""F-cycle multigrid (numMG < 0, MappedMultiGrid.H:577-619): a recursive F-cycle, pre-smoothing, |numMG| V-cycles with the
"m_cycle = 1" hack, post-smoothing -- one cycle against the oracle (to round-off: 1e-12), and a whole solve's history."""
import numpy as np
import pytest

from helpers import download_valid, make_oracle_solver, make_problem, upload, valid_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("numMG", [-1, -2])
def test_fcycle_matches_and_solve_history(oracle, numMG):
    from somar_amd import AMRPressureSolver
    from somar_amd import api as F
    so = oracle
    dom, grids, dx, Jgup, Jinv = make_problem(so, (32, 32, 16), 16, "stretched", (False, True, False), (2.0, 1.0, 0.5))
    amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv)
    amr.numMG = numMG
    amr.mg.cycle_type = numMG
    s = AMRPressureSolver()
    p = s._p
    s.setAMRMGParameters(p.imin, p.imax, p.eps, -1, p.num_smooth_precond, 2, 2, 2, p.precond_mode, 1, numMG, p.hang,
                         p.norm_thresh, 0)
    s.define(dom.box.lo, dom.box.hi, dom.periodic, dx, [(g.lo, g.hi) for g in grids])
    for q in range(s.num_local_patches):
        _, _, gi = s.patch_box(q)
        jg = [np.asfortranarray(Jgup[gi][d].a[..., d]) for d in range(3)]
        s.setMetricOrtho(q, jg[0], jg[1], jg[2], np.asfortranarray(Jinv[gi].a[..., 0]))
    s.finalize()
    try:
        res = so.random_field(grids, 12345, (0, 0, 0), dom.box)
        so.remove_weighted_mean(res, Jinv)
        corr = so.LevelData(grids, 1, (1, 1, 1))
        amr.mg.init(corr, res)
        amr.mg.one_cycle(corr, res)
        upload(s, F.F_RES, res)
        s.setVal(F.F_CORR, 0.0)
        s.vcycle(F.F_CORR, F.F_RES)
        # (the depth-0 level has 16384 cells: its zero-average mean is a tree sum on the GPU, hence round-off, not bits)
        scale = max(float(np.abs(b).max()) for b in valid_of(corr))
        for a, b in zip(download_valid(s, F.F_CORR, grids), valid_of(corr)):
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-12 * scale)
        x = so.LevelData(grids, 1, (1, 1, 1))
        amr.set_solver_parameters(2, 2, 2, numMG, 20, 1e-6, 1e-15, 1e-30)
        amr.solve(x, res)
        upload(s, F.F_RHS, res)
        st = s.solveResident(True, False)
        assert st["iters"] == amr.iters and st["exitStatus"] == amr.exitStatus
        np.testing.assert_allclose(st["history"], amr.history, rtol=1e-10, atol=1e-10 * amr.history[0])
    finally:
        s.undefine()
