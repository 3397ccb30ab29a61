"""LineGSRBIter2D (CH_SPACEDIM = 2 line relaxation, GSRBF.ChF:1529-1724) on the GPU (k_line_gsrb_2d) against the oracle."""
import numpy as np
import pytest

from oracle import somar_oracle as so
from tests.helpers import download_valid, max_rel_diff, upload, valid_of

pytestmark = pytest.mark.gpu
D, N = 1, 0

CASES = [
    # n (x, y), box x-extent, periodic x, L, metric, bc types [(x lo, hi), (y lo, hi)]
    ((32, 16), 16, False, (1.0, 0.2), "stretched", [(N, N), (N, N)]),
    ((48, 24), 16, True, (2.0, 0.1), "stretched", [(N, N), (N, N)]),
    ((32, 16), 32, False, (1.0, 0.2), "cartesian", [(D, N), (D, D)]),       # Dirichlet ends of the columns: coeff1 = 2
    ((32, 16), 16, False, (1.0, 0.2), "sheared", [(N, N), (N, N)]),         # both cross terms, from the extrapolated copy
    ((40, 12), 8, False, (1.0, 0.3), "sheared", [(N, N), (N, N)]),
]


def _setup(case, alpha=0.0, beta=1.0, **kw):
    from somar_amd import AMRPressureSolver
    n, bx, perx, L, metric, types = case
    dom = so.Domain(so.Box((0, 0, 0), (n[0] - 1, n[1] - 1, 0)), (perx, False, False))
    grids = so.split_domain(dom.box, (bx, n[1], 1))          # columns are never split in the vertical (direction 1)
    dx = (L[0] / n[0], L[1] / n[1], 1.0)
    full = metric == "sheared"
    if full:
        Jgup, Jinv = so.make_full_metric_2d(grids, dx, L, dom, amp=(0.05, 0.04))
    else:
        Jgup, Jinv = so.make_diagonal_metric(grids, dx, (L[0], L[1], 1.0), 2, metric, domain=dom)
    diri = any(t == D for pr in types for t in pr)
    bc = so.BCHolder([list(types[0]), list(types[1]), [N, N]], [[0.0, 0.0]] * 3) if diri else so.BCHolder()
    fac = so.Factory(dom, grids, dx, bc, Jgup, Jinv, alpha=alpha, beta=beta, isDiagonal=not full, ndim=2,
                     relaxMode=so.RELAX_LINE_GSRB, **kw)
    s = AMRPressureSolver()
    s.setSpaceDim(2)
    p = s._p
    s.setAMRMGParameters(p.imin, p.imax, p.eps, kw.get("maxDepth", -1), p.num_smooth_precond, 2, 2, 2, p.precond_mode, 3,
                         p.num_mg, p.hang, p.norm_thresh, 0)
    bct = [types[0][0], types[0][1], types[1][0], types[1][1], N, N]
    s.define(dom.box.lo, dom.box.hi, dom.periodic, dx, [(g.lo, g.hi) for g in grids], alpha=alpha, beta=beta,
             bc_type=bct if diri else None)
    for q in range(s.num_local_patches):
        _, _, gi = s.patch_box(q)
        if full:
            s.setMetricFull(q, np.asfortranarray(Jgup[gi][0].a), np.asfortranarray(Jgup[gi][1].a), None,
                            np.asfortranarray(Jinv[gi].a[..., 0]))
        else:
            s.setMetricOrtho(q, np.asfortranarray(Jgup[gi][0].a[..., 0]), np.asfortranarray(Jgup[gi][1].a[..., 1]), None,
                             np.asfortranarray(Jinv[gi].a[..., 0]))
    s.finalize()
    return dom, grids, fac, Jinv, s


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("alpha_beta", [(0.0, 1.0), (1.0, -0.05)])
def test_line2d_sweeps_bit_exact(case, alpha_beta):
    from somar_amd import api as F
    a, b = alpha_beta
    dom, grids, fac, Jinv, gpu = _setup(case, a, b, maxDepth=0)
    try:
        op = fac.mg_new_op(0, None)
        phi = so.random_field(grids, 41, (1, 1, 0), dom.box)
        rhs = so.random_field(grids, 42, (0, 0, 0), dom.box)
        upload(gpu, F.F_PHI, phi)
        upload(gpu, F.F_RHS, rhs)
        op.relax(phi, rhs, 2)
        gpu.relax(0, F.F_PHI, F.F_RHS, 2)
        for g, w in zip(download_valid(gpu, F.F_PHI, grids), valid_of(phi)):
            np.testing.assert_array_equal(g, w)
    finally:
        gpu.undefine()


@pytest.mark.parametrize("case", [CASES[0], CASES[2], CASES[3]])
def test_line2d_solve_history_matches(case):
    from somar_amd import api as F
    dom, grids, fac, Jinv, gpu = _setup(case)
    try:
        amr = so.AMRMultiGrid(fac, so.BiCGStab())
        rhs = so.random_field(grids, 12345, (0, 0, 0), dom.box)
        if not any(t == D for pr in case[5] for t in pr):
            so.remove_weighted_mean(rhs, Jinv)
        phi = so.LevelData(grids, 1, (1, 1, 0))
        amr.solve(phi, rhs)
        upload(gpu, F.F_RHS, rhs)
        st = gpu.solveResident(True, False)
        assert st["iters"] == amr.iters and st["exitStatus"] == amr.exitStatus
        np.testing.assert_allclose(st["history"], amr.history, rtol=0, atol=1e-10 * amr.history[0])
        assert max_rel_diff(download_valid(gpu, F.F_PHI, grids), valid_of(phi)) < 1e-7
    finally:
        gpu.undefine()


def test_line2d_refuses_columns_split_in_the_vertical():
    from somar_amd import AMRPressureSolver, SomarError
    s = AMRPressureSolver()
    s.setSpaceDim(2)
    p = s._p
    s.setAMRMGParameters(p.imin, p.imax, p.eps, -1, p.num_smooth_precond, 2, 2, 2, p.precond_mode, 3, p.num_mg, p.hang,
                         p.norm_thresh, 0)
    with pytest.raises(SomarError, match="start at the bottom"):
        s.define((0, 0, 0), (15, 15, 0), (False, False, False), (0.1, 0.1, 1.0),
                 [((0, 0, 0), (15, 7, 0)), ((0, 8, 0), (15, 15, 0))])
