"""Round-2 features at BASELINE sizes, checked through size-independent properties (the oracle is too slow there):
  * inflow / outflow sides of the velocity BC at C2's 512^3: a uniform through-flow is divergence free in EVERY cell, with
    solid walls the same field has a source and a sink layer that cancel;
  * leptic columns with a Dirichlet (free-surface) top at C3's base size 512 x 512 x 64 on a thin domain: no horizontal solve,
    every order a dptsv pass, the orders converge, and the reported norm is the true residual of the level operator."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_through_flow_is_divergence_free_at_512_cubed():
    from somar_amd import AMRPressureSolver
    from somar_amd import api as F
    n = 512
    dx = (1.0 / n,) * 3
    s = AMRPressureSolver()
    p = s._p
    s.setAMRMGParameters(p.imin, p.imax, p.eps, 0, p.num_smooth_precond, 2, 2, 2, p.precond_mode, 1, p.num_mg, p.hang,
                         p.norm_thresh, 0)
    boxes = [((0, 0, k), (n - 1, n - 1, k + 255)) for k in (0, 256)]
    s.define((0, 0, 0), (n - 1,) * 3, (False, False, False), dx, boxes)
    s.setMetricUniform(1.0, 1.0, 1.0, 1.0)
    s.finalize()
    try:
        U = 0.75
        for q in range(s.num_local_patches):
            lo, hi, _ = s.patch_box(q)
            shp = tuple(h - a + 3 for a, h in zip(lo, hi))
            v = np.zeros(shp + (3,), order="F")
            v[..., 0] = U
            s.uploadCCVel(q, v, (1, 1, 1))
        s.setVelBC([1, 2, 0, 0, 0, 0], [U, 0.0, 0.0, 0.0, 0.0, 0.0])
        s.divergenceCC(F.F_RHS, 1.0, True)
        assert s.norm(F.F_RHS, 0) == 0.0
        s.setVelBC([0] * 6, [0.0] * 6)
        s.divergenceCC(F.F_RHS, 1.0, True)
        assert s.norm(F.F_RHS, 0) == pytest.approx(U / dx[0], rel=1e-12)     # the wall layers: -+U / dx
        s.setVal(F.F_SCRATCH, 1.0)
        assert abs(s.dotProduct(F.F_RHS, F.F_SCRATCH)) < 1e-6 * U / dx[0]    # and they cancel
    finally:
        s.undefine()


def test_dirichlet_topped_leptic_solve_at_c3_base_size():
    from somar_amd import LevelLepticSolver
    from somar_amd.api import F_PHI, F_RES, F_RHS, F_SCRATCH
    n, H, box = (512, 512, 64), 0.002, 128     # lepticity dx / H = 15: hash-rough data still gains an order per order
    L = (15.0, 15.0, H)
    dx = tuple(L[d] / n[d] for d in range(3))
    boxes = [((i, j, 0), (i + box - 1, j + box - 1, n[2] - 1)) for j in range(0, n[1], box) for i in range(0, n[0], box)]
    s = LevelLepticSolver()
    s.params.max_order, s.params.domain_height = 3, H
    s.define((0, 0, 0), tuple(x - 1 for x in n), (False, False, False), dx, boxes, bc_type=[0, 0, 0, 0, 0, 1])
    try:
        assert s.horiz is None                      # gatherVerticalBCTypes switched the horizontal problem off
        s.level.setMetricUniform(1.0, 1.0, 1.0, 1.0)
        s.finalize()
        s.level.fillHash(F_RHS, 12345)
        s.level.setVal(F_PHI, 0.0)
        st = s.solve(True)
        h = st["resNorms"]
        assert st["horizSolves"] == 0 and st["exitStatus"] in (0, 1)
        assert all(b < 0.3 * a for a, b in zip(h, h[1:])) and h[-1] < 1e-2 * h[0], h
        # J = 1: the reported norm is max |rhs - L[phi]| of the level's own operator
        s.level.residualBC(F_RES, F_PHI, F_RHS, True)
        assert s.level.norm(F_RES, 0) == pytest.approx(h[-1], rel=1e-8)
        s.level.setVal(F_SCRATCH, 0.0)
    finally:
        s.undefine()
