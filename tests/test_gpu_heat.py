"""Viscous / diffusive Helmholtz solves on the GPU (somar_solver_set_alpha_beta, somar_heat_step) vs the oracle's
restatement of MappedAMRPoissonOp::setAlphaAndBeta and the level backward-Euler / Crank-Nicolson / TGA integrators
(for TGA the compared history is the second solve's)."""
import numpy as np
import pytest

from helpers import download_valid, make_gpu_solver, make_problem, max_rel_diff, upload, valid_of

pytestmark = pytest.mark.gpu
D, N = 1, 0

CASES = [
    # n, box, periodic, bc types, bc values, nu
    ((32, 16, 16), (16, 8, 8), (False, False, False), [(D, D), (D, D), (D, D)], [(0.0, 0.0)] * 3, 1e-2),
    ((16, 16, 8), 8, (False, True, False), [(D, D), (N, N), (D, N)], [(0.3, -0.2), (0.0, 0.0), (0.1, 0.0)], 5e-2),
]


@pytest.fixture(autouse=True, params=["small", "large"])
def _kernel_path(request, monkeypatch):
    """'large': the kernels levels above 64^3 cells run -- fused red+black sweep, marching residual, fused restriction"""
    if request.param == "large":
        monkeypatch.setenv("SOMAR_FUSED_MIN_CELLS", "0")


def _setup(so, case):
    n, bs, per, types, values, nu = case
    dom, grids, dx, Jgup, Jinv = make_problem(so, n, bs, "stretched", per, (1.0, 1.0, 0.5))
    bc = so.BCHolder([list(t) for t in types], [list(v) for v in values])
    fac = so.Factory(dom, grids, dx, bc, Jgup, Jinv, alpha=1.0, beta=nu)
    amr = so.AMRMultiGrid(fac, so.BiCGStab())
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv, alpha=1.0, beta=nu, bc_type=[t for p in types for t in p],
                          bc_values=[v for p in values for v in p])
    return dom, grids, amr, gpu


@pytest.mark.parametrize("case", CASES)
def test_set_alpha_and_beta_reaches_every_depth_bit_exact(oracle, case):
    from somar_amd import api as F
    so = oracle
    dom, grids, amr, gpu = _setup(so, case)
    try:
        for a, b in ((1.0, -0.37), (1.0, 0.125)):
            so.reset_solver_alpha_and_beta(amr, a, b)
            gpu.setAlphaAndBeta(a, b)
            for d in range(min(amr.mg.depth, 3)):
                op = amr.mg.ops[d]
                g = op.grids
                phi = so.random_field(g, 7 + d, (1, 1, 1), op.domain.box)
                rhs = so.random_field(g, 8 + d, (0, 0, 0), op.domain.box)
                fc, fr, fs = ((F.F_PHI, F.F_RHS, F.F_RES) if d == 0 else
                              (F.FIELD(d, F.F_CORR), F.FIELD(d, F.F_RES), F.FIELD(d, F.F_SCRATCH)))
                upload(gpu, fc, phi, depth=d)
                upload(gpu, fr, rhs, depth=d)
                res = so.LevelData(g, 1)
                op.residual(res, phi, rhs, True)
                gpu.residual(d, fs, fc, fr)
                for x, y in zip(download_valid(gpu, fs, g, d), valid_of(res)):
                    np.testing.assert_array_equal(x, y)
                op.relax(phi, rhs, 2)
                gpu.relax(d, fc, fr, 2)
                for x, y in zip(download_valid(gpu, fc, g, d), valid_of(phi)):
                    np.testing.assert_array_equal(x, y)
    finally:
        gpu.undefine()


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("scheme", [0, 1, 2])
def test_heat_step_matches_oracle(oracle, case, scheme):
    from somar_amd import api as F
    so = oracle
    dom, grids, amr, gpu = _setup(so, case)
    try:
        dt = 0.2
        old = so.random_field(grids, 3, (1, 1, 1), dom.box)
        src = so.random_field(grids, 4, (0, 0, 0), dom.box)
        upload(gpu, F.F_HEAT_OLD, old)
        upload(gpu, F.F_HEAT_SRC, src)
        new = so.LevelData(grids, 1, (1, 1, 1))
        step = [so.level_backward_euler, so.level_crank_nicolson, so.level_tga][scheme]
        step(amr, new, old, src, dt)
        st = gpu.heatStep(scheme, dt)
        assert st["iters"] == amr.iters and st["exitStatus"] == amr.exitStatus
        np.testing.assert_allclose(st["history"], amr.history, rtol=1e-10, atol=1e-13 * amr.history[0])
        assert st["history"][-1] <= 1e-6 * st["history"][0]
        assert max_rel_diff(download_valid(gpu, F.F_PHI, grids), valid_of(new)) < 1e-9
        # a second step from the new state (the coefficients are reset, not compounded)
        upload(gpu, F.F_HEAT_OLD, new)
        new2 = so.LevelData(grids, 1, (1, 1, 1))
        step(amr, new2, new, src, 0.5 * dt)
        st = gpu.heatStep(scheme, 0.5 * dt)
        assert st["iters"] == amr.iters
        np.testing.assert_allclose(st["history"], amr.history, rtol=1e-10, atol=1e-13 * amr.history[0])
        assert max_rel_diff(download_valid(gpu, F.F_PHI, grids), valid_of(new2)) < 1e-9
    finally:
        gpu.undefine()


def test_tga_from_an_initial_guess(oracle):
    """zeroPhi = false: both TGA solves start from the caller's phiNew (the reference saves it in `phis`)"""
    from somar_amd import api as F
    so = oracle
    dom, grids, amr, gpu = _setup(so, CASES[1])
    try:
        old = so.random_field(grids, 3, (1, 1, 1), dom.box)
        src = so.random_field(grids, 4, (0, 0, 0), dom.box)
        guess = so.random_field(grids, 5, (1, 1, 1), dom.box)
        upload(gpu, F.F_HEAT_OLD, old)
        upload(gpu, F.F_HEAT_SRC, src)
        upload(gpu, F.F_PHI, guess)
        so.level_tga(amr, guess, old, src, 0.2, zeroPhi=False)
        st = gpu.heatStep(2, 0.2, zeroPhi=False)
        assert st["iters"] == amr.iters and st["exitStatus"] == amr.exitStatus
        np.testing.assert_allclose(st["history"], amr.history, rtol=1e-10, atol=1e-13 * amr.history[0])
        assert max_rel_diff(download_valid(gpu, F.F_PHI, grids), valid_of(guess)) < 1e-9
    finally:
        gpu.undefine()


@pytest.mark.parametrize("case", CASES)
def test_inhomogeneous_operator_and_residual_bit_exact(oracle, case):
    """applyOp / residual with a_homogeneous = false: the Dirichlet values enter through the ghosts (one component of
    VelocityAMRPoissonOp::applyOpI with viscous walls)"""
    from somar_amd import api as F
    so = oracle
    dom, grids, amr, gpu = _setup(so, case)
    try:
        phi = so.random_field(grids, 11, (1, 1, 1), dom.box)
        rhs = so.random_field(grids, 12, (0, 0, 0), dom.box)
        upload(gpu, F.F_PHI, phi)
        upload(gpu, F.F_RHS, rhs)
        for homog in (False, True):
            lhs = so.LevelData(grids, 1)
            amr.op.apply_op(lhs, phi, homog)
            gpu.applyOpBC(F.F_RES, F.F_PHI, homog)
            for x, y in zip(download_valid(gpu, F.F_RES, grids), valid_of(lhs)):
                np.testing.assert_array_equal(x, y)
            amr.op.residual(lhs, phi, rhs, homog)
            gpu.residualBC(F.F_RES, F.F_PHI, F.F_RHS, homog)
            for x, y in zip(download_valid(gpu, F.F_RES, grids), valid_of(lhs)):
                np.testing.assert_array_equal(x, y)
    finally:
        gpu.undefine()


def test_set_alpha_and_beta_on_a_level_of_a_hierarchy_fails_loudly(oracle):
    """the flux-register scales of an AMR hierarchy carry beta: changing it behind their back must be an error, not a
    silently inconsistent composite operator"""
    from oracle import somar_amr as am
    from somar_amd import SomarError
    from helpers import make_amr_levels, make_gpu_amr
    so = oracle
    fb = [[so.Box((8, 8, 4), (23, 23, 11))]]
    levels = make_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.5), (False, False, False), [(2, 2, 2)], fb)
    gpu = make_gpu_amr(levels, [(2, 2, 2)])
    try:
        for v in gpu.levels:
            with pytest.raises(SomarError):
                v.setAlphaAndBeta(1.0, -0.1)
            with pytest.raises(SomarError):
                v.heatStep(0, 0.1)
    finally:
        gpu.undefine()
