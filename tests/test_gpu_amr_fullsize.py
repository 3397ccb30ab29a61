"""BASELINE configs C3 and C4 at FULL size (C3: 512x512x64 base + a (2,2,1)-refined central half in x, 50 M cells; C4:
1024x1024x128 base + two (2,2,1) levels, 940 M cells, here on ONE GPU; Cartesian, y periodic) -- too big for the oracle,
so checked through size-independent properties:
  * the refluxed composite operator is conservative: its volume integral over the closed/periodic domain vanishes;
  * a composite solve of a compatible right-hand side converges with the reference's stopping logic, monotonically."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("config", ["c3", "c4"])
def test_composite_operator_conserves_and_solve_converges(config):
    """c3: 512x512x64 + one level (50 M cells); c4: BASELINE C4 on ONE GPU -- 1024x1024x128 + two (2,2,1) levels, 940 M cells"""
    from bench_amr import build_hierarchy
    from somar_amd import api as F
    gpu, levels, cells, _, dx0, ratios = build_hierarchy(config, 1)
    try:
        if config == "c3":
            assert cells == [512 * 512 * 64, 512 * 1024 * 64]
        else:
            assert cells == [1024 * 1024 * 128, 1024 * 2048 * 128, 1024 * 4096 * 128]
        nlev = len(levels)
        for l, v in enumerate(gpu.levels):
            v.fillHash(F.F_PHI, 5 + l)
            v.setVal(F.F_RHS, 0.0)
        # RES = 0 - L_composite[phi] on every level, covered coarse cells zeroed
        for ilev in range(nlev):
            gpu.residualLevel(nlev - 1, 0, ilev)
        for l in range(nlev - 1):
            gpu.zeroCovered(l, F.F_RES)
        total, mag = 0.0, 0.0
        dx = list(dx0)
        for l, v in enumerate(gpu.levels):
            if l > 0:
                dx = [a / b for a, b in zip(dx, ratios[l - 1])]
            v.setVal(F.F_SCRATCH, 1.0)
            total += v.dotProduct(F.F_RES, F.F_SCRATCH) * float(np.prod(dx))   # J = 1: sum of L * dV
            mag = max(mag, v.norm(F.F_RES, 0))
        volume = 15.0 * 3.0 * 2.0
        assert abs(total) < 1e-11 * mag * volume
        # a compatible right-hand side by construction: RHS := RES = -L[phi] (covered coarse cells zero: never used)
        for v in gpu.levels:
            for q in range(v.num_local_patches):
                v.upload(F.F_RHS, q, v.download(F.F_RES, q, (0, 0, 0)), (0, 0, 0))
        # the first AMR V-cycle on this compatible right-hand side contracts strongly (0.02-0.03 at every size tried)
        for v in gpu.levels:
            v.setVal(F.F_CORR, 0.0)
        gpu.vcycleAMR(nlev - 1, 0)
        r0 = max(v.norm(F.F_RES, 0) for v in gpu.levels)
        for ilev in range(nlev):
            gpu.residualLevel(nlev - 1, 0, ilev, res_field=F.F_SCRATCH, phi_field=F.F_CORR, rhs_field=F.F_RES)
        for l in range(nlev - 1):
            gpu.zeroCovered(l, F.F_SCRATCH)
        r1 = max(v.norm(F.F_SCRATCH, 0) for v in gpu.levels)
        assert r1 < 0.1 * r0
        st = gpu.solveAMR(nlev - 1, 0)
        h = st["history"]
        if config == "c3":
            assert st["exitStatus"] == 1 and h[-1] <= 1e-6 * h[0]
            assert all(b < a for a, b in zip(h, h[1:]))
        else:
            # Three levels of (2,2,1) refinement on this 5:1 (x:y) anisotropic grid: point GSRB + the reference's
            # semicoarsening rule (which coarsens x and y together) + piecewise-constant transfers reduce the residual
            # ~100x in three cycles and then STALL (exit status 4 = hang) -- the same history shape at 1/64, 1/8 and full
            # size on the GPU and, at 1/4096 of the size, in the oracle (tools/diag_c4.py).  Reference
            # behaviour on this (our) configuration, not a size effect; what is asserted is what holds everywhere.
            assert st["exitStatus"] in (1, 4)
            assert min(h) < 2e-2 * h[0]
            assert all(b < a for a, b in zip(h[:3], h[1:4]))
    finally:
        gpu.undefine()


def test_composite_tga_step_conserves_the_heat_content_at_c3_size():
    """MappedAMRTGA::oneStep (somar_amr_tga_step) on BASELINE C3 at full size (50 M cells, two levels, Neumann / periodic):
    the refluxed composite operator integrates to zero, so the composite integral of phi grows by dt x the integral of the
    source -- which holds only if the flux-register scales follow beta through the step's four coefficient changes.  The
    solves stop at the solver's eps (1e-6 of their initial residual), which bounds the defect."""
    from bench_amr import build_hierarchy
    from somar_amd import api as F
    nu, dt, S = 0.05, 0.2, 0.5
    gpu, levels, cells, _, dx0, ratios = build_hierarchy("c3", 1, alpha=1.0, beta=nu)
    try:
        nlev = len(levels)
        dV = []
        dx = list(dx0)
        for l in range(nlev):
            if l > 0:
                dx = [a / b for a, b in zip(dx, ratios[l - 1])]
            dV.append(float(np.prod(dx)))
        for l, v in enumerate(gpu.levels):
            v.fillHash(F.F_HEAT_OLD, 31 + l)
            v.setVal(F.F_HEAT_SRC, S)
            v.setVal(F.F_PHI, 0.0)

        def integral(field):
            for l in range(nlev - 1):
                gpu.zeroCovered(l, field)
            for v in gpu.levels:
                v.setVal(F.F_SCRATCH, 1.0)      # the step uses the scratch field (computeAMROperator's zero right-hand side)
            return sum(v.dotProduct(field, F.F_SCRATCH) * dV[l] for l, v in enumerate(gpu.levels))

        i_old = integral(F.F_HEAT_OLD)      # covered coarse cells never reach an uncovered result (the reflux replaces their fluxes)
        volume = 15.0 * 3.0 * 2.0
        st = gpu.tgaStepAMR(nlev - 1, 0, dt)
        assert st["exitStatus"] == 1
        h = st["history"]
        assert h[-1] <= 1e-6 * h[0]
        i_new = integral(F.F_PHI)
        assert abs(i_new - (i_old + dt * S * volume)) < 2e-5 * volume
        assert abs(i_new - i_old) > 0.5 * dt * S * volume
        # the coefficients are still the last solve's (1, -mu1 dt nu): a composite residual with them works as well
        gpu.setAlphaAndBetaAMR(1.0, 1.0)
    finally:
        gpu.undefine()
