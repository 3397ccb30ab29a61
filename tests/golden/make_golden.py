#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_small.npz: outputs of the CPU ORACLE (oracle/, this repository's restatement of
the reference algorithm) on small seeded problems.

These are NOT outputs of the reference: UNC-CFD/somar ships no golden vectors for this path and cannot be built in
this environment.  The file pins the oracle against ITSELF, so that an edit to oracle/ that changes a
single bit of its results is noticed (tests/test_oracle_golden.py), and gives later rounds a fixed target.

    python tests/golden/make_golden.py            # rewrites oracle_small.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))


def compute():
    from oracle import somar_amr as am
    from oracle import somar_oracle as so
    from helpers import make_amr_levels, make_oracle_solver, make_problem
    out = {}
    # 1. single level, stretched diagonal metric, periodic y: GSRB sweep, residual, restriction, V-cycle history
    dom, grids, dx, Jgup, Jinv = make_problem(so, (16, 16, 8), 8, "stretched", (False, True, False), (2.0, 1.0, 0.5))
    amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv)
    op = amr.mg.ops[0]
    phi = so.random_field(grids, 3, (1, 1, 1), dom.box)
    rhs = so.random_field(grids, 4, (0, 0, 0), dom.box)
    op.relax(phi, rhs, 2)
    out["gsrb2_box0"] = phi[0].view(grids[0])[..., 0].copy()
    res = so.LevelData(grids, 1)
    op.residual(res, phi, rhs, True)
    out["residual_box1"] = res[1].view(grids[1])[..., 0].copy()
    cres = op.create_coarser(res)
    op.restrict_residual(cres, phi, rhs)
    out["restrict_box0"] = cres[0].view(cres.grids[0])[..., 0].copy()
    b = so.random_field(grids, 12345, (0, 0, 0), dom.box)
    so.remove_weighted_mean(b, Jinv)
    x = so.LevelData(grids, 1, (1, 1, 1))
    amr.solve(x, b)
    out["solve_history"] = np.array(amr.history)
    out["solve_iters_exit"] = np.array([amr.iters, amr.exitStatus])
    # 2. vertical-line GSRB
    fac = so.Factory(dom, grids, dx, so.BCHolder(), Jgup, Jinv, relaxMode=so.RELAX_LINE_GSRB, maxDepth=0)
    opl = fac.mg_new_op(0, None)
    phi = so.random_field(grids, 5, (1, 1, 1), dom.box)
    opl.relax(phi, rhs, 1)
    out["line_gsrb_box0"] = phi[0].view(grids[0])[..., 0].copy()
    # 3. non-diagonal metric: 19-point residual and GSRB
    Jgf, Jif = so.make_full_metric(grids, dx, (2.0, 1.0, 0.5), dom)
    opf = so.Factory(dom, grids, dx, so.BCHolder(), Jgf, Jif, isDiagonal=False, maxDepth=0).mg_new_op(0, None)
    phi = so.random_field(grids, 6, (1, 1, 1), dom.box)
    resf = so.LevelData(grids, 1)
    opf.residual(resf, phi, rhs, True)
    out["full_residual_box0"] = resf[0].view(grids[0])[..., 0].copy()
    opf.relax(phi, rhs, 1)
    out["full_gsrb_box0"] = phi[0].view(grids[0])[..., 0].copy()
    # 4. two AMR levels: quadratic CF interpolation, refluxed composite residual
    fb = [[so.Box((8, 8, 4), (23, 23, 11))]]
    levels = make_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.5), (False, False, False), [(2, 2, 2)], fb)
    comp = am.AMRComposite(levels, [(2, 2, 2)], so.BCHolder(), so.BiCGStab())
    phis = [so.random_field(L.grids, 5 + l, (1, 1, 1), L.domain.box) for l, L in enumerate(levels)]
    rhss = [so.random_field(L.grids, 50 + l, (0, 0, 0), L.domain.box) for l, L in enumerate(levels)]
    ress = [so.LevelData(L.grids, 1) for L in levels]
    comp.init(phis, rhss, 1, 0)
    comp.compute_amr_residual(ress, phis, rhss, 1, 0, True)
    out["amr_res_level0_box0"] = ress[0][0].view(levels[0].grids[0])[..., 0].copy()
    out["amr_res_level1"] = ress[1][0].view(levels[1].grids[0])[..., 0].copy()
    # (edge / vertex ghosts next to the CF faces included: interpCFGhosts ends with ExtrapolateCFEV)
    out["amr_fine_phi_with_cf_ghosts"] = phis[1][0].a[..., 0].copy()
    # 5. refinement by (4,1,1): one AMR V-cycle (forced MG depth + mini V-cycle on the fine level)
    fb = [[so.Box((16, 0, 0), (31, 15, 7)), so.Box((32, 0, 0), (47, 15, 7))]]
    levels = make_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.5), (False, True, False), [(4, 1, 1)], fb)
    comp = am.AMRComposite(levels, [(4, 1, 1)], so.BCHolder(), so.BiCGStab())
    zero = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
    res2 = [so.random_field(L.grids, 70 + l, (0, 0, 0), L.domain.box) for l, L in enumerate(levels)]
    comp.zero_covered(0, res2[0])
    comp.init(zero, res2, 1, 0)
    comp.set_bottom_solver(1, 0)
    corr = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
    comp.amr_vcycle(corr, res2, 1, 1, 0)
    out["amr_ratio4_vcycle_corr_level1_box0"] = corr[1][0].view(levels[1].grids[0])[..., 0].copy()
    out["amr_ratio4_fine_mg_ratios"] = np.array(comp.mg[1].mgRefRatios)
    # 6. leptic level solver on a thin domain: residual norms per order, final phi
    from oracle import somar_leptic as sl
    H = 0.005
    n, L = (32, 32, 8), (1.0, 1.0, H)
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (False, False, False))
    grids = so.split_domain(dom.box, (16, 16, 8))
    dx = tuple(L[d] / n[d] for d in range(3))
    Jgup, Jinv = so.make_diagonal_metric(grids, dx, L, 3, "stretched", domain=dom)
    rhs = so.random_field(grids, 3, domainBox=dom.box)
    so.remove_weighted_mean(rhs, Jinv)
    amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv)
    lep = sl.LevelLepticSolver(amr.op, maxOrder=3, domainHeight=H)
    phi = so.LevelData(grids, 1, (1, 1, 1))
    status = lep.solve(phi, rhs)
    out["leptic_res_norms"] = np.array(lep.resNorms)
    out["leptic_status_horiz_full"] = np.array([status, lep.horizSolves, int(lep.usedFullSolver)])
    out["leptic_phi_box0"] = phi[0].view(grids[0])[..., 0].copy()
    # 7. cell-centred level projection: divergence (CellToEdge + wall faces) / dt, gradient + EdgeToCell + correction
    from helpers import smooth_cc_velocity
    dom, grids, dx, Jgup, Jinv = make_problem(so, (16, 16, 8), 8, "stretched", (False, True, False), (2.0, 1.0, 0.5))
    vel = smooth_cc_velocity(so, dom, grids, (1, 1, 1))
    div = so.LevelData(grids, 1)
    so.level_divergence_cc(div, vel, Jinv, grids, dom, dx)
    out["cc_div_over_dt_box0"] = div[0].a[..., 0] / 0.37
    phi = so.random_field(grids, 17, (1, 1, 1), dom.box)
    corr = so.LevelData(grids, 3)
    so.level_gradient_cc(corr, phi, grids, dom, Jgup, dx)
    out["cc_corrected_vel_box1"] = vel[1].view(grids[1]) + (-0.37) * corr[1].a
    # 8. Helmholtz operator with Dirichlet walls after setAlphaAndBeta(1, -0.37): two GSRB sweeps; one TGA step
    bc = so.BCHolder([[1, 1], [0, 0], [1, 0]], [[0.3, -0.2], [0.0, 0.0], [0.1, 0.0]])
    dom, grids, dx, Jgup, Jinv = make_problem(so, (16, 16, 8), 8, "stretched", (False, True, False), (1.0, 1.0, 0.5))
    amr = so.AMRMultiGrid(so.Factory(dom, grids, dx, bc, Jgup, Jinv, alpha=1.0, beta=5e-2), so.BiCGStab())
    so.reset_solver_alpha_and_beta(amr, 1.0, -0.37)
    phi = so.random_field(grids, 7, (1, 1, 1), dom.box)
    rhs = so.random_field(grids, 8, (0, 0, 0), dom.box)
    amr.op.relax(phi, rhs, 2)
    out["helm_gsrb2_box0"] = phi[0].view(grids[0])[..., 0].copy()
    old = so.random_field(grids, 3, (1, 1, 1), dom.box)
    src = so.random_field(grids, 4, (0, 0, 0), dom.box)
    new = so.LevelData(grids, 1, (1, 1, 1))
    so.level_tga(amr, new, old, src, 0.2)
    out["tga_history"] = np.array(amr.history)
    out["tga_iters_exit"] = np.array([amr.iters, amr.exitStatus])
    out["tga_phi_box0"] = new[0].view(grids[0])[..., 0].copy()
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "oracle_small.npz"), **compute())
    print("wrote", os.path.join(HERE, "oracle_small.npz"))
