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
    # 9. (round 2) BathymetricBaseMap from a nodal depth: J g^{zeta b} on the zeta-faces and 1/J of box 0, and the 19-point
    #    operator applied with that metric
    from oracle import somar_maps as sm
    n, L, bs = (16, 16, 8), (4.0, 2.0, 1.0), (8, 8, 8)
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (False, False, False))
    grids = so.split_domain(dom.box, bs)
    dx = tuple(L[d] / n[d] for d in range(3))
    dlo, dn = (-1, -1), (n[0] + 4, n[1] + 4)
    x = (np.arange(dlo[0], dlo[0] + dn[0]) * dx[0])[:, None]
    y = (np.arange(dlo[1], dlo[1] + dn[1]) * dx[1])[None, :]
    depth = 0.15 + 0.02 * x - 0.03 * y + 0.25 * np.exp(-((x - 1.7) ** 2 + (y - 0.9) ** 2) / 0.5)
    m = sm.BathymetricMap(dx, L, depth, dlo)
    out["bathy_depth_nodes"] = depth
    out["bathy_jgup_zeta_box0"] = sm.fill_jgup(m, grids[0], 2)
    out["bathy_jinv_box0"] = sm.fill_jinv(m, grids[0])
    Jgup = so.FluxData(grids, 3, 3)
    Jinv = so.LevelData(grids, 1, (0, 0, 0))
    for i, g in enumerate(grids):
        for mu in range(3):
            Jgup[i][mu].a[...] = sm.fill_jgup(m, g, mu)
        Jinv[i].a[..., 0] = sm.fill_jinv(m, g)
    fac = so.Factory(dom, grids, dx, so.BCHolder(), Jgup, Jinv, isDiagonal=False)
    mg = so.MultiGrid(fac, so.BiCGStab())
    phi = so.random_field(grids, 7, (1, 1, 1), dom.box)
    lph = so.LevelData(grids, 1)
    mg.ops[0].apply_op(lph, phi, True)
    out["bathy_applyop_box0"] = lph[0].view(grids[0])[..., 0].copy()
    # 10. (round 2) leptic columns with a Dirichlet top: LepticLapackVerticalSolver + dptsv, no horizontal solves
    n, L = (16, 16, 8), (1.0, 1.0, 0.005)
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (False, False, False))
    grids = so.split_domain(dom.box, (8, 8, 8))
    dx = tuple(L[d] / n[d] for d in range(3))
    Jgup, Jinv = so.make_diagonal_metric(grids, dx, L, 3, "stretched", domain=dom)
    bc = so.BCHolder([[0, 0], [0, 0], [0, 1]], [[0.0, 0.0], [0.0, 0.0], [0.0, 0.3]])
    op = so.AMRMultiGrid(so.Factory(dom, grids, dx, bc, Jgup, Jinv), so.BiCGStab()).op
    lep = sl.LevelLepticSolver(op, maxOrder=3, domainHeight=L[2])
    rhs = so.random_field(grids, 9, domainBox=dom.box)
    phi = so.LevelData(grids, 1, (1, 1, 1))
    status = lep.solve(phi, rhs, False)
    out["leptic_diri_res_norms"] = np.array(lep.resNorms)
    out["leptic_diri_status_full"] = np.array([status, int(lep.usedFullSolver)])
    out["leptic_diri_phi_box0"] = phi[0].view(grids[0])[..., 0].copy()
    # 11. (round 2) inflow / outflow sides of BasicVelocityBCGhostClass in the cell-centred divergence
    dom, grids, dx, Jgup, Jinv = make_problem(so, (16, 16, 8), 8, "stretched", (False, True, False), (2.0, 1.0, 0.5))
    vel = smooth_cc_velocity(so, dom, grids, (1, 1, 1))
    div = so.LevelData(grids, 1)
    so.level_divergence_cc(div, vel, Jinv, grids, dom, dx, velbc=([1, 2, 0, 0, 2, 1], [0.7, 0.0, 0.0, 0.0, 0.0, -0.3]))
    out["cc_div_inflow_outflow_box0"] = div[0].a[..., 0].copy()
    # 12. (round 2) MappedAMRTGA::oneStep over two levels: history of the last solve, fine-level solution
    ratios = [(2, 2, 2)]
    fine = [[so.Box((8, 8, 4), (23, 15, 11)), so.Box((8, 16, 4), (23, 23, 11))]]
    levels = make_amr_levels(so, am, (16, 16, 8), (1.0, 1.0, 0.5), (False, False, False), ratios, fine, cbox=8)
    bc = so.BCHolder([[1, 1], [1, 0], [1, 1]], [[0.1, 0.0], [0.0, 0.0], [0.0, -0.2]])
    comp = am.AMRComposite(levels, ratios, bc, so.BiCGStab(), alpha=1.0, beta=0.05)
    old = [so.random_field(Lv.grids, 3 + l, (1, 1, 1), Lv.domain.box) for l, Lv in enumerate(levels)]
    src = [so.random_field(Lv.grids, 13 + l, (1, 1, 1), Lv.domain.box) for l, Lv in enumerate(levels)]
    new = [so.random_field(Lv.grids, 23 + l, (1, 1, 1), Lv.domain.box) for l, Lv in enumerate(levels)]
    am.amr_tga_one_step(comp, new, old, src, 0.2, 0, 1)
    out["amr_tga_history"] = np.array(comp.history)
    out["amr_tga_iters_exit"] = np.array([comp.iters, comp.exitStatus])
    out["amr_tga_fine_phi_box0"] = new[1][0].view(levels[1].grids[0])[..., 0].copy()
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "oracle_small.npz"), **compute())
    print("wrote", os.path.join(HERE, "oracle_small.npz"))
