"""The HIP path against the committed vectors of tests/golden/oracle_small.npz (oracle outputs, see make_golden.py):
same seeded inputs, results compared bit for bit without running the oracle's kernels."""
import os

import numpy as np
import pytest

from helpers import download_valid, make_amr_levels, make_gpu_amr, make_gpu_solver, make_problem, upload

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_small.npz"))


def test_single_level_kernels_match_golden(oracle):
    from somar_amd import api as F
    so = oracle
    dom, grids, dx, Jgup, Jinv = make_problem(so, (16, 16, 8), 8, "stretched", (False, True, False), (2.0, 1.0, 0.5))
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv)
    try:
        upload(gpu, F.F_PHI, so.random_field(grids, 3, (1, 1, 1), dom.box))
        rhs = so.random_field(grids, 4, (0, 0, 0), dom.box)
        upload(gpu, F.F_RHS, rhs)
        gpu.relax(0, F.F_PHI, F.F_RHS, 2)
        np.testing.assert_array_equal(download_valid(gpu, F.F_PHI, grids)[0], GOLD["gsrb2_box0"])
        gpu.residual(0, F.F_RES, F.F_PHI, F.F_RHS)
        np.testing.assert_array_equal(download_valid(gpu, F.F_RES, grids)[1], GOLD["residual_box1"])
        gpu.restrictResidual(0, F.FIELD(1, F.F_RES), F.F_PHI, F.F_RHS)
        cg = [g.coarsen(gpu.mgRefRatios()[0]) for g in grids]
        np.testing.assert_array_equal(download_valid(gpu, F.FIELD(1, F.F_RES), cg, 1)[0], GOLD["restrict_box0"])
        b = so.random_field(grids, 12345, (0, 0, 0), dom.box)
        so.remove_weighted_mean(b, Jinv)
        upload(gpu, F.F_RHS, b)
        st = gpu.solveResident(True, False)
        assert [st["iters"], st["exitStatus"]] == list(GOLD["solve_iters_exit"])
        np.testing.assert_allclose(st["history"], GOLD["solve_history"], rtol=1e-10)
    finally:
        gpu.undefine()
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv, relaxMode=3, maxDepth=0)
    try:
        upload(gpu, F.F_PHI, so.random_field(grids, 5, (1, 1, 1), dom.box))
        upload(gpu, F.F_RHS, rhs)
        gpu.relax(0, F.F_PHI, F.F_RHS, 1)
        np.testing.assert_array_equal(download_valid(gpu, F.F_PHI, grids)[0], GOLD["line_gsrb_box0"])
    finally:
        gpu.undefine()


def test_full_metric_kernels_match_golden(oracle):
    from somar_amd import AMRPressureSolver
    from somar_amd import api as F
    so = oracle
    dom, grids, dx, _, _ = make_problem(so, (16, 16, 8), 8, "stretched", (False, True, False), (2.0, 1.0, 0.5))
    Jgf, Jif = so.make_full_metric(grids, dx, (2.0, 1.0, 0.5), dom)
    s = AMRPressureSolver()
    p = s._p
    s.setAMRMGParameters(p.imin, p.imax, p.eps, 0, p.num_smooth_precond, 2, 2, 2, p.precond_mode, 1, p.num_mg, p.hang,
                         p.norm_thresh, 0)
    s.define(dom.box.lo, dom.box.hi, dom.periodic, dx, [(g.lo, g.hi) for g in grids])
    try:
        for q in range(s.num_local_patches):
            _, _, gi = s.patch_box(q)
            s.setMetricFull(q, *[np.asfortranarray(Jgf[gi][d].a) for d in range(3)], np.asfortranarray(Jif[gi].a[..., 0]))
        s.finalize()
        upload(s, F.F_PHI, so.random_field(grids, 6, (1, 1, 1), dom.box))
        upload(s, F.F_RHS, so.random_field(grids, 4, (0, 0, 0), dom.box))
        s.residual(0, F.F_RES, F.F_PHI, F.F_RHS)
        np.testing.assert_array_equal(download_valid(s, F.F_RES, grids)[0], GOLD["full_residual_box0"])
        s.relax(0, F.F_PHI, F.F_RHS, 1)
        np.testing.assert_array_equal(download_valid(s, F.F_PHI, grids)[0], GOLD["full_gsrb_box0"])
    finally:
        s.undefine()


def test_amr_residual_matches_golden(oracle):
    from oracle import somar_amr as am
    from somar_amd import api as F
    so = oracle
    fb = [[so.Box((8, 8, 4), (23, 23, 11))]]
    levels = make_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.5), (False, False, False), [(2, 2, 2)], fb)
    gpu = make_gpu_amr(levels, [(2, 2, 2)])
    try:
        for l, v in enumerate(gpu.levels):
            upload(v, F.F_PHI, so.random_field(levels[l].grids, 5 + l, (1, 1, 1), levels[l].domain.box))
            upload(v, F.F_RHS, so.random_field(levels[l].grids, 50 + l, (0, 0, 0), levels[l].domain.box))
        for ilev in (0, 1):
            gpu.residualLevel(1, 0, ilev)
        gpu.zeroCovered(0, F.F_RES)
        np.testing.assert_array_equal(download_valid(gpu.levels[0], F.F_RES, levels[0].grids)[0], GOLD["amr_res_level0_box0"])
        np.testing.assert_array_equal(download_valid(gpu.levels[1], F.F_RES, levels[1].grids)[0], GOLD["amr_res_level1"])
        got = gpu.levels[1].download(F.F_PHI, 0, (1, 1, 1))
        want = GOLD["amr_fine_phi_with_cf_ghosts"]
        # faces only: edge / corner ghosts of the host array are not defined by the CF interpolation
        for sl in [(0, slice(1, -1), slice(1, -1)), (-1, slice(1, -1), slice(1, -1)), (slice(1, -1), 0, slice(1, -1)),
                   (slice(1, -1), -1, slice(1, -1)), (slice(1, -1), slice(1, -1), 0), (slice(1, -1), slice(1, -1), -1)]:
            np.testing.assert_array_equal(got[sl], want[sl])
    finally:
        gpu.undefine()


def test_ratio4_vcycle_and_leptic_match_golden(oracle):
    """Refinement by (4,1,1) (forced MG depth + mini V-cycle) and the leptic level solver against the committed
    oracle vectors: inputs are rebuilt from seeds, only the GPU computes."""
    from oracle import somar_amr as am
    from somar_amd import api as F
    from helpers import download_valid, make_amr_levels, make_gpu_amr, upload
    so = oracle
    fb = [[so.Box((16, 0, 0), (31, 15, 7)), so.Box((32, 0, 0), (47, 15, 7))]]
    levels = make_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.5), (False, True, False), [(4, 1, 1)], fb)
    gpu = make_gpu_amr(levels, [(4, 1, 1)])
    try:
        assert np.array_equal(np.array(gpu.levels[1].mgRefRatios()), GOLD["amr_ratio4_fine_mg_ratios"])
        res2 = [so.random_field(L.grids, 70 + l, (0, 0, 0), L.domain.box) for l, L in enumerate(levels)]
        for fbx in levels[1].grids:   # zeroCovered on level 0
            for cb, f in zip(res2[0].grids, res2[0].fabs):
                reg = fbx.coarsen((4, 1, 1)) & cb
                if not reg.isEmpty():
                    f.view(reg)[...] = 0.0
        for l, v in enumerate(gpu.levels):
            upload(v, F.F_RES, res2[l])
            v.setVal(F.F_CORR, 0.0)
        gpu.vcycleAMR(1, 0)
        got = download_valid(gpu.levels[1], F.F_CORR, levels[1].grids)[0]
        np.testing.assert_array_equal(got, GOLD["amr_ratio4_vcycle_corr_level1_box0"])
    finally:
        gpu.undefine()
    # leptic
    from somar_amd import LevelLepticSolver
    H, n = 0.005, (32, 32, 8)
    L = (1.0, 1.0, H)
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (False, False, False))
    grids = so.split_domain(dom.box, (16, 16, 8))
    dx = tuple(L[d] / n[d] for d in range(3))
    Jgup, Jinv = so.make_diagonal_metric(grids, dx, L, 3, "stretched", domain=dom)
    rhs = so.random_field(grids, 3, domainBox=dom.box)
    so.remove_weighted_mean(rhs, Jinv)
    s = LevelLepticSolver()
    s.params.max_order = 3
    s.params.domain_height = H
    s.define(dom.box.lo, dom.box.hi, dom.periodic, dx, [(g.lo, g.hi) for g in grids])
    for p_ in range(s.level.num_local_patches):
        _, _, gi = s.level.patch_box(p_)
        jg = [np.asfortranarray(Jgup[gi][d].a[..., d]) for d in range(3)]
        s.level.setMetricOrtho(p_, jg[0], jg[1], jg[2], np.asfortranarray(Jinv[gi].a[..., 0]))
    s.finalize()
    s.level.setVal(F.F_PHI, 0.0)
    upload(s.level, F.F_RHS, rhs)
    st = s.solve()
    assert [st["exitStatus"], st["horizSolves"], int(st["usedFullSolver"])] == list(GOLD["leptic_status_horiz_full"])
    if not st["usedFullSolver"]:
        assert st["resNorms"] == list(GOLD["leptic_res_norms"])
        np.testing.assert_array_equal(download_valid(s.level, F.F_PHI, grids)[0], GOLD["leptic_phi_box0"])
    else:
        np.testing.assert_allclose(st["resNorms"], GOLD["leptic_res_norms"], rtol=1e-9)
        scale = float(np.max(np.abs(GOLD["leptic_phi_box0"])))
        np.testing.assert_allclose(download_valid(s.level, F.F_PHI, grids)[0], GOLD["leptic_phi_box0"], rtol=0,
                                   atol=1e-11 * scale)


def test_cc_projection_and_helmholtz_match_golden(oracle):
    from somar_amd import api as F
    from helpers import smooth_cc_velocity
    so = oracle
    dom, grids, dx, Jgup, Jinv = make_problem(so, (16, 16, 8), 8, "stretched", (False, True, False), (2.0, 1.0, 0.5))
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv)
    try:
        ghost = (1, 1, 1)
        vel = smooth_cc_velocity(so, dom, grids, ghost)
        for p in range(gpu.num_local_patches):
            gpu.uploadCCVel(p, vel[gpu.patch_box(p)[2]].a, ghost)
        gpu.divergenceCC(F.F_RHS, 0.37)
        np.testing.assert_array_equal(download_valid(gpu, F.F_RHS, grids)[0], GOLD["cc_div_over_dt_box0"])
        upload(gpu, F.F_PHI, so.random_field(grids, 17, (1, 1, 1), dom.box))
        gpu.ccCorrect(F.F_PHI, 0.37)
        for p in range(gpu.num_local_patches):
            if gpu.patch_box(p)[2] == 1:
                buf = vel[1].a.copy(order="F")
                gpu.downloadCCVel(p, buf, ghost)
                np.testing.assert_array_equal(buf[1:-1, 1:-1, 1:-1], GOLD["cc_corrected_vel_box1"])
    finally:
        gpu.undefine()
    dom, grids, dx, Jgup, Jinv = make_problem(so, (16, 16, 8), 8, "stretched", (False, True, False), (1.0, 1.0, 0.5))
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv, alpha=1.0, beta=5e-2, bc_type=[1, 1, 0, 0, 1, 0],
                          bc_values=[0.3, -0.2, 0.0, 0.0, 0.1, 0.0])
    try:
        gpu.setAlphaAndBeta(1.0, -0.37)
        upload(gpu, F.F_PHI, so.random_field(grids, 7, (1, 1, 1), dom.box))
        upload(gpu, F.F_RHS, so.random_field(grids, 8, (0, 0, 0), dom.box))
        gpu.relax(0, F.F_PHI, F.F_RHS, 2)
        np.testing.assert_array_equal(download_valid(gpu, F.F_PHI, grids)[0], GOLD["helm_gsrb2_box0"])
        upload(gpu, F.F_HEAT_OLD, so.random_field(grids, 3, (1, 1, 1), dom.box))
        upload(gpu, F.F_HEAT_SRC, so.random_field(grids, 4, (0, 0, 0), dom.box))
        st = gpu.heatStep(2, 0.2)
        assert [st["iters"], st["exitStatus"]] == list(GOLD["tga_iters_exit"])
        np.testing.assert_allclose(st["history"], GOLD["tga_history"], rtol=1e-10)
        np.testing.assert_allclose(download_valid(gpu, F.F_PHI, grids)[0], GOLD["tga_phi_box0"], rtol=0, atol=1e-9)
    finally:
        gpu.undefine()


def test_round2_features_match_golden(oracle):
    """the device-produced bathymetric metric (seen through the 19-point operator), the Dirichlet-topped leptic solve, the
    inflow / outflow sides of the cell-centred divergence and the composite TGA step against the committed vectors"""
    from somar_amd import AMRPressureSolver, LevelLepticSolver
    from somar_amd import api as F
    from helpers import smooth_cc_velocity
    so = oracle
    from oracle import somar_amr as sa
    # 9. bathymetric map on the device
    n, L, bs = (16, 16, 8), (4.0, 2.0, 1.0), (8, 8, 8)
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (False, False, False))
    grids = so.split_domain(dom.box, bs)
    dx = tuple(L[d] / n[d] for d in range(3))
    s = AMRPressureSolver()
    p = s._p
    s.setAMRMGParameters(p.imin, p.imax, p.eps, -1, p.num_smooth_precond, 2, 2, 2, p.precond_mode, 1, p.num_mg, p.hang,
                         p.norm_thresh, 0)
    s.define(dom.box.lo, dom.box.hi, dom.periodic, dx, [(g.lo, g.hi) for g in grids])
    s.setMetricMap(F.MAP_BATHYMETRIC, L, GOLD["bathy_depth_nodes"], (-1, -1))
    s.finalize()
    try:
        upload(s, F.F_PHI, so.random_field(grids, 7, (1, 1, 1), dom.box))
        s.applyOp(0, F.F_RES, F.F_PHI)
        np.testing.assert_array_equal(download_valid(s, F.F_RES, grids)[0], GOLD["bathy_applyop_box0"])
    finally:
        s.undefine()
    # 10. leptic columns with a Dirichlet top
    n, L = (16, 16, 8), (1.0, 1.0, 0.005)
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (False, False, False))
    grids = so.split_domain(dom.box, (8, 8, 8))
    dx = tuple(L[d] / n[d] for d in range(3))
    Jgup, Jinv = so.make_diagonal_metric(grids, dx, L, 3, "stretched", domain=dom)
    lep = LevelLepticSolver()
    lep.params.max_order, lep.params.domain_height = 3, L[2]
    lep.define(dom.box.lo, dom.box.hi, dom.periodic, dx, [(g.lo, g.hi) for g in grids], bc_type=[0, 0, 0, 0, 0, 1])
    try:
        lv = lep.level
        lv.setBCValues([0.0, 0.0, 0.0, 0.0, 0.0, 0.3])
        for q in range(lv.num_local_patches):
            _, _, gi = lv.patch_box(q)
            jg = [np.asfortranarray(Jgup[gi][d].a[..., d]) for d in range(3)]
            lv.setMetricOrtho(q, jg[0], jg[1], jg[2], np.asfortranarray(Jinv[gi].a[..., 0]))
        lep.finalize()
        lv.setVal(F.F_PHI, 0.0)
        upload(lv, F.F_RHS, so.random_field(grids, 9, domainBox=dom.box))
        st = lep.solve(False)
        assert [st["exitStatus"], int(st["usedFullSolver"])] == list(GOLD["leptic_diri_status_full"])
        if not st["usedFullSolver"]:
            assert st["resNorms"] == list(GOLD["leptic_diri_res_norms"])
            np.testing.assert_array_equal(download_valid(lv, F.F_PHI, grids)[0], GOLD["leptic_diri_phi_box0"])
        else:
            np.testing.assert_allclose(st["resNorms"], GOLD["leptic_diri_res_norms"], rtol=1e-9)
    finally:
        lep.undefine()
    # 11. inflow / outflow sides in the cell-centred divergence
    dom, grids, dx, Jgup, Jinv = make_problem(so, (16, 16, 8), 8, "stretched", (False, True, False), (2.0, 1.0, 0.5))
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv)
    try:
        gpu.setVelBC([1, 2, 0, 0, 2, 1], [0.7, 0.0, 0.0, 0.0, 0.0, -0.3])
        vel = smooth_cc_velocity(so, dom, grids, (1, 1, 1))
        for q in range(gpu.num_local_patches):
            gpu.uploadCCVel(q, vel[gpu.patch_box(q)[2]].a, (1, 1, 1))
        gpu.divergenceCC(F.F_RHS, 1.0, True)
        np.testing.assert_array_equal(download_valid(gpu, F.F_RHS, grids)[0], GOLD["cc_div_inflow_outflow_box0"])
    finally:
        gpu.undefine()
    # 12. composite TGA step
    ratios = [(2, 2, 2)]
    fine = [[so.Box((8, 8, 4), (23, 15, 11)), so.Box((8, 16, 4), (23, 23, 11))]]
    levels = make_amr_levels(so, sa, (16, 16, 8), (1.0, 1.0, 0.5), (False, False, False), ratios, fine, cbox=8)
    a = AMRPressureSolver()
    p = a._p
    a.setAMRMGParameters(p.imin, p.imax, p.eps, -1, p.num_smooth_precond, 2, 2, 2, p.precond_mode, 1, p.num_mg, p.hang,
                         p.norm_thresh, 0)
    L0 = levels[0]
    a.defineAMR(L0.domain.box.lo, L0.domain.box.hi, L0.domain.periodic, L0.dx, ratios,
                [[(g.lo, g.hi) for g in Lv.grids] for Lv in levels], alpha=1.0, beta=0.05, bc_type=[1, 1, 1, 0, 1, 1])
    for v in a.levels:
        v.setBCValues([0.1, 0.0, 0.0, 0.0, 0.0, -0.2])
    for Lv, v in zip(levels, a.levels):
        for q in range(v.num_local_patches):
            _, _, gi = v.patch_box(q)
            jg = [np.asfortranarray(Lv.Jgup[gi][d].a[..., d]) for d in range(3)]
            v.setMetricOrtho(q, jg[0], jg[1], jg[2], np.asfortranarray(Lv.Jinv[gi].a[..., 0]))
    a.finalize()
    try:
        for l, Lv in enumerate(levels):
            upload(a.levels[l], F.F_HEAT_OLD, so.random_field(Lv.grids, 3 + l, (1, 1, 1), Lv.domain.box))
            upload(a.levels[l], F.F_HEAT_SRC, so.random_field(Lv.grids, 13 + l, (1, 1, 1), Lv.domain.box))
            upload(a.levels[l], F.F_PHI, so.random_field(Lv.grids, 23 + l, (1, 1, 1), Lv.domain.box))
        st = a.tgaStepAMR(1, 0, 0.2)
        assert [st["iters"], st["exitStatus"]] == list(GOLD["amr_tga_iters_exit"])
        np.testing.assert_allclose(st["history"], GOLD["amr_tga_history"], rtol=1e-10)
        got = download_valid(a.levels[1], F.F_PHI, levels[1].grids)[0]
        np.testing.assert_allclose(got, GOLD["amr_tga_fine_phi_box0"], rtol=0, atol=1e-9)
    finally:
        a.undefine()
