"""MAC level projection on the GPU vs the oracle (diagonal metric).

Kernel-level pieces (divergence, gradient+correction) are bit-exact; the full projection inherits the
solve's tolerance (residual history to 1e-10, solution to ~1e-8 of its magnitude)."""
import numpy as np
import pytest

from helpers import download_valid, make_gpu_solver, make_oracle_solver, make_problem, max_rel_diff, upload, valid_of

pytestmark = pytest.mark.gpu

CASES = [
    ((32, 16, 16), 16, "cartesian", (True, True, True), (2.0, 1.0, 1.0)),
    ((32, 16, 16), (16, 8, 16), "stretched", (False, True, False), (2.0, 1.0, 1.0)),
    ((24, 20, 12), (12, 20, 4), "stretched", (False, False, False), (1.0, 1.0, 0.5)),
]


def _velocity(so, dom, grids):
    """a smooth face field, identical on faces shared by two boxes, zero normal flux on physical walls"""
    vel = so.FluxData(grids, 1, 3)
    n = dom.box.size()
    for i, g in enumerate(grids):
        for d in range(3):
            fb = vel[i][d].box
            I, J, K = np.meshgrid(*[np.arange(fb.lo[a], fb.hi[a] + 1) for a in range(3)], indexing="ij")
            v = (np.sin(2 * np.pi * np.mod(I, n[0]) / n[0] + 0.1 * d) * np.cos(2 * np.pi * np.mod(J, n[1]) / n[1])
                 * np.cos(2 * np.pi * np.mod(K, n[2]) / n[2] + d))
            if not dom.periodic[d]:
                idx = [I, J, K][d]
                v = np.where((idx == dom.box.lo[d]) | (idx == dom.box.hi[d] + 1), 0.0, v)
            vel[i][d].a[..., 0] = v
    return vel


@pytest.mark.parametrize("case", CASES)
def test_divergence_and_gradient_correction_bit_exact(oracle, case):
    from somar_amd import api as F
    so = oracle
    n, boxsz, variant, periodic, L = case
    dom, grids, dx, Jgup, Jinv = make_problem(so, n, boxsz, variant, periodic, L)
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv)
    vel = _velocity(so, dom, grids)
    for p in range(gpu.num_local_patches):
        _, _, gi = gpu.patch_box(p)
        for d in range(3):
            gpu.uploadVel(d, p, np.asfortranarray(vel[gi][d].a[..., 0]))
    # divergence / dt
    dt = 0.37
    div = so.LevelData(grids, 1)
    so.level_divergence_mac(div, vel, Jinv, grids, dx)
    for f in div.fabs:
        f.a /= dt
    gpu.divergenceMAC(F.F_RHS, dt)
    for g, w in zip(download_valid(gpu, F.F_RHS, grids), valid_of(div)):
        np.testing.assert_array_equal(g, w)
    # gradient + correction with a given phi
    phi = so.random_field(grids, 17, (1, 1, 1), dom.box)
    upload(gpu, F.F_PHI, phi)
    corr = so.FluxData(grids, 1, 3)
    so.level_gradient_mac(corr, phi, grids, dom, Jgup, dx)
    for i in range(len(grids)):
        for d in range(3):
            vel[i][d].a += (-dt) * corr[i][d].a
    gpu.macCorrect(F.F_PHI, dt)
    for p in range(gpu.num_local_patches):
        _, _, gi = gpu.patch_box(p)
        for d in range(3):
            np.testing.assert_array_equal(gpu.downloadVel(d, p), vel[gi][d].a[..., 0])
    gpu.undefine()


@pytest.mark.parametrize("case", CASES[:2])
def test_level_projection_matches_oracle(oracle, case):
    so = oracle
    n, boxsz, variant, periodic, L = case
    dom, grids, dx, Jgup, Jinv = make_problem(so, n, boxsz, variant, periodic, L)
    amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv)
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv)
    vel = _velocity(so, dom, grids)
    gvel = [[np.asfortranarray(vel[gpu.patch_box(p)[2]][d].a[..., 0]).copy(order="F")
             for p in range(gpu.num_local_patches)] for d in range(3)]
    phi = so.LevelData(grids, 1, (1, 1, 1))
    dt = 0.5
    so.mac_level_project(amr, vel, phi, dt)
    st = gpu.levelProject(gvel, dt)
    assert st["iters"] == amr.iters and st["exitStatus"] == amr.exitStatus
    np.testing.assert_allclose(st["history"], amr.history, rtol=1e-10, atol=1e-10 * amr.history[0])
    for d in range(3):
        want = [vel[gpu.patch_box(p)[2]][d].a[..., 0] for p in range(gpu.num_local_patches)]
        assert max_rel_diff(gvel[d], want) < 1e-8
    # the projected field is discretely divergence-free away from physical walls (where the reference's
    # extrapolated boundary gradient leaves a residual that its velocity BC removes later)
    div = so.LevelData(grids, 1)
    gv = so.FluxData(grids, 1, 3)
    for p in range(gpu.num_local_patches):
        gi = gpu.patch_box(p)[2]
        for d in range(3):
            gv[gi][d].a[..., 0] = gvel[d][p]
    so.level_divergence_mac(div, gv, Jinv, grids, dx)
    inter = dom.box.grow([0 if periodic[d] else -1 for d in range(3)])
    worst = max(float(np.max(np.abs(f.view(g & inter)))) for g, f in zip(grids, div.fabs))
    assert worst < 1e-4 * amr.history[0] * dt
    gpu.undefine()


# ---- non-diagonal metric: MAPPEDMACGRAD with fillExtrap-type extrap and the extrapolation BC on phi's own ghosts ----
FULL_CASES = [
    ((16, 16, 8), 8, (False, True, False), (2.0, 1.0, 0.5)),
    ((24, 16, 8), (12, 8, 8), (False, False, False), (1.5, 1.0, 0.5)),
]


def _full_setup(so, case):
    from somar_amd import AMRPressureSolver
    n, bs, per, L = case
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), per)
    grids = so.split_domain(dom.box, bs)
    dx = tuple(L[d] / n[d] for d in range(3))
    Jgup, Jinv = so.make_full_metric(grids, dx, L, dom)
    fac = so.Factory(dom, grids, dx, so.BCHolder(), Jgup, Jinv, isDiagonal=False)
    s = AMRPressureSolver()
    s.define(dom.box.lo, dom.box.hi, dom.periodic, dx, [(g.lo, g.hi) for g in grids])
    for q in range(s.num_local_patches):
        _, _, gi = s.patch_box(q)
        s.setMetricFull(q, *[np.asfortranarray(Jgup[gi][d].a) for d in range(3)], np.asfortranarray(Jinv[gi].a[..., 0]))
    s.finalize()
    return dom, grids, dx, Jgup, Jinv, fac, s


@pytest.mark.parametrize("case", FULL_CASES)
def test_full_metric_gradient_correction_bit_exact_and_projection(oracle, case):
    from somar_amd import api as F
    so = oracle
    dom, grids, dx, Jgup, Jinv, fac, gpu = _full_setup(so, case)
    try:
        amr = so.AMRMultiGrid(fac, so.BiCGStab())
        vel = _velocity(so, dom, grids)
        for p in range(gpu.num_local_patches):
            _, _, gi = gpu.patch_box(p)
            for d in range(3):
                gpu.uploadVel(d, p, np.asfortranarray(vel[gi][d].a[..., 0]))
        dt = 0.37
        phi = so.random_field(grids, 17, (1, 1, 1), dom.box)
        upload(gpu, F.F_PHI, phi)
        corr = so.FluxData(grids, 1, 3)
        so.level_gradient_mac(corr, phi, grids, dom, Jgup, dx, op=amr.op)
        ref = [[vel[i][d].a.copy() for d in range(3)] for i in range(len(grids))]
        for i in range(len(grids)):
            for d in range(3):
                ref[i][d] += (-dt) * corr[i][d].a
        gpu.macCorrect(F.F_PHI, dt)
        for p in range(gpu.num_local_patches):
            _, _, gi = gpu.patch_box(p)
            for d in range(3):
                np.testing.assert_array_equal(gpu.downloadVel(d, p), ref[gi][d][..., 0])
        # whole projection
        gvel = [[np.asfortranarray(vel[gpu.patch_box(p)[2]][d].a[..., 0]).copy(order="F")
                 for p in range(gpu.num_local_patches)] for d in range(3)]
        phi2 = so.LevelData(grids, 1, (1, 1, 1))
        so.mac_level_project(amr, vel, phi2, 0.5)
        st = gpu.levelProject(gvel, 0.5)
        assert st["iters"] == amr.iters and st["exitStatus"] == amr.exitStatus
        np.testing.assert_allclose(st["history"], amr.history, rtol=1e-9, atol=1e-10 * amr.history[0])
        for d in range(3):
            want = [vel[gpu.patch_box(p)[2]][d].a[..., 0] for p in range(gpu.num_local_patches)]
            assert max_rel_diff(gvel[d], want) < 1e-7
    finally:
        gpu.undefine()


# ---- cell-centred level projection (LevelCCProjector): CellToEdge + wall BC + divergence, gradient + EdgeToCell ----
def _cc_velocity(so, dom, grids, ghost):
    from helpers import smooth_cc_velocity
    return smooth_cc_velocity(so, dom, grids, ghost)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("ghost", [(1, 1, 1), (2, 1, 2)])
def test_cc_divergence_and_gradient_correction_bit_exact(oracle, case, ghost):
    from somar_amd import api as F
    so = oracle
    n, boxsz, variant, periodic, L = case
    dom, grids, dx, Jgup, Jinv = make_problem(so, n, boxsz, variant, periodic, L)
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv)
    vel = _cc_velocity(so, dom, grids, ghost)
    for p in range(gpu.num_local_patches):
        gpu.uploadCCVel(p, vel[gpu.patch_box(p)[2]].a, ghost)
    dt = 0.37
    for wall in (True, False):
        div = so.LevelData(grids, 1)
        so.level_divergence_cc(div, vel, Jinv, grids, dom, dx, wall=wall)
        for f in div.fabs:
            f.a /= dt
        gpu.divergenceCC(F.F_RHS, dt, wall)
        for g, w in zip(download_valid(gpu, F.F_RHS, grids), valid_of(div)):
            np.testing.assert_array_equal(g, w)
    # gradient + EdgeToCell + correction with a given phi
    phi = so.random_field(grids, 17, (1, 1, 1), dom.box)
    upload(gpu, F.F_PHI, phi)
    corr = so.LevelData(grids, 3)
    so.level_gradient_cc(corr, phi, grids, dom, Jgup, dx)
    want = [f.a.copy(order="F") for f in vel.fabs]
    for i, g in enumerate(grids):
        vel[i].view(g)[...] += (-dt) * corr[i].a
    gpu.ccCorrect(F.F_PHI, dt)
    for p in range(gpu.num_local_patches):
        gi = gpu.patch_box(p)[2]
        gpu.downloadCCVel(p, want[gi], ghost)               # valid cells overwritten, ghosts untouched
        np.testing.assert_array_equal(want[gi], vel[gi].a)
    gpu.undefine()


@pytest.mark.parametrize("case", CASES[:2])
def test_cc_level_projection_matches_oracle(oracle, case):
    so = oracle
    n, boxsz, variant, periodic, L = case
    dom, grids, dx, Jgup, Jinv = make_problem(so, n, boxsz, variant, periodic, L)
    amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv)
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv)
    ghost = (1, 1, 1)
    vel = _cc_velocity(so, dom, grids, ghost)
    gvel = [vel[gpu.patch_box(p)[2]].a.copy(order="F") for p in range(gpu.num_local_patches)]
    phi = so.LevelData(grids, 1, (1, 1, 1))
    dt = 0.5
    so.cc_level_project(amr, vel, phi, dt)
    st = gpu.levelProjectCC(gvel, ghost, dt)
    assert st["iters"] == amr.iters and st["exitStatus"] == amr.exitStatus
    np.testing.assert_allclose(st["history"], amr.history, rtol=1e-10, atol=1e-10 * amr.history[0])
    want = [vel[gpu.patch_box(p)[2]].a for p in range(gpu.num_local_patches)]
    assert max_rel_diff(gvel, want) < 1e-8
    gpu.undefine()


@pytest.mark.parametrize("case", FULL_CASES)
def test_cc_correction_with_a_non_diagonal_metric_bit_exact(oracle, case):
    from somar_amd import api as F
    so = oracle
    dom, grids, dx, Jgup, Jinv, fac, gpu = _full_setup(so, case)
    try:
        amr = so.AMRMultiGrid(fac, so.BiCGStab())
        ghost = (1, 1, 1)
        vel = _cc_velocity(so, dom, grids, ghost)
        for p in range(gpu.num_local_patches):
            gpu.uploadCCVel(p, vel[gpu.patch_box(p)[2]].a, ghost)
        dt = 0.37
        phi = so.random_field(grids, 17, (1, 1, 1), dom.box)
        upload(gpu, F.F_PHI, phi)
        corr = so.LevelData(grids, 3)
        so.level_gradient_cc(corr, phi, grids, dom, Jgup, dx, op=amr.op)
        got = [f.a.copy(order="F") for f in vel.fabs]
        for i, g in enumerate(grids):
            vel[i].view(g)[...] += (-dt) * corr[i].a
        gpu.ccCorrect(F.F_PHI, dt)
        for p in range(gpu.num_local_patches):
            gi = gpu.patch_box(p)[2]
            gpu.downloadCCVel(p, got[gi], ghost)
            np.testing.assert_array_equal(got[gi], vel[gi].a)
    finally:
        gpu.undefine()


@pytest.mark.parametrize("case", CASES)
def test_mac_velocity_wall_bc_bit_exact(oracle, case):
    """somar_vel_wall_bc = the solid-wall velocity BC levelDivergenceMAC applies in place: wall-normal faces zero, all
    other faces untouched, periodic directions skipped"""
    so = oracle
    n, boxsz, variant, periodic, L = case
    dom, grids, dx, Jgup, Jinv = make_problem(so, n, boxsz, variant, periodic, L)
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv)
    try:
        vel = so.FluxData(grids, 1, 3)
        rng = np.random.default_rng(5)
        for i in range(len(grids)):
            for d in range(3):
                vel[i][d].a[...] = rng.uniform(0.5, 1.5, vel[i][d].a.shape)
        for p in range(gpu.num_local_patches):
            gi = gpu.patch_box(p)[2]
            for d in range(3):
                gpu.uploadVel(d, p, np.asfortranarray(vel[gi][d].a[..., 0]))
        so.set_wall_normal_flux(vel, grids, dom)
        gpu.velWallBC()
        nz = 0
        for p in range(gpu.num_local_patches):
            gi = gpu.patch_box(p)[2]
            for d in range(3):
                got = gpu.downloadVel(d, p)
                np.testing.assert_array_equal(got, vel[gi][d].a[..., 0])
                nz += int((got == 0.0).sum())
        assert (nz > 0) == (not all(periodic))
    finally:
        gpu.undefine()


VELBC = ([1, 2, 0, 0, 2, 1], [0.7, 0.0, 0.0, 0.0, 0.0, -0.3])


@pytest.mark.parametrize("case", CASES)
def test_inflow_outflow_velocity_bc_bit_exact_on_both_centrings(oracle, case):
    """somar_solver_set_vel_bc: BasicVelocityBCGhostClass's inflow (prescribed value) and outflow (order-0 extrapolation of
    the next face inside) sides on the MAC velocity and on the faces the cell-centred divergence averages"""
    from somar_amd import api as F
    so = oracle
    n, boxsz, variant, periodic, L = case
    dom, grids, dx, Jgup, Jinv = make_problem(so, n, boxsz, variant, periodic, L)
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv)
    try:
        kind, value = VELBC
        gpu.setVelBC(kind, value)
        vel = so.FluxData(grids, 1, 3)
        rng = np.random.default_rng(5)
        for i in range(len(grids)):
            for d in range(3):
                vel[i][d].a[...] = rng.uniform(0.5, 1.5, vel[i][d].a.shape)
        for p in range(gpu.num_local_patches):
            gi = gpu.patch_box(p)[2]
            for d in range(3):
                gpu.uploadVel(d, p, np.asfortranarray(vel[gi][d].a[..., 0]))
        so.set_normal_flux_bc(vel, grids, dom, kind, value)
        gpu.velWallBC()
        for p in range(gpu.num_local_patches):
            gi = gpu.patch_box(p)[2]
            for d in range(3):
                np.testing.assert_array_equal(gpu.downloadVel(d, p), vel[gi][d].a[..., 0])
        # cell-centred divergence with the same sides
        ghost = (1, 1, 1)
        cc = _cc_velocity(so, dom, grids, ghost)
        for p in range(gpu.num_local_patches):
            gpu.uploadCCVel(p, cc[gpu.patch_box(p)[2]].a, ghost)
        div = so.LevelData(grids, 1)
        so.level_divergence_cc(div, cc, Jinv, grids, dom, dx, wall=True, velbc=VELBC)
        gpu.divergenceCC(F.F_RHS, 1.0, True)
        for g, w in zip(download_valid(gpu, F.F_RHS, grids), valid_of(div)):
            np.testing.assert_array_equal(g, w)
        # back to solid walls
        gpu.setVelBC([0] * 6, [0.0] * 6)
        div = so.LevelData(grids, 1)
        so.level_divergence_cc(div, cc, Jinv, grids, dom, dx, wall=True)
        gpu.divergenceCC(F.F_RHS, 1.0, True)
        for g, w in zip(download_valid(gpu, F.F_RHS, grids), valid_of(div)):
            np.testing.assert_array_equal(g, w)
    finally:
        gpu.undefine()


def test_mult_and_div_by_j_on_the_resident_velocities(oracle):
    """LevelGeometry::multByJ / divByJ (geometry/LevelGeometryUtil.cpp:287-339, 372-420, 456-...): data *= J, data *= Jinv on
    the device, for the cell-centred velocity (ghost layer included) and the MAC velocity -- the same products, bit for bit."""
    from somar_amd import api as F
    from helpers import make_gpu_solver, make_problem, smooth_cc_velocity
    so = oracle
    dom, grids, dx, Jgup, Jinv = make_problem(so, (16, 16, 8), 8, "stretched", (False, True, False), (2.0, 1.0, 0.5))
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv)
    try:
        rng = np.random.default_rng(9)
        ghost = (1, 1, 1)
        vel = smooth_cc_velocity(so, dom, grids, ghost)
        Js = {}
        for p in range(gpu.num_local_patches):
            _, _, gi = gpu.patch_box(p)
            J = np.asfortranarray(rng.uniform(0.5, 2.0, vel[gi].a.shape[:3]))
            Js[gi] = (J, np.asfortranarray(1.0 / J), vel[gi].a * J[..., None])
            gpu.uploadCCVel(p, vel[gi].a, ghost)
            gpu.setCCJ(p, Js[gi][0], Js[gi][1], ghost)
        gpu.multByJ(1)
        for p in range(gpu.num_local_patches):
            _, _, gi = gpu.patch_box(p)
            buf = np.zeros(vel[gi].a.shape, order="F")
            gpu.downloadCCVel(p, buf, ghost)
            sl = grids[gi].slices(vel[gi].box.lo)
            np.testing.assert_array_equal(buf[sl], Js[gi][2][sl])
        gpu.divByJ(1)
        for p in range(gpu.num_local_patches):
            _, _, gi = gpu.patch_box(p)
            buf = np.zeros(vel[gi].a.shape, order="F")
            gpu.downloadCCVel(p, buf, ghost)
            sl = grids[gi].slices(vel[gi].box.lo)
            np.testing.assert_array_equal(buf[sl], (Js[gi][2] * Js[gi][1][..., None])[sl])
        # MAC velocity
        faces = {}
        for p in range(gpu.num_local_patches):
            _, _, gi = gpu.patch_box(p)
            for d in range(3):
                shp = tuple(n + (1 if a == d else 0) for a, n in enumerate(grids[gi].size()))
                u = np.asfortranarray(rng.uniform(-1, 1, shp))
                J = np.asfortranarray(rng.uniform(0.5, 2.0, shp))
                gpu.uploadVel(d, p, u)
                gpu.setFaceJ(d, p, J, np.asfortranarray(1.0 / J))
                faces[(p, d)] = (u, J)
        gpu.multByJ(0)
        for (p, d), (u, J) in faces.items():
            np.testing.assert_array_equal(gpu.downloadVel(d, p), u * J)
        gpu.divByJ(0)
        for (p, d), (u, J) in faces.items():
            np.testing.assert_array_equal(gpu.downloadVel(d, p), (u * J) * (1.0 / J))
    finally:
        gpu.undefine()
