"""Level projections on a refined level of a hierarchy (BaseProjector::levelProject with a coarser level) on the GPU vs
oracle/somar_amr.py::level_project: MAC (LevelMACProjector) and cell-centred (LevelCCProjector, whose divergence starts
with the velocity's quadratic coarse-fine interpolation).  The solve is the level solve with the coarse pressure as CF data;
histories to 1e-10, projected velocities to 1e-8 of their magnitude."""
import numpy as np
import pytest

from helpers import make_amr_levels, make_gpu_amr, max_rel_diff, upload

pytestmark = pytest.mark.gpu

LAYOUTS = [
    ((False, False, False), [(2, 2, 2)], [[((8, 8, 4), (23, 23, 11))]]),
    ((True, False, False), [(2, 2, 1)], [[((0, 8, 0), (15, 23, 7)), ((24, 8, 0), (31, 23, 7))]]),
    # a fine slab against the walls: wall faces on the fine level, one-sided CF stencils
    ((False, False, False), [(2, 2, 2)], [[((8, 0, 0), (23, 15, 15)), ((8, 16, 0), (23, 31, 15))]]),
]


@pytest.fixture(scope="module")
def am(oracle):
    from oracle import somar_amr
    return somar_amr


def _setup(so, am, layout):
    periodic, ratios, boxes = layout
    fb = [[so.Box(lo, hi) for lo, hi in lev] for lev in boxes]
    levels = make_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.5), periodic, ratios, fb)
    comp = am.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab())
    gpu = make_gpu_amr(levels, ratios)
    return levels, comp, gpu


def _smooth(so, L, ncomp, ghost, phase):
    n = L.domain.box.size()
    u = so.LevelData(L.grids, ncomp, ghost)
    for f in u.fabs:
        I, J, K = np.meshgrid(*[np.arange(f.box.lo[a], f.box.hi[a] + 1) for a in range(3)], indexing="ij")
        for d in range(ncomp):
            f.a[..., d] = (np.sin(2 * np.pi * (I + 0.5) / n[0] + 0.1 * d + phase) * np.cos(2 * np.pi * (J + 0.5) / n[1] + 0.3)
                           * np.cos(2 * np.pi * (K + 0.5) / n[2] + d)) + 0.25 * d
    return u


@pytest.mark.parametrize("layout", LAYOUTS)
def test_cc_level_projection_on_the_fine_level(oracle, am, layout):
    from somar_amd import api as F
    so = oracle
    levels, comp, gpu = _setup(so, am, layout)
    try:
        ghost = (1, 1, 1)
        vc = _smooth(so, levels[0], 3, ghost, 0.0)
        vf = _smooth(so, levels[1], 3, ghost, 0.0)
        phi = [so.random_field(L.grids, 5 + l, (1, 1, 1), L.domain.box) for l, L in enumerate(levels)]
        for l, v in enumerate(gpu.levels):
            upload(v, F.F_PHI, phi[l])
            src = vc if l == 0 else vf
            for p in range(v.num_local_patches):
                v.uploadCCVel(p, src[v.patch_box(p)[2]].a, ghost)
        dt = 0.5
        am.level_project(comp, 1, vf, phi, dt, "cc", vc)
        st = gpu.levelProjectAMR(1, 1, dt)
        assert st["iters"] == comp.iters and st["exitStatus"] == comp.exitStatus
        np.testing.assert_allclose(st["history"], comp.history, rtol=1e-10, atol=1e-10 * comp.history[0])
        v = gpu.levels[1]
        got, want = [], []
        for p in range(v.num_local_patches):
            gi = v.patch_box(p)[2]
            buf = np.zeros(vf[gi].a.shape, order="F")
            v.downloadCCVel(p, buf, ghost)
            g = levels[1].grids[gi]
            sl = g.slices(vf[gi].box.lo)
            got.append(buf[sl])
            want.append(vf[gi].a[sl])
        assert max_rel_diff(got, want) < 1e-8
    finally:
        gpu.undefine()


@pytest.mark.parametrize("layout", LAYOUTS[:2])
def test_mac_level_projection_on_the_fine_level(oracle, am, layout):
    from somar_amd import api as F
    so = oracle
    levels, comp, gpu = _setup(so, am, layout)
    try:
        L1 = levels[1]
        n = L1.domain.box.size()
        vel = so.FluxData(L1.grids, 1, 3)
        for i in range(len(L1.grids)):
            for d in range(3):
                fb = vel[i][d].box
                I, J, K = np.meshgrid(*[np.arange(fb.lo[a], fb.hi[a] + 1) for a in range(3)], indexing="ij")
                vel[i][d].a[..., 0] = (np.sin(2 * np.pi * np.mod(I, n[0]) / n[0] + 0.1 * d)
                                       * np.cos(2 * np.pi * np.mod(J, n[1]) / n[1]) * np.cos(2 * np.pi * np.mod(K, n[2]) / n[2] + d))
        phi = [so.random_field(L.grids, 5 + l, (1, 1, 1), L.domain.box) for l, L in enumerate(levels)]
        for l, v in enumerate(gpu.levels):
            upload(v, F.F_PHI, phi[l])
        v = gpu.levels[1]
        for p in range(v.num_local_patches):
            gi = v.patch_box(p)[2]
            for d in range(3):
                v.uploadVel(d, p, np.asfortranarray(vel[gi][d].a[..., 0]))
        dt = 0.5
        am.level_project(comp, 1, vel, phi, dt, "mac")
        st = gpu.levelProjectAMR(1, 0, dt)
        assert st["iters"] == comp.iters and st["exitStatus"] == comp.exitStatus
        np.testing.assert_allclose(st["history"], comp.history, rtol=1e-10, atol=1e-10 * comp.history[0])
        for d in range(3):
            got = [v.downloadVel(d, p) for p in range(v.num_local_patches)]
            want = [vel[v.patch_box(p)[2]][d].a[..., 0] for p in range(v.num_local_patches)]
            assert max_rel_diff(got, want) < 1e-8
    finally:
        gpu.undefine()


# ---- non-diagonal metric on a REFINED level: ExtrapolateCFEV after the pressure's CF interpolation, then singleBoxMacGrad's
#      full sequence (Gradient.cpp:104-114, 946-1101) next to coarse-fine faces ----
@pytest.mark.parametrize("centring", ["cc", "mac"])
def test_level_projection_on_a_refined_level_with_a_non_diagonal_metric(oracle, am, centring):
    from somar_amd import api as F
    from helpers import make_full_amr_levels
    so = oracle
    fb = [[so.Box((8, 8, 4), (23, 23, 11))]]
    levels = make_full_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.5), (False, False, False), [(2, 2, 2)], fb, cbox=8, ndim=3)
    comp = am.AMRComposite(levels, [(2, 2, 2)], so.BCHolder(), so.BiCGStab(), isDiagonal=False)
    gpu = make_gpu_amr(levels, [(2, 2, 2)], ndim=3, full=True)
    try:
        ghost = (1, 1, 1)
        phi = [so.random_field(L.grids, 5 + l, (1, 1, 1), L.domain.box) for l, L in enumerate(levels)]
        for l, v in enumerate(gpu.levels):
            upload(v, F.F_PHI, phi[l])
        v = gpu.levels[1]
        dt = 0.5
        if centring == "cc":
            vc = _smooth(so, levels[0], 3, ghost, 0.0)
            vf = _smooth(so, levels[1], 3, ghost, 0.0)
            for l, w in enumerate(gpu.levels):
                src = vc if l == 0 else vf
                for p in range(w.num_local_patches):
                    w.uploadCCVel(p, src[w.patch_box(p)[2]].a, ghost)
            am.level_project(comp, 1, vf, phi, dt, "cc", vc)
            st = gpu.levelProjectAMR(1, 1, dt)
        else:
            L1 = levels[1]
            n = L1.domain.box.size()
            vel = so.FluxData(L1.grids, 1, 3)
            for i in range(len(L1.grids)):
                for d in range(3):
                    fbx = vel[i][d].box
                    I, J, K = np.meshgrid(*[np.arange(fbx.lo[a], fbx.hi[a] + 1) for a in range(3)], indexing="ij")
                    vel[i][d].a[..., 0] = (np.sin(2 * np.pi * I / n[0] + 0.1 * d) * np.cos(2 * np.pi * J / n[1])
                                           * np.cos(2 * np.pi * K / n[2] + d))
            for p in range(v.num_local_patches):
                gi = v.patch_box(p)[2]
                for d in range(3):
                    v.uploadVel(d, p, np.asfortranarray(vel[gi][d].a[..., 0]))
            am.level_project(comp, 1, vel, phi, dt, "mac")
            st = gpu.levelProjectAMR(1, 0, dt)
        assert st["iters"] == comp.iters and st["exitStatus"] == comp.exitStatus
        np.testing.assert_allclose(st["history"], comp.history, rtol=1e-10, atol=1e-10 * comp.history[0])
        if centring == "cc":
            got, want = [], []
            for p in range(v.num_local_patches):
                gi = v.patch_box(p)[2]
                buf = np.zeros(vf[gi].a.shape, order="F")
                v.downloadCCVel(p, buf, ghost)
                sl = levels[1].grids[gi].slices(vf[gi].box.lo)
                got.append(buf[sl])
                want.append(vf[gi].a[sl])
            assert max_rel_diff(got, want) < 1e-8
        else:
            for d in range(3):
                got = [v.downloadVel(d, p) for p in range(v.num_local_patches)]
                want = [vel[v.patch_box(p)[2]][d].a[..., 0] for p in range(v.num_local_patches)]
                assert max_rel_diff(got, want) < 1e-8
    finally:
        gpu.undefine()
