"""The COMPOSITE cell-centred projector (AMRCCProjector: sync / init / regrid projection) on the GPU vs
oracle/somar_ccproj.py: compDivergenceCC with refluxing and compGradientCC with one-sided coarse-fine faces bit for bit
(given the same pressure), the averaging down bit for bit, and the whole projection (composite solve in between):
same iterations / exit status, residual history to 1e-10, projected velocities to 1e-8 of their magnitude."""
import numpy as np
import pytest

from helpers import make_amr_levels, make_full_amr_levels, make_gpu_amr, max_rel_diff, smooth_cc_velocity, upload

pytestmark = pytest.mark.gpu

LAYOUTS = [
    ((False, False, False), [(2, 2, 2)], [[((8, 8, 4), (23, 23, 11))]]),
    ((True, False, False), [(2, 2, 1)], [[((0, 8, 0), (15, 23, 7)), ((24, 8, 0), (31, 23, 7))]]),
    ((False, True, False), [(2, 2, 1), (2, 2, 1)], [[((8, 0, 0), (23, 31, 7))], [((24, 0, 0), (39, 63, 7))]]),
    # a fine slab against the walls, two fine boxes side by side
    ((False, False, False), [(2, 2, 2)], [[((8, 0, 0), (23, 15, 15)), ((8, 16, 0), (23, 31, 15))]]),
]
GHOST = (1, 1, 1)


@pytest.fixture(scope="module")
def am(oracle):
    from oracle import somar_amr
    return somar_amr


def _setup(so, am, layout, full=False):
    periodic, ratios, boxes = layout
    fb = [[so.Box(lo, hi) for lo, hi in lev] for lev in boxes]
    if full:
        levels = make_full_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.5), periodic, ratios, fb, cbox=8, ndim=3)
    else:
        levels = make_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.5), periodic, ratios, fb)
    comp = am.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab(), isDiagonal=not full)
    gpu = make_gpu_amr(levels, ratios, full=full)
    return levels, comp, gpu


def _upload_vel(gpu, vel):
    for l, v in enumerate(gpu.levels):
        for p in range(v.num_local_patches):
            v.uploadCCVel(p, vel[l][v.patch_box(p)[2]].a, GHOST)


def _download_vel(gpu, levels, vel):
    """-> per level lists (got, want) of the valid-region velocity arrays"""
    out = []
    for l, v in enumerate(gpu.levels):
        got, want = [], []
        for p in range(v.num_local_patches):
            gi = v.patch_box(p)[2]
            buf = np.zeros(vel[l][gi].a.shape, order="F")
            v.downloadCCVel(p, buf, GHOST)
            sl = levels[l].grids[gi].slices(vel[l][gi].box.lo)
            got.append(buf[sl])
            want.append(vel[l][gi].a[sl])
        out.append((got, want))
    return out


@pytest.mark.parametrize("layout", LAYOUTS)
def test_composite_divergence_bit_exact(oracle, am, layout):
    from somar_amd import api as F
    from oracle import somar_ccproj as cp
    so = oracle
    levels, comp, gpu = _setup(so, am, layout)
    try:
        lmax = len(levels) - 1
        vel = [smooth_cc_velocity(so, L.domain, L.grids, GHOST) for L in levels]
        _upload_vel(gpu, vel)
        for l in range(lmax + 1):
            div = so.LevelData(levels[l].grids, 1)
            so.exchange(vel[l], levels[l].domain, vel[l].ghost)
            cp.comp_divergence_cc(comp, l, div, vel[l], vel[l - 1] if l > 0 else None, vel[l + 1] if l < lmax else None)
            gpu.compDivergenceCC(l, lmax, F.F_RHS)
            if l < lmax:
                # coarse cells under the finer level: the reference leaves flux-register debris there (its reverse copier
                # also serves the register cells that lie inside a neighbouring fine box); the solve never reads them
                comp.zero_covered(l, div)
                gpu.zeroCovered(l, F.F_RHS)
            v = gpu.levels[l]
            for p in range(v.num_local_patches):
                gi = v.patch_box(p)[2]
                np.testing.assert_array_equal(v.download(F.F_RHS, p, (0, 0, 0)), div[gi].a[..., 0], err_msg="level %d" % l)
    finally:
        gpu.undefine()


@pytest.mark.parametrize("layout", LAYOUTS)
def test_composite_gradient_correction_and_averaging_bit_exact(oracle, am, layout):
    from somar_amd import api as F
    from oracle import somar_ccproj as cp
    so = oracle
    levels, comp, gpu = _setup(so, am, layout)
    try:
        lmax = len(levels) - 1
        vel = [smooth_cc_velocity(so, L.domain, L.grids, GHOST) for L in levels]
        phi = [so.random_field(L.grids, 5 + l, (1, 1, 1), L.domain.box) for l, L in enumerate(levels)]
        _upload_vel(gpu, vel)
        for l, v in enumerate(gpu.levels):
            upload(v, F.F_PHI, phi[l])
        dt = 0.5
        for l in range(lmax, -1, -1):
            corr = so.LevelData(levels[l].grids, 3)
            so.exchange(phi[l], levels[l].domain, phi[l].ghost)
            cp.comp_gradient_cc(comp, l, corr, phi[l], phi[l - 1] if l > 0 else None, phi[l + 1] if l < lmax else None)
            for i, g in enumerate(levels[l].grids):
                vel[l][i].view(g)[...] += -dt * corr[i].a
            gpu.compGradCorrectCC(l, lmax, F.F_PHI, dt)
            if l < lmax:
                cp.average_to_coarse(comp, l, vel[l], cp._valid_only(vel[l + 1]))
                gpu.averageDownCCVel(l)
        for l, (got, want) in enumerate(_download_vel(gpu, levels, vel)):
            for a, b in zip(got, want):
                np.testing.assert_array_equal(a, b, err_msg="level %d" % l)
    finally:
        gpu.undefine()


@pytest.mark.parametrize("layout", LAYOUTS[:3])
def test_composite_projection_matches(oracle, am, layout):
    from oracle import somar_ccproj as cp
    so = oracle
    levels, comp, gpu = _setup(so, am, layout)
    try:
        lmax = len(levels) - 1
        vel = [smooth_cc_velocity(so, L.domain, L.grids, GHOST) for L in levels]
        phi = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
        _upload_vel(gpu, vel)
        cp.amr_cc_project(comp, vel, phi, 0, lmax, 0.5, zeroPhi=True)
        st = gpu.projectAMRCC(0, lmax, 0.5, zeroPressure=True)
        assert st["iters"] == comp.iters and st["exitStatus"] == comp.exitStatus
        np.testing.assert_allclose(st["history"], comp.history, rtol=1e-10, atol=1e-10 * comp.history[0])
        for l, (got, want) in enumerate(_download_vel(gpu, levels, vel)):
            assert max_rel_diff(got, want) < 1e-8, l
    finally:
        gpu.undefine()


def test_composite_projection_with_a_non_diagonal_metric(oracle, am):
    from oracle import somar_ccproj as cp
    so = oracle
    levels, comp, gpu = _setup(so, am, LAYOUTS[0], full=True)
    try:
        vel = [smooth_cc_velocity(so, L.domain, L.grids, GHOST) for L in levels]
        phi = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
        _upload_vel(gpu, vel)
        cp.amr_cc_project(comp, vel, phi, 0, 1, 0.5, zeroPhi=True)
        st = gpu.projectAMRCC(0, 1, 0.5, zeroPressure=True)
        assert st["iters"] == comp.iters and st["exitStatus"] == comp.exitStatus
        np.testing.assert_allclose(st["history"], comp.history, rtol=1e-9, atol=1e-9 * comp.history[0])
        for l, (got, want) in enumerate(_download_vel(gpu, levels, vel)):
            assert max_rel_diff(got, want) < 1e-7, l
    finally:
        gpu.undefine()
