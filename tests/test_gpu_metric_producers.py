"""Metric producers on the device: GeoSourceInterface::fill_Jgup's generic algebra and the Cartesian
map's constants, against the oracle's restatement (oracle/somar_oracle.py::geo_fill_jgup) and against host-side uploads."""
import numpy as np
import pytest

from oracle import somar_oracle as so
from tests.helpers import download_valid, make_gpu_solver, upload

pytestmark = pytest.mark.gpu


def _jacobians(n, seed):
    rng = np.random.default_rng(seed)
    dx = np.eye(3)[None, :, :] + 0.3 * rng.uniform(-1, 1, (n, 3, 3))     # well conditioned, every entry non-zero
    J = np.linalg.det(dx)
    return dx, J


@pytest.mark.parametrize("scale", [1.0, 0.37])
def test_generic_jgup_algebra_bit_exact(scale):
    from somar_amd import api as F
    dx, J = _jacobians(5000, 3)
    for mu in range(3):
        got = F.jgup_from_dxdxi(dx, J, mu, scale)
        want = so.geo_fill_jgup(dx, J, mu, scale)
        np.testing.assert_array_equal(got, want)
    # symmetry of the tensor it produces: J g^{01} from face direction 0 == J g^{10} from direction 1 (same centring here)
    g0, g1 = F.jgup_from_dxdxi(dx, J, 0), F.jgup_from_dxdxi(dx, J, 1)
    np.testing.assert_allclose(g0[:, 1], g1[:, 0], rtol=1e-13)


def test_generic_algebra_reproduces_the_terrain_following_metric():
    """z = d(xi, eta) + (1 - d/H) zeta (BathymetricBaseMap's form): J = z_zeta, J g^{xi xi} = z_zeta, J g^{xi zeta} = -z_xi,
    J g^{zeta zeta} = (1 + z_xi^2 + z_eta^2) / z_zeta -- SURVEY.md 8d's C5 construction"""
    from somar_amd import api as F
    rng = np.random.default_rng(7)
    n = 2000
    zx, zy, zz = rng.uniform(-0.3, 0.3, n), rng.uniform(-0.3, 0.3, n), rng.uniform(0.5, 0.8, n)
    dx = np.zeros((n, 3, 3))
    dx[:, 0, 0] = dx[:, 1, 1] = 1.0
    dx[:, 2, 0], dx[:, 2, 1], dx[:, 2, 2] = zx, zy, zz
    g = [F.jgup_from_dxdxi(dx, zz, mu) for mu in range(3)]
    np.testing.assert_allclose(g[0], np.stack([zz, 0 * zz, -zx], 1), atol=1e-14)
    np.testing.assert_allclose(g[1], np.stack([0 * zz, zz, -zy], 1), atol=1e-14)
    np.testing.assert_allclose(g[2], np.stack([-zx, -zy, (1 + zx ** 2 + zy ** 2) / zz], 1), rtol=1e-13)


def test_uniform_producer_equals_uploaded_constant_arrays():
    """setMetricUniform writes what setMetricOrtho uploads for a Cartesian map: same hierarchy, same detection, same solve"""
    from somar_amd import AMRPressureSolver
    from somar_amd.api import F_PHI, F_RHS
    n, box = (32, 32, 16), (16, 32, 16)
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (False, True, False))
    grids = so.split_domain(dom.box, box)
    dx = (1.0 / 32, 1.0 / 32, 0.5 / 16)
    c = (1.0, 1.0, 1.0, 1.0)
    Jgup, Jinv = so.make_diagonal_metric(grids, dx, (1.0, 1.0, 0.5), 3, "cartesian", domain=dom)
    rhs = so.random_field(grids, 5, domainBox=dom.box)
    so.remove_weighted_mean(rhs, Jinv)
    up = make_gpu_solver(dom, grids, dx, Jgup, Jinv)
    s = AMRPressureSolver()
    p = s._p
    s.setAMRMGParameters(p.imin, p.imax, p.eps, -1, p.num_smooth_precond, 2, 2, 2, p.precond_mode, 1, p.num_mg, p.hang,
                         p.norm_thresh, 0)
    s.define(dom.box.lo, dom.box.hi, dom.periodic, dx, [(g.lo, g.hi) for g in grids])
    s.setMetricUniform(*c)
    s.finalize()
    try:
        assert s.metricUniform(0) == c and up.metricUniform(0) == c
        assert s.depth() == up.depth()
        hist = []
        for g in (s, up):
            upload(g, F_RHS, rhs)
            hist.append(g.solveResident(True, False)["history"])
        assert hist[0] == hist[1]
        for a, b in zip(download_valid(s, F_PHI, grids), download_valid(up, F_PHI, grids)):
            np.testing.assert_array_equal(a, b)
    finally:
        s.undefine()
        up.undefine()


# ---- coordinate maps evaluated on the device (somar_solver_set_metric_map) ------------------------------------------------
def _twin(dom, grids, dx, install):
    from somar_amd import AMRPressureSolver
    s = AMRPressureSolver()
    p = s._p
    s.setAMRMGParameters(p.imin, p.imax, p.eps, -1, p.num_smooth_precond, 2, 2, 2, p.precond_mode, 1, p.num_mg, p.hang,
                         p.norm_thresh, 0)
    s.define(dom.box.lo, dom.box.hi, dom.periodic, dx, [(g.lo, g.hi) for g in grids])
    install(s)
    s.finalize()
    return s


def _compare_operators(a, b, grids, dom, exact):
    from somar_amd import api as F
    phi = so.random_field(grids, 7, (1, 1, 1), dom.box)
    out = []
    for s in (a, b):
        upload(s, F.F_PHI, phi)
        s.applyOp(0, F.F_RES, F.F_PHI)
        out.append(download_valid(s, F.F_RES, grids))
    for x, y in zip(*out):
        if exact:
            np.testing.assert_array_equal(x, y)
        else:
            np.testing.assert_allclose(x, y, rtol=0, atol=1e-12 * float(np.max(np.abs(y))))
    assert a.depth() == b.depth()


def test_bathymetric_map_on_the_device_equals_the_uploaded_oracle_metric():
    """BathymetricBaseMap's dx/dXi from a NODAL depth (a bump plus a slope, every derivative non-zero), CONVERTFAB to the
    face / cell centrings (AVG3IX's misprint included) and GeoSourceInterface's algebra: the device fills what the oracle's
    restatement fills, bit for bit -- seen through the 19-point operator of two solvers, one fed each way -- and on every
    multigrid depth (same coarsened metric)."""
    from oracle import somar_maps as sm
    from somar_amd import api as F
    n, L, bs = (32, 16, 8), (4.0, 2.0, 1.0), (16, 8, 8)
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (False, False, False))
    grids = so.split_domain(dom.box, bs)
    dx = tuple(L[d] / n[d] for d in range(3))
    dlo, dn = (-1, -1), (n[0] + 4, n[1] + 4)
    x = (np.arange(dlo[0], dlo[0] + dn[0]) * dx[0])[:, None]
    y = (np.arange(dlo[1], dlo[1] + dn[1]) * dx[1])[None, :]
    depth = 0.15 + 0.02 * x - 0.03 * y + 0.25 * np.exp(-((x - 1.7) ** 2 + (y - 0.9) ** 2) / 0.5)
    m = sm.BathymetricMap(dx, L, depth, dlo)

    def upload_metric(s):
        for q in range(s.num_local_patches):
            _, _, gi = s.patch_box(q)
            g = grids[gi]
            jg = [np.asfortranarray(sm.fill_jgup(m, g, mu)) for mu in range(3)]
            s.setMetricFull(q, jg[0], jg[1], jg[2], np.asfortranarray(sm.fill_jinv(m, g)))

    a = _twin(dom, grids, dx, upload_metric)
    b = _twin(dom, grids, dx, lambda s: s.setMetricMap(F.MAP_BATHYMETRIC, L, depth, dlo))
    try:
        _compare_operators(a, b, grids, dom, exact=True)
        rhs = so.random_field(grids, 5, domainBox=dom.box)
        hist = []
        for s in (a, b):
            upload(s, F.F_RHS, rhs)
            hist.append(s.solveResident(True, False)["history"])
        assert hist[0] == hist[1]
    finally:
        a.undefine()
        b.undefine()


def test_cylindrical_map_on_the_device_matches_the_oracle_metric():
    """CylindricalMap (diagonal): J g^{rr} = r, J g^{theta theta} = 1/r, J g^{zz} = r, 1/J = 1/r through cos / sin of the
    device's libm (a few ulp from numpy's), periodic in theta"""
    from oracle import somar_maps as sm
    from somar_amd import api as F
    n, bs = (16, 32, 8), (8, 16, 8)
    dom = so.Domain(so.Box((8, 0, 0), (8 + n[0] - 1, n[1] - 1, n[2] - 1)), (False, True, False))
    grids = so.split_domain(dom.box, bs)
    dx = (0.05, 2 * np.pi / n[1], 0.1)
    m = sm.CylindricalMap(dx)

    def upload_metric(s):
        for q in range(s.num_local_patches):
            _, _, gi = s.patch_box(q)
            g = grids[gi]
            jg = [np.asfortranarray(sm.fill_jgup(m, g, mu)[..., mu]) for mu in range(3)]
            s.setMetricOrtho(q, jg[0], jg[1], jg[2], np.asfortranarray(sm.fill_jinv(m, g)))

    a = _twin(dom, grids, dx, upload_metric)
    b = _twin(dom, grids, dx, lambda s: s.setMetricMap(F.MAP_CYLINDRICAL))
    try:
        _compare_operators(a, b, grids, dom, exact=False)
    finally:
        a.undefine()
        b.undefine()


def test_twisted_map_on_the_device_matches_the_oracle_metric():
    """TwistedMap (m_twistType 0, TWISTED0_FILL_DXDXI / TWISTED0_FILL_J): every off-diagonal dx/dXi non-zero, the analytic
    Jacobian; cos / sin of the device's libm against numpy's: 1e-12 of the operator, fully periodic as the map is"""
    from oracle import somar_maps as sm
    from somar_amd import api as F
    n, bs = (16, 16, 16), 8
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (True, True, True))
    grids = so.split_domain(dom.box, bs)
    dx = tuple(1.0 / a for a in n)
    pert = (0.05, 0.04, 0.03)
    m = sm.TwistedMap(dx, pert)

    def upload_metric(s):
        for q in range(s.num_local_patches):
            _, _, gi = s.patch_box(q)
            g = grids[gi]
            jg = [np.asfortranarray(sm.fill_jgup(m, g, mu)) for mu in range(3)]
            s.setMetricFull(q, jg[0], jg[1], jg[2], np.asfortranarray(sm.fill_jinv(m, g)))

    a = _twin(dom, grids, dx, upload_metric)
    b = _twin(dom, grids, dx, lambda s: s.setMetricMap(F.MAP_TWISTED, pert))
    try:
        _compare_operators(a, b, grids, dom, exact=False)
    finally:
        a.undefine()
        b.undefine()


def test_twisted_map_type_1_on_the_device_matches_the_oracle_metric():
    """TwistedMap with m_twistType 1: analytic coordinates, dx/dXi and J from GeoSourceInterface's finite-difference defaults
    (staggered coordinate differences, DEFAULT_FILL_J_3D + CellToEdge).  Device libm against numpy's: 1e-12 of the operator."""
    from oracle import somar_maps as sm
    from somar_amd import api as F
    n, bs = (16, 16, 8), 8
    L = (1.0, 2.0, 0.5)
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (False, False, False))
    grids = so.split_domain(dom.box, bs)
    dx = tuple(L[d] / n[d] for d in range(3))
    pert = (0.05, 0.08, 0.02)
    m = sm.TwistedMap1(dx, pert, L)

    def upload_metric(s):
        for q in range(s.num_local_patches):
            _, _, gi = s.patch_box(q)
            g = grids[gi]
            jg = [np.asfortranarray(sm.fill_jgup(m, g, mu)) for mu in range(3)]
            s.setMetricFull(q, jg[0], jg[1], jg[2], np.asfortranarray(sm.fill_jinv(m, g)))

    a = _twin(dom, grids, dx, upload_metric)
    b = _twin(dom, grids, dx, lambda s: s.setMetricMap(F.MAP_TWISTED1, pert))
    try:
        _compare_operators(a, b, grids, dom, exact=False)
    finally:
        a.undefine()
        b.undefine()


def test_map_producer_argument_checks():
    from somar_amd import SomarError
    from somar_amd import api as F
    n = (16, 16, 8)
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (False, False, False))
    grids = so.split_domain(dom.box, 8)
    dx = (0.1, 0.1, 0.1)
    with pytest.raises(SomarError, match="nodal depth must cover"):
        _twin(dom, grids, dx, lambda s: s.setMetricMap(F.MAP_BATHYMETRIC, (1.6, 1.6, 0.8), np.zeros((17, 17)), (0, 0)))
    with pytest.raises(SomarError, match="map kind"):
        _twin(dom, grids, dx, lambda s: s.setMetricMap(7))
