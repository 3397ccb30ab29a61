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
