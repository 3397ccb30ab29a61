"""Known answers for the coordinate maps' metric producers (oracle/somar_maps.py: CylindricalMap, BathymetricBaseMap,
CONVERTFAB).  The reference ships no fixtures: parity unpinned w.r.t. reference tests; pinned here by the analytic metrics
of the two maps and by the averaging identities CONVERTFAB has to satisfy -- including the one it does NOT satisfy because
of AVG3IX's misprinted eighth term (utils/AddlFortranMacros.H:88), which the restatement reproduces."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def sm(oracle):
    from oracle import somar_maps
    return somar_maps


def test_convert_fab_is_exact_on_linear_fields_in_one_and_two_directions(oracle, sm):
    def f(i, j, k):
        return 1.0 + 0.5 * i - 0.25 * j + 0.125 * k
    I, J, K = np.meshgrid(np.arange(3, 7), np.arange(-2, 2), np.arange(0, 3), indexing="ij")
    # node -> x-face (average over j, k forward), node -> xy-edge... every 1- and 2-direction combination
    for S, T in [((1, 1, 1), (1, 0, 0)), ((1, 1, 1), (0, 1, 0)), ((1, 1, 1), (0, 0, 1)), ((0, 1, 1), (0, 0, 1)),
                 ((0, 1, 1), (0, 1, 0)), ((1, 0, 1), (1, 0, 0)), ((0, 1, 1), (1, 1, 1)), ((0, 0, 0), (1, 0, 0))]:
        got = sm.convert_fab(f, I, J, K, S, T)
        off = [0.5 * (S[d] - T[d]) for d in range(3)]
        want = f(I + off[0], J + off[1], K + off[2])
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-14)


def test_convert_fab_in_three_directions_reproduces_the_avg3ix_misprint(oracle, sm):
    I, J, K = np.meshgrid(np.arange(0, 2), np.arange(0, 2), np.arange(0, 2), indexing="ij")

    def fx(i, j, k):
        return 1.0 * i + 0.0 * j

    def fy(i, j, k):
        return 1.0 * j + 0.0 * i

    def fz(i, j, k):
        return 1.0 * k + 0.0 * i
    S, T = (1, 1, 1), (0, 0, 0)
    # linear in x or z: the doubled corner (ii + kk) and the missing corner (ii + jj + kk) weigh the same -> still exact
    np.testing.assert_allclose(sm.convert_fab(fx, I, J, K, S, T), I + 0.5, atol=1e-15)
    np.testing.assert_allclose(sm.convert_fab(fz, I, J, K, S, T), K + 0.5, atol=1e-15)
    # linear in y: 3 of the 8 terms sit at j + 1 instead of 4 -> j + 3/8, not j + 1/2
    np.testing.assert_allclose(sm.convert_fab(fy, I, J, K, S, T), J + 0.375, atol=1e-15)


def test_cylindrical_metric_is_r_one_over_r_r(oracle, sm):
    so = oracle
    dXi = (0.05, 2 * np.pi / 32, 0.1)
    m = sm.CylindricalMap(dXi)
    valid = so.Box((4, 0, 0), (11, 7, 3))
    for mu in range(3):
        g = sm.fill_jgup(m, valid, mu)
        fb = valid.faces(mu)
        i = np.arange(fb.lo[0], fb.hi[0] + 1)
        r = dXi[0] * (i + (0.0 if mu == 0 else 0.5))
        want = {0: r, 1: 1.0 / r, 2: r}[mu]
        np.testing.assert_allclose(g[..., mu], np.broadcast_to(want[:, None, None], g.shape[:3]), rtol=1e-13)
        for nu in range(3):
            if nu != mu:
                assert np.all(g[..., nu] == 0.0)
    i = np.arange(valid.lo[0], valid.hi[0] + 1)
    r = dXi[0] * (i + 0.5)
    np.testing.assert_allclose(sm.fill_jinv(m, valid), np.broadcast_to((1.0 / r)[:, None, None], valid.size()), rtol=1e-15)


def _plane_depth(dXi, lo, n, a, bx, by):
    i = np.arange(lo[0], lo[0] + n[0])[:, None] * dXi[0]
    j = np.arange(lo[1], lo[1] + n[1])[None, :] * dXi[1]
    return a + bx * i + by * j


def test_bathymetric_metric_of_a_plane_bottom_is_the_terrain_following_metric(oracle, sm):
    """z = d + (1 - d/H) zeta with d = a + bx x + by y: z_zeta = 1 - d/H, z_xi = (1 - zeta/H) bx, z_eta = (1 - zeta/H) by;
    J g^{xi xi} = J g^{eta eta} = z_zeta, J g^{xi zeta} = -z_xi, J g^{eta zeta} = -z_eta, J g^{xi eta} = 0,
    J g^{zeta zeta} = (1 + z_xi^2 + z_eta^2) / z_zeta.  Face centrings average a plane exactly."""
    so = oracle
    L, n = (4.0, 2.0, 1.0), (16, 8, 8)
    dXi = tuple(L[d] / n[d] for d in range(3))
    a, bx, by = 0.2, 0.05, -0.08
    dlo, dn = (-1, -1), (n[0] + 4, n[1] + 4)
    depth = _plane_depth(dXi, dlo, dn, a, bx, by)
    m = sm.BathymetricMap(dXi, L, depth, dlo)
    valid = so.Box((0, 0, 0), tuple(x - 1 for x in n))
    H = L[2]
    for mu in range(3):
        fb = valid.faces(mu)
        I, J, K = np.meshgrid(*[np.arange(fb.lo[d], fb.hi[d] + 1) for d in range(3)], indexing="ij")
        off = [0.5] * 3
        off[mu] = 0.0
        x, y, zeta = (I + off[0]) * dXi[0], (J + off[1]) * dXi[1], (K + off[2]) * dXi[2]
        d = a + bx * x + by * y
        zz, zx, zy = 1.0 - d / H, (1.0 - zeta / H) * bx, (1.0 - zeta / H) * by
        want = {0: [zz, 0 * zz, -zx], 1: [0 * zz, zz, -zy], 2: [-zx, -zy, (1.0 + zx * zx + zy * zy) / zz]}[mu]
        g = sm.fill_jgup(m, valid, mu)
        for nu in range(3):
            np.testing.assert_allclose(g[..., nu], want[nu], rtol=1e-12, atol=1e-14)


def test_bathymetric_cell_centred_j_carries_the_misprint(oracle, sm):
    so = oracle
    L, n = (4.0, 2.0, 1.0), (8, 8, 4)
    dXi = tuple(L[d] / n[d] for d in range(3))
    valid = so.Box((0, 0, 0), tuple(x - 1 for x in n))
    dlo, dn = (-1, -1), (n[0] + 4, n[1] + 4)
    I, J, K = np.meshgrid(*[np.arange(valid.lo[d], valid.hi[d] + 1) for d in range(3)], indexing="ij")
    # depth varying in x only: exact cell-centre value
    mx = sm.BathymetricMap(dXi, L, _plane_depth(dXi, dlo, dn, 0.2, 0.05, 0.0), dlo)
    d = 0.2 + 0.05 * (I + 0.5) * dXi[0]
    np.testing.assert_allclose(sm.fill_jinv(mx, valid), 1.0 / (1.0 - d / L[2]), rtol=1e-13)
    # depth varying in y: the average sits at j + 3/8
    my = sm.BathymetricMap(dXi, L, _plane_depth(dXi, dlo, dn, 0.2, 0.0, -0.08), dlo)
    d = 0.2 - 0.08 * (J + 0.375) * dXi[1]
    np.testing.assert_allclose(sm.fill_jinv(my, valid), 1.0 / (1.0 - d / L[2]), rtol=1e-13)


def test_twisted_map_jacobian_is_the_determinant_of_its_dxdxi(oracle, sm):
    """TWISTED0_FILL_J is the closed form of det(dx/dXi) of TWISTED0_FILL_DXDXI's entries (TwistedMap.cpp:204-207: "without the
    analytic version machine-epsilon sized noise is introduced"); and dx/dXi are the derivatives of the map's coordinates
    (TWISTED0_FILL_PHYSCOOR) to second order in the spacing"""
    so = oracle
    n = 24
    dXi = (1.0 / n,) * 3
    pert = (0.05, -0.04, 0.03)
    m = sm.TwistedMap(dXi, pert)
    valid = so.Box((0, 0, 0), (n - 1, n - 1, n - 1))
    I, J, K = np.meshgrid(*[np.arange(n)] * 3, indexing="ij")
    for T in ((0, 0, 0), (1, 0, 0), (0, 0, 1)):
        D = np.empty(I.shape + (3, 3))
        for mu in range(3):
            for nu in range(3):
                D[..., mu, nu] = m.dxdXi(mu, nu, T, I, J, K)
        np.testing.assert_allclose(m.J(T, I, J, K), np.linalg.det(D), rtol=1e-13)
    assert m.J((0, 0, 0), I, J, K).min() > 0.5

    def x(mu, xi):   # TWISTED0_FILL_PHYSCOOR
        nu, o = (mu + 1) % 3, (mu + 2) % 3
        return xi[mu] + pert[mu] * np.sin(2 * np.pi * xi[nu]) * np.sin(2 * np.pi * xi[o])
    xi = [(I + 0.5) * dXi[0], (J + 0.5) * dXi[1], (K + 0.5) * dXi[2]]
    h = 1e-5
    for mu in range(3):
        for nu in range(3):
            xp = [a + (h if d == nu else 0.0) for d, a in enumerate(xi)]
            xm = [a - (h if d == nu else 0.0) for d, a in enumerate(xi)]
            fd = (x(mu, xp) - x(mu, xm)) / (2 * h)
            np.testing.assert_allclose(m.dxdXi(mu, nu, (0, 0, 0), I, J, K), fd, atol=1e-8)
    # the metric it feeds: symmetric positive J g^{ab} at a common point would need a common centring; on each face family
    # the diagonal component is positive
    for mu in range(3):
        assert sm.fill_jgup(m, valid, mu)[..., mu].min() > 0.0


def test_twisted_map_type_1_defaults_converge_to_the_analytic_jacobian(oracle, sm):
    """m_twistType 1 goes through GeoSourceInterface's finite-difference defaults: the differenced dx/dXi must approach the analytic
    derivative of TWISTED1_FILL_PHYSCOOR's coordinates at second order (16^3 -> 32^3: about four times smaller), and the default J
    (face-coordinate differences) is the determinant of those differenced derivatives."""
    L, pert = (1.0, 2.0, 0.5), (0.05, 0.08, 0.02)
    errs = []
    for n in (16, 32):
        dXi = tuple(L[d] / n for d in range(3))
        m = sm.TwistedMap1(dXi, pert, L)
        I, J, K = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
        T = (0, 0, 0)
        Xi = [dXi[0] * (I + 0.5), dXi[1] * (J + 0.5), dXi[2] * (K + 0.5)]
        k = [np.pi / L[d] for d in range(3)]
        ph = 0.25 * np.pi
        # analytic d x^0 / d Xi^0 and d x^0 / d Xi^1 of x^0 = Xi0 + p0 sin(k0 Xi0) cos(2 k1 Xi1 + ph) cos(2 k2 Xi2 + ph)
        d00 = 1.0 + pert[0] * k[0] * np.cos(k[0] * Xi[0]) * np.cos(2 * k[1] * Xi[1] + ph) * np.cos(2 * k[2] * Xi[2] + ph)
        d01 = -pert[0] * 2 * k[1] * np.sin(k[0] * Xi[0]) * np.sin(2 * k[1] * Xi[1] + ph) * np.cos(2 * k[2] * Xi[2] + ph)
        e = max(np.abs(m.dxdXi(0, 0, T, I, J, K) - d00).max(), np.abs(m.dxdXi(0, 1, T, I, J, K) - d01).max())
        A = np.empty(I.shape + (3, 3))
        for r in range(3):
            for s_ in range(3):
                A[..., r, s_] = m.dxdXi(r, s_, T, I, J, K)
        eJ = np.abs(m.J(T, I, J, K) - np.linalg.det(A)).max()
        errs.append((e, eJ))
    assert errs[1][0] < 0.3 * errs[0][0] and errs[1][0] < 1e-3
    # cell-centred, the default J IS the determinant of the differenced dx/dXi (the same face-coordinate differences)
    assert errs[0][1] < 1e-13 and errs[1][1] < 1e-13


def test_analytic_bathymetries_of_ledge_and_beam_generator_maps(sm):
    """The host helpers behind SOMAR_MAP_BATHYMETRIC's nodal depth (somar_bathymetry_ledge / _beam_generator, no GPU involved)
    against the numpy restatement, plus what the shapes must satisfy: the ledge's cubic meets both plateaus with zero slope, the
    ridge is even in x, continuous at its six break points, zero outside and peaks at l* sin(angle) - tan(angle) P / 2."""
    from somar_amd import api as F
    x = np.linspace(-0.3, 1.4, 3001)
    for order in (1, 3):
        got = F.ledge_bathymetry(x, None, order, 1.0, 0.4, 0.2, 0.9)
        np.testing.assert_allclose(got, sm.ledge_bathymetry(x, order, 1.0, 0.4, 0.2, 0.9), rtol=1e-15, atol=0)
        assert got[0] == 1.0 and got[-1] == 0.4 and np.abs(np.diff(got)).max() < 2e-3
    cub = F.ledge_bathymetry(np.array([0.2, 0.2 + 1e-6, 0.9 - 1e-6, 0.9]), None, 3, 1.0, 0.4, 0.2, 0.9)
    assert abs(cub[1] - cub[0]) < 1e-10 and abs(cub[3] - cub[2]) < 1e-10      # zero slope where the cubic meets the plateaus
    bump = F.ledge_bathymetry(np.array([0.2, 0.7]), np.array([0.9, 0.9]), 3, 0.5, 0.4, 0.2, 0.9)
    np.testing.assert_allclose(bump, [1.0, np.exp(-1.0)], rtol=1e-15)
    Lx, ang = 40.5, np.deg2rad(21.0)
    xs = np.linspace(-3.0, 3.0, 6001)
    r = F.beam_generator_bathymetry(xs, Lx, ang)
    np.testing.assert_allclose(r, sm.beam_generator_bathymetry(xs, Lx, ang), rtol=1e-15, atol=0)
    np.testing.assert_allclose(r, r[::-1], rtol=0, atol=1e-14)                   # even
    assert r[0] == 0.0 and r[-1] == 0.0 and np.abs(np.diff(r)).max() < 1e-3      # flat outside, continuous
    lp, Bp, Pp = 0.009714, 0.01173, 0.0183542
    lstar = lp * Lx + (Bp * Lx + Pp * Lx) / np.cos(ang)
    assert abs(r.max() - (lstar * np.sin(ang) - 0.5 * np.tan(ang) * Pp * Lx)) < 1e-12


def test_dem_interpolators_against_scipy():
    """DEMMap's interpolation cores as host helpers: the natural cubic spline (CubicSpline::solve / interp, the Numerical
    Recipes tridiagonal solve) against scipy's CubicSpline(bc_type="natural"), BilinearInterp2D against scipy's
    RegularGridInterpolator -- independent implementations of the same mathematics; nodes reproduce the data exactly."""
    from scipy.interpolate import CubicSpline, RegularGridInterpolator
    from somar_amd import api as F
    rng = np.random.default_rng(5)
    xd = np.cumsum(rng.uniform(0.5, 1.5, 40))
    fd = np.sin(0.3 * xd) + 0.1 * rng.standard_normal(40)
    x = np.sort(rng.uniform(xd[0], xd[-1], 500))
    got = F.dem_cubic_spline(x, xd, fd)
    np.testing.assert_allclose(got, CubicSpline(xd, fd, bc_type="natural")(x), rtol=0, atol=2e-13)
    np.testing.assert_allclose(F.dem_cubic_spline(xd, xd, fd), fd, rtol=0, atol=1e-15)
    yd = np.cumsum(rng.uniform(0.5, 1.5, 30))
    f2 = rng.standard_normal((40, 30))
    px, py = rng.uniform(xd[0], xd[-1], 400), rng.uniform(yd[0], yd[-1], 400)
    want = RegularGridInterpolator((xd, yd), f2, method="linear")(np.stack([px, py], axis=1))
    np.testing.assert_allclose(F.dem_bilinear(px, py, xd, yd, f2), want, rtol=0, atol=2e-14)
    gx, gy = np.meshgrid(xd, yd, indexing="ij")
    np.testing.assert_allclose(F.dem_bilinear(gx.ravel(), gy.ravel(), xd, yd, f2), f2.ravel(), rtol=0, atol=1e-15)


def test_dem_hermite_interpolant_as_the_reference_writes_it():
    """HermiteInterp2D with Create_Level_DEM_3D's difference tables: nodes reproduce the data; on a UNIT-spaced grid (where the
    reference's unscaled nodal derivatives are the right ones) a plane a + b x + c y is reproduced exactly (the simplified
    interpolant blends its cubic Hermite edges with smoothstep weights, so a twist term x y is not); and a numpy restatement of the same
    statements agrees to roundoff on a ragged grid."""
    from somar_amd import api as F
    rng = np.random.default_rng(8)
    xd, yd = np.arange(12.0), np.arange(9.0)
    gx, gy = np.meshgrid(xd, yd, indexing="ij")
    f = 1.5 - 0.7 * gx + 0.3 * gy
    px, py = rng.uniform(0, 11, 300), rng.uniform(0, 8, 300)
    np.testing.assert_allclose(F.dem_bilinear(px, py, xd, yd, f, hermite=True), 1.5 - 0.7 * px + 0.3 * py, rtol=0, atol=1e-13)
    xr, yr = np.cumsum(rng.uniform(0.5, 1.5, 14)), np.cumsum(rng.uniform(0.5, 1.5, 11))
    g = rng.standard_normal((14, 11))
    rx, ry = np.meshgrid(xr, yr, indexing="ij")
    np.testing.assert_allclose(F.dem_bilinear(rx.ravel(), ry.ravel(), xr, yr, g, hermite=True), g.ravel(), rtol=0, atol=1e-14)
    # the same statements in numpy
    fx, fy = np.empty_like(g), np.empty_like(g)
    fx[0] = (g[1] - g[0]) / (xr[1] - xr[0]); fx[-1] = (g[-1] - g[-2]) / (xr[-1] - xr[-2])
    fx[1:-1] = (g[2:] - g[:-2]) / (xr[2:] - xr[:-2])[:, None]
    fy[:, 0] = (g[:, 1] - g[:, 0]) / (yr[1] - yr[0]); fy[:, -1] = (g[:, -1] - g[:, -2]) / (yr[-1] - yr[-2])
    fy[:, 1:-1] = (g[:, 2:] - g[:, :-2]) / (yr[2:] - yr[:-2])[None, :]
    qx, qy = rng.uniform(xr[0], xr[-1], 200), rng.uniform(yr[0], yr[-1], 200)
    il = np.clip(np.searchsorted(xr, qx, side="right") - 1, 0, 12); jl = np.clip(np.searchsorted(yr, qy, side="right") - 1, 0, 9)
    u = (qx - xr[il]) / (xr[il + 1] - xr[il]); v = (qy - yr[jl]) / (yr[jl + 1] - yr[jl])
    h = lambda t: (1.0 - (3.0 - 2.0 * t) * t * t, (3.0 - 2.0 * t) * t * t, ((t - 2.0) * t + 1.0) * t, (t - 1.0) * t * t)
    h1u, h2u, h3u, h4u = h(u); h1v, h2v, h3v, h4v = h(v)
    A, B, Cc, D = (il, jl), (il + 1, jl), (il, jl + 1), (il + 1, jl + 1)
    fAB = g[A] * h1u + g[B] * h2u + fx[A] * h3u + fx[B] * h4u
    fCD = g[Cc] * h1u + g[D] * h2u + fx[Cc] * h3u + fx[D] * h4u
    fAC = g[A] * h1v + g[Cc] * h2v + fy[A] * h3v + fy[Cc] * h4v
    fBD = g[B] * h1v + g[D] * h2v + fy[B] * h3v + fy[D] * h4v
    want = fAB * h1v + fCD * h2v + fAC * h1u + fBD * h2u - g[A] * h1u * h1v - g[B] * h2u * h1v - g[Cc] * h1u * h2v - g[D] * h2u * h2v
    np.testing.assert_allclose(F.dem_bilinear(qx, qy, xr, yr, g, hermite=True), want, rtol=0, atol=1e-13)
