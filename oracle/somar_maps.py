"""CPU restatement (TEST INFRASTRUCTURE, never imported by the product) of the coordinate maps' metric producers
(SURVEY.md 8f rank 3), numpy over whole boxes, one statement per Fortran statement so the roundings are the reference's:

  CylindricalMap::fill_dxdXi / fill_J           geometry/maps/CylindricalMap.cpp:125-190, CylindricalMapF.ChF
                                                (CYLINDRICAL_FILL_DXDXI, CYLINDRICAL_FILL_J); isDiagonal() = true
  BathymetricBaseMap::fill_dxdXi / fill_J       geometry/maps/BathymetricBaseMap.cpp:133-313 (the !isDiagonal() branches),
                                                BathymetricBaseMapF.ChF (FILL_BATHYDXDXI, FILL_BATHYDZDXI, FILL_BATHYDZDZETA,
                                                VERTPHI(f) = f, DVERTPHI(f) = one, HORIZPHI(f) = f)
  CONVERTFAB                                    calculus/interpolation/ConvertFABF.ChF:32-150 with AVG1IX / AVG2IX / AVG3IX of
                                                utils/AddlFortranMacros.H:74-88
  GeoSourceInterface::fill_Jgup / fill_Jinv     geometry/GeoSourceInterface.cpp:296-311, 417-450 (somar_oracle.geo_fill_jgup)

Parity unpinned w.r.t. reference tests (the reference ships none); tests/test_oracle_maps.py pins it by known answers.

Reference quirk reproduced, not fixed: AVG3IX's eighth term reads f(i0+ii0+jj0+kk0, i1+ii1+kk1+kk1, i2+ii2+jj2+kk2) -- kk1
twice where jj1 belongs (AddlFortranMacros.H:88) -- so an average over all three directions takes the corner (ii + kk) twice
and never the corner (ii + jj + kk).  It is what BathymetricBaseMap's cell-centred J (hence J^{-1}) goes through.

The nodal depth (BathymetricBaseMap::fill_bathymetry, a virtual of the DEM / Ledge / BeamGenerator maps) is an INPUT here:
a 2-D array over nodes [dlo, dlo + shape) of the level's index space."""
import numpy as np

from . import somar_oracle as so

CYLINDRICAL, BATHYMETRIC, TWISTED = 1, 2, 3


def _idx(lo, hi):
    return np.meshgrid(*[np.arange(lo[d], hi[d] + 1) for d in range(3)], indexing="ij")


def convert_fab(f, I, J, K, srcType, destType):
    """dest(i) of CONVERTFAB for a pointwise-evaluable source f(i0, i1, i2)"""
    ii = (srcType[0] - destType[0], 0, 0)
    jj = (0, srcType[1] - destType[1], 0)
    kk = (0, 0, srcType[2] - destType[2])
    num = abs(ii[0]) + abs(jj[1]) + abs(kk[2])

    def at(*offs):
        o = [sum(x[d] for x in offs) for d in range(3)]
        return f(I + o[0], J + o[1], K + o[2])

    if num == 0:
        return at()
    if num == 1:
        o = ii if ii[0] else (jj if jj[1] else kk)
        return 0.5 * (at() + at(o))
    if num == 2:
        if ii[0] == 0:
            a, b = jj, kk
        elif jj[1] == 0:
            a, b = kk, ii
        else:
            a, b = ii, jj
        return 0.25 * (((at() + at(a)) + at(b)) + at(a, b))
    # AVG3IX, eighth term as written: (i0+ii0, i1, i2+kk2)
    s = at() + at(ii)
    s = s + at(jj)
    s = s + at(ii, jj)
    s = s + at(kk)
    s = s + at(ii, kk)
    s = s + at(jj, kk)
    s = s + at(ii, kk)
    return 0.125 * s


class CylindricalMap:
    diagonal = True

    def __init__(self, dXi):
        self.dXi = tuple(float(x) for x in dXi)

    def dxdXi(self, mu, nu, T, I, J, K, scale=1.0):
        shape = np.broadcast(I, J, K).shape
        if mu == 2 or nu == 2:
            return np.full(shape, scale if mu == nu else 0.0)
        off0 = (1.0 - T[0]) * 0.5
        off1 = (1.0 - T[1]) * 0.5
        Xi0 = self.dXi[0] * (I + off0)
        Xi1 = self.dXi[1] * (J + off1)
        if mu == 0:
            v = scale * np.cos(Xi1) if nu == 0 else -scale * Xi0 * np.sin(Xi1)
        else:
            v = scale * np.sin(Xi1) if nu == 0 else scale * Xi0 * np.cos(Xi1)
        return np.broadcast_to(v, shape).copy()

    def J(self, T, I, J, K, scale=1.0):
        off0 = (1.0 - T[0]) * 0.5
        scaleDXi0 = scale * self.dXi[0]
        return np.broadcast_to(scaleDXi0 * (I + off0), np.broadcast(I, J, K).shape).copy()


class BathymetricMap:
    diagonal = False

    def __init__(self, dXi, L, depth, dlo):
        self.dXi, self.L = tuple(float(x) for x in dXi), tuple(float(x) for x in L)
        self.depth, self.dlo = np.asarray(depth, dtype=np.float64), tuple(dlo)

    def _d(self, a, b):
        return self.depth[a - self.dlo[0], b - self.dlo[1]]

    def dxdXi(self, mu, nu, T, I, J, K, scale=1.0):
        shape = np.broadcast(I, J, K).shape
        if mu != 2:
            if nu != 2 and mu == nu:
                offsetF = 0.5 * (1.0 - float(T[nu])) + 0.5
                offsetB = 0.5 * (1.0 - float(T[nu])) - 0.5
                twoDXiOnL = 2.0 * self.dXi[nu] / self.L[nu]
                invDXi = scale / twoDXiOnL
                i = (I, J)[nu].astype(np.float64)
                XiF = twoDXiOnL * (i + offsetF)
                XiB = twoDXiOnL * (i + offsetB)
                return np.broadcast_to((XiF - XiB) * invDXi, shape).copy()
            return np.zeros(shape)
        H, dZeta = self.L[2], self.dXi[2]
        if nu != 2:
            E = [1, 1, 1]
            E[nu] = 0
            invDXi = 1.0 / self.dXi[nu]
            e = (1, 0) if nu == 0 else (0, 1)

            def edge(a, b, c):
                zetaFrac = c.astype(np.float64) * dZeta / H
                DDepthDXi = (self._d(a + e[0], b + e[1]) - self._d(a, b)) * invDXi
                return scale * (1.0 - zetaFrac) * DDepthDXi
            return np.broadcast_to(convert_fab(edge, I, J, K, E, T), shape).copy()

        def node(a, b, c):
            depthFrac = self._d(a, b) / H
            return np.broadcast_to(scale * (1.0 - depthFrac) * 1.0, np.broadcast(a, b, c).shape)
        return np.broadcast_to(convert_fab(node, I, J, K, (1, 1, 1), T), shape).copy()

    def J(self, T, I, J, K, scale=1.0):
        d = self.dxdXi(2, 2, T, I, J, K, scale)
        d = d * self.dxdXi(0, 0, T, I, J, K, 1.0)
        d = d * self.dxdXi(1, 1, T, I, J, K, 1.0)
        return d


class TwistedMap:
    """TwistedMap with m_twistType 0 (geometry/maps/TwistedMap.cpp:160-260): x^mu = xi^mu + pert_mu sin(2 pi xi^nu) sin(2 pi xi^o);
    fill_dxdXi = setVal(scale) on the diagonal and TWISTED0_FILL_DXDXI off it (TwistedMapF.ChF:181-262), fill_J =
    TWISTED0_FILL_J (TwistedMapF.ChF:270-340), one statement per Fortran statement"""
    diagonal = False

    def __init__(self, dXi, pert):
        self.dXi, self.pert = tuple(float(x) for x in dXi), tuple(float(x) for x in pert)

    def dxdXi(self, mu, nu, T, I, J, K, scale=1.0):
        shape = np.broadcast(I, J, K).shape
        if mu == nu:
            return np.full(shape, scale)
        twoPi = 2.0 * np.pi
        o = 3 - mu - nu
        x = (I, J, K)
        offn, offo = (1.0 - T[nu]) * 0.5, (1.0 - T[o]) * 0.5
        scaledPert = twoPi * scale * self.pert[mu]
        v = scaledPert * np.cos(twoPi * self.dXi[nu] * (x[nu] + offn)) * np.sin(twoPi * self.dXi[o] * (x[o] + offo))
        return np.broadcast_to(v, shape).copy()

    def J(self, T, I, J, K, scale=1.0):
        Pi = np.pi
        twoPi = 2.0 * Pi
        twoPiPi = twoPi * Pi
        p = self.pert
        pertProd = Pi * p[0] * p[1] * p[2]
        Xi0 = twoPi * self.dXi[0] * (I + (1.0 - T[0]) * 0.5)
        Xi1 = twoPi * self.dXi[1] * (J + (1.0 - T[1]) * 0.5)
        Xi2 = twoPi * self.dXi[2] * (K + (1.0 - T[2]) * 0.5)
        cCos2 = p[2] * np.cos(Xi2)
        SinXi2, Sin2Xi2 = np.sin(Xi2), np.sin(2.0 * Xi2)
        aSin22 = p[0] * SinXi2 ** 2
        CosXi1, SinXi1, Sin2Xi1 = np.cos(Xi1), np.sin(Xi1), np.sin(2.0 * Xi1)
        SinProd = pertProd * Sin2Xi1 * Sin2Xi2
        CosProd = -2.0 * p[0] * cCos2 * SinXi1 ** 2
        twobCos1 = -2.0 * p[1] * CosXi1
        CosXi0, SinXi0, Sin2Xi0 = np.cos(Xi0), np.sin(Xi0), np.sin(2.0 * Xi0)
        v = scale * (1.0 + twoPiPi * (twobCos1 * (cCos2 * SinXi0 ** 2 + CosXi0 * aSin22) + (CosXi0 * CosProd + Sin2Xi0 * SinProd)))
        return np.broadcast_to(v, np.broadcast(I, J, K).shape).copy()


class TwistedMap1:
    """TwistedMap with m_twistType 1 (geometry/maps/TwistedMap.cpp:99-113, 160-260): only the coordinate functions are analytic
    (TWISTED1_FILL_PHYSCOOR, TwistedMapF.ChF:356-430); fill_dxdXi and fill_J are GeoSourceInterface's defaults
    (GeoSourceInterface.cpp:65-113, 122-202): x^mu on the box staggered in nu, differenced by SIMPLECCDERIV / SIMPLEFCDERIV
    (GeoSourceInterfaceF.ChF:195-246); J cell-centred from the face coordinates (DEFAULT_FILL_J_3D, :67-105), on a box that is
    face-centred in one direction the average of the two cells beside the face (Chombo CellToEdge), times scale / prod(dXi)."""
    diagonal = False

    def __init__(self, dXi, pert, L):
        self.dXi = tuple(float(x) for x in dXi)
        self.pert = tuple(float(x) for x in pert)
        self.L = tuple(float(x) for x in L)

    def x(self, mu, Tc, I, J, K):
        Pi = np.pi
        Xi0 = self.dXi[0] * (I + (1.0 - Tc[0]) * 0.5)
        Xi1 = self.dXi[1] * (J + (1.0 - Tc[1]) * 0.5)
        Xi2 = self.dXi[2] * (K + (1.0 - Tc[2]) * 0.5)
        k0, k1, k2 = Pi / self.L[0], Pi / self.L[1], Pi / self.L[2]
        phi0 = phi1 = phi2 = 0.25 * Pi
        pert = self.pert[mu]
        if mu == 0:
            return Xi0 + pert * np.sin(k0 * Xi0) * np.cos(2.0 * k1 * Xi1 + phi1) * np.cos(2.0 * k2 * Xi2 + phi2)
        if mu == 1:
            return Xi1 + pert * np.cos(2.0 * k0 * Xi0 + phi0) * np.sin(k1 * Xi1) * np.cos(2.0 * k2 * Xi2 + phi2)
        return Xi2 + pert * np.cos(2.0 * k0 * Xi0 + phi0) * np.cos(2.0 * k1 * Xi1 + phi1) * np.sin(k2 * Xi2)

    def dxdXi(self, mu, nu, T, I, J, K, scale=1.0):
        Tx = list(T)
        Tx[nu] = 1 - T[nu]
        e = [1 if d == nu else 0 for d in range(3)]
        scaleOnDXi = scale / self.dXi[nu]
        if T[nu] == 0:
            v = (self.x(mu, Tx, I + e[0], J + e[1], K + e[2]) - self.x(mu, Tx, I, J, K)) * scaleOnDXi
        else:
            v = (self.x(mu, Tx, I, J, K) - self.x(mu, Tx, I - e[0], J - e[1], K - e[2])) * scaleOnDXi
        return np.broadcast_to(v, np.broadcast(I, J, K).shape).copy()

    def _ccj(self, I, J, K):
        d = [[None] * 3 for _ in range(3)]
        for nu in range(3):
            Tf = [0, 0, 0]
            Tf[nu] = 1
            e = [1 if q == nu else 0 for q in range(3)]
            for mu in range(3):
                d[nu][mu] = self.x(mu, Tf, I + e[0], J + e[1], K + e[2]) - self.x(mu, Tf, I, J, K)
        xXi, yXi, zXi = d[0]
        xNu, yNu, zNu = d[1]
        xZeta, yZeta, zZeta = d[2]
        return xXi * (yNu * zZeta - yZeta * zNu) + xNu * (yZeta * zXi - yXi * zZeta) + xZeta * (yXi * zNu - yNu * zXi)

    def J(self, T, I, J, K, scale=1.0):
        f = 0 if T[0] else (1 if T[1] else (2 if T[2] else -1))
        if f < 0:
            v = self._ccj(I, J, K)
        else:
            e = [1 if q == f else 0 for q in range(3)]
            v = 0.5 * (self._ccj(I, J, K) + self._ccj(I - e[0], J - e[1], K - e[2]))
        v = v * (scale / (self.dXi[0] * self.dXi[1] * self.dXi[2]))
        return np.broadcast_to(v, np.broadcast(I, J, K).shape).copy()


def fill_jgup(m, valid, mu):
    """LevelGeometry's FC J g^{mu nu} on faces(valid, mu): -> array (faces..., 3) (for a diagonal map the nu != mu
    components are zero, GeoSourceInterface.cpp:431-436)"""
    fb = valid.faces(mu)
    I, J, K = _idx(fb.lo, fb.hi)
    T = [0, 0, 0]
    T[mu] = 1
    n = I.size
    dx = np.empty((n, 3, 3))
    for r in range(3):
        for s in range(3):
            dx[:, r, s] = m.dxdXi(r, s, T, I, J, K).reshape(-1)
    detJ = m.J(T, I, J, K).reshape(-1)
    g = so.geo_fill_jgup(dx, detJ, mu)
    if m.diagonal:
        for nu in range(3):
            if nu != mu:
                g[:, nu] = 0.0
    return g.reshape(I.shape + (3,))


def fill_jinv(m, valid):
    """fill_Jinv = fill_J then FArrayBox::invert(1.0) on the cell-centred valid box"""
    I, J, K = _idx(valid.lo, valid.hi)
    return 1.0 / m.J((0, 0, 0), I, J, K)


def ledge_bathymetry(x, order, hl, hr, xl, xr):
    """LedgeMap::fill_bathymetry, CH_SPACEDIM == 2 branch (geometry/maps/LedgeMap.cpp:38-60, 118-160)"""
    x = np.asarray(x, dtype=np.float64)
    dh, dx = hr - hl, xr - xl
    invdx3 = dx ** -3.0
    if order == 1:
        mid = (hr - xr * dh / dx) + x * (dh / dx)
    else:
        c0 = hr + dh * (3.0 * xl - xr) * xr * xr * invdx3
        c1 = -6.0 * dh * xl * xr * invdx3
        c2 = 3.0 * dh * (xl + xr) * invdx3
        c3 = -2.0 * dh * invdx3
        mid = c0 + x * (c1 + x * (c2 + x * c3))
    return np.where(x < xl, hl, np.where(x > xr, hr, mid))


def beam_generator_bathymetry(x, Lx, angle):
    """FILL_BeamGeneratorMapBATHYMETRY (geometry/maps/BeamGeneratorMapF.ChF:51-166), Masoud's lab-scale PARAMETER set"""
    x = np.asarray(x, dtype=np.float64)
    lp, Bp, Pp = 0.009714, 0.01173, 0.0183542
    sa, ca, ta = np.sin(angle), np.cos(angle), np.tan(angle)
    l, B, P = lp * Lx, Bp * Lx, Pp * Lx
    lstar = l + (B + P) / ca
    C1, C2, C3, C4, C5, C6 = -lstar * ca - B, -lstar * ca + B, -P, P, lstar * ca - B, lstar * ca + B
    b0 = 0.25 * ta * (B + lstar * ca) * (B + lstar * ca) / B
    b1 = -0.5 * ta * (B + lstar * ca) / B
    b2 = 0.25 * ta / B
    p0 = lstar * sa - 0.5 * ta * P
    p2 = -0.5 * ta / P
    out = np.zeros_like(x)
    m = (C1 < x) & (x < C2)
    out[m] = b2 * x[m] * x[m] - b1 * x[m] + b0
    m = (C2 <= x) & (x <= C3)
    out[m] = lstar * sa + ta * x[m]
    m = (C3 < x) & (x < C4)
    out[m] = p2 * x[m] * x[m] + p0
    m = (C4 <= x) & (x <= C5)
    out[m] = lstar * sa - ta * x[m]
    m = (C5 < x) & (x < C6)
    out[m] = b2 * x[m] * x[m] + b1 * x[m] + b0
    return out
