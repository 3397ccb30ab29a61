"""
oracle/somar_amr.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the multi-level (AMR) part of SOMAR's pressure solve, on top of oracle/somar_oracle.py:

  CFRegion / CFIVS (Chombo, patched copies in MappedChombo/HeaderOverrides)       -> class CFRegion
  homogeneousCFInterp                    calculus/interpolation/HomogeneousCFInterp.cpp:30-201
  MappedQuadCFInterp + MappedQuadCFStencil MappedChombo/MappedQuadCFInterp.cpp:68-622,
                                         MappedChombo/MappedCFStencil.cpp:224-307, 378-597, 831-1232,
                                         MappedChombo/MappedQuadCFInterpF.ChF:9-127
  MappedLevelFluxRegister                MappedChombo/MappedLevelFluxRegister.cpp:80-657, ...F.ChF:9-44
  MappedAMRPoissonOp AMR* members        calculus/AMRElliptic/MappedAMRPoissonOp.cpp:1311-1707
  MappedAMRMultiGrid AMRVCycle & friends calculus/AMRElliptic/MappedAMRMultiGrid.H:736-927, 979-1215, 1320-1598

Scope: diagonal metric, zero-Neumann/periodic physical BCs, AMR refinement ratios with entries in {1, 2, 4}
(entries > 2: mini V-cycles through forced MG depths), LevelGSRB.  Parity unpinned w.r.t. reference tests (there are none); pinned by
tests/test_oracle_amr.py (exactness of the quadratic CF interpolation on quadratics, conservation of the
refluxed composite operator, composite convergence).
"""
import ctypes as C

import numpy as np

from . import somar_oracle as so
from .somar_oracle import Box, Domain, Fab, LevelData, _b, _iv, _ivc, _rv, lib


# ----------------------------------------------------------------------------
# helpers on layouts
# ----------------------------------------------------------------------------
def periodic_images(domain, boxes):
    """CFStencil::buildPeriodicVector: every box plus its periodic images."""
    out = []
    for sh in so.periodic_shifts(domain):
        for b in boxes:
            out.append(b.shift(sh))
    return out


def mask_minus_boxes(region, boxes):
    """boolean array over `region`: True where NOT covered by any of `boxes`."""
    m = np.ones(region.size(), dtype=bool)
    for b in boxes:
        r = region & b
        if not r.isEmpty():
            m[r.slices(region.lo)] = False
    return m


class CFRegion:
    """For every box / direction / side: the ghost cells (1 deep) that abut a coarser level, i.e. lie inside
    the (periodically extended) domain and are covered by no box of this level."""

    def __init__(self, grids, domain):
        self.grids, self.domain = list(grids), domain
        imgs = periodic_images(domain, self.grids)
        dom = domain.box.grow([1 if p else 0 for p in domain.periodic])
        self.ivs = {}
        for i, g in enumerate(self.grids):
            for d in range(3):
                for s in (0, 1):
                    gb = g.adjCell(d, s, 1) & dom
                    if gb.isEmpty():
                        self.ivs[(i, d, s)] = (gb, None)
                        continue
                    m = mask_minus_boxes(gb, imgs)
                    self.ivs[(i, d, s)] = (gb, m if m.any() else None)

    def coarsen(self, r):
        return CFRegion([g.coarsen(r) for g in self.grids], self.domain.coarsen(r))

    def has_cf(self):
        return any(v[1] is not None for v in self.ivs.values())

    # ---- homogeneousCFInterp (HomogeneousCFInterp.cpp:96-201; formula :56, 72-73) ---------------------
    def homogeneous_cf_interp(self, phi, dx, dxCrse, activeDirs):
        for i, g in enumerate(self.grids):
            f = phi[i]
            for d in range(3):
                if activeDirs[d] == 0 or phi.ghost[d] < 1:
                    continue
                for s in (0, 1):
                    gb, m = self.ivs[(i, d, s)]
                    if m is None:
                        continue
                    sgn = 1 if s else -1
                    e = [0, 0, 0]
                    e[d] = sgn
                    Df, Dc = dx[d], dxCrse[d]
                    ghost = f.view(gb)[..., 0]
                    pb = f.view(gb.shift([-a for a in e]))[..., 0]
                    if f.box.size()[d] == 3:
                        factor = 1.0 - 2.0 * Df / (Df + Dc)
                        new = factor * pb
                    else:
                        pa = f.view(gb.shift([-2 * a for a in e]))[..., 0]
                        c1 = 2.0 * (Dc - Df) / (Dc + Df)
                        c2 = -(Dc - Df) / (Dc + 3.0 * Df)
                        new = c1 * pb + c2 * pa
                    ghost[m] = new[m]

    def extrapolate_cf_ev(self, phi, order, activeDirs):
        """ExtrapolateCFEV (LevelData version), extrapolation/ExtrapolationUtils.cpp:165-372: edge and vertex ghosts
        next to the coarse-fine faces, which only the cross terms of a non-diagonal metric read.  Works on the
        bounding box (minBox) of each face's CF cells, as the reference does.  Order 2 only (the callers' choice)."""
        assert order == 2
        ndim = 3 if activeDirs[2] else 2

        def quad(f, near, step):
            """state(near - step) := quadraticExtrap(state(near), state(near + step), state(near + 2 step))"""
            g = lambda q: float(f.view(Box(q, q))[0, 0, 0, 0])
            add = lambda a, k: [a[t] + k * step[t] for t in range(3)]
            dst = add(near, -1)
            if any(dst[t] < f.box.lo[t] or dst[t] > f.box.hi[t] for t in range(3)):
                return
            f.view(Box(dst, dst))[...] = 3.0 * (g(near) - g(add(near, 1))) + g(add(near, 2))

        for d in range(3):
            if activeDirs[d] == 0:
                continue
            for i in range(len(self.grids)):
                f = phi[i]
                for s in (0, 1):
                    gb, m = self.ivs[(i, d, s)]
                    if m is None:
                        continue
                    idx = np.argwhere(m)
                    lo = [gb.lo[t] + int(idx[:, t].min()) for t in range(3)]
                    hi = [gb.lo[t] + int(idx[:, t].max()) for t in range(3)]
                    faceBox = Box(lo, hi) & f.box
                    if faceBox.isEmpty():
                        continue
                    if ndim == 2:
                        vdir = 1 - d
                        if activeDirs[vdir] == 0:
                            continue
                        v = [0, 0, 0]
                        v[vdir] = 1
                        quad(f, list(faceBox.lo), v)
                        quad(f, list(faceBox.hi), [-a for a in v])
                        continue
                    for edir in range(3):
                        if edir == d or activeDirs[edir] == 0:
                            continue
                        for es in (0, 1):
                            edgeBox = faceBox.adjCell(edir, es, 1) & f.box
                            if edgeBox.isEmpty():
                                continue
                            elo, ehi = _b(edgeBox)
                            rc = lib().orc_extrapolatefacenoev(*f.fra(), *f.fran(), elo, ehi, edir, 1 if es else -1, order)
                            assert rc == 0
                            vdir = 3 - edir - d
                            if activeDirs[vdir] == 0:
                                continue
                            v = [0, 0, 0]
                            v[vdir] = 1
                            quad(f, list(edgeBox.lo), v)
                            quad(f, list(edgeBox.hi), [-a for a in v])


# ----------------------------------------------------------------------------
# MappedQuadCFInterp
# ----------------------------------------------------------------------------
class QuadCFStencil:
    """MappedQuadCFStencil for one fine box, one direction, one side (MappedCFStencil.cpp:831-1232)."""

    def __init__(self, fineDomain, grid, fineImgs, crseImgsOfFine, coarBoxes, r, direction, side, dims=3):
        self.dir, self.side = direction, side
        self.empty = True
        edge = grid.adjCell(direction, side, 1) & fineDomain.box.grow([1 if p else 0 for p in fineDomain.periodic])
        # MappedCFStencil::define_2: edgebox = a_fineDomain & edgebox (ProblemDomain & keeps periodic images)
        if edge.isEmpty():
            return
        m = mask_minus_boxes(edge, fineImgs)
        if not m.any():
            return
        self.fineBox, self.fineMask = edge, m
        self.packed = bool(m.all())
        crseDomain = fineDomain.coarsen(r)
        # coarse IVS = coarsen(fine IVS) & coarse domain
        cbox = edge.coarsen(r)
        cm = np.zeros(cbox.size(), dtype=bool)
        idx = np.argwhere(m)
        for q in idx:
            iv = [edge.lo[a] + int(q[a]) for a in range(3)]
            civ = [iv[a] // r[a] for a in range(3)]
            cm[tuple(civ[a] - cbox.lo[a] for a in range(3))] = True
        cdomGrown = crseDomain.box.grow([1 if p else 0 for p in crseDomain.periodic])
        keep = cbox & cdomGrown
        if keep.isEmpty():
            return
        self.coarBox = keep
        self.coarMask = cm[keep.slices(cbox.lo)]
        if not self.coarMask.any():
            return
        self.empty = False
        tran = [a for a in range(dims) if a != direction]
        coarseGrid = grid.coarsen(r)
        g2 = coarseGrid.adjCell(direction, side, 1)
        g1 = coarseGrid.adjCell(direction, side, 1)
        for t in tran:
            g2 = g2.growDir(t, 2)
            g1 = g1.growDir(t, 1)
        g2 = g2 & cdomGrown_for(crseDomain, 2)
        g1 = g1 & cdomGrown_for(crseDomain, 2)
        self.allGoodBox = g2
        allGood = mask_minus_boxes(g2, crseImgsOfFine)
        # edge of the earth: allGood cells not covered by the coarse level (incl. periodic images)
        covered = ~mask_minus_boxes(g2, periodic_images(crseDomain, coarBoxes))
        allGood &= covered
        self.allGood = allGood
        # standard = allGood & g1, shrunk by one in each tangential direction (IntVectSet::grow(-1))
        std = np.zeros(g2.size(), dtype=bool)
        r1 = g1 & g2
        std[r1.slices(g2.lo)] = allGood[r1.slices(g2.lo)]
        for t in tran:
            std = shrink_mask(std, t)
        self.standard = std

    def good(self, iv):
        b = self.allGoodBox
        if any(iv[a] < b.lo[a] or iv[a] > b.hi[a] for a in range(3)):
            return False
        return bool(self.allGood[tuple(iv[a] - b.lo[a] for a in range(3))])

    def is_standard(self, iv):
        b = self.allGoodBox
        if any(iv[a] < b.lo[a] or iv[a] > b.hi[a] for a in range(3)):
            return False
        return bool(self.standard[tuple(iv[a] - b.lo[a] for a in range(3))])


def cdomGrown_for(crseDomain, n):
    """ProblemDomain & box: no clipping in periodic directions."""
    return crseDomain.box.grow([n + 4 if p else 0 for p in crseDomain.periodic])


def shrink_mask(m, axis):
    """IntVectSet::grow(axis, -1): a cell survives iff it and both its axis-neighbours are in the set."""
    out = m.copy()
    lo = np.zeros_like(m)
    hi = np.zeros_like(m)
    sl_src = [slice(None)] * 3
    sl_dst = [slice(None)] * 3
    sl_src[axis], sl_dst[axis] = slice(0, -1), slice(1, None)
    lo[tuple(sl_dst)] = m[tuple(sl_src)]
    sl_src[axis], sl_dst[axis] = slice(1, None), slice(0, -1)
    hi[tuple(sl_dst)] = m[tuple(sl_src)]
    return out & lo & hi


class QuadCFInterp:
    """MappedQuadCFInterp::define / coarseFineInterp."""

    def __init__(self, fineGrids, coarGrids, dxFine, refRatio, fineDomain, ndim=3):
        self.ndim = ndim
        self.fineGrids, self.coarGrids = list(fineGrids), list(coarGrids)
        self.dxf = tuple(dxFine)
        self.r = _iv(refRatio)
        self.dxc = tuple(a * b for a, b in zip(self.dxf, self.r))
        self.fineDomain = fineDomain
        self.crseDomain = fineDomain.coarsen(self.r)
        nfine = sum(g.numPts() for g in fineGrids) // int(np.prod(self.r))
        self.level = 0 if nfine == sum(g.numPts() for g in coarGrids) else 1
        if self.level == 0:
            return
        self.bufBoxes = [g.coarsen(self.r).grow(2) for g in fineGrids]
        fineImgs = periodic_images(fineDomain, self.fineGrids)
        crseImgs = [b.coarsen(self.r) for b in fineImgs]
        self.sten = {}
        for i, g in enumerate(self.fineGrids):
            for d in range(ndim):
                for s in (0, 1):
                    self.sten[(i, d, s)] = QuadCFStencil(fineDomain, g, fineImgs, crseImgs, self.coarGrids, self.r, d, s,
                                                         dims=ndim)

    def fill_buffer(self, phic):
        """a_phic.copyTo(m_coarBuffer, m_copier): coarse valid data (and periodic images) into the grown
        coarsened-fine boxes; cells not covered by the coarse level stay NaN (never read by a valid stencil)."""
        bufs = []
        shifts = so.periodic_shifts(self.crseDomain)
        for bb in self.bufBoxes:
            f = Fab(bb, 1, np.nan)
            for cb, cf in zip(phic.grids, phic.fabs):
                for sh in shifts:
                    r = bb & cb.shift(sh)
                    if not r.isEmpty():
                        f.view(r)[...] = cf.view(r.shift([-a for a in sh]))
            bufs.append(f)
        return bufs

    # -- derivative evaluation (MappedCFStencil.cpp:378-597 with the stencils of :993-1232) ------------------
    def _first_second(self, st, phic, iv, t):
        """-> (first, second, dropOrd contribution) at coarse cell iv in tangential direction t"""
        e = [0, 0, 0]
        e[t] = 1
        p = lambda k: float(phic.view(Box([iv[a] + k * e[a] for a in range(3)], [iv[a] + k * e[a] for a in range(3)]))[0, 0, 0, 0])
        h = self.dxc[t]
        if st.is_standard(iv):
            first = (p(1) - p(-1)) / (2.0 * h)
            second = (p(1) + p(-1) - 2.0 * p(0)) / (h * h)
            return first, second
        drop = st._drop.get(tuple(iv), False)
        g = lambda k: st.good([iv[a] + k * e[a] for a in range(3)])
        # weights are accumulated in box order (low index first), MappedCFStencil.cpp:1233-1251
        if not drop:
            if g(-1) and g(0) and g(1):
                second = (0.0 + 1.0 * p(-1) + -2.0 * p(0) + 1.0 * p(1)) / (h * h)
                first = (0.0 + -0.5 * p(-1) + 0.0 * p(0) + 0.5 * p(1)) / h
                return first, second
            if g(0) and g(1) and g(2):
                second = (0.0 + 1.0 * p(0) + -2.0 * p(1) + 1.0 * p(2)) / (h * h)
                first = (0.0 + (-3.0 / 2.0) * p(0) + (4.0 / 2.0) * p(1) + (-1.0 / 2.0) * p(2)) / h
                return first, second
            if g(-2) and g(-1) and g(0):
                second = (0.0 + 1.0 * p(-2) + -2.0 * p(-1) + 1.0 * p(0)) / (h * h)
                first = (0.0 + (1.0 / 2.0) * p(-2) + (-4.0 / 2.0) * p(-1) + (3.0 / 2.0) * p(0)) / h
                return first, second
            st._drop[tuple(iv)] = True  # m_dropOrd(iv) = true (:1175)
            if g(1):
                return (0.0 + -1.0 * p(0) + 1.0 * p(1)) / h, 0.0
            if g(-1):
                return (0.0 + -1.0 * p(-1) + 1.0 * p(0)) / h, 0.0
            return (0.0 + 0.0 * p(0)) / h, 0.0
        return None, 0.0  # dropped before this direction was visited: no first-derivative stencil was built

    def _mixed(self, st, phic, iv, t1, t2):
        """computeMixedDerivative (MappedCFStencil.cpp:514-597).  t1 < t2 are the tangential directions
        (buildStencils: itran1 = vinttran[0] = t1, itran2 = vinttran[1] = t2)."""
        def p(a, b):
            q = list(iv)
            q[t1] += a
            q[t2] += b
            return float(phic.view(Box(q, q))[0, 0, 0, 0])
        if st.is_standard(iv):
            # itran2 -> basex, itran1 -> basey in computeMixedDerivative; the product of spacings is symmetric
            return (p(1, 1) + p(-1, -1) - p(1, -1) - p(-1, 1)) / (4.0 * self.dxc[t2] * self.dxc[t1])

        def g(a, b):
            q = list(iv)
            q[t1] += a
            q[t2] += b
            return st.good(q)
        # buildStencils (:1010-1075): four 2x2 boxes with low corners (in (t1,t2) offsets)
        #   fabur: (-1, 0)   fabul: (0, 0)   fablr: (0, -1)   fabll: (-1, -1)        [names as in the source]
        # each carrying  -1 at lo and lo+e1+e2,  +1 at lo+e1 and lo+e2  (fabur.setVal(-1); +1 at e1, e2),
        # added in the order ur, ul, lr, ll when the whole box is "all good"; DerivStencil::accum merges a
        # repeated point into its first entry; finally the weights are divided by the number of boxes used.
        entries, weights = [], {}
        nused = 0
        for (la, lb) in ((-1, 0), (0, 0), (0, -1), (-1, -1)):
            if all(g(la + da, lb + db) for da in (0, 1) for db in (0, 1)):
                nused += 1
                for db in (0, 1):          # BoxIterator: lower direction index (t1) fastest
                    for da in (0, 1):
                        w = -1.0 if (da == db) else 1.0
                        key = (la + da, lb + db)
                        if key in weights:
                            weights[key] += w
                        else:
                            weights[key] = w
                            entries.append(key)
        if nused == 0:
            st._drop[tuple(iv)] = True
            return 0.0
        val = 0.0
        for key in entries:
            val += (weights[key] / float(nused)) * p(*key)
        return val / (self.dxc[t1] * self.dxc[t2])

    def coarse_fine_interp(self, phif, phic):
        """MappedQuadCFInterp::coarseFineInterp(LevelData&, const LevelData&), :579-622"""
        if self.level == 0:
            return
        bufs = self.fill_buffer(phic)
        for d in range(self.ndim):
            if phif.ghost[d] == 0:
                continue
            for i in range(len(self.fineGrids)):
                for s in (0, 1):
                    st = self.sten[(i, d, s)]
                    if not st.empty:
                        self._interp_one(phif[i], bufs[i], st)

    def _interp_one(self, f, phic, st):
        d, s = st.dir, st.side
        ihilo = 1 if s else -1
        tran = [a for a in range(self.ndim) if a != d]
        # CH_SPACEDIM == 2 (MappedQuadCFInterp.cpp:300-310, 386-400): one tangential direction, no mixed derivative
        you1, you2 = (tran[0], tran[1]) if self.ndim == 3 else (tran[0], None)
        st._drop = {}
        # slopes on the coarse IVS.  buildStencils visits the mixed stencil first (it decides dropOrd), then
        # the tangential directions in order
        slope, curva, mixed = {}, {}, {}
        cidx = np.argwhere(st.coarMask)
        for q in cidx:
            iv = [st.coarBox.lo[a] + int(q[a]) for a in range(3)]
            key = tuple(iv)
            mixed[key] = self._mixed(st, phic, iv, you1, you2) if you2 is not None else 0.0
            sl, cu = {}, {}
            for t in tran:
                fst, sec = self._first_second(st, phic, iv, t)
                sl[t], cu[t] = fst, sec
            if st._drop.get(key, False) and not st.is_standard(iv):
                # m_dropOrd: second and mixed derivatives are zero (:411, :572); first derivatives use whatever
                # stencil the build left (one-sided 2-point if the drop happened in that direction, else the
                # 3-point one built before the drop)
                for t in tran:
                    cu[t] = 0.0
                    if sl[t] is None:
                        sl[t] = 0.0
                mixed[key] = 0.0
            slope[key], curva[key] = sl, cu
        # phi* at the fine ghost cells, then quadratic interpolation in the normal direction
        e = [0, 0, 0]
        e[d] = ihilo
        nref = self.r[d]
        h = self.dxf[d]
        fidx = np.argwhere(st.fineMask)
        for q in fidx:
            ivf = [st.fineBox.lo[a] + int(q[a]) for a in range(3)]
            ivc = tuple(ivf[a] // self.r[a] for a in range(3))
            pc = float(phic.view(Box(ivc, ivc))[0, 0, 0, 0])
            xs = []
            for t in (you1, you2):
                if t is None:
                    xs.append(0.0)
                    continue
                xf = (ivf[t] + 0.5) * self.dxf[t]
                xc = (ivc[t] + 0.5) * self.dxc[t]
                xs.append(xf - xc)
            x1, x2 = xs
            sl, cu = slope[ivc], curva[ivc]
            if you2 is None:
                # SpaceDim 2: always the C++ path (MAPPEDPHISTAR is an error there); update2 = update3 = 0
                update1 = x1 * sl[you1] + 0.5 * x1 * x1 * cu[you1]
                pstar = pc + update1 + 0.0 + 0.0
            elif st.packed:
                # MAPPEDPHISTAR (MappedQuadCFInterpF.ChF:51-127)
                pstar = pc + (x1 * sl[you1] + 0.5 * x1 * x1 * cu[you1]) + (x2 * sl[you2] + 0.5 * x2 * x2 * cu[you2]) \
                    + x1 * x2 * mixed[ivc]
            else:
                update1 = x1 * sl[you1] + 0.5 * x1 * x1 * cu[you1]
                update2 = x2 * sl[you2] + 0.5 * x2 * x2 * cu[you2]
                update3 = x1 * x2 * mixed[ivc]
                pstar = pc + update1 + update2 + update3
            at = lambda k: float(f.view(Box([ivf[a] + k * e[a] for a in range(3)], [ivf[a] + k * e[a] for a in range(3)]))[0, 0, 0, 0])
            pa, pb = at(-2), at(-1)
            if st.packed:
                # mappedquadinterp (MappedQuadCFInterpF.ChF:9-49)
                frac = 2.0 / (h * h)
                denom = float(nref * nref + 4 * nref + 3)
                mult = frac / denom
                invh = 1.0 / h
                x = 2.0 * h
                xsq = 4.0 * h * h
                a_ = mult * (2.0 * pstar + (nref + 1) * pa - (nref + 3) * pb)
                b_ = (pb - pa) * invh - a_ * h
                val = xsq * a_ + b_ * x + pa
            else:
                a_ = (2.0 / h / h) * (2.0 * pstar + pa * (nref + 1.0) - pb * (nref + 3.0)) / (nref * nref + 4 * nref + 3.0)
                b_ = (pb - pa) / h - a_ * h
                x = 2.0 * h
                val = a_ * x * x + b_ * x + pa
            f.view(Box(ivf, ivf))[...] = val


# ----------------------------------------------------------------------------
# MappedLevelFluxRegister
# ----------------------------------------------------------------------------
class FluxRegister:
    """MappedLevelFluxRegister (MappedChombo/MappedLevelFluxRegister.cpp): coarse register on the coarse
    grids, fine register (1 ghost) on the coarsened fine grids; define :80-188, incrementCoarse :298-347,
    incrementFine :366-448 (+ MAPPEDINCREMENTFINE, ...F.ChF:9-44), reflux :560-604, 608-650."""

    def __init__(self, fineGrids, crseGrids, fineDomain, r, ncomp=1):
        self.r = _iv(r)
        self.crseGrids = list(crseGrids)
        self.cfGrids = [g.coarsen(self.r) for g in fineGrids]
        self.crseDomain = fineDomain.coarsen(self.r)
        npts = sum(g.numPts() for g in self.crseGrids) - sum(g.numPts() for g in self.cfGrids)
        self.defined = npts != 0  # the "temporary flux register optimization", :97-109
        if not self.defined:
            return
        self.coarFlux = LevelData(self.crseGrids, ncomp, (0, 0, 0))
        self.fineFlux = LevelData(self.cfGrids, ncomp, (1, 1, 1))
        # m_coarseLocations[dir + side*SpaceDim][coarse box]: the coarse cells just outside a fine box (:124-187)
        shifts = so.periodic_shifts(self.crseDomain)
        shifts.sort(key=lambda s: s != (0, 0, 0))  # unshifted boxes first, then periodic images
        self.locs = {}
        for ci, cb in enumerate(self.crseGrids):
            for d in range(3):
                for s in (0, 1):
                    v = []
                    for sh in shifts:
                        for fb in self.cfGrids:
                            b = fb.shift(sh).adjCell(d, s, 1) & cb
                            if not b.isEmpty():
                                v.append(b)
                    self.locs[(ci, d, s)] = v
        self.shifts = shifts

    def set_to_zero(self):
        if self.defined:
            so.ld_set(self.coarFlux, 0.0)
            so.ld_set(self.fineFlux, 0.0)

    def increment_coarse(self, flux, scale, ci, d):
        """flux: face-centred Fab over the coarse box's faces in direction d.  Side Lo = cells on the low
        side of a fine box, which see the interface through their HIGH face."""
        if not self.defined:
            return
        coarse = self.coarFlux[ci]
        for s in (0, 1):
            sc = -(1 if s else -1) * scale  # scale = -sign(sd) * a_scale
            for b in self.locs[(ci, d, s)]:
                # shiftHalf(dir, sign): Lo -> cell c reads face c+1;  Hi -> cell c reads face c
                fb = b.shift([1 if (a == d and s == 0) else 0 for a in range(3)])
                coarse.view(b)[...] += sc * flux.view(fb)

    def increment_fine(self, flux, scale, fi, d, s):
        """flux: face-centred Fab holding the boundary faces (bdryBox) of fine box fi on side s of dir d."""
        if not self.defined:
            return
        r = self.r
        denom = float(r[0] * r[1] * r[2] // r[d])
        sc = (1 if s else -1) * scale / denom
        cFine = self.fineFlux[fi]
        clip = self.cfGrids[fi].refine(r)
        fineBox = clip.adjCell(d, s, 1)  # the fine cells just outside, ∩ shifted flux box
        # shifted flux: Lo: face i -> cell i-1 ; Hi: face i -> cell i
        fshift = [(-1 if s == 0 else 0) if a == d else 0 for a in range(3)]
        fcells = flux.box.shift(fshift)
        hi = list(fcells.hi)
        # a bdryBox is one face thick: as cells, it is one cell thick
        fineBox = fineBox & Box(fcells.lo, hi)
        if fineBox.isEmpty():
            return
        cbox = fineBox.coarsen(r)
        dst = cFine.view(cbox)
        src = flux.view(fineBox.shift([-a for a in fshift]))
        # MAPPEDINCREMENTFINE: Fortran-order loop over fine cells -> per coarse cell the fine values arrive
        # ordered by (k, j, i) offset
        for o2 in range(r[2] if d != 2 else 1):
            for o1 in range(r[1] if d != 1 else 1):
                for o0 in range(r[0] if d != 0 else 1):
                    sl = [slice(o0, None, r[0]), slice(o1, None, r[1]), slice(o2, None, r[2])]
                    sl[d] = slice(None)
                    dst[...] = dst + sc * src[tuple(sl)]

    def reflux(self, LofPhi, Jinv):
        """reflux(a_uCoarse, a_scale=1, ..., a_beta=Jinv), :608-650."""
        if not self.defined:
            return
        inc = LevelData(self.crseGrids, LofPhi.ncomp, LofPhi.ghost)
        for ci, cb in enumerate(self.crseGrids):
            inc[ci].view(cb)[...] += -1.0 * self.coarFlux[ci].view(cb)
        # m_fineFlux.copyTo(..., m_reverseCopier, AddOp(scale=-1)): ghosted coarsened-fine boxes -> coarse valid
        for ci, cb in enumerate(self.crseGrids):
            for sh in self.shifts:
                for fi, fb in enumerate(self.cfGrids):
                    src = fb.grow(1).shift(sh)
                    reg = src & cb
                    if reg.isEmpty():
                        continue
                    inc[ci].view(reg)[...] += -1.0 * self.fineFlux[fi].view(reg.shift([-a for a in sh]))
        for ci, cb in enumerate(self.crseGrids):
            inc[ci].view(cb)[...] *= Jinv[ci].view(cb)
            LofPhi[ci].view(cb)[...] += inc[ci].view(cb)


# ----------------------------------------------------------------------------
# AMR levels: MappedAMRPoissonOp's AMR* members and MappedAMRMultiGrid
# ----------------------------------------------------------------------------
class NoOpSolver:
    """Chombo NoOpSolver: 'solves' by zeroing the unknown."""

    def define(self, op, homogeneous):
        self.op = op

    def set_convergence_metrics(self, metric, tol):
        pass

    def solve(self, phi, rhs):
        so.ld_set(phi, 0.0)


class AMRLevel:
    """Everything one AMR level owns: geometry, the level operator (MappedAMRPoissonOp via AMRnewOp,
    MappedAMRPoissonOpFactory.cpp:710-880), its MG hierarchy, CF interpolator and flux register."""

    def __init__(self, domain, grids, dx, Jgup, Jinv):
        self.domain, self.grids, self.dx, self.Jgup, self.Jinv = domain, list(grids), tuple(dx), Jgup, Jinv


class AMRComposite:
    """MappedAMRMultiGrid<LevelData<FArrayBox>> over several AMR levels (define :1407-1490)."""

    def __init__(self, levels, refRatios, bc, bottomSolver, alpha=0.0, beta=1.0, maxDepth=-1,
                 relaxMode=so.RELAX_LEVEL_GSRB, precondIters=2, amrmg_eps=1e-6, ndim=3, isDiagonal=True):
        self.ndim = ndim
        self.levels, self.refRatios = levels, [_iv(r) for r in refRatios]
        n = len(levels)
        assert len(self.refRatios) >= n - 1
        for r in self.refRatios[:n - 1]:
            assert all(x in (1, 2, 4) for x in r), "refinement ratios are 1, 2 or 4 per direction"
        self.eps, self.hang, self.normThresh = 1e-6, 1e-15, 1e-30
        self.imin, self.iterMax = 5, 20
        self.pre = self.post = self.bottom = 2
        self.numMG = 1
        self.convergenceMetric = 0.0
        self.bottomSolverEpsCushion = 1.0
        self.bottomSolver = bottomSolver
        self.nosolve = NoOpSolver()
        self.ops, self.mg = [], []
        for l, L in enumerate(levels):
            cf = CFRegion(L.grids, L.domain) if l > 0 else None
            dxCrse = levels[l - 1].dx if l > 0 else None
            fac = so.Factory(L.domain, L.grids, L.dx, bc, L.Jgup, L.Jinv, alpha=alpha, beta=beta, maxDepth=maxDepth,
                             precondIters=precondIters, relaxMode=relaxMode, amrmg_eps=amrmg_eps, dxCrse=dxCrse, cf=cf,
                             ndim=ndim, isDiagonal=isDiagonal)
            # the mini V-cycle's coarsening pattern, MappedAMRMultiGrid.H:1455-1482
            force = None
            if l > 0:
                force = []
                r = list(self.refRatios[l - 1])
                while max(r) > 2:
                    this = [1, 1, 1]
                    for d in range(3):
                        if r[d] > 2:
                            r[d] //= 2
                            this[d] = 2
                    if this[0] * this[1] * this[2] > 1:
                        force.append(tuple(this))
                force.reverse()
            mg = so.MultiGrid(fac, NoOpSolver(), maxDepth, forceAllMGRefRatios=force)
            op = mg.ops[0]
            op.level = l
            op.refToCoarser = self.refRatios[l - 1] if l > 0 else None
            op.refToFiner = self.refRatios[l] if l < n - 1 else None
            # define(...) with a coarser level: m_interpWithCoarser (MappedAMRPoissonOp.cpp:193-196, 236-239)
            op.quad = (QuadCFInterp(L.grids, levels[l - 1].grids, L.dx, op.refToCoarser, L.domain, ndim=ndim)
                       if l > 0 else None)
            # define(...) with a finer level: m_levfluxreg (:141-147, 243-249)
            op.fluxreg = (FluxRegister(levels[l + 1].grids, L.grids, levels[l + 1].domain, op.refToFiner)
                          if l < n - 1 else None)
            self.ops.append(op)
            self.mg.append(mg)
        self.exitStatus, self.history, self.iters = 0, [], 0

    def set_solver_parameters(self, pre, post, bottom, numMG, iterMax, eps, hang, normThresh):
        self.pre, self.post, self.bottom, self.numMG = pre, post, bottom, numMG
        self.iterMax, self.eps, self.hang, self.normThresh = iterMax, eps, hang, normThresh

    # ---- MappedAMRPoissonOp AMR members ---------------------------------------------------------------
    def interp_cf_ghosts(self, l, phi, phiCoarse):
        """interpCFGhosts(phi, &phiCoarse, false), MappedAMRPoissonOp.cpp:2170-2216."""
        op = self.ops[l]
        op.quad.coarse_fine_interp(phi, phiCoarse)
        op.cf.extrapolate_cf_ev(phi, 2, op.activeDirs)

    def amr_operator(self, l, LofPhi, phiFine, phi, phiCoarse, homogeneous):
        """AMROperator / NC / NF, :1373-1450."""
        op = self.ops[l]
        if phiCoarse is not None:
            self.interp_cf_ghosts(l, phi, phiCoarse)
        op.apply_op_i(LofPhi, phi, homogeneous)
        if phiFine is not None:
            self.reflux(l, phiFine, phi, LofPhi)

    def amr_residual(self, l, resid, phiFine, phi, phiCoarse, rhs, homogeneous):
        """AMRResidual / NC, :1311-1345:  L first, then axby(res, res, rhs, -1, 1)."""
        self.amr_operator(l, resid, phiFine, phi, phiCoarse, homogeneous)
        so.ld_axby(resid, resid, rhs, -1.0, 1.0)

    def amr_residual_nf(self, l, resid, phi, phiCoarse, rhs, homogeneous):
        """AMRResidualNF, :1351-1366 (goes through residualI -> SUBTRACTOP)."""
        if phiCoarse is not None:
            self.interp_cf_ghosts(l, phi, phiCoarse)
        self.ops[l].residual_i(resid, phi, rhs, homogeneous)

    def reflux(self, l, phiFine, phi, LofPhi):
        """reflux, :1615-1707."""
        op, fop = self.ops[l], self.ops[l + 1]
        fr = op.fluxreg
        if not fr.defined:
            # hasCF() == isAllDefined() is false: no increments; interpCFGhosts on the fine level still runs
            self.interp_cf_ghosts(l + 1, phiFine, phi)
            return
        fr.set_to_zero()
        for ci, cb in enumerate(op.grids):
            for d in range(op.ndim):
                flux = Fab(cb.faces(d), phi.ncomp, np.nan)
                extrap = Fab(phi[ci].box, phi.ncomp, np.nan)
                op.fill_extrap(extrap, phi[ci], 2)
                op.get_flux_complete(flux, phi[ci], extrap, cb.faces(d), ci, d)
                fr.increment_coarse(flux, op.beta / op.dx[d], ci, d)
        self.interp_cf_ghosts(l + 1, phiFine, phi)
        for fi, fb in enumerate(fop.grids):
            for d in range(op.ndim):
                for s in (0, 1):
                    fluxBox = fb.edgeCells(d, s).faces(d)
                    # bdryBox(fineRegion, d, side, 1): the one layer of faces on that side
                    lo, hi = list(fluxBox.lo), list(fluxBox.hi)
                    if s == 0:
                        hi[d] = lo[d]
                    else:
                        lo[d] = hi[d]
                    fluxBox = Box(lo, hi)
                    flux = Fab(fluxBox, phi.ncomp, np.nan)
                    extrap = Fab(phiFine[fi].box, phi.ncomp, np.nan)
                    fop.fill_extrap(extrap, phiFine[fi], 2)
                    fop.get_flux_complete(flux, phiFine[fi], extrap, fluxBox, fi, d, 1)
                    fr.increment_fine(flux, op.beta / op.dx[d], fi, d, s)
        fr.reflux(LofPhi, op.Jinv)

    def restrict(self, l, resC, fineRes):
        """FullWeightingPS::restrict with the AMR ratio, MGStrategies/RestrictionStrategy.cpp:36-99."""
        op = self.ops[l]
        r = op.refToCoarser
        for i, cg in enumerate(resC.grids):
            lo, hi = _b(cg)
            lib().orc_mappedaverage2(*resC[i].fra(), *fineRes[i].fran(), *op.Jinv[i].fra1(0), lo, hi, _ivc(r))

    def amr_restrict_s(self, l, resC, residual, correction, coarseCorrection, scratch):
        """AMRRestrictS, :1482-1498."""
        self.amr_residual_nf(l, scratch, correction, coarseCorrection, residual, True)
        self.restrict(l, resC, scratch)

    def amr_prolong_s(self, l, correction, coarseCorrection):
        """AMRProlongS, :1529-1546: copy coarse correction onto the coarsened fine grids, then the level's
        prolongation strategy with the AMR ratio."""
        op = self.ops[l]
        r = op.refToCoarser
        eCoar = LevelData([g.coarsen(r) for g in op.grids], correction.ncomp, coarseCorrection.ghost)
        copy_valid(eCoar, coarseCorrection)
        keep = op.mgCrseRefRatio
        op.mgCrseRefRatio = r
        try:
            op.prolong_increment(correction, eCoar)
        finally:
            op.mgCrseRefRatio = keep

    def amr_update_residual(self, l, residual, correction, coarseCorrection):
        """AMRUpdateResidual, :1553-1565."""
        oldRes = so.ld_create(residual)
        so.ld_assign(oldRes, residual)
        self.amr_residual_nf(l, residual, correction, coarseCorrection, oldRes, True)

    def zero_covered(self, l, resid):
        """zeroCovered -> LevelDataOps::copyToZero over the coarsened finer grids."""
        r = self.ops[l + 1].refToCoarser
        for cb, f in zip(resid.grids, resid.fabs):
            for fb in self.ops[l + 1].grids:
                reg = fb.coarsen(r) & cb
                if not reg.isEmpty():
                    f.view(reg)[...] = 0.0

    # ---- MappedAMRMultiGrid ---------------------------------------------------------------------------
    def relax(self, l, correction, residual, n):
        """relax, MappedAMRMultiGrid.H:736-766: plain smoothing, or -- when MG depths lie between this level and the
        next coarser AMR level (a ratio entry > 2) -- a mini V-cycle over those depths."""
        op, mg = self.ops[l], self.mg[l]
        if op.refToCoarser is not None and max(op.refToCoarser) > 2:
            assert mg.maxForcedDepth > 0
            keep = mg.depth
            mg.depth = mg.maxForcedDepth + 1
            mg.pre, mg.post, mg.bottom, mg.cycle_type = self.pre, self.post, self.bottom, self.numMG
            try:
                mg.cycle(0, correction, residual)
            finally:
                mg.depth = keep
        else:
            op.relax(correction, residual, n)

    def compute_amr_residual_level(self, resid, phi, rhs, l_max, l_base, ilev, homogeneous):
        """computeAMRResidualLevel, :884-927."""
        if l_max != l_base:
            if ilev == l_max:
                self.amr_residual_nf(l_max, resid[l_max], phi[l_max], phi[l_max - 1], rhs[l_max], homogeneous)
            elif ilev == l_base and l_base == 0:
                self.amr_residual(0, resid[0], phi[1], phi[0], None, rhs[0], homogeneous)
            else:
                self.amr_residual(ilev, resid[ilev], phi[ilev + 1], phi[ilev], phi[ilev - 1], rhs[ilev], homogeneous)
        else:
            if l_base == 0:
                self.ops[0].residual(resid[0], phi[0], rhs[0], homogeneous)
            else:
                self.amr_residual_nf(l_max, resid[l_max], phi[l_max], phi[l_max - 1], rhs[l_max], homogeneous)

    def compute_amr_residual(self, resid, phi, rhs, l_max, l_base, homogeneous, computeNorm=True):
        """computeAMRResidual, :793-836."""
        rnorm = 0.0
        for ilev in range(l_base, l_max + 1):
            self.compute_amr_residual_level(resid, phi, rhs, l_max, l_base, ilev, homogeneous)
            if computeNorm:
                if ilev != l_max:
                    self.zero_covered(ilev, resid[ilev])
                rnorm = max(self.ops[ilev].local_max_norm(resid[ilev]), rnorm)
        return rnorm

    def init(self, phi, rhs, l_max, l_base):
        """init, :1320-1352."""
        self.m_correction = [None] * len(self.levels)
        self.m_residual = [None] * len(self.levels)
        self.m_resC = [None] * len(self.levels)
        for i in range(l_base, l_max + 1):
            self.m_correction[i] = so.ld_create(phi[i])
            self.m_residual[i] = so.ld_create(rhs[i])
            if i == l_base:
                self.mg[i].init(phi[i], rhs[i])
            else:
                if self.mg[i].maxForcedDepth > 0:   # "This triggers the mini V-cycles when the AMR ref ratio > 2."
                    keep = self.mg[i].depth
                    self.mg[i].depth = self.mg[i].maxForcedDepth + 1
                    self.mg[i].init(phi[i], rhs[i])
                    self.mg[i].depth = keep
                r = self.ops[i].refToCoarser
                self.m_resC[i] = LevelData([g.coarsen(r) for g in self.ops[i].grids], rhs[i].ncomp, rhs[i].ghost)

    def set_bottom_solver(self, l_max, l_base):
        """setBottomSolver, :1376-1390."""
        for l in range(l_base, l_max + 1):
            self.mg[l].pre, self.mg[l].post, self.mg[l].bottom = self.pre, self.post, self.bottom
            self.mg[l].cycle_type = self.numMG
            self.mg[l].bottomSolver = self.nosolve
        mg = self.mg[l_base]
        mg.bottomSolver = self.bottomSolver
        self.bottomSolver.define(mg.ops[-1], True)

    def amr_vcycle(self, uberCorrection, uberResidual, ilev, l_max, l_base):
        """AMRVCycle, :1498-1597."""
        ops = self.ops
        if ilev == l_max:
            for l in range(l_base, l_max + 1):
                so.ld_assign(self.m_residual[l], uberResidual[l])  # assignLocal
                so.ld_set(self.m_correction[l], 0.0)
        if l_max == l_base:
            self.mg[l_base].one_cycle(uberCorrection[ilev], uberResidual[ilev])
        elif ilev == l_base:
            self.mg[l_base].one_cycle(self.m_correction[ilev], self.m_residual[ilev])
            so.ld_incr(uberCorrection[ilev], self.m_correction[ilev], 1.0)
        else:
            self.relax(ilev, self.m_correction[ilev], self.m_residual[ilev], self.pre)
            so.ld_incr(uberCorrection[ilev], self.m_correction[ilev], 1.0)
            so.ld_set(self.m_correction[ilev - 1], 0.0)
            self.compute_amr_residual_level(self.m_residual, uberCorrection, uberResidual, l_max, l_base, ilev - 1, True)
            # the scratch argument IS uberCorrection[ilev]: it is clobbered here and rebuilt at the end
            self.amr_restrict_s(ilev, self.m_resC[ilev], self.m_residual[ilev], self.m_correction[ilev],
                                self.m_correction[ilev - 1], uberCorrection[ilev])
            copy_valid(self.m_residual[ilev - 1], self.m_resC[ilev])  # assignCopier
            for _ in range(self.numMG):
                self.amr_vcycle(uberCorrection, uberResidual, ilev - 1, l_max, l_base)
            self.amr_prolong_s(ilev, self.m_correction[ilev], self.m_correction[ilev - 1])
            self.amr_update_residual(ilev, self.m_residual[ilev], self.m_correction[ilev], self.m_correction[ilev - 1])
            dCorr = uberCorrection[ilev]
            so.ld_set(dCorr, 0.0)
            self.relax(ilev, dCorr, self.m_residual[ilev], self.post)
            so.ld_incr(self.m_correction[ilev], dCorr, 1.0)
            so.ld_assign(uberCorrection[ilev], self.m_correction[ilev])  # assignLocal

    def solve(self, phi, rhs, l_max, l_base, zeroPhi=True, forceHomogeneous=False):
        """solve -> init + solveNoInit + solveNoInitResid, :933-1183."""
        ops = self.ops
        self.init(phi, rhs, l_max, l_base)
        self.set_bottom_solver(l_max, l_base)
        lowlim = l_base - 1 if l_base > 0 else l_base
        nl = len(self.levels)
        uberCorrection, uberResidual, bestPhi = [None] * nl, [None] * nl, [None] * nl
        for l in range(lowlim, l_max + 1):
            uberCorrection[l] = so.ld_create(phi[l])
            if l >= l_base:
                uberResidual[l] = so.ld_create(rhs[l])
            bestPhi[l] = so.ld_create(phi[l])
        if zeroPhi:
            for l in range(l_base, l_max + 1):
                so.ld_set(phi[l], 0.0)
        for l in range(lowlim, l_max + 1):
            so.ld_assign(bestPhi[l], phi[l])
        initial_rnorm = self.compute_amr_residual(uberResidual, phi, rhs, l_max, l_base, forceHomogeneous)
        if self.convergenceMetric != 0.0:
            initial_rnorm = self.convergenceMetric
        rnorm = initial_rnorm
        norm_last = 2 * initial_rnorm
        best_rnorm = rnorm
        useBestPhi = False
        somethingConverged = False
        self.bottomSolver.set_convergence_metrics(initial_rnorm, self.bottomSolverEpsCushion * self.eps)
        it = 0
        self.history = [rnorm]
        goNorm = rnorm > self.normThresh
        goRedu = rnorm > self.eps * initial_rnorm
        goIter = it < self.iterMax
        goHang = it < self.imin or rnorm < (1 - self.hang) * norm_last
        while goIter and goRedu and goHang and goNorm:
            norm_last = rnorm
            self.amr_vcycle(uberCorrection, uberResidual, l_max, l_max, l_base)
            # postVCycleOps, :1189-1215
            for l in range(l_base, l_max + 1):
                so.ld_incr(phi[l], uberCorrection[l], 1.0)
                so.ld_set(uberCorrection[l], 0.0)
            rnorm = self.compute_amr_residual(uberResidual, phi, rhs, l_max, l_base, forceHomogeneous)
            it += 1
            self.history.append(rnorm)
            if rnorm <= best_rnorm:
                best_rnorm = rnorm
                for l in range(l_base, l_max + 1):
                    so.ld_assign(bestPhi[l], phi[l])
                useBestPhi = False
                somethingConverged = True
            else:
                useBestPhi = True
            goNorm = rnorm > self.normThresh
            goRedu = rnorm > self.eps * initial_rnorm
            goIter = it < self.iterMax
            goHang = it < self.imin or rnorm < (1 - self.hang) * norm_last
        if useBestPhi:
            rnorm = best_rnorm
            for l in range(l_base, l_max + 1):
                so.ld_assign(phi[l], bestPhi[l])
        if rnorm > 10.0 * initial_rnorm and rnorm > 10.0 * self.eps:
            raise RuntimeError("kaboom")
        if (not somethingConverged) and rnorm >= initial_rnorm and rnorm >= self.eps:
            raise RuntimeError("MappedAMRMultiGrid solver blew up")
        self.exitStatus = int(not goRedu) + int(not goIter) * 2 + int(not goHang) * 4 + int(not goNorm) * 8
        self.iters, self.final_rnorm, self.initial_rnorm = it, rnorm, initial_rnorm
        return rnorm


def copy_valid(dst, src):
    """LevelData::copyTo between two layouts of the same index space: valid cells of src -> valid cells of dst."""
    for db, df in zip(dst.grids, dst.fabs):
        for sb, sf in zip(src.grids, src.fabs):
            reg = db & sb
            if not reg.isEmpty():
                df.view(reg)[...] = sf.view(reg)


# ----------------------------------------------------------------------------
# Level projection on level l of a hierarchy: BaseProjector<T>::levelProject -> project(lmin = lmax = l)
# (projection/BaseProjectorI.H:176-366) with the coarser level supplying coarse-fine values.
#   'mac': LevelMACProjector (LevelMACProjector.cpp:156-241) -- levelDivergenceMAC needs nothing from the coarser level;
#          computeGrad -> levelGradientMAC(edgeGrad, phi, crsePhi, cfInterp): coarseFineInterp(phi, crsePhi), exchange,
#          extrapolation BC, MAC gradient (Gradient.cpp:85-206)
#   'cc' : LevelCCProjector (LevelCCProjector.cpp:163-255) -- levelDivergenceCC first interpolates the velocity's CF
#          ghosts, m_velCFInterp.coarseFineInterp(u, uCrse) (Divergence.cpp:372-375; MappedQuadCFInterp with SpaceDim
#          comps = component by component), then CellToEdge etc.; the gradient as above + EdgeToCell
# The solve in between is AMRPressureSolver::solve(lmin = lmax = l) = the level solve with phi[l-1] as CF data.
# With a non-diagonal metric levelGradientMAC adds ExtrapolateCFEV after the interpolation and the gradient is
# singleBoxMacGrad's full sequence (so.level_gradient_mac with the level operator).
# ----------------------------------------------------------------------------
def level_project(comp, l, vel, phi, dt, centring="mac", velCoarse=None, zeroPhi=True, wall=True):
    op = comp.ops[l]
    L = comp.levels[l]
    nd = comp.ndim
    rhs = [None] * len(comp.levels)
    rhs[l] = so.LevelData(L.grids, 1)
    if centring == "cc":
        if l > 0:
            Lc = comp.levels[l - 1]
            for c in range(nd):
                f1 = so.LevelData(L.grids, 1, vel.ghost)
                c1 = so.LevelData(Lc.grids, 1, velCoarse.ghost)
                for a, b in zip(f1.fabs, vel.fabs):
                    a.a[..., 0] = b.a[..., c]
                for a, b in zip(c1.fabs, velCoarse.fabs):
                    a.a[..., 0] = b.a[..., c]
                op.quad.coarse_fine_interp(f1, c1)
                for a, b in zip(f1.fabs, vel.fabs):
                    b.a[..., c] = a.a[..., 0]
        so.level_divergence_cc(rhs[l], vel, L.Jinv, L.grids, L.domain, L.dx, nd, wall)
    else:
        if wall:
            # levelDivergenceMAC's a_fluxBC = m_divBC overwrites the caller's wall-normal faces (Divergence.cpp:73-100)
            so.set_wall_normal_flux(vel, L.grids, L.domain, nd)
        so.level_divergence_mac(rhs[l], vel, L.Jinv, L.grids, L.dx, nd)
    if dt != 0.0:
        for f in rhs[l].fabs:
            f.a /= dt
    comp.solve(phi, rhs, l, l, zeroPhi=zeroPhi)
    if l > 0:
        op.quad.coarse_fine_interp(phi[l], phi[l - 1])
        if not op.isDiagonal:
            op.cf.extrapolate_cf_ev(phi[l], 2, op.activeDirs)       # levelGradientMAC, Gradient.cpp:110-113
    dtScale = -1.0 if dt == 0.0 else -dt
    if centring == "cc":
        corr = so.LevelData(L.grids, nd)
        so.level_gradient_cc(corr, phi[l], L.grids, L.domain, L.Jgup, L.dx, nd, op=op)
        for i, g in enumerate(L.grids):
            vel[i].view(g)[...] += dtScale * corr[i].a
    else:
        corr = so.FluxData(L.grids, 1, nd)
        so.level_gradient_mac(corr, phi[l], L.grids, L.domain, L.Jgup, L.dx, nd, op=op)
        for i in range(len(L.grids)):
            for d in range(nd):
                vel[i][d].a += dtScale * corr[i][d].a
    return rhs[l]


# ----------------------------------------------------------------------------
# Viscous / diffusive Helmholtz steps on a level of a hierarchy (the multi-level part):
#   MappedLevelBackwardEuler::updateSoln                    AMRParabolic/MappedLevelBackwardEuler.cpp:52-158
#   MappedLevelCrankNicolson::updateSoln                    AMRParabolic/MappedLevelCrankNicolson.cpp:52-152
#   MappedLevelTGA::updateSolnWithTimeIndependentOp         AMRParabolic/MappedLevelTGA.cpp:231-387
#   applyHelm (AMROperatorNF with the time-interpolated coarse data), solveHelm (m_solver->solve(phi, rhs, l, l)),
#   incrementFlux, timeInterp, resetSolverAlphaAndBeta      AMRParabolic/MappedBaseLevelHeatSolver.cpp:154-300
# The flux-register increments after the solve use the caller's registers (AMRNavierStokes owns them) and a_flux, which
# incrementFlux fills with J Grad(phi) -- getFlux(FluxBox&, ...) applies NO beta, MappedAMRPoissonOp.cpp:2129-2151 -- of
# every stage: returned here face by face so an adapter can run incrementCoarse / incrementFine as before.
# ----------------------------------------------------------------------------
def amr_reset_alpha_beta(comp, a, b):
    """resetSolverAlphaAndBeta: every op of the solver (all levels, all depths)."""
    for mg in comp.mg:
        for op in mg.ops:
            if not hasattr(op, "aCoef"):
                op.aCoef, op.bCoef = op.alpha, op.beta
            op.alpha = a * op.aCoef
            op.beta = b * op.bCoef


def time_interp(oldData, newData, time, oldTime, newTime):
    """timeInterp, MappedBaseLevelHeatSolver.cpp:273-300"""
    out = so.ld_create(oldData)
    so.ld_set(out, 0.0)
    diff = newTime - oldTime
    if diff < 1.0e-10:
        so.ld_incr(out, oldData, 1.0)
    else:
        factor = (time - oldTime) / (newTime - oldTime)
        so.ld_incr(out, oldData, 1.0 - factor)
        so.ld_incr(out, newData, factor)
    return out


def increment_flux(comp, l, flux, phi, setToZero):
    """incrementFlux: thisFlux (+)= getFlux(phi) = J Grad(phi) on every face of every box (ghosts of phi as they are)."""
    op = comp.ops[l]
    for i, valid in enumerate(op.grids):
        phiF = phi[i]
        extrap = so.Fab(phiF.box, phi.ncomp, np.nan)
        op.fill_extrap(extrap, phiF, 2)
        for d in range(op.ndim):
            tmp = so.Fab(valid.faces(d), phi.ncomp, 0.0)
            op.get_flux_complete(tmp, phiF, extrap, valid.faces(d), i, d)
            if setToZero:
                flux[i][d].a[...] = 0.0
            flux[i][d].a[..., 0] += tmp.a[..., 0]


def amr_level_heat(comp, l, scheme, phiNew, phiOld, src, crseOld=None, crseNew=None, oldTime=0.0, crseOldTime=0.0,
                   crseNewTime=0.0, dt=0.0, zeroPhi=True, flux=None):
    """One level step on level l of the hierarchy; phi of level l-1 for the coarse-fine values is the time interpolation of
    (crseOld, crseNew).  scheme 0 backward Euler, 1 Crank-Nicolson, 2 TGA.  flux: optional so.FluxData of level l that
    receives a_flux.  Returns nothing; comp.history / iters / exitStatus are the LAST solve's."""
    op = comp.ops[l]
    nl = len(comp.levels)
    grids = comp.levels[l].grids

    def coarse_at(t):
        return time_interp(crseOld, crseNew, t, crseOldTime, crseNewTime) if l > 0 else None

    def apply_helm(ans, phi, phiC, mu, homogeneous):
        amr_reset_alpha_beta(comp, 1.0, mu * dt)          # m_ops[l]->setAlphaAndBeta(1, mu dt)
        if phiC is None or l == 0:
            op.apply_op(ans, phi, homogeneous)
        else:
            comp.amr_operator(l, ans, None, phi, phiC, homogeneous)

    def solve_helm(phi, phiC, rhs, mu):
        if zeroPhi:
            so.ld_set(phi, 0.0)
        P, R = [None] * nl, [None] * nl
        P[l], R[l] = phi, rhs
        if l > 0:
            P[l - 1] = phiC
        amr_reset_alpha_beta(comp, 1.0, -dt * mu)
        comp.solve(P, R, l, l, zeroPhi=zeroPhi)

    def incr_flux(phi, setToZero):
        if flux is not None:
            increment_flux(comp, l, flux, phi, setToZero)

    rhst = so.ld_create(src)
    so.ld_set(rhst, 0.0)
    if scheme == 0:
        phit = so.ld_create(phiNew)
        so.ld_set(phit, 0.0)
        if zeroPhi:
            so.ld_set(phiNew, 0.0)
        so.ld_incr(phit, phiOld, 1.0)
        so.ld_incr(rhst, phit, 1.0)
        coarse = coarse_at(oldTime)                       # as written: a_oldTime, not the new time (:112-113)
        solve_helm(phiNew, coarse, rhst, 1.0)
        incr_flux(phiNew, True)
    elif scheme == 1:
        phit = so.ld_create(phiNew)
        so.ld_set(phit, 0.0)
        if zeroPhi:
            so.ld_set(phiNew, 0.0)
        coarse = coarse_at(oldTime)
        apply_helm(phit, phiOld, coarse, 0.5, False)
        so.ld_incr(rhst, src, dt)
        so.ld_incr(rhst, phit, 1.0)
        solve_helm(phiNew, coarse, rhst, 0.5)
        incr_flux(phiNew, True)
    else:
        mu1, mu2, mu3, mu4, r1 = so.tga_coefficients()
        srct = so.ld_create(phiNew)
        phis = so.ld_create(phiNew)
        so.ld_set(srct, 0.0)
        so.ld_incr(srct, src, dt)
        if not zeroPhi:
            so.ld_set(phis, 0.0)
            so.ld_incr(phis, phiNew, 1.0)
        # setSourceGhostCells(srct): only ghosts, all of which applyOp refills (exchange, homogeneous CF / BC values)
        apply_helm(rhst, srct, None, mu4, True)
        incr_flux(srct, True)
        coarse = coarse_at(oldTime)
        apply_helm(phiNew, phiOld, coarse, mu3, False)
        incr_flux(phiOld, False)
        so.ld_incr(rhst, phiNew, 1.0)
        coarse = coarse_at(oldTime + (1.0 - r1) * dt)
        if not zeroPhi:
            so.ld_set(phiNew, 0.0)
            so.ld_incr(phiNew, phis, 1.0)
        solve_helm(phiNew, coarse, rhst, mu2)
        incr_flux(phiNew, False)
        for dF, f in zip(rhst.fabs, phiNew.fabs):
            dF.copy_from(f)
        coarse = coarse_at(oldTime + dt)
        if not zeroPhi:
            so.ld_set(phiNew, 0.0)
            so.ld_incr(phiNew, phis, 1.0)
        solve_helm(phiNew, coarse, rhst, mu1)
        incr_flux(phiNew, False)


# ----------------------------------------------------------------------------
# MappedAMRTGA<T>::oneStep -- the COMPOSITE TGA step over levels l_base..l_max (AMRElliptic/MappedAMRTGA.H:417-497).
#   applyHelm  (:499-523): resetAlphaAndBeta(1, mu dt); m_solver->computeAMROperator(ans, phi, l_max, l_base, homogeneous)
#   solveHelm  (:525-546): resetAlphaAndBeta(1, -mu dt); m_solver->solveNoInit(ans, rhs, l_max, l_base, zeroPhi = false)
#   computeAMROperator (MappedAMRMultiGrid.H:862-878): computeAMRResidual against a zero m_residual with
#   a_computeNorm = false (no zeroCovered), then scale(-1).
#   divideByIdentityCoef / diagonalScale: no-ops of MappedAMRPoissonOp (MappedAMRPoissonOp.cpp:814-825).
# The register scale of the refluxing inside is the coarse operator's CURRENT beta / dx (reflux(), above).
# No driver of the reference calls this class (AMRNavierStokes advances level by level, amr_level_heat above).
# ----------------------------------------------------------------------------
def compute_amr_operator(comp, lph, phi, l_max, l_base, homogeneous):
    nl = len(comp.levels)
    zero = [None] * nl
    for l in range(l_base, l_max + 1):
        zero[l] = so.ld_create(lph[l])
        so.ld_set(zero[l], 0.0)
    comp.compute_amr_residual(lph, phi, zero, l_max, l_base, homogeneous, computeNorm=False)
    for l in range(l_base, l_max + 1):
        so.ld_scale(lph[l], -1.0)


def amr_tga_one_step(comp, phiNew, phiOld, source, dt, l_base, l_max):
    """phiNew / phiOld / source: per-level lists.  comp.history / iters / exitStatus are the LAST solve's.
    l_base must be 0: createData allocates m_srct for l_base..l_max only (:388-403) and computeAMROperator on it reads
    *m_srct[l_base - 1] for level l_base's coarse-fine values (MappedAMRMultiGrid.H:907-909) -- a null pointer."""
    assert l_base == 0, "MappedAMRTGA::oneStep with l_base > 0 dereferences a null m_srct[l_base - 1]"
    mu1, mu2, mu3, mu4, _ = so.tga_coefficients()
    nl = len(comp.levels)
    rng = range(l_base, l_max + 1)
    rhst, srct = [None] * nl, [None] * nl
    for l in rng:
        rhst[l] = so.ld_create(source[l])
        srct[l] = so.ld_create(phiNew[l])
        so.ld_set(srct[l], 0.0)
        so.ld_incr(srct[l], source[l], 1.0)

    def apply_helm(ans, phi, mu, homogeneous):
        amr_reset_alpha_beta(comp, 1.0, mu * dt)
        compute_amr_operator(comp, ans, phi, l_max, l_base, homogeneous)

    def solve_helm(ans, rhs, mu):
        amr_reset_alpha_beta(comp, 1.0, -mu * dt)
        comp.solve(ans, rhs, l_max, l_base, zeroPhi=False)

    apply_helm(rhst, srct, mu4, True)
    for l in rng:
        so.ld_scale(rhst[l], dt)
    apply_helm(phiNew, phiOld, mu3, False)
    for l in rng:
        so.ld_incr(rhst[l], phiNew[l], 1.0)
    for l in rng:
        so.ld_assign(phiNew[l], phiOld[l])
    solve_helm(phiNew, rhst, mu2)
    for l in rng:
        so.ld_assign(rhst[l], phiNew[l])
    for l in rng:
        so.ld_assign(phiNew[l], phiOld[l])
    solve_helm(phiNew, rhst, mu1)
