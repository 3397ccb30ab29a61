"""oracle/somar_leptic.py -- TEST INFRASTRUCTURE: CPU restatement of the reference's leptic level solver.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this module.

  LevelLepticSolver::define / solve      calculus/LepticSolver/LevelLepticSolver.cpp:147-437, 646-956
  computeHorizRHS                        :981-1097
  levelVertHorizGradient                 :1107-1176      (diagonal metric: boundary data := 0; else LEPTICVERTHORIZGRAD,
                                                          LevelLepticSolverF.ChF:59-99, after ExtrapolateFaceAndCopy in z)
  computeVerticalExcess                  :1183-1240
  verticalLineSolver                     :1248-1421      (Neumann-Neumann columns: TriDiagPoissonNN1DFAB)
  horizontalSolver, add*Correction       :1427-1517
  setZeroAvg                             :1668-1710
  UNMAPPEDVERTINTEGRAL, ADDEXTRUSION     utils/SubspaceF.ChF:33-110
  LEPTICACCUMDIV                         calculus/LepticSolver/LevelLepticSolverF.ChF:285-331
  createVertAvgFCJgupPtr / CCJinvPtr     geometry/LevelGeometryBasics.cpp:409-433, 500-569

Scope (the same as the HIP path, csrc/leptic.cpp): one level whose boxes are vertically complete columns
(the layout LepticBoxUtils::createVerticalSolverGrids produces), so the reference's orig <-> vertical and
flat <-> horizontal re-layouts are identities; diagonal OR non-diagonal metric (the latter: MAPPEDMACGRAD with its cross
terms in computeHorizRHS after extrapAllGhosts + exchanges, the vertical boundary data of levelVertHorizGradient, the
vertically averaged horizontal block J g^{ab}, a, b < 2, as a 9-point flat problem solved at EVERY order); Neumann (or any non-periodic) physical
boundaries in the vertical and non-periodic horizontal directions (the reference never fills the face gradient on
a periodic horizontal boundary, LevelLepticSolver.cpp:997-1001 + :1056-1061); coarse-fine boundaries on the LATERAL sides
of the columns only (homogeneousCFInterp / ExtrapolateCFEV in computeHorizRHS, :1003-1012; the J-scaled operator, the full
multigrid and the flat problem carry the level's CFRegion and dxCrse, :216-266, :381) -- with the reference's (2,2,1)
refinements a fine level's columns always span the domain, so the CF vertical ends of LepticLapackVerticalSolver never occur.

  AMRLepticSolver (AMRLepticSolver.cpp:68-196 define/init, :200-424 solve*, :430-529 AMRVCycle, :535-672 residuals):
  the composite V-cycle of MappedAMRMultiGrid with LevelLepticSolver::solve in place of relax, restated as written --
  including that the base level solves for a_uberCorrection from a_uberResidual (not from the restricted m_residual) and
  that m_correction[l_base] stays zero, so nothing is prolonged from the base level (:444-449, :486-487).
Parity unpinned w.r.t. reference tests: the reference ships none; pinned by the
analytic known answers in tests/test_oracle_leptic.py.
"""
import ctypes as C

import numpy as np

from . import somar_oracle as so
from . import somar_amr as sa

EXIT_NONE, EXIT_CONVERGE, EXIT_ITER, EXIT_HANG, EXIT_DIVERGE, EXIT_KABOOM = -1, 0, 1, 2, 3, 4


def flatten_box(b, pos):
    """LepticBoxUtils::FlattenTransform, LepticBoxUtils.cpp:39-49"""
    return so.Box((b.lo[0], b.lo[1], pos), (b.hi[0], b.hi[1], pos))


def vert_avg_metric(grids, Jgup, domBox, isDiagonal=True, which=None):
    """The horizontal problem's metric: vertical average of the horizontal block J g^{ab}, a, b < 2 (diagonal metric:
    only a == b is non-zero), J^{-1} := 1.  createVertAvgFCJgupPtr (geometry/LevelGeometryBasics.cpp:500-569):
    UNMAPPEDVERTINTEGRAL, levelSum += func * (1/Nz), k ascending."""
    pos = domBox.lo[2]
    which = list(range(len(grids))) if which is None else list(which)   # the grids that get a flat twin (m_flatDI)
    flat = [flatten_box(grids[i], pos) for i in which]
    hJg = so.FluxData(flat, 2, ndim=2)
    for h, i in enumerate(which):
        g = grids[i]
        scale = 1.0 / float(g.size()[2])
        for d in range(2):
            for b in range(2):
                if isDiagonal and b != d:
                    continue
                src = Jgup[i][d].a[..., b]
                acc = np.zeros(src.shape[:2])
                for k in range(src.shape[2]):
                    acc = acc + src[:, :, k] * scale
                hJg[h][d].a[:, :, 0, b] = acc
    hJinv = so.LevelData(flat, 1, (0, 0, 0), fill=1.0)
    return flat, hJg, hJinv


VBC_NEUM, VBC_DIRI, VBC_CF = 0, 1, 2   # BCType::Neum / Diri / CF as verticalLineSolver sees them


def dptsv(D, E, B):
    """LAPACK dptsv (EXTERNAL, declared utils/lapack.H, called LevelLepticSolverF.ChF:262): dpttrf's L D L^T factorisation
    (unrolled by four there, the same operations in the same order) followed by dptts2, for many systems at once
    (last axis = the unknowns).  In place: D, E hold the factors, B the solution.  Raises if a pivot is not positive
    (INFO != 0; the reference tolerates INFO == N only)."""
    n = D.shape[-1]
    for i in range(n - 1):
        assert np.all(D[..., i] > 0.0), "dptsv: INFO = %d" % (i + 1)
        ei = E[..., i].copy()
        E[..., i] = ei / D[..., i]
        D[..., i + 1] = D[..., i + 1] - E[..., i] * ei
    assert np.all(D[..., n - 1] > 0.0), "dptsv: INFO = N"
    for i in range(1, n):
        B[..., i] = B[..., i] - B[..., i - 1] * E[..., i - 1]
    B[..., n - 1] = B[..., n - 1] / D[..., n - 1]
    for i in range(n - 2, -1, -1):
        B[..., i] = B[..., i] / D[..., i] - B[..., i + 1] * E[..., i]


class LevelLepticSolver:
    def __init__(self, origOp, maxOrder=4, hang=1e-15, normType=0, horizRhsTol=1e-14, domainHeight=None,
                 horiz=None, horizBottom=None, full=None, fullBottom=None):
        # setDefaultParameters, LevelLepticSolver.cpp:461-508
        hz = dict(imin=5, imax=20, pre=4, bottom=4, post=4, precond=2, relaxMode=so.RELAX_LEVEL_GSRB, maxDepth=-1,
                  eps=1e-12, hang=1e-15, normThresh=1e-30)
        hzb = dict(imax=80, eps=1e-12, numRestarts=5, hang=1e-15)
        fl = dict(imin=5, imax=20, pre=4, bottom=4, post=4, precond=4, relaxMode=so.RELAX_LINE_GSRB, maxDepth=-1,
                  eps=1e-6, hang=1e-15, normThresh=1e-30)
        flb = dict(imax=80, numRestarts=5)
        hz.update(horiz or {}); hzb.update(horizBottom or {}); fl.update(full or {}); flb.update(fullBottom or {})
        assert normType == 0, "only the max norm (the reference's default) is restated"
        self.maxOrder, self.hang, self.normType, self.horizRhsTol = maxOrder, hang, normType, horizRhsTol
        op = self.origOp = origOp
        assert op.ndim == 3
        self.isDiagonal = op.isDiagonal
        # m_CFRegion.define(m_grids, m_domain), m_dxCrse = lepticOpPtr->getDxCrse()                 :164, :216-217
        self.cf, self.dxCrse = op.cf, op.dxCrse
        dom, grids, dx = op.domain, op.grids, op.dx
        self.domain, self.grids, self.dx = dom, grids, dx
        domBox = dom.box
        assert not any(dom.periodic), "periodic directions are not supported by the reference's leptic path"
        # gatherVerticalBCTypes                                                                     :1523-1640
        self.vertBCTypes = []
        for i, g in enumerate(grids):
            t = []
            for side in (0, 1):
                atDom = (g.lo[2] == domBox.lo[2]) if side == 0 else (g.hi[2] == domBox.hi[2])
                if atDom:
                    t.append(VBC_NEUM if op.bc.types[2][side] == so.BC_NEUM else VBC_DIRI)
                else:
                    gb, m = (self.cf.ivs[(i, 2, side)] if self.cf is not None else (None, None))
                    assert m is not None and bool(m.all()), "Vertical grids are ill-formed"
                    t.append(VBC_CF)
            self.vertBCTypes.append(tuple(t))
        self.doHorizSolve = any(t == (VBC_NEUM, VBC_NEUM) for t in self.vertBCTypes)
        # m_flatDI: the grids that span the domain vertically (vertSpanCheck, :318-333); with Neumann walls these are the
        # Neumann-Neumann columns.  m_flatDIComplement: the rest -- they take no part in the horizontal problem.
        self.flatDI = [i for i, g in enumerate(grids) if g.lo[2] == domBox.lo[2] and g.hi[2] == domBox.hi[2]]
        if self.doHorizSolve:
            assert all((self.vertBCTypes[i] == (VBC_NEUM, VBC_NEUM)) == (i in self.flatDI) for i in range(len(grids))), \
                "a column that spans the domain without being Neumann-Neumann next to Neumann-Neumann ones is not restated"
            if len(self.flatDI) != len(grids):
                # levelVertHorizGradient hands the non-spanning columns' Neumann ends boundary data as well (non-zero only
                # with cross terms): restated for the diagonal metric, where it vanishes
                assert op.isDiagonal, "mixed column kinds are restated for diagonal metrics"
        self.H = dx[2] * domBox.size()[2] if domainHeight is None else domainHeight
        # vertical grids == original grids; Jgup is shared, Jinv := 1 (the residual equation is scaled by J)
        self.Jgup = op.Jgup
        self.Jinv1 = so.LevelData(grids, 1, (0, 0, 0), fill=1.0)
        # full 3-D MG solver (also supplies m_opPtr: alpha 0, beta 1)            :254-300
        fac = so.Factory(dom, grids, dx, op.bc, self.Jgup, self.Jinv1, alpha=0.0, beta=1.0, isDiagonal=op.isDiagonal, ndim=3,
                         maxDepth=fl["maxDepth"], precondIters=fl["precond"], relaxMode=fl["relaxMode"],
                         dxCrse=self.dxCrse, cf=self.cf)
        bot = so.BiCGStab(imax=flb["imax"], numRestarts=flb["numRestarts"], normType=normType, hang=1e-8)
        self.mgSolver = so.AMRMultiGrid(fac, bot, fl["maxDepth"])
        self.mgSolver.imin = fl["imin"]
        self.mgSolver.set_solver_parameters(fl["pre"], fl["post"], fl["bottom"], 1, fl["imax"], fl["eps"], fl["hang"],
                                            fl["normThresh"])
        self.op = self.mgSolver.op
        self.exitStatus = EXIT_NONE
        self.resNorms = []
        self.usedFullSolver = False
        self.horizSolves = 0
        self.flatGrids = [flatten_box(g, g.lo[2]) for g in grids]
        if not self.doHorizSolve:
            return
        # horizontal structures                                                   :304-432
        # createHorizontalSolverGrids: the flattened spanning boxes (LepticBoxUtils.cpp:100-117)
        self.horizGrids, hJg, hJinv = vert_avg_metric(grids, self.Jgup, domBox, op.isDiagonal, self.flatDI)
        self.horizDomain = so.Domain(flatten_box(domBox, domBox.lo[2]), dom.periodic)
        hcf = sa.CFRegion(self.horizGrids, self.horizDomain) if self.cf is not None else None
        hfac = so.Factory(self.horizDomain, self.horizGrids, dx, so.BCHolder(), hJg, hJinv, alpha=0.0, beta=1.0,
                          isDiagonal=op.isDiagonal, ndim=2, maxDepth=hz["maxDepth"], precondIters=hz["precond"],
                          relaxMode=hz["relaxMode"], dxCrse=self.dxCrse, cf=hcf)   # forceDxCrse(m_dxCrse), :381
        hbot = so.BiCGStab(imax=hzb["imax"], eps=hzb["eps"], numRestarts=hzb["numRestarts"], hang=hzb["hang"],
                           normType=normType)
        self.horizSolver = so.AMRMultiGrid(hfac, hbot, hz["maxDepth"])
        self.horizSolver.imin = hz["imin"]
        self.horizSolver.set_solver_parameters(hz["pre"], hz["post"], hz["bottom"], 1, hz["imax"], hz["eps"],
                                               hz["hang"], hz["normThresh"])
        npts = sum(g.numPts() for g in self.horizGrids)
        self.horizRemoveAvg = npts == self.horizDomain.box.numPts()
        self.exitStatus = EXIT_NONE
        self.resNorms = []
        self.usedFullSolver = False
        self.horizSolves = 0

    # -- pieces ---------------------------------------------------------------------------------------
    def compute_vertical_excess(self, excess, rhs, bcLo, bcHi):
        """excess = hiNeumBC - loNeumBC - Integral[rhs]                                :1183-1240"""
        dzScale = -1.0 * self.dx[2]
        for i, g in enumerate(self.grids):
            if i not in self.flatDI:
                excess[i][...] = 0.0                  # setToZero(a_excess, m_flatDIComplement), :1203
                continue
            e = bcHi[i].copy()
            e = e + (-1.0) * bcLo[i]
            r = rhs[i].view(g)[..., 0]
            for k in range(r.shape[2]):
                e = e + r[:, :, k] * dzScale
            excess[i][...] = e

    def lapack_vertical_solver(self, i, vertPhi, vertRhs):
        """LepticLapackVerticalSolver (LevelLepticSolverF.ChF:161-283) on box i: the symmetric tridiagonal system of a column
        whose ends are Neumann / Dirichlet / coarse-fine (linear interpolation), solved by LAPACK dptsv (EXTERNAL; restated
        from the published dpttrf + dptts2 loops, pinned against SciPy's dptsv in tests/test_oracle_leptic.py).  DU is
        assembled by the reference but not handed to dptsv.  All columns of the box at once, k sequential."""
        g = self.grids[i]
        dz, dzCrse = self.dx[2], (self.dxCrse[2] if self.dxCrse is not None else 0.0)
        lo, hi = self.vertBCTypes[i]
        kmax = g.size()[2]
        invdzsq = 1.0 / (dz * dz)
        Jf = self.Jgup[i][2]
        fb = g.faces(2)
        Jgzz = Jf.view(fb)[..., 2]                       # Jgzz(IDX(k)), k = 0..kmax: face k is the low face of cell k
        rhs = vertRhs[i].view(g)[..., 0]
        D = np.empty(rhs.shape)
        DL = np.empty(rhs.shape[:2] + (kmax - 1,))
        B = -rhs
        for k in range(1, kmax + 1):
            D[:, :, k - 1] = (Jgzz[:, :, k - 1] + Jgzz[:, :, k]) * invdzsq
            if k < kmax:
                DL[:, :, k - 1] = -Jgzz[:, :, k] * invdzsq
        alpha = 1.0 - 2.0 * dz / (dzCrse + dz)
        if lo == VBC_NEUM:
            D[:, :, 0] = Jgzz[:, :, 1] * invdzsq
        elif lo == VBC_DIRI:
            D[:, :, 0] = (2.0 * Jgzz[:, :, 0] + Jgzz[:, :, 1]) * invdzsq
        else:
            D[:, :, 0] = ((1.0 - alpha) * Jgzz[:, :, 0] + Jgzz[:, :, 1]) * invdzsq
        if hi == VBC_NEUM:
            D[:, :, kmax - 1] = Jgzz[:, :, kmax - 1] * invdzsq
        elif hi == VBC_DIRI:
            D[:, :, kmax - 1] = (Jgzz[:, :, kmax - 1] + 2.0 * Jgzz[:, :, kmax]) * invdzsq
        else:
            D[:, :, kmax - 1] = (Jgzz[:, :, kmax - 1] + (1.0 - alpha) * Jgzz[:, :, kmax]) * invdzsq
        dptsv(D, DL, B)
        vertPhi[i].view(g)[..., 0] = B

    def vertical_line_solver(self, vertPhi, vertRhs, bcLo, bcHi):
        """verticalLineSolver                                                          :1248-1421"""
        dz = self.dx[2]
        L = so.lib()
        if not self.doHorizSolve:
            # no column is Neumann-Neumann; a Neumann end rolls in the boundary data, which stays zero without horizontal
            # solves (bdryData.setVal(0.0), never touched again): rhs + 0 is rhs
            for i in range(len(self.grids)):
                self.lapack_vertical_solver(i, vertPhi, vertRhs)
            return
        for i, g in enumerate(self.grids):
            if self.vertBCTypes[i] != (VBC_NEUM, VBC_NEUM):
                # a column of the complement (mixed layout): its boundary data are zero for a diagonal metric
                self.lapack_vertical_solver(i, vertPhi, vertRhs)
                continue
            Nz = g.size()[2]
            r = vertRhs[i].view(g)[..., 0]
            # roll the BC values in: rhs -/+ NeumBCVal/dz (rollInFAB := 0; plus(bc, scale); rhs.plus(rollIn, 1))
            rollLo = 0.0 + bcLo[i] * (1.0 / dz)     # scale = -isign/dz, isign = -1
            rollHi = 0.0 + bcHi[i] * (-1.0 / dz)
            r[:, :, 0] = r[:, :, 0] + rollLo * 1.0
            r[:, :, Nz - 1] = r[:, :, Nz - 1] + rollHi * 1.0
            bottom = so.Box(g.lo, (g.hi[0], g.hi[1], g.lo[2]))
            blo, bhi = so._b(bottom)
            JgzF = self.Jgup[i][2]
            L.orc_tridiagpoissonnn1dfab(*vertPhi[i].fra1(0), *vertRhs[i].fra1(0), *JgzF.fra1(2), blo, bhi, Nz,
                                        C.c_double(dz), 2)
            # unroll (rhs.plus(rollIn, -1)): NOT bit-identical to the rhs before, as in the reference
            r[:, :, 0] = r[:, :, 0] + rollLo * (-1.0)
            r[:, :, Nz - 1] = r[:, :, Nz - 1] + rollHi * (-1.0)

    def compute_horiz_rhs(self, flatRhs, phi):
        """-d_m bar(Jg^{mm} d_m phi), diagonal metric; boundary faces carry the (zero) boundary data   :981-1097"""
        dom = self.domain
        if not self.isDiagonal:
            self.compute_horiz_rhs_full(flatRhs, phi)
            return
        # extrapAllGhosts(phi, 2) only matters on faces whose gradient is then overwritten by boundary data or
        # which the exchange refills; the exchange supplies the neighbour values
        if self.cf is not None:
            self.cf.homogeneous_cf_interp(phi, self.dx, self.dxCrse, (1, 1, 1))   # :1005
        so.exchange(phi, dom, phi.ghost)
        for i, g in enumerate(self.grids):
            if i not in self.flatDI:
                flatRhs[i][...] = 0.0                 # setValLevel(a_rhs, 0.0); only m_flatDI is filled (:1019-1021)
                continue
            Nz = g.size()[2]
            dzScale = 1.0 / float(Nz)
            acc_rhs = np.zeros(g.size()[:2])
            pF = phi[i]
            for d in range(2):
                fb = g.faces(d)
                Jg = self.Jgup[i][d].view(fb, d)
                e = [1 if q == d else 0 for q in range(3)]
                hi_cells = so.Box(fb.lo, fb.hi)
                lo_cells = hi_cells.shift([-x for x in e])
                dxinv = 1.0 / self.dx[d]
                grad = dxinv * Jg * (pF.view(hi_cells)[..., 0] - pF.view(lo_cells)[..., 0])
                # faces on the physical boundary: gradPhiFAB.copy(bcFAB), zero for every order
                if g.lo[d] == dom.box.lo[d]:
                    sl = [slice(None)] * 3
                    sl[d] = 0
                    grad[tuple(sl)] = 0.0
                if g.hi[d] == dom.box.hi[d]:
                    sl = [slice(None)] * 3
                    sl[d] = grad.shape[d] - 1
                    grad[tuple(sl)] = 0.0
                avg = np.zeros(grad.shape[:2])
                for k in range(Nz):
                    avg = avg + grad[:, :, k] * dzScale
                dxScale = -1.0 / self.dx[d]
                if d == 0:
                    acc_rhs = acc_rhs + (avg[1:, :] - avg[:-1, :]) * dxScale
                else:
                    acc_rhs = acc_rhs + (avg[:, 1:] - avg[:, :-1]) * dxScale
            flatRhs[i][...] = acc_rhs

    def compute_horiz_rhs_full(self, flatRhs, phi):
        """computeHorizRHS with a non-diagonal metric (:981-1097): extrapAllGhosts(phi, 2) (every ghost layer of every box
        by order-2 extrapolation from the BOX's valid cells, x sides, then y with the box grown in x, then z;
        ExtrapolationUtils.cpp:388-420), exchange (+ corner exchange), MAPPEDMACGRAD with phi as its own extrap on the
        faces inside the domain, zero on the domain's side faces (the boundary data), vertical average, -divergence."""
        dom = self.domain
        for i, g in enumerate(self.grids):
            valid = g
            for d in range(3):
                for side in (0, 1):
                    so.extrapolate_face_no_ev(phi[i], phi[i], valid, d, side, 2)
                valid = valid.growDir(d, 1)
        if self.cf is not None:
            self.cf.homogeneous_cf_interp(phi, self.dx, self.dxCrse, (1, 1, 1))   # :1005
        so.exchange(phi, dom, phi.ghost)
        if self.cf is not None:
            self.cf.extrapolate_cf_ev(phi, 2, (1, 1, 1))                          # :1008-1013
            so.exchange(phi, dom, phi.ghost)
        for i, g in enumerate(self.grids):
            Nz = g.size()[2]
            dzScale = 1.0 / float(Nz)
            acc_rhs = np.zeros(g.size()[:2])
            pF = phi[i]
            for d in range(2):
                fb = g.faces(d)
                # interior faces: faceBox & grow(surroundingNodes(domBox, d), d, -1)
                lo, hi = list(fb.lo), list(fb.hi)
                lo[d] = max(lo[d], dom.box.lo[d] + 1)
                hi[d] = min(hi[d], dom.box.hi[d])
                grad = so.Fab(fb, 1, 0.0)    # boundary faces: gradPhiFAB.copy(bcFAB) = 0
                inner = so.Box(lo, hi)
                if not inner.isEmpty():
                    blo, bhi = so._b(inner)
                    so.lib().orc_mappedgetflux(*grad.fra(), *pF.fran(), *pF.fran(), *self.Jgup[i][d].fran(), blo, bhi,
                                               C.c_double(1.0), so._rv(self.dx), d)
                gv = grad.a[..., 0]
                avg = np.zeros(gv.shape[:2])
                for k in range(Nz):
                    avg = avg + gv[:, :, k] * dzScale
                dxScale = -1.0 / self.dx[d]
                if d == 0:
                    acc_rhs = acc_rhs + (avg[1:, :] - avg[:-1, :]) * dxScale
                else:
                    acc_rhs = acc_rhs + (avg[:, 1:] - avg[:, :-1]) * dxScale
            flatRhs[i][...] = acc_rhs

    def level_vert_horiz_gradient(self, bcLo, bcHi, phi, scale):
        """levelVertHorizGradient (:1107-1176), non-diagonal metric: per box and vertical side, ExtrapolateFaceAndCopy of phi
        in z (order 2, from the box's FAB clipped to the domain, in place), then LEPTICVERTHORIZGRAD on the boundary face."""
        dom = self.domain
        for side, bcs in ((0, bcLo), (1, bcHi)):
            isign = 1 if side else -1
            for i, g in enumerate(self.grids):
                pF = phi[i]
                domValid = pF.box & dom.box
                so.extrapolate_face_and_copy(pF, pF, domValid, 2, side, 2)
                kface = g.lo[2] if side == 0 else g.hi[2] + 1
                fb = so.Box((g.lo[0], g.lo[1], kface), (g.hi[0], g.hi[1], kface))
                out = np.zeros(g.size()[:2], order="F")
                JgzF = self.Jgup[i][2]
                flo, fhi = so._b(fb)
                so.lib().orc_lepticverthorizgrad(out.ctypes.data_as(C.POINTER(C.c_double)), flo, fhi, *pF.fra1(0),
                                                 *JgzF.fran(), flo, fhi, isign, so._rv(self.dx), C.c_double(scale))
                bcs[i][...] = out

    @staticmethod
    def set_zero_avg(phi):
        tot, vol = 0.0, 0
        for g, f in zip(phi.grids, phi.fabs):
            tot += so._seqsum(f.view(g))
            vol += g.numPts()
        avg = tot / float(vol)
        for f in phi.fabs:
            f.a[...] -= avg

    # -- solve ----------------------------------------------------------------------------------------
    def solve(self, a_phi, a_rhs, homogeneous=False):
        """LevelLepticSolver::solve                                                     :646-956"""
        grids, op = self.grids, self.op
        maxOrder, H = self.maxOrder, self.H
        phiTotal = so.LevelData(grids, 1, (1, 1, 1))
        vertPhi = so.LevelData(grids, 1, (1, 1, 1))
        rhs = so.LevelData(grids, 1, (0, 0, 0))
        tmpRhs = so.LevelData(grids, 1, (0, 0, 0))
        flat2 = [g.size()[:2] for g in grids]
        excess = [np.zeros(s) for s in flat2]
        flatRhs = [np.zeros(s) for s in flat2]
        bcLo = [np.zeros(s) for s in flat2]
        bcHi = [np.zeros(s) for s in flat2]
        horizPhi = so.LevelData(self.horizGrids, 1, (1, 1, 0)) if self.doHorizSolve else None
        horizRhs = so.LevelData(self.horizGrids, 1, (0, 0, 0)) if self.doHorizSolve else None
        useExcess = useHorizPhi = self.doHorizSolve

        # J * residual
        res = so.LevelData(grids, 1, (0, 0, 0))
        self.origOp.residual(res, a_phi, a_rhs, homogeneous)
        for i, g in enumerate(grids):
            rhs[i].view(g)[...] = res[i].view(g) / self.origOp.Jinv[i].view(g)

        self.resNorms = [so.ld_norm(rhs, self.normType)]
        resNorm = self.resNorms[0]
        so.ld_set(phiTotal, 0.0)
        self.usedFullSolver = False
        self.horizSolves = 0

        for order in range(maxOrder + 1):
            if self.doHorizSolve:
                if order >= 1:   # levelVertHorizGradient: zero for a diagonal metric
                    for a in bcLo + bcHi:
                        a[...] = 0.0
                    if not self.isDiagonal:
                        self.level_vert_horiz_gradient(bcLo, bcHi, vertPhi, -1.0)
                if order >= 1 and useExcess:
                    for i in range(len(grids)):
                        bcHi[i][...] = bcHi[i] + excess[i] * 1.0
                if useExcess:
                    self.compute_vertical_excess(excess, rhs, bcLo, bcHi)
                    if order == 1:
                        useExcess = False
                if order == 0 and useExcess:
                    for i in range(len(grids)):
                        bcHi[i][...] = bcHi[i] + excess[i] * (-1.0)

            self.vertical_line_solver(vertPhi, rhs, bcLo, bcHi)

            if useHorizPhi:
                self.compute_horiz_rhs(flatRhs, vertPhi)
                if useExcess:
                    for i in self.flatDI:
                        flatRhs[i][...] = flatRhs[i] + excess[i] * (-1.0 / H)
                so.ld_set(horizRhs, 0.0)
                for h, i in enumerate(self.flatDI):      # flatRhs.addTo(horizRhs) through m_flatToHorizCopier
                    horizRhs[h].view(self.horizGrids[h])[:, :, 0, 0] += flatRhs[i]
                horizRhsNorm = so.ld_norm(horizRhs, self.normType)
                if self.horizRhsTol * self.resNorms[0] > horizRhsNorm:
                    useHorizPhi = False

            if useHorizPhi:
                self.horizSolver.solve(horizPhi, horizRhs, zeroPhi=True, forceHomogeneous=True)
                if self.horizRemoveAvg:
                    self.set_zero_avg(horizPhi)
                self.horizSolves += 1
                for h, i in enumerate(self.flatDI):   # ADDEXTRUSION on the spanning grids (addHorizontalCorrection, :1504)
                    g = grids[i]
                    vertPhi[i].view(g)[..., 0] += horizPhi[h].view(self.horizGrids[h])[:, :, 0, 0][:, :, None]

            op.residual(tmpRhs, vertPhi, rhs, True)
            resNorm = so.ld_norm(tmpRhs, self.normType)
            relResNorm = resNorm / self.resNorms[0]
            prevRelResNorm = self.resNorms[-1] / self.resNorms[0]
            redu = prevRelResNorm - relResNorm
            if redu <= self.hang and order == maxOrder:
                self.mgSolver.solve(vertPhi, rhs, zeroPhi=True, forceHomogeneous=True)
                op.residual(tmpRhs, vertPhi, rhs, True)
                resNorm = so.ld_norm(tmpRhs, self.normType)
                relResNorm = resNorm / self.resNorms[0]
                self.usedFullSolver = True
            rhs, tmpRhs = tmpRhs, rhs
            self.resNorms.append(resNorm)

            redu = prevRelResNorm - relResNorm
            if redu > self.hang or order < maxOrder:
                for i, g in enumerate(grids):
                    phiTotal[i].view(g)[...] += vertPhi[i].view(g)
                self.exitStatus = EXIT_CONVERGE if order < maxOrder - 1 else EXIT_ITER
            elif -redu > self.hang:
                self.exitStatus = EXIT_KABOOM if order == 0 else EXIT_DIVERGE
                break
            else:
                self.exitStatus = EXIT_KABOOM if order == 0 else EXIT_HANG
                break
            if self.isDiagonal:
                useHorizPhi = False   # LevelGeometry::isDiagonal()

        self.last = {"vertPhi": vertPhi, "horizPhi": horizPhi, "horizRhs": horizRhs, "phiTotal": phiTotal}
        if self.exitStatus != EXIT_KABOOM:
            for i, g in enumerate(grids):
                a_phi[i].view(g)[...] += phiTotal[i].view(g) * 1.0
        return self.exitStatus



class AMRLepticSolver(sa.AMRComposite):
    """AMRLepticSolver (calculus/LepticSolver/AMRLepticSolver.cpp): the level operators, CF interpolators and flux
    registers are MappedAMRMultiGrid's (AMRnewOp per level, :68-110); every level owns a LevelLepticSolver defined
    homogeneous with no coarse phi (:185-195)."""

    def __init__(self, levels, refRatios, bc, leptic=None, baseFromRestricted=False, **kw):
        """baseFromRestricted=True is NOT the reference: the base level then solves m_correction from m_residual, as
        MappedAMRMultiGrid's V-cycle does (MappedAMRMultiGrid.H:1517-1521); it exists to show that the divergence of the
        reference's multi-level leptic V-cycle comes from that one branch and not from the restated pieces."""
        super().__init__(levels, refRatios, bc, sa.NoOpSolver(), **kw)
        self.baseFromRestricted = baseFromRestricted
        self.lepticParams = dict(leptic or {})
        self.eps, self.hang, self.normThresh, self.imin, self.iterMax, self.numMG = 1e-6, 1e-15, 1e-30, 5, 20, 1   # :30-42
        self.leptic = [None] * len(levels)

    def set_solver_parameters(self, numMG, iterMax, eps, hang, normThresh):   # :55-66
        self.numMG, self.iterMax, self.eps, self.hang, self.normThresh = numMG, iterMax, eps, hang, normThresh

    def init(self, phi, rhs, l_max, l_base):
        """init, :163-196"""
        n = len(self.levels)
        self.m_correction, self.m_residual, self.m_resC = [None] * n, [None] * n, [None] * n
        for i in range(l_base, l_max + 1):
            self.m_correction[i] = so.ld_create(phi[i])
            self.m_residual[i] = so.ld_create(rhs[i])
            if i != l_base:
                r = self.ops[i].refToCoarser
                self.m_resC[i] = so.LevelData([g.coarsen(r) for g in self.ops[i].grids], rhs[i].ncomp, rhs[i].ghost)
        self.leptic = [None] * n
        for l in range(l_base, l_max + 1):
            self.leptic[l] = LevelLepticSolver(self.ops[l], **self.lepticParams)

    def amr_vcycle(self, uberCorrection, uberResidual, ilev, l_max, l_base):
        """AMRVCycle, :430-529"""
        if ilev == l_max:
            for l in range(l_base, l_max + 1):
                so.ld_assign(self.m_residual[l], uberResidual[l])
                so.ld_set(self.m_correction[l], 0.0)
        if l_max == l_base:
            self.leptic[l_base].solve(uberCorrection[ilev], uberResidual[ilev], True)
        elif ilev == l_base:
            if self.baseFromRestricted:
                self.leptic[l_base].solve(self.m_correction[ilev], self.m_residual[ilev], True)
            else:
                self.leptic[l_base].solve(uberCorrection[ilev], uberResidual[ilev], True)   # as written, :444-449
            so.ld_incr(uberCorrection[ilev], self.m_correction[ilev], 1.0)
        else:
            self.leptic[ilev].solve(self.m_correction[ilev], self.m_residual[ilev], True)
            so.ld_incr(uberCorrection[ilev], self.m_correction[ilev], 1.0)
            so.ld_set(self.m_correction[ilev - 1], 0.0)
            self.compute_amr_residual_level(self.m_residual, uberCorrection, uberResidual, l_max, l_base, ilev - 1, True)
            self.amr_restrict_s(ilev, self.m_resC[ilev], self.m_residual[ilev], self.m_correction[ilev],
                                self.m_correction[ilev - 1], uberCorrection[ilev])
            sa.copy_valid(self.m_residual[ilev - 1], self.m_resC[ilev])
            for _ in range(self.numMG):
                self.amr_vcycle(uberCorrection, uberResidual, ilev - 1, l_max, l_base)
            self.amr_prolong_s(ilev, self.m_correction[ilev], self.m_correction[ilev - 1])
            self.amr_update_residual(ilev, self.m_residual[ilev], self.m_correction[ilev], self.m_correction[ilev - 1])
            dCorr = uberCorrection[ilev]
            so.ld_set(dCorr, 0.0)
            self.leptic[ilev].solve(dCorr, self.m_residual[ilev], True)
            so.ld_incr(self.m_correction[ilev], dCorr, 1.0)
            so.ld_assign(uberCorrection[ilev], self.m_correction[ilev])

    def solve(self, phi, rhs, l_max, l_base, zeroPhi=True, forceHomogeneous=False):
        """solve -> init + solveNoInit + solveNoInitResid, :200-424"""
        self.init(phi, rhs, l_max, l_base)
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
        rnorm, norm_last, best_rnorm = initial_rnorm, 2 * initial_rnorm, initial_rnorm
        useBestPhi = False
        it = 0
        self.history = [rnorm]
        goNorm = rnorm > self.normThresh
        goRedu = rnorm > self.eps * initial_rnorm
        goIter = it < self.iterMax
        goHang = it < self.imin or rnorm < (1 - self.hang) * norm_last
        while goIter and goRedu and goHang and goNorm:
            norm_last = rnorm
            self.amr_vcycle(uberCorrection, uberResidual, l_max, l_max, l_base)
            for l in range(l_base, l_max + 1):   # postVCycleOps, :535-563
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
            raise RuntimeError("kaboom")   # :377-382; the "blew up" case below it only prints (:384-388)
        self.exitStatus = int(not goRedu) + int(not goIter) * 2 + int(not goHang) * 4 + int(not goNorm) * 8
        self.iters, self.final_rnorm, self.initial_rnorm = it, rnorm, initial_rnorm
        return rnorm
