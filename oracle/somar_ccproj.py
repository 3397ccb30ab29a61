"""TEST INFRASTRUCTURE, NOT PRODUCT CODE: CPU restatement of SOMAR's COMPOSITE cell-centred projector -- the sync /
initialisation / post-regrid projection (NavierStokes/AMRNavierStokesSync.cpp:280-295), paths relative to
/root/reference/src:

  AMRCCProjector::computeDiv / computeGrad / applyCorrection   projection/AMRCCProjector.cpp:204-377
  BaseProjector<FArrayBox>::project (lmin..lmax)               projection/BaseProjectorI.H:176-299
  Divergence::compDivergenceCC (the branch built by default:   calculus/DivCurlGrad/Divergence.cpp:514-585, 697-838
      USE_SIMPLE_STENCIL is commented out, :37)
  Gradient::compGradientCC (default branch), levelGradientMAC  calculus/DivCurlGrad/Gradient.cpp:560-591, 707-842, 85-206
  CRSEONESIDEGRAD                                              calculus/DivCurlGrad/DivCurlGradF.ChF:626-697
  Mask::buildMask                                              calculus/DivCurlGrad/Mask.cpp:16-59 (values MASKVAL.H)
  MappedCoarseAverage::averageToCoarse (unweighted)            MappedChombo/MappedCoarseAverage.cpp:137-198, 283-290
      + UNMAPPEDAVERAGE                                        MappedChombo/MappedCoarseAverageF.ChF:7-41
  MappedLevelFluxRegister (incrementCoarse / incrementFine / reflux with Jinv)   oracle/somar_amr.py::FluxRegister

Chombo 3.1 pieces that are EXTERNAL to the reference and restated from the published source: CellToEdge / EdgeToCell
(see somar_oracle.py), LevelData<FluxBox>::exchange = per motion item FluxBox::copy over surroundingNodes(region, dir)
for every dir.  The reference ships no fixture for this path: parity unpinned; tests/test_oracle_ccproj_amr.py pins the
restatement by properties (conservation of the composite divergence, exactness on linear fields, consistency with the
level projector when there is one level)."""
import ctypes as C

import numpy as np

from . import somar_amr as am
from . import somar_oracle as so
from .somar_oracle import Box, Fab, LevelData

MASKCOVERED, MASKPHYS, MASKCOPY, MASKCOARSE = -2, -1, 0, 1   # calculus/DivCurlGrad/MASKVAL.H


def quad_interp_comps(quad, fine, coarse):
    """MappedQuadCFInterp defined with ncomp comps: every component goes through the same stencils"""
    for c in range(fine.ncomp):
        f1 = LevelData(fine.grids, 1, fine.ghost)
        c1 = LevelData(coarse.grids, 1, coarse.ghost)
        for a, b in zip(f1.fabs, fine.fabs):
            a.a[..., 0] = b.a[..., c]
        for a, b in zip(c1.fabs, coarse.fabs):
            a.a[..., 0] = b.a[..., c]
        quad.coarse_fine_interp(f1, c1)
        for a, b in zip(f1.fabs, fine.fabs):
            b.a[..., c] = a.a[..., 0]


def comp_divergence_cc(comp, l, div, u, uCrse, uFine, wall=True):
    """Divergence::compDivergenceCC, Divergence.cpp:697-838.  u: LevelData (SpaceDim comps, >= 1 ghost) = J*u at cell
    centres; its CF ghosts (and uFine's) are filled here, as in the reference.  -> the face velocities used."""
    op, L, nd = comp.ops[l], comp.levels[l], comp.ndim
    if uCrse is not None:
        quad_interp_comps(op.quad, u, uCrse)                     # a_cfInterpCrse.coarseFineInterp(a_u, *a_uCrsePtr)
    uEdge = so.FluxData(L.grids, 1, nd)
    so.cell_to_edge(u, uEdge, nd)
    if wall:
        so.set_wall_normal_flux(uEdge, L.grids, L.domain, nd)    # levelDivergenceMAC's a_fluxBC, in place (:73-100)
    so.level_divergence_mac(div, uEdge, L.Jinv, L.grids, L.dx, nd)
    if uFine is not None:
        fop = comp.ops[l + 1]
        fr = op.fluxreg
        fr.set_to_zero()
        quad_interp_comps(fop.quad, uFine, u)                    # a_cfInterpFine.coarseFineInterp(*a_uFinePtr, a_u)
        for ci in range(len(L.grids)):
            for d in range(nd):
                fr.increment_coarse(uEdge[ci][d], 1.0 / L.dx[d], ci, d)
        for fi, fb in enumerate(fop.grids):
            for d in range(nd):
                for s in (0, 1):
                    # the two fine cells astride the box side -> CellToEdge -> the one layer of boundary faces
                    sh = [0, 0, 0]
                    if s == 0:
                        sh[d] = 1
                        ccEdgeBox = fb.adjCell(d, 0, 2).shift(sh)
                    else:
                        sh[d] = -1
                        ccEdgeBox = fb.adjCell(d, 1, 2).shift(sh)
                    lo, hi = list(fb.lo), list(fb.hi)
                    if s == 0:
                        hi[d] = lo[d]
                    else:
                        lo[d] = hi[d] = hi[d] + 1
                    edgeBox = Box(lo, hi)                          # face index i = low face of cell i
                    fineF = uFine[fi]
                    assert fineF.box.contains(ccEdgeBox)
                    edge = Fab(edgeBox, 1, np.nan)
                    back = [0, 0, 0]
                    back[d] = -1
                    edge.a[..., 0] = 0.5 * (fineF.view(edgeBox, d) + fineF.view(edgeBox.shift(back), d))
                    fr.increment_fine(edge, 1.0 / L.dx[d], fi, d, s)   # scale = 1 / (COARSE dx), :805
        fr.reflux(div, L.Jinv)
    return uEdge


def build_mask(box, domain, grids, fineGrids, nRefFine):
    """Mask::buildMask on `box` (Mask.cpp:16-59); plain boxes, no periodic images, as there"""
    m = np.full(box.size(), MASKPHYS, dtype=np.int32, order="F")

    def put(reg, v):
        reg = reg & box
        if not reg.isEmpty():
            m[reg.slices(box.lo)] = v

    # domainInterior &= a_dProblem: a ProblemDomain keeps everything along its periodic directions
    lo = [box.lo[d] if domain.periodic[d] else max(box.lo[d], domain.box.lo[d]) for d in range(3)]
    hi = [box.hi[d] if domain.periodic[d] else min(box.hi[d], domain.box.hi[d]) for d in range(3)]
    put(Box(lo, hi), MASKCOARSE)
    for g in grids:
        put(g, MASKCOPY)
    if fineGrids is not None:
        for g in fineGrids:
            put(g.coarsen(nRefFine), MASKCOVERED)
    return m


def exchange_faces(edge, grids, domain, nd):
    """LevelData<FluxBox>::exchange with Copier(grids, grids, domain, ghost 1, exchange = true): per motion item (a
    cell region of the destination's ghost layer covered by another box or a periodic image) every face direction is
    copied over surroundingNodes(region, dir)."""
    shifts = so.periodic_shifts(domain)
    for di, db in enumerate(grids):
        gbox = db.grow(1)
        for si, sb in enumerate(grids):
            for sh in shifts:
                if si == di and sh == (0, 0, 0):
                    continue
                r = gbox & sb.shift(sh)
                if r.isEmpty():
                    continue
                for d in range(nd):
                    rf = r.faces(d)
                    edge[di][d].view(rf)[...] = edge[si][d].view(rf.shift([-s for s in sh]))


def comp_gradient_cc(comp, l, grad, phi, phiCrse, phiFine):
    """Gradient::compGradientCC, Gradient.cpp:707-842: face gradients (levelGradientMAC with the coarse level's CF
    values), one-sided faces next to the finer level (CRSEONESIDEGRAD), EdgeToCell.  grad: LevelData, SpaceDim comps."""
    op, L, nd = comp.ops[l], comp.levels[l], comp.ndim
    grids = L.grids
    if phiCrse is not None:                                       # levelGradientMAC, :104-114
        op.quad.coarse_fine_interp(phi, phiCrse)
        if not op.isDiagonal:
            op.cf.extrapolate_cf_ev(phi, 2, op.activeDirs)
    inner = so.FluxData(grids, 1, nd)
    so.level_gradient_mac(inner, phi, grids, L.domain, L.Jgup, L.dx, nd, op=op)
    # LevelData<FluxBox> edgeGrad(grids, 1, ghost 1): only the faces of the valid box are computed (:139-140)
    edge = [[Fab(g.grow(1).faces(d), 1, np.nan) for d in range(nd)] for g in grids]
    for i, g in enumerate(grids):
        for d in range(nd):
            edge[i][d].view(g.faces(d))[...] = inner[i][d].a
    if phiFine is not None:
        fineGrids = comp.levels[l + 1].grids
        r = comp.refRatios[l]
        exchange_faces(edge, grids, L.domain, nd)
        crseFine = [g.coarsen(r) for g in fineGrids]
        for i, thisGradBox in enumerate(grids):
            mbox = thisGradBox.grow(2)
            mask = build_mask(mbox, L.domain, grids, fineGrids, r)
            for cfb in crseFine:
                overlap = thisGradBox & cfb.grow(1)
                if overlap.isEmpty():
                    continue
                for d in range(nd):
                    up = [0, 0, 0]
                    up[d] = 1
                    loEdge = (cfb.adjCell(d, 0, 1) & thisGradBox)
                    hiEdge = (cfb.adjCell(d, 1, 1) & thisGradBox)
                    # shiftHalf(dir, +1): cell c -> its HIGH face (index c + 1); shiftHalf(dir, -1): its LOW face (index c)
                    loEdge = loEdge.shift(up)
                    do_lo = 0 if overlap.lo[d] <= thisGradBox.lo[d] else 1
                    do_hi = 0 if overlap.hi[d] >= thisGradBox.hi[d] else 1
                    if loEdge.isEmpty():
                        do_lo = 0
                    if hiEdge.isEmpty():
                        do_hi = 0
                    e = edge[i][d]
                    so.lib().orc_crseonesidegrad(e.p(), e.lo(), e.hi(), mask.ctypes.data_as(C.POINTER(C.c_int)),
                                                 so._ivc(mbox.lo), so._ivc(mbox.hi), so._ivc(loEdge.lo), so._ivc(loEdge.hi),
                                                 so._ivc(hiEdge.lo), so._ivc(hiEdge.hi), d, do_lo, do_hi)
    # EdgeToCell(edgeGrad, a_grad)
    for i, g in enumerate(grids):
        for d in range(nd):
            up = [0, 0, 0]
            up[d] = 1
            grad[i].view(g, d)[...] = 0.5 * (edge[i][d].view(g, 0) + edge[i][d].view(g.shift(up), 0))
    return edge


def average_to_coarse(comp, l, crse, fine):
    """MappedCoarseAverage::averageToCoarse(crse, fine, ., considerCellSizes = false): UNMAPPEDAVERAGE per fine box into
    the coarsened fine layout, copyTo -> the valid cells of the coarse level under the fine one"""
    r = comp.refRatios[l]
    cgrids = [g.coarsen(r) for g in fine.grids]
    cf = LevelData(cgrids, fine.ncomp, (0, 0, 0))
    for i, cg in enumerate(cgrids):
        lo, hi = so._b(cg)
        so.lib().orc_unmappedaverage(*cf[i].fra(), *fine[i].fran(), lo, hi, so._ivc(r))
    shifts = so.periodic_shifts(comp.levels[l].domain)
    for db, df in zip(crse.grids, crse.fabs):
        for sb, sf in zip(cgrids, cf.fabs):
            for sh in shifts:
                reg = db & sb.shift(sh)
                if not reg.isEmpty():
                    df.view(reg)[...] = sf.view(reg.shift([-s for s in sh]))


def amr_cc_project(comp, vel, phi, lmin, lmax, dt, zeroPhi=False, forceHomogeneous=False, wall=True):
    """BaseProjector<FArrayBox>::project over levels lmin..lmax with AMRCCProjector's pieces (velocity in flux form,
    a_velIsFlux = true).  vel[l]: LevelData (SpaceDim comps, >= 1 ghost) or None outside [lmin-1, lmax]; phi[l] likewise
    (1 comp, 1 ghost).  Projects vel in place and leaves the pressure in phi.  -> the right-hand sides."""
    nlev = len(comp.levels)
    nd = comp.ndim
    rhs = [None] * nlev
    for lev in range(lmin, lmax + 1):
        L = comp.levels[lev]
        rhs[lev] = LevelData(L.grids, 1)
        crse = vel[lev - 1] if lev > 0 else None
        fine = vel[lev + 1] if lev < lmax else None
        so.exchange(vel[lev], L.domain, vel[lev].ghost)           # "Just in case...", AMRCCProjector.cpp:241-243
        comp_divergence_cc(comp, lev, rhs[lev], vel[lev], crse, fine, wall)
    if dt != 0.0:
        for lev in range(lmin, lmax + 1):
            for f in rhs[lev].fabs:
                f.a /= dt
    comp.solve(phi, rhs, lmax, lmin, zeroPhi=zeroPhi, forceHomogeneous=forceHomogeneous)
    corr = [None] * nlev
    for lev in range(lmin, lmax + 1):
        L = comp.levels[lev]
        corr[lev] = LevelData(L.grids, nd)
        so.exchange(phi[lev], L.domain, phi[lev].ghost)           # Copier + CornerCopier, AMRCCProjector.cpp:303-313
        comp_gradient_cc(comp, lev, corr[lev], phi[lev], phi[lev - 1] if lev > 0 else None,
                         phi[lev + 1] if lev < lmax else None)
    dtScale = -1.0 if dt == 0.0 else -dt
    for lev in range(lmax, lmin - 1, -1):
        for i, g in enumerate(comp.levels[lev].grids):
            vel[lev][i].view(g)[...] += dtScale * corr[lev][i].a
        if lev < lmax:
            average_to_coarse(comp, lev, vel[lev], _valid_only(vel[lev + 1]))
    return rhs


def _valid_only(ld):
    out = LevelData(ld.grids, ld.ncomp, (0, 0, 0))
    for g, a, b in zip(ld.grids, out.fabs, ld.fabs):
        a.a[...] = b.view(g)
    return out


def composite_sum(comp, fields, lmax, weight_jinv=True):
    """sum over the VALID region of the hierarchy (coarse cells under a finer level excluded) of field * J * dV"""
    tot = 0.0
    for l in range(lmax + 1):
        L = comp.levels[l]
        dV = float(np.prod(L.dx[:comp.ndim]))
        for i, g in enumerate(L.grids):
            w = np.ones(g.size())
            if l < lmax:
                r = comp.refRatios[l]
                for fg in comp.levels[l + 1].grids:
                    c = fg.coarsen(r) & g
                    if not c.isEmpty():
                        w[c.slices(g.lo)] = 0.0
            J = 1.0 / L.Jinv[i].view(g, 0) if weight_jinv else 1.0
            tot += float(np.sum(fields[l][i].view(g, 0) * J * w)) * dV
    return tot
