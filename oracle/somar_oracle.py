"""
oracle/somar_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the C++ orchestration of SOMAR's pressure-projection hot
path (reference = /root/reference/src, cited per function as file:line), driving
the Fortran-kernel restatements in oracle/kernels.c through ctypes.  numpy holds
the data; every arithmetic loop on the path lives in kernels.c so the operation
order is the reference's.  Only tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() may import this module; somar_amd/ never does.

Parity status: the reference has no tests/golden vectors and cannot be built
here (Chombo 3.1 absent) => "parity unpinned" w.r.t. reference tests; pinned by
analytic known answers (tests/test_oracle_kats.py).  Third-party arithmetic that
is NOT in /root/reference and is restated from its published algorithm:
Chombo 3.1 BiCGStabSolver (class BiCGStab below), LevelDataOps (dot/norm/axby),
Copier/exchange semantics.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BC_NONE, BC_NEUM, BC_DIRI = -1, 0, 1  # calculus/BCInterface/BCDescriptor.H:34-39

RELAX_JACOBI, RELAX_LEVEL_GSRB, RELAX_LOOSE_GSRB, RELAX_LINE_GSRB = 0, 1, 2, 3  # utils/ProblemContext.H:322-340
S_MAX_COARSE = 4  # AMRElliptic/MappedAMRPoissonOp.cpp:55


def build(force=False):
    """Compile oracle/kernels.c -> oracle/liboracle.so (gcc, no FMA contraction)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "kernels.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-std=c99",
                               "-shared", "-fPIC", "-o", so, src, "-lm"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


# ----------------------------------------------------------------------------
# boxlite: the few Chombo BoxTools semantics the path needs
# ----------------------------------------------------------------------------
class Box:
    __slots__ = ("lo", "hi")

    def __init__(self, lo, hi):
        self.lo = tuple(int(v) for v in lo)
        self.hi = tuple(int(v) for v in hi)

    def __repr__(self):
        return "Box(%s,%s)" % (self.lo, self.hi)

    def __eq__(self, o):
        return self.lo == o.lo and self.hi == o.hi

    def __hash__(self):
        return hash((self.lo, self.hi))

    def size(self):
        return tuple(h - l + 1 for l, h in zip(self.lo, self.hi))

    def isEmpty(self):
        return any(h < l for l, h in zip(self.lo, self.hi))

    def numPts(self):
        return 0 if self.isEmpty() else int(np.prod(self.size(), dtype=np.int64))

    def grow(self, g):
        g = _iv(g)
        return Box([l - a for l, a in zip(self.lo, g)], [h + a for h, a in zip(self.hi, g)])

    def growDir(self, d, n):
        g = [0, 0, 0]
        g[d] = n
        return self.grow(g)

    def __and__(self, o):
        return Box([max(a, b) for a, b in zip(self.lo, o.lo)], [min(a, b) for a, b in zip(self.hi, o.hi)])

    def intersects(self, o):
        return not (self & o).isEmpty()

    def contains(self, o):
        return o.isEmpty() or all(a <= b for a, b in zip(self.lo, o.lo)) and all(a >= b for a, b in zip(self.hi, o.hi))

    def shift(self, sh):
        sh = _iv(sh)
        return Box([l + s for l, s in zip(self.lo, sh)], [h + s for h, s in zip(self.hi, sh)])

    def coarsen(self, r):
        r = _iv(r)
        return Box([l // a for l, a in zip(self.lo, r)], [h // a for h, a in zip(self.hi, r)])

    def refine(self, r):
        r = _iv(r)
        return Box([l * a for l, a in zip(self.lo, r)], [(h + 1) * a - 1 for h, a in zip(self.hi, r)])

    def faces(self, d):
        """surroundingNodes(box, d): face i = low face of cell i."""
        hi = list(self.hi)
        hi[d] += 1
        return Box(self.lo, hi)

    def adjCell(self, d, side, n=1):
        """adjCellBox: the n cells just outside this box on `side` (0 lo / 1 hi) of dir d."""
        lo, hi = list(self.lo), list(self.hi)
        if side == 0:
            lo[d], hi[d] = self.lo[d] - n, self.lo[d] - 1
        else:
            lo[d], hi[d] = self.hi[d] + 1, self.hi[d] + n
        return Box(lo, hi)

    def edgeCells(self, d, side):
        """The 1-cell layer just INSIDE this box on that side."""
        lo, hi = list(self.lo), list(self.hi)
        if side == 0:
            hi[d] = lo[d]
        else:
            lo[d] = hi[d]
        return Box(lo, hi)

    def slices(self, origin_lo):
        return tuple(slice(l - o, h - o + 1) for l, h, o in zip(self.lo, self.hi, origin_lo))


def _iv(v):
    if isinstance(v, (int, np.integer)):
        return (int(v),) * 3
    return tuple(int(a) for a in v)


def coarsenable(boxes, r):
    r = _iv(r)
    return all(b.coarsen(r).refine(r) == b for b in boxes)


class Domain:
    def __init__(self, box, periodic=(False, False, False)):
        self.box = box
        self.periodic = tuple(bool(p) for p in periodic)

    def coarsen(self, r):
        return Domain(self.box.coarsen(r), self.periodic)

    def refine(self, r):
        return Domain(self.box.refine(r), self.periodic)


class Fab:
    """BaseFab<Real>: column-major, component slowest."""

    def __init__(self, box, ncomp=1, fill=0.0):
        self.box = box
        self.ncomp = ncomp
        self.a = np.full(box.size() + (ncomp,), fill, dtype=np.float64, order="F")

    def view(self, region, comp=None):
        s = region.slices(self.box.lo)
        return self.a[s + (slice(None) if comp is None else comp,)]

    def copy_from(self, src, region=None):
        r = (self.box & src.box) if region is None else region
        if r.isEmpty():
            return
        self.view(r)[...] = src.view(r)

    # ctypes marshalling
    def p(self):
        return self.a.ctypes.data_as(C.POINTER(C.c_double))

    def lo(self):
        return (C.c_int * 3)(*self.box.lo)

    def hi(self):
        return (C.c_int * 3)(*self.box.hi)

    def fra(self):
        return (self.p(), self.lo(), self.hi(), C.c_int(self.ncomp))

    def fra1(self, comp=0):
        off = comp * int(np.prod(self.box.size(), dtype=np.int64))
        ptr = C.cast(C.addressof(self.p().contents) + 8 * off, C.POINTER(C.c_double))
        return (ptr, self.lo(), self.hi())

    def fran(self):
        """pointer,lo,hi with all comps addressable (callee knows ncomp)."""
        return (self.p(), self.lo(), self.hi())


def _b(box):
    return ((C.c_int * 3)(*box.lo), (C.c_int * 3)(*box.hi))


def _rv(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def _ivc(v):
    return (C.c_int * 3)(*[int(x) for x in v])


class LevelData:
    """LevelData<FArrayBox> on a list of disjoint boxes (all local in the oracle)."""

    def __init__(self, grids, ncomp=1, ghost=(0, 0, 0), fill=0.0):
        self.grids = list(grids)
        self.ncomp = ncomp
        self.ghost = _iv(ghost)
        self.fabs = [Fab(b.grow(self.ghost), ncomp, fill) for b in self.grids]

    def __getitem__(self, i):
        return self.fabs[i]

    def __len__(self):
        return len(self.fabs)


class FluxData:
    """LevelData<FluxBox>: per box SpaceDim face-centred Fabs."""

    def __init__(self, grids, ncomp, ndim=3, fill=0.0):
        self.grids = list(grids)
        self.ncomp = ncomp
        self.fabs = [[Fab(b.faces(d), ncomp, fill) for d in range(ndim)] for b in self.grids]

    def __getitem__(self, i):
        return self.fabs[i]


# ---- LevelDataOps (Chombo 3.1, EXTERNAL; whole-FAB semantics incl. ghosts) ----
def ld_create(like):
    return LevelData(like.grids, like.ncomp, like.ghost)


def ld_set(a, v):
    for f in a.fabs:
        f.a[...] = v


def ld_assign(dst, src):
    for d, s in zip(dst.fabs, src.fabs):
        d.copy_from(s)


def ld_incr(dst, x, scale):
    for d, s in zip(dst.fabs, x.fabs):
        r = d.box & s.box
        d.view(r)[...] += scale * s.view(r)


def ld_axby(dst, x, y, a, b):
    for d, fx, fy in zip(dst.fabs, x.fabs, y.fabs):
        r = (d.box & fx.box) & fy.box
        d.view(r)[...] = a * fx.view(r) + b * fy.view(r)


def ld_scale(a, s):
    for f in a.fabs:
        f.a *= s


def _seqsum(v):
    # sequential (Fortran-order) summation, as FArrayBox::dotProduct / sumPow do
    return float(np.add.accumulate(np.ravel(v, order="F"))[-1]) if v.size else 0.0


def ld_dot(a, b):
    """LevelDataOps::dotProduct: sum over valid cells, no metric."""
    tot = 0.0
    for g, fa, fb in zip(a.grids, a.fabs, b.fabs):
        tot += _seqsum(fa.view(g) * fb.view(g))
    return tot


def ld_norm(a, order):
    """CH_XD::norm(a, interval, p): p=0 max |a|; p=1 sum |a|; else (sum |a|^p)^(1/p), no dx."""
    if order == 0:
        return max([float(np.max(np.abs(f.view(g)))) for g, f in zip(a.grids, a.fabs)] + [0.0])
    tot = 0.0
    for g, f in zip(a.grids, a.fabs):
        tot += _seqsum(np.abs(f.view(g)) ** order)
    return tot if order == 1 else tot ** (1.0 / order)


# ----------------------------------------------------------------------------
# exchange: Copier(grids, grids, domain, ghost, exchange=true) semantics --
# every ghost cell covered by another box's valid region (or a periodic image,
# incl. the box itself) is overwritten with that valid value.
# ----------------------------------------------------------------------------
def periodic_shifts(domain):
    n = domain.box.size()
    rng = [(-n[d], 0, n[d]) if domain.periodic[d] else (0,) for d in range(3)]
    return [(a, b, c) for a in rng[0] for b in rng[1] for c in rng[2]]


def exchange(ld, domain, ghost=None):
    ghost = ld.ghost if ghost is None else _iv(ghost)
    shifts = periodic_shifts(domain)
    for di, (db, dfab) in enumerate(zip(ld.grids, ld.fabs)):
        gbox = db.grow(ghost) & dfab.box
        for si, (sb, sfab) in enumerate(zip(ld.grids, ld.fabs)):
            for sh in shifts:
                if si == di and sh == (0, 0, 0):
                    continue
                img = sb.shift(sh)
                r = gbox & img
                if r.isEmpty():
                    continue
                dfab.view(r)[...] = sfab.view(r.shift([-s for s in sh]))


# ----------------------------------------------------------------------------
# BCs for the pressure solve: constant Neumann ghost + zero boundary flux
# (BCutil/PhysBCUtil.cpp:1404-1422; BCInterface/EllipticBCUtils.cpp:128-214,
#  431-542; EllipticBCInterface.cpp:107-218)
# ----------------------------------------------------------------------------
class BCHolder:
    def __init__(self, types=None, values=None):
        # types[d][side], values[d][side]
        self.types = types if types is not None else [[BC_NEUM, BC_NEUM] for _ in range(3)]
        self.values = values if values is not None else [[0.0, 0.0] for _ in range(3)]

    def stencil(self, region, domain, d, side):
        """BCDescriptor::stencil, BCDescriptor.H:220-229."""
        end = region.lo[d] if side == 0 else region.hi[d]
        dend = domain.box.lo[d] if side == 0 else domain.box.hi[d]
        if (not domain.periodic[d]) and end == dend:
            return self.types[d][side]
        return BC_NONE


def extrapolate_face_no_ev(dest, src, valid, d, side, order):
    """ExtrapolateFaceNoEV, extrapolation/ExtrapolationUtils.cpp:34-67 (cell-centred case)."""
    to = valid.adjCell(d, side, 1) & src.box
    if to.isEmpty():
        return
    lo, hi = _b(to)
    rc = lib().orc_extrapolatefacenoev(*dest.fra(), *src.fran(), lo, hi, d, 1 if side else -1, order)
    assert rc == 0


def extrapolate_face_and_copy(dest, src, valid, d, side, order, num_layers=1):
    """ExtrapolateFaceAndCopy, ExtrapolationUtils.cpp:109-155."""
    extrapolate_face_no_ev(dest, src, valid, d, side, order)
    ghostBox = valid.adjCell(d, side, 1) & src.box
    sgn = 1 if side else -1
    sh = [0, 0, 0]
    sh[d] = -sgn
    nearBox = ghostBox.shift(sh)
    # growDir(dir, flip(side), numLayers-1)
    if num_layers > 1:
        lo, hi = list(nearBox.lo), list(nearBox.hi)
        if side == 1:
            lo[d] -= num_layers - 1
        else:
            hi[d] += num_layers - 1
        nearBox = Box(lo, hi)
    if dest is not src:
        dest.copy_from(src, nearBox)
    for e in range(3):
        if e == d:
            continue
        if valid.lo[e] == valid.hi[e] and dest.box.lo[e] == dest.box.hi[e]:
            continue  # flat (2-D) direction: SpaceDim == 2 has no such edir
        for es in (0, 1):
            extrapolate_face_no_ev(dest, dest, ghostBox, e, es, order)
            extrapolate_face_no_ev(dest, dest, nearBox, e, es, order)
        ghostBox = ghostBox.growDir(e, 1) & src.box
        nearBox = nearBox.growDir(e, 1) & src.box


def set_side_neum_bc(state, valid, domain, value, d, side, Jgup_d, extrap, dx, is_diagonal, ndim=3):
    """setSideNeumBC, BCInterface/EllipticBCUtils.cpp:128-214 (a_homogeneous is unused there)."""
    if domain.periodic[d]:
        return
    vend = valid.lo[d] if side == 0 else valid.hi[d]
    dend = domain.box.lo[d] if side == 0 else domain.box.hi[d]
    if vend != dend:
        return
    ghostBox = valid.adjCell(d, side, 1) & state.box
    if ghostBox.isEmpty():
        return
    lo, hi = _b(ghostBox)
    sgn = 1 if side else -1
    if is_diagonal:
        lib().orc_ellipticconstneumbcghostortho(*state.fra(), *Jgup_d.fra1(d), lo, hi,
                                                C.c_double(value), d, sgn, C.c_double(dx[d]))
    else:
        ex = extrap if extrap is not None else Fab(state.box, state.ncomp)
        extrapolate_face_and_copy(ex, state, valid, d, side, 2)
        fn = lib().orc_ellipticconstneumbcghost if ndim == 3 else lib().orc_ellipticconstneumbcghost2d
        fn(*state.fra(), *ex.fran(), *Jgup_d.fran(), lo, hi, C.c_double(value), d, sgn, _rv(dx))


def set_side_diri_bc(state, valid, domain, value, d, side, homogeneous, order):
    """setSideDiriBC (cell-centred branch), BCInterface/EllipticBCUtils.cpp:41-119; EllipticConstDiriBCGhostClass
    calls it with order 1 (:414-421)."""
    if domain.periodic[d]:
        return
    vend = valid.lo[d] if side == 0 else valid.hi[d]
    dend = domain.box.lo[d] if side == 0 else domain.box.hi[d]
    if vend != dend:
        return
    ghostBox = valid.adjCell(d, side, 1) & state.box
    if ghostBox.isEmpty():
        return
    lo, hi = _b(ghostBox)
    bcval = 0.0 if homogeneous else value
    rc = lib().orc_ellipticconstdiribcghost(*state.fra(), lo, hi, C.c_double(bcval), d, 1 if side else -1, order)
    assert rc == 0


def bc_set_ghosts(bc, state, extrap, valid, domain, dx, Jgup, homogeneous, is_diagonal, ndim=3):
    """EllipticConstNeumBCGhostClass::operator(), EllipticBCUtils.cpp:431-482."""
    for d in range(ndim):
        if domain.periodic[d]:
            continue
        for side in (0, 1):
            if bc.types[d][side] == BC_NEUM:
                set_side_neum_bc(state, valid, domain, bc.values[d][side], d, side, Jgup[d], extrap, dx, is_diagonal, ndim)
            elif bc.types[d][side] == BC_DIRI:
                set_side_diri_bc(state, valid, domain, bc.values[d][side], d, side, homogeneous, 1)
            else:
                raise NotImplementedError("BC type %r" % (bc.types[d][side],))


def bc_set_fluxes(bc, flux, valid, domain, homogeneous, ndim=3):
    """EllipticConstNeumBCFluxClass::operator() -> setSideDiriBC on the face FAB,
    EllipticBCUtils.cpp:491-542, 41-119: boundary faces := (homogeneous ? 0 : value)."""
    for d in range(ndim):
        if domain.periodic[d]:
            continue
        for side in (0, 1):
            vend = valid.lo[d] if side == 0 else valid.hi[d]
            dend = domain.box.lo[d] if side == 0 else domain.box.hi[d]
            if vend != dend:
                continue
            if bc.types[d][side] != BC_NEUM:   # a Dirichlet side has no flux method: the flux comes from the ghost
                continue
            fb = valid.faces(d)
            lo, hi = list(fb.lo), list(fb.hi)
            if side == 0:
                hi[d] = lo[d]
            else:
                lo[d] = hi[d]
            bcval = 0.0 if homogeneous else bc.values[d][side]
            flux[d].view(Box(lo, hi))[...] = bcval


# ----------------------------------------------------------------------------
# RelaxationMethod / LevelGSRB / Jacobi
# ----------------------------------------------------------------------------
class BoundaryBoxData:
    __slots__ = ("index", "valid", "validBdry", "stencil")


def collect_boundary_data(grids, domain, bc, activeDirs, simple=False):
    """RelaxationMethod::collectBoundaryData, RelaxationMethods/RelaxationMethod.cpp:83-364.
    simple=False -> m_boundaryBoxData (domain-boundary shells only);
    simple=True  -> m_simpleBoundaryBoxData (all box-boundary shells, LooseGSRB)."""
    domBox = domain.box
    domBdry = [[domBox.edgeCells(d, s) for s in (0, 1)] for d in range(3)]
    domInterior = domBox.grow([-a for a in activeDirs])
    out = []

    def sten(b):
        return [[bc.stencil(b, domain, d, s) for s in (0, 1)] for d in range(3)]

    def push(idx, valid, b):
        e = BoundaryBoxData()
        e.index, e.valid, e.validBdry, e.stencil = idx, valid, b, sten(b)
        out.append(e)

    for idx, valid in enumerate(grids):
        interior = valid.grow([-a for a in activeDirs]) if simple else (valid & domInterior)
        for fdir in range(3):
            if activeDirs[fdir] == 0:
                continue
            for fs in (0, 1):
                faceBox = interior.adjCell(fdir, fs, 1) & valid
                if not simple and not domBdry[fdir][fs].intersects(faceBox):
                    continue
                if simple and faceBox.isEmpty():
                    continue
                push(idx, valid, faceBox)
                for edir in range(fdir + 1, 3):
                    if activeDirs[edir] == 0:
                        continue
                    for es in (0, 1):
                        edgeBox = faceBox.adjCell(edir, es, 1) & valid
                        if not simple and not domBdry[edir][es].intersects(edgeBox):
                            continue
                        if simple and edgeBox.isEmpty():
                            continue
                        push(idx, valid, edgeBox)
                        vdir = 3 - fdir - edir
                        if vdir <= edir:
                            continue
                        if activeDirs[vdir] == 0:
                            continue
                        for vs in (0, 1):
                            vertexBox = edgeBox.adjCell(vdir, vs, 1) & valid
                            if not simple and not domBdry[vdir][vs].intersects(vertexBox):
                                continue
                            if simple and vertexBox.isEmpty():
                                continue
                            push(idx, valid, vertexBox)
    return out


class Relaxer:
    """RelaxationMethod base: define() + fillGhostsAndExtrapolate(), RelaxationMethod.cpp:25-76, 376-435."""

    def __init__(self, op):
        self.op = op
        self.extrap = LevelData(op.grids, 1, op.activeDirs)
        self.bdry = collect_boundary_data(op.grids, op.domain, op.bc, op.activeDirs, simple=False)
        self.simple_bdry = collect_boundary_data(op.grids, op.domain, op.bc, op.activeDirs, simple=True)

    def fill_ghosts_and_extrapolate(self, phi, doCF=True, doExtrap=True, doBCs=True):
        op = self.op
        extrapOrder = 1
        if doCF and op.cf is not None:
            op.cf.homogeneous_cf_interp(phi, op.dx, op.dxCrse, op.activeDirs)
        for i, valid in enumerate(op.grids):
            phiF, exF = phi[i], self.extrap[i]
            if (not op.isDiagonal) and doExtrap:
                exF.copy_from(phiF)
                domValid = op.domain.box & exF.box
                for fdir in range(op.ndim):
                    if op.activeDirs[fdir] == 0:
                        continue
                    for fs in (0, 1):
                        extrapolate_face_and_copy(exF, exF, domValid, fdir, fs, extrapOrder)
                    domValid = domValid.growDir(fdir, 1)
            if doBCs:
                bc_set_ghosts(op.bc, phiF, exF, valid, op.domain, op.dx, op.Jgup[i], True, op.isDiagonal, op.ndim)


class LevelGSRB(Relaxer):
    """LevelGSRB::relax, RelaxationMethods/GSRB.cpp:58-98; fullStencilGSRB :341-478;
    boundaryGSRB :489-653."""

    def relax(self, phi, rhs):
        op = self.op
        domInterior = op.domain.box.grow([-a for a in op.activeDirs])
        for whichPass in (0, 1):
            exchange(phi, op.domain, op.activeDirs)
            self.fill_ghosts_and_extrapolate(phi)
            for i, g in enumerate(op.grids):
                self.full_stencil_gsrb(phi[i], rhs[i], g & domInterior, i, whichPass)
            self.boundary_gsrb(phi, rhs, whichPass, False)

    def full_stencil_gsrb(self, phiF, rhsF, region, i, whichPass):
        op = self.op
        if region.isEmpty():
            return
        L = lib()
        Jg, Jinv, lapd, ex = op.Jgup[i], op.Jinv[i], op.lapDiag[i], self.extrap[i]
        lo, hi = _b(region)
        a, b = C.c_double(op.alpha), C.c_double(op.beta)
        if op.ndim == 3:
            if op.isDiagonal:
                L.orc_gsrbiter3dortho(*phiF.fra(), *rhsF.fran(), *Jg[0].fra1(0), *Jg[1].fra1(1), *Jg[2].fra1(2),
                                      *Jinv.fra1(0), *lapd.fra1(0), lo, hi, _rv(op.dx), a, b, whichPass)
            else:
                L.orc_gsrbiter3d(*phiF.fra(), *ex.fran(), *rhsF.fran(), *Jg[0].fran(), *Jg[1].fran(), *Jg[2].fran(),
                                 *Jinv.fra1(0), *lapd.fra1(0), lo, hi, _rv(op.dx), a, b, whichPass)
        else:
            if op.isDiagonal:
                L.orc_gsrbiter2dortho(*phiF.fra(), *rhsF.fran(), *Jg[0].fra1(0), *Jg[1].fra1(1),
                                      *Jinv.fra1(0), *lapd.fra1(0), lo, hi, _rv(op.dx), a, b, whichPass)
            else:
                L.orc_gsrbiter2d(*phiF.fra(), *ex.fran(), *rhsF.fran(), *Jg[0].fran(), *Jg[1].fran(),
                                 *Jinv.fra1(0), *lapd.fra1(0), lo, hi, _rv(op.dx), a, b, whichPass)

    def boundary_gsrb(self, phi, rhs, whichPass, doAll):
        op = self.op
        L = lib()
        a, b = C.c_double(op.alpha), C.c_double(op.beta)
        for e in (self.simple_bdry if doAll else self.bdry):
            i = e.index
            Jg, Jinv, ex = op.Jgup[i], op.Jinv[i], self.extrap[i]
            lo, hi = _b(e.validBdry)
            st = (C.c_int * 6)(e.stencil[0][0], e.stencil[0][1], e.stencil[1][0], e.stencil[1][1],
                               e.stencil[2][0], e.stencil[2][1])
            if op.ndim == 3:
                if op.isDiagonal:
                    L.orc_gsrbboundaryiter3dortho(*phi[i].fra(), *rhs[i].fran(), *Jg[0].fra1(0), *Jg[1].fra1(1),
                                                  *Jg[2].fra1(2), *Jinv.fra1(0), lo, hi, _rv(op.dx), a, b, st,
                                                  whichPass)
                else:
                    L.orc_gsrbboundaryiter3d(*phi[i].fra(), *ex.fran(), *rhs[i].fran(), *Jg[0].fran(),
                                             *Jg[1].fran(), *Jg[2].fran(), *Jinv.fra1(0), lo, hi, _rv(op.dx),
                                             a, b, st, whichPass)
            else:
                if op.isDiagonal:
                    L.orc_gsrbboundaryiter2dortho(*phi[i].fra(), *rhs[i].fran(), *Jg[0].fra1(0), *Jg[1].fra1(1),
                                                  *Jinv.fra1(0), lo, hi, _rv(op.dx), a, b, st, whichPass)
                else:
                    L.orc_gsrbboundaryiter2d(*phi[i].fra(), *ex.fran(), *rhs[i].fran(), *Jg[0].fran(), *Jg[1].fran(),
                                             *Jinv.fra1(0), lo, hi, _rv(op.dx), a, b, st, whichPass)


class LooseGSRB(LevelGSRB):
    """LooseGSRB::relax, GSRB.cpp:104-141: one exchange per sweep.  exchangeBegin posts the ghost copies before
    the interior phase and exchangeEnd lands them after it; the copies only carry cells of the box shells
    (ghost depth 1), which the interior phase does not touch, so landing them up front gives the same data."""

    def relax(self, phi, rhs):
        op = self.op
        exchange(phi, op.domain, op.activeDirs)
        self.fill_ghosts_and_extrapolate(phi)
        for i, g in enumerate(op.grids):
            interior = g.grow([-a for a in op.activeDirs])
            self.full_stencil_gsrb(phi[i], rhs[i], interior, i, 0)
            self.full_stencil_gsrb(phi[i], rhs[i], interior, i, 1)
        self.boundary_gsrb(phi, rhs, 0, True)
        self.boundary_gsrb(phi, rhs, 1, True)


class LineGSRB(Relaxer):
    """LineGSRB::relax, RelaxationMethods/GSRB.cpp:148-330 (3-D): vertical line relaxation, dgtsv per column.
    Uses the intended region-bound Neumann test (deviation Q1, see oracle/kernels.c)."""

    def relax(self, phi, rhs):
        op = self.op
        L = lib()
        if op.ndim == 2:
            # CH_SPACEDIM = 2 build: LineGSRBIter2D, the vertical is direction 1 (GSRB.cpp:262-289, GSRBF.ChF:1529-1724)
            for whichPass in (0, 1):
                exchange(phi, op.domain, op.activeDirs)
                self.fill_ghosts_and_extrapolate(phi)
                for i, valid in enumerate(op.grids):
                    st = [op.bc.stencil(valid, op.domain, d, s) for d in range(2) for s in (0, 1)]
                    ex = self.extrap[i] if not op.isDiagonal else phi[i]
                    lo, hi = _b(valid)
                    Jg = op.Jgup[i]
                    dzCrse = op.dxCrse[1] if op.dxCrse is not None else float("nan")
                    info = L.orc_linegsrbiter2d(*phi[i].fra1(0), *ex.fra1(0), *rhs[i].fra1(0), *Jg[0].fran(), *Jg[1].fran(),
                                                *op.Jinv[i].fra1(0), lo, hi, _rv(op.dx), C.c_double(dzCrse),
                                                C.c_double(op.alpha), C.c_double(op.beta), whichPass, (C.c_int * 4)(*st))
                    assert info == 0, "LineGSRBIter2D: INFO = %d" % info
            return
        assert op.ndim == 3 and op.activeDirs == (1, 1, 1)
        for whichPass in (0, 1):
            exchange(phi, op.domain, op.activeDirs)
            self.fill_ghosts_and_extrapolate(phi)
            for i, valid in enumerate(op.grids):
                st = [op.bc.stencil(valid, op.domain, d, s) for d in range(3) for s in (0, 1)]
                ex = self.extrap[i] if not op.isDiagonal else phi[i]   # diagonal metric: Jg^{ab}=0 kills the cross terms
                lo, hi = _b(valid)
                Jg = op.Jgup[i]
                dzCrse = op.dxCrse[2] if op.dxCrse is not None else float("nan")
                info = L.orc_linegsrbiter3d(*phi[i].fra1(0), *ex.fra1(0), *rhs[i].fra1(0), *Jg[0].fran(), *Jg[1].fran(),
                                            *Jg[2].fran(), *op.Jinv[i].fra1(0), lo, hi, _rv(op.dx), C.c_double(dzCrse),
                                            C.c_double(op.alpha), C.c_double(op.beta), whichPass, (C.c_int * 6)(*st))
                assert info == 0, "dgtsv INFO = %d" % info


class Jacobi(Relaxer):
    """Jacobi::relax, RelaxationMethods/Jacobi.cpp:54-90."""

    def relax(self, phi, rhs):
        op = self.op
        resid = ld_create(rhs)
        op.residual(resid, phi, rhs, True)
        for i, g in enumerate(op.grids):
            lo, hi = _b(g)
            lib().orc_jacobiiter(*phi[i].fra(), *resid[i].fran(), *op.lapDiag[i].fra1(0), lo, hi,
                                 C.c_double(op.alpha), C.c_double(op.beta))


# ----------------------------------------------------------------------------
# MappedAMRPoissonOp (single level part)
# ----------------------------------------------------------------------------
class PoissonOp:
    """MappedAMRPoissonOp, AMRElliptic/MappedAMRPoissonOp.cpp."""

    def __init__(self, grids, domain, dx, bc, Jgup, Jinv, lapDiag, alpha, beta, isDiagonal,
                 ndim=3, dxCrse=None, cf=None, relaxMode=RELAX_LEVEL_GSRB, precondIters=2):
        self.grids, self.domain, self.dx, self.bc = list(grids), domain, tuple(dx), bc
        self.Jgup, self.Jinv, self.lapDiag = Jgup, Jinv, lapDiag
        self.alpha, self.beta, self.isDiagonal, self.ndim = alpha, beta, isDiagonal, ndim
        self.activeDirs = (1, 1, 1) if ndim == 3 else (1, 1, 0)
        self.dxCrse, self.cf = dxCrse, cf
        self.mgCrseRefRatio = None
        self.precondIters = precondIters
        # m_validDomain: domain grown by 1 in periodic dirs (MappedAMRPoissonOp.cpp:278-283)
        self.validDomain = domain.box.grow([1 if (domain.periodic[d] and self.activeDirs[d]) else 0 for d in range(3)])
        if relaxMode == RELAX_JACOBI:
            self.relaxer = Jacobi(self)
        elif relaxMode == RELAX_LEVEL_GSRB:
            self.relaxer = LevelGSRB(self)
        elif relaxMode == RELAX_LOOSE_GSRB:
            self.relaxer = LooseGSRB(self)
        elif relaxMode == RELAX_LINE_GSRB:
            self.relaxer = LineGSRB(self)
        else:
            raise NotImplementedError("relaxMode %d" % relaxMode)
        self.zeroAvg = False
        self.dxProduct = float(np.prod(self.dx[:ndim]))

    # -- ghosts
    def exchange_complete(self, phi):
        """exchangeComplete, MappedAMRPoissonOp.cpp:2222-2238: faces+edges+corners."""
        exchange(phi, self.domain, phi.ghost)

    def interp_cf_ghosts_homog(self, phi):
        """interpCFGhosts(homogeneous), MappedAMRPoissonOp.cpp:2193-2201."""
        if self.cf is not None:
            self.cf.homogeneous_cf_interp(phi, self.dx, self.dxCrse, self.activeDirs)
            self.cf.extrapolate_cf_ev(phi, 2, self.activeDirs)

    def fill_extrap(self, extrap, phiF, order):
        """fillExtrap, MappedAMRPoissonOp.cpp:2244-2270."""
        if self.isDiagonal:
            return
        stateBox = phiF.box
        validPhi = stateBox & self.validDomain
        extrap.copy_from(phiF, validPhi)
        for fdir in range(self.ndim):
            extrapolate_face_and_copy(extrap, extrap, validPhi, fdir, 0, order)
            extrapolate_face_and_copy(extrap, extrap, validPhi, fdir, 1, order)
            validPhi = validPhi.growDir(fdir, 1) & stateBox

    # -- operator
    def apply_op_i(self, lhs, phi, homogeneous):
        """applyOpI (USE_SYNC_METHOD branch), MappedAMRPoissonOp.cpp:772-898."""
        L = lib()
        self.exchange_complete(phi)
        for i, valid in enumerate(self.grids):
            phiF, lhsF, Jg, Jinv = phi[i], lhs[i], self.Jgup[i], self.Jinv[i]
            flux = [Fab(valid.faces(d), phi.ncomp, np.nan) for d in range(self.ndim)]
            extrap = Fab(phiF.box, phi.ncomp, np.nan)
            self.fill_extrap(extrap, phiF, 2)
            bc_set_ghosts(self.bc, phiF, extrap, valid, self.domain, self.dx, Jg, homogeneous, self.isDiagonal, self.ndim)
            for d in range(self.ndim):
                self.get_flux_complete(flux[d], phiF, extrap, valid.faces(d), i, d)
            bc_set_fluxes(self.bc, flux, valid, self.domain, homogeneous, self.ndim)
            for d in range(self.ndim):
                flux[d].a *= self.beta  # fluxFB *= m_beta  (:841)
            lo, hi = _b(valid)
            if self.ndim == 3:
                L.orc_mappedfluxdivergence3d(*lhsF.fra(), *flux[0].fran(), *flux[1].fran(), *flux[2].fran(),
                                             *Jinv.fra1(0), lo, hi, _rv(self.dx))
            else:
                L.orc_mappedfluxdivergence2d(*lhsF.fra(), *flux[0].fran(), *flux[1].fran(),
                                             *Jinv.fra1(0), lo, hi, _rv(self.dx))
            if self.alpha != 0.0:
                L.orc_axbyip(*lhsF.fra(), *phiF.fran(), C.c_double(self.alpha), C.c_double(1.0), lo, hi)

    def get_flux_complete(self, fluxF, phiF, extrap, edgebox, i, d, ref=1):
        """getFluxComplete, MappedAMRPoissonOp.cpp:2048-2122."""
        lo, hi = _b(edgebox)
        Jg = self.Jgup[i][d]
        if self.isDiagonal:
            scale = ref / self.dx[d]
            lib().orc_mappedgetfluxortho(*fluxF.fra(), *phiF.fran(), *Jg.fra1(d), lo, hi, C.c_double(scale), d)
        else:
            if self.ndim != 3:
                lib().orc_mappedgetflux2d(*fluxF.fra(), *phiF.fran(), *extrap.fran(), *Jg.fran(), lo, hi,
                                          C.c_double(float(ref)), _rv(self.dx), d)
                return
            lib().orc_mappedgetflux(*fluxF.fra(), *phiF.fran(), *extrap.fran(), *Jg.fran(), lo, hi,
                                    C.c_double(float(ref)), _rv(self.dx), d)

    def apply_op(self, lhs, phi, homogeneous):
        """applyOp, MappedAMRPoissonOp.cpp:740-765."""
        self.interp_cf_ghosts_homog(phi)
        self.apply_op_i(lhs, phi, homogeneous)

    def residual_i(self, lhs, phi, rhs, homogeneous):
        """residualI, MappedAMRPoissonOp.cpp:646-677."""
        self.apply_op_i(lhs, phi, homogeneous)
        for i, g in enumerate(self.grids):
            lo, hi = _b(g)
            lib().orc_subtractop(*lhs[i].fra(), *rhs[i].fran(), *lhs[i].fran(), lo, hi)

    def residual(self, lhs, phi, rhs, homogeneous):
        """residual, MappedAMRPoissonOp.cpp:628-640."""
        self.interp_cf_ghosts_homog(phi)
        self.residual_i(lhs, phi, rhs, homogeneous)

    def pre_cond(self, phi, rhs):
        """preCond (DiagRelax), MappedAMRPoissonOp.cpp:684-734."""
        if self.precondIters == 0:
            ld_assign(phi, rhs)
            return
        for i, g in enumerate(self.grids):
            lo, hi = _b(g)
            lib().orc_diagprecond(*phi[i].fra(), *rhs[i].fran(), *self.lapDiag[i].fra1(0), lo, hi,
                                  C.c_double(self.alpha), C.c_double(self.beta))
        for _ in range(self.precondIters):
            self.relaxer.relax(phi, rhs)

    def relax(self, e, residual, iterations):
        """relax, MappedAMRPoissonOp.cpp:1733-1743."""
        for _ in range(iterations):
            self.relaxer.relax(e, residual)

    # -- MG transfer
    def coarse_grids(self):
        return [g.coarsen(self.mgCrseRefRatio) for g in self.grids]

    def create_coarser(self, fine, ghosted=None):
        """createCoarser, MappedAMRPoissonOp.cpp:1256-1273 (ghost vector copied from fine)."""
        return LevelData(self.coarse_grids(), fine.ncomp, fine.ghost)

    def restrict_residual(self, resCoarse, phiFine, rhsFine):
        """restrictResidual, MappedAMRPoissonOp.cpp:1281-1304 + FullWeightingPS::restrict,
        MGStrategies/RestrictionStrategy.cpp:36-99."""
        resFine = LevelData(self.grids, phiFine.ncomp, (0, 0, 0))
        self.residual(resFine, phiFine, rhsFine, True)
        r = self.mgCrseRefRatio
        for i, cg in enumerate(resCoarse.grids):
            lo, hi = _b(cg)
            lib().orc_mappedaverage2(*resCoarse[i].fra(), *resFine[i].fran(), *self.Jinv[i].fra1(0), lo, hi, _ivc(r))

    def prolong_increment(self, phiFine, corrCoarse):
        """ConstInterpPS / ZeroAvgConstInterpPS::prolongIncrement,
        MGStrategies/ProlongationStrategy.cpp:49-84, 91-164."""
        m = self.mgCrseRefRatio
        vol, s = C.c_double(0.0), C.c_double(0.0)
        for i, fineValid in enumerate(self.grids):
            fiv = fineValid.lo
            civ = tuple(a // b for a, b in zip(fiv, m))
            fF, cF = phiFine[i], corrCoarse[i]
            # CHF_FRA_SHIFT: same data, box shifted by -fiv / -civ
            fl = _ivc([a - b for a, b in zip(fF.box.lo, fiv)])
            fh = _ivc([a - b for a, b in zip(fF.box.hi, fiv)])
            cl = _ivc([a - b for a, b in zip(cF.box.lo, civ)])
            ch = _ivc([a - b for a, b in zip(cF.box.hi, civ)])
            reg = fineValid.shift([-a for a in fiv])
            lo, hi = _b(reg)
            if self.zeroAvg:
                J = self.Jinv[i]
                jl = _ivc([a - b for a, b in zip(J.box.lo, fiv)])
                jh = _ivc([a - b for a, b in zip(J.box.hi, fiv)])
                lib().orc_constinterpwithavgps(fF.p(), fl, fh, C.c_int(fF.ncomp), cF.p(), cl, ch, lo, hi, _ivc(m),
                                               J.p(), jl, jh, C.c_double(self.dxProduct), C.byref(vol), C.byref(s))
            else:
                lib().orc_constinterpps(fF.p(), fl, fh, C.c_int(fF.ncomp), cF.p(), cl, ch, lo, hi, _ivc(m))
        if self.zeroAvg:
            avgPhi = s.value / vol.value
            for f in phiFine.fabs:
                f.a -= avgPhi  # a_phiThisLevel[dit] -= avgPhi : whole FAB incl. ghosts

    def local_max_norm(self, x):
        return ld_norm(x, 0)


# ----------------------------------------------------------------------------
# Metric coarsening + factory (MappedAMRPoissonOpFactory)
# ----------------------------------------------------------------------------
def fill_lap_diag(lapDiag, Jgup, Jinv, grids, dx, ndim):
    """FILLMAPPEDLAPDIAG*D call sites, MappedAMRPoissonOpFactory.cpp:985-1034, 1189-1234."""
    for i, g in enumerate(grids):
        lo, hi = _b(g)
        if ndim == 3:
            lib().orc_fillmappedlapdiag3d(*lapDiag[i].fra1(0), *Jgup[i][0].fran(), *Jgup[i][1].fran(),
                                          *Jgup[i][2].fran(), *Jinv[i].fra1(0), lo, hi, _rv(dx))
        else:
            lib().orc_fillmappedlapdiag2d(*lapDiag[i].fra1(0), *Jgup[i][0].fran(), *Jgup[i][1].fran(),
                                          *Jinv[i].fra1(0), lo, hi, _rv(dx))


def coarsen_metric(fineGrids, fineJgup, fineJinv, mgRefRatio, ndim):
    """fill_MGfields (a_MGdepth > 0 branch), MappedAMRPoissonOpFactory.cpp:1164-1234:
    Jgup^c = arithmetic mean of the fine faces lying on the coarse face
    (MappedChombo/MappedCoarseAverage.cpp:447-521 -> UNMAPPEDAVERAGEFACE),
    Jinv^c = harmonic mean (UNMAPPEDAVERAGEHARMONIC)."""
    r = _iv(mgRefRatio)
    crseGrids = [g.coarsen(r) for g in fineGrids]
    Jgup = FluxData(crseGrids, fineJgup.ncomp, ndim)
    Jinv = LevelData(crseGrids, 1, (0, 0, 0))
    for i, cg in enumerate(crseGrids):
        for d in range(ndim):
            cF, fF = Jgup[i][d], fineJgup[i][d]
            lo, hi = _b(cF.box)
            lib().orc_unmappedaverageface(*cF.fra(), *fF.fran(), lo, hi, d, _ivc(r))
        lo, hi = _b(cg)
        lib().orc_unmappedaverageharmonic(*Jinv[i].fra(), *fineJinv[i].fran(), lo, hi, _ivc(r))
    return crseGrids, Jgup, Jinv


def choose_mg_ref_ratio(dx, ndim):
    """Semicoarsening rule, MappedAMRPoissonOpFactory.cpp:476-495."""
    maxDx = max(dx[:ndim])
    r = [1, 1, 1]
    for d in range(ndim):
        if dx[d] <= maxDx / 2.0:
            r[d] = 2
    if r[0] * r[1] * r[2] == 1:
        r = [2, 2, 2]
        if ndim == 2:
            r[2] = 1
    return tuple(r)


class Factory:
    """MappedAMRPoissonOpFactory (single AMR level so far): MGnewOp, Factory.cpp:363-702."""

    def __init__(self, domain, grids, dx, bc, Jgup, Jinv, alpha=0.0, beta=1.0, isDiagonal=True, ndim=3,
                 maxDepth=-1, precondIters=2, relaxMode=RELAX_LEVEL_GSRB, amrmg_eps=1e-6, dxCrse=None, cf=None):
        self.domain, self.grids, self.dx, self.bc = domain, list(grids), tuple(dx), bc
        self.alpha, self.beta, self.isDiagonal, self.ndim = alpha, beta, isDiagonal, ndim
        self.maxDepth, self.precondIters, self.relaxMode = maxDepth, precondIters, relaxMode
        self.amrmg_eps = amrmg_eps
        self.dxCrse, self.cf = dxCrse, cf
        self.maskedMaxCoarse = (S_MAX_COARSE, S_MAX_COARSE, S_MAX_COARSE if ndim == 3 else 1)
        lapDiag = LevelData(grids, 1, (0, 0, 0))
        fill_lap_diag(lapDiag, Jgup, Jinv, grids, dx, ndim)
        self.metrics = [(list(grids), Jgup, Jinv, lapDiag)]  # m_vvJgup[ref][depth] ...

    def mg_new_op(self, depth, allMGRefRatios, forceAll=None):
        if self.maxDepth >= 0 and depth > self.maxDepth:
            if forceAll is not None and depth <= len(forceAll):
                raise ValueError("You must make the maxDepth large enough to accomodate the mini V-cycles")
            return None
        ndim = self.ndim
        domain, dx = self.domain, list(self.dx)
        coarsening = [1, 1, 1]
        mgRefRatio = (1, 1, 1)
        mmc = self.maskedMaxCoarse
        forced = forceAll is not None and depth <= len(forceAll)
        for i in range(depth):
            if forced:
                mgRefRatio = tuple(forceAll[i])        # Factory.cpp:414-424: the mini V-cycle's coarsening pattern
            elif allMGRefRatios is not None and i < depth - 1:
                mgRefRatio = allMGRefRatios[i]
            else:
                mgRefRatio = choose_mg_ref_ratio(dx, ndim)
            domain = domain.coarsen(mgRefRatio)
            dx = [a * b for a, b in zip(dx, mgRefRatio)]
            coarsening = [a * b for a, b in zip(coarsening, mgRefRatio)]
        if forced and int(np.prod(coarsening)) > 1 and not coarsenable(self.grids, [a * b for a, b in zip(coarsening, mmc)]):
            raise ValueError("Could not coarsen grids for mini V-cycle. Your block factor needs to be at least %d"
                             % max(a * b for a, b in zip(coarsening, mmc)))
        if (not forced) and int(np.prod(coarsening)) > 1 and not coarsenable(self.grids, [a * b for a, b in zip(coarsening, mmc)]):
            # fallback, Factory.cpp:504-550
            domain = domain.refine(mgRefRatio)
            dx = [a / b for a, b in zip(dx, mgRefRatio)]
            coarsening = [a // b for a, b in zip(coarsening, mgRefRatio)]
            r = [1, 1, 1]
            for d in range(ndim):
                r[d] = 2
                if not coarsenable(self.grids, [a * b * c for a, b, c in zip(coarsening, mmc, r)]):
                    r[d] = 1
            if r[0] * r[1] * r[2] == 1:
                return None
            refDir = 0
            while refDir < ndim and r[refDir] != 1:
                refDir += 1
            aspect = [x / dx[refDir] for x in dx]
            for d in range(ndim):
                if r[d] > 1 and aspect[d] > 0.5:
                    r[d] = 1
            if r[0] * r[1] * r[2] == 1:
                return None
            mgRefRatio = tuple(r)
            domain = domain.coarsen(mgRefRatio)
            dx = [a * b for a, b in zip(dx, mgRefRatio)]
            coarsening = [a * b for a, b in zip(coarsening, mgRefRatio)]
            if int(np.prod(coarsening)) > 1 and not coarsenable(self.grids, [a * b for a, b in zip(coarsening, mmc)]):
                return None
        if allMGRefRatios is not None and depth > 0:
            allMGRefRatios.append(tuple(mgRefRatio))
        # validateMetricPtrs / fill_MGfields
        while len(self.metrics) <= depth:
            fg, fJg, fJi, _ = self.metrics[-1]
            cg, cJg, cJi = coarsen_metric(fg, fJg, fJi, mgRefRatio, ndim)
            lap = LevelData(cg, 1, (0, 0, 0))
            fill_lap_diag(lap, cJg, cJi, cg, dx, ndim)
            self.metrics.append((cg, cJg, cJi, lap))
        grids, Jgup, Jinv, lapDiag = self.metrics[depth]
        cf = self.cf.coarsen(coarsening) if self.cf is not None else None
        op = PoissonOp(grids, domain, dx, self.bc, Jgup, Jinv, lapDiag, self.alpha, self.beta, self.isDiagonal,
                       ndim, self.dxCrse, cf, self.relaxMode, self.precondIters)
        op.mgDepth = depth
        # null-space probe, Factory.cpp:659-693
        phi = LevelData(grids, 1, op.activeDirs)
        rhs = LevelData(grids, 1, (0, 0, 0))
        res = LevelData(grids, 1, (0, 0, 0))
        ld_set(phi, 0.0)
        op.apply_op(rhs, phi, True)
        ld_set(phi, 1.0)
        op.residual(res, phi, rhs, True)
        maxNorm = abs(max(float(np.max(f.view(g))) for g, f in zip(res.grids, res.fabs)))
        op.zeroAvg = maxNorm < 0.01 * self.amrmg_eps
        return op


# ----------------------------------------------------------------------------
# Chombo 3.1 BiCGStabSolver<LevelData<FArrayBox>> (EXTERNAL, restated from the
# published Chombo 3.1 source lib/src/AMRElliptic/BiCGStabSolver.H; configured at
# projection/AMRPressureSolver.cpp:253-265, defaults utils/ProblemContext.cpp:1207-1231)
# ----------------------------------------------------------------------------
class BiCGStab:
    def __init__(self, imax=80, eps=1e-6, reps=1e-12, hang=1e-15, small=1e-30, numRestarts=5, normType=2,
                 verbosity=0):
        self.imax, self.eps, self.reps, self.hang, self.small = imax, eps, reps, hang, small
        self.numRestarts, self.normType, self.verbosity = numRestarts, normType, verbosity
        self.convergenceMetric = -1.0
        self.homogeneous = True
        self.op = None
        self.exitStatus = 0
        self.iters = 0

    def define(self, op, homogeneous):
        self.op, self.homogeneous = op, homogeneous

    def set_convergence_metrics(self, metric, tol):
        self.convergenceMetric, self.eps = metric, tol

    def solve(self, phi, rhs):
        op = self.op
        r, r_tilde = ld_create(rhs), ld_create(rhs)
        e, p, p_tilde, s_tilde = ld_create(phi), ld_create(phi), ld_create(phi), ld_create(phi)
        t, v = ld_create(rhs), ld_create(rhs)
        recount = 0
        op.residual(r, phi, rhs, self.homogeneous)
        ld_assign(r_tilde, r)
        ld_set(e, 0.0)
        ld_set(p_tilde, 0.0)
        ld_set(s_tilde, 0.0)
        i = 0
        rho = [0.0, 0.0, 0.0, 0.0]
        norm = [ld_norm(r, self.normType), 0.0]
        initial_norm = norm[0]
        initial_rnorm = norm[0]
        norm[1] = norm[0]
        alpha, beta, omega = [0.0, 0.0], [0.0, 0.0], [0.0, 0.0]
        init = True
        restarts = 0
        if self.convergenceMetric > 0:
            initial_norm = self.convergenceMetric
        self.exitStatus = -1
        while (i < self.imax and norm[0] > self.eps * norm[1]) and (norm[1] > 0):
            i += 1
            norm[1] = norm[0]
            alpha[1], beta[1], omega[1] = alpha[0], beta[0], omega[0]
            rho[3], rho[2] = rho[2], rho[1]
            rho[1] = ld_dot(r_tilde, r)
            if rho[1] == 0.0:
                ld_incr(phi, e, 1.0)
                self.exitStatus = 2
                self.iters = i
                return
            if init:
                ld_assign(p, r)
                init = False
            else:
                beta[1] = (rho[1] / rho[2]) * (alpha[1] / omega[1])
                ld_scale(p, beta[1])
                ld_incr(p, v, -beta[1] * omega[1])
                ld_incr(p, r, 1.0)
            op.pre_cond(p_tilde, p)
            op.apply_op(v, p_tilde, True)
            m = ld_dot(r_tilde, v)
            alpha[0] = rho[1] / m
            if abs(m) > self.small * abs(rho[1]):
                ld_incr(r, v, -alpha[0])
                norm[0] = ld_norm(r, self.normType)
                ld_incr(e, p_tilde, alpha[0])
            else:
                ld_set(r, 0.0)
                norm[0] = 0.0
            if norm[0] > self.eps * initial_norm and norm[0] > self.reps * initial_rnorm:
                op.pre_cond(s_tilde, r)
                op.apply_op(t, s_tilde, True)
                omega[0] = ld_dot(t, r) / ld_dot(t, t)
                ld_incr(e, s_tilde, omega[0])
                ld_incr(r, t, -omega[0])
                norm[0] = ld_norm(r, self.normType)
            if norm[0] <= self.eps * initial_norm or norm[0] <= self.reps * initial_rnorm:
                self.exitStatus = 1
                break
            if omega[0] == 0.0 or norm[0] > (1 - self.hang) * norm[1]:
                if recount == 0:
                    recount = 1
                else:
                    recount = 0
                    ld_incr(phi, e, 1.0)
                    if restarts == self.numRestarts:
                        self.exitStatus = 3
                        self.iters = i
                        return
                    op.residual(r, phi, rhs, self.homogeneous)
                    norm[0] = ld_norm(r, self.normType)
                    rho = [0.0, 0.0, 0.0, 0.0]
                    alpha[0] = beta[0] = omega[0] = 0.0
                    ld_assign(r_tilde, r)
                    ld_set(e, 0.0)
                    restarts += 1
                    init = True
        ld_incr(phi, e, 1.0)
        self.iters = i


# ----------------------------------------------------------------------------
# MappedMultiGrid / MappedAMRMultiGrid (single level)
# ----------------------------------------------------------------------------
class MultiGrid:
    """MappedMultiGrid<T>, AMRElliptic/MappedMultiGrid.H:328-404 (define), 421-434 (init),
    555-653 (cycle)."""

    def __init__(self, factory, bottomSolver, maxDepth=-1, pre=2, post=2, bottom=2, cycle=1, forceAllMGRefRatios=None):
        self.pre, self.post, self.bottom, self.cycle_type = pre, post, bottom, cycle
        self.ops = []
        self.mgRefRatios = []
        self.maxForcedDepth = len(forceAllMGRefRatios) if forceAllMGRefRatios is not None else 0
        nextOp = factory.mg_new_op(0, None)
        depth = 0
        while nextOp is not None:
            self.ops.append(nextOp)
            depth += 1
            if depth < maxDepth or maxDepth < 0:
                fineOp = nextOp
                nextOp = factory.mg_new_op(depth, self.mgRefRatios, forceAllMGRefRatios)
                if nextOp is not None:
                    fineOp.mgCrseRefRatio = self.mgRefRatios[-1]
            else:
                nextOp = None
        self.depth = depth
        self.bottomSolver = bottomSolver
        self.bottomSolver.define(self.ops[-1], True)
        self.bottomCells = self.ops[-1].domain.box.numPts()
        self.residual = [None] * depth
        self.correction = [None] * depth

    def init(self, e, residual):
        if self.depth > 1:
            self.residual[1] = self.ops[0].create_coarser(residual)
            self.correction[1] = self.ops[0].create_coarser(e)
        for i in range(2, self.depth):
            self.residual[i] = self.ops[i - 1].create_coarser(self.residual[i - 1])
            self.correction[i] = self.ops[i - 1].create_coarser(self.correction[i - 1])

    def one_cycle(self, e, residual):
        self.cycle(0, e, residual)  # m_homogeneous == true

    def cycle(self, depth, correction, residual):
        op = self.ops[depth]
        if depth == self.depth - 1:
            if self.bottomCells == 1:
                op.relax(correction, residual, 1)
            else:
                op.relax(correction, residual, self.bottom)
                self.bottomSolver.solve(correction, residual)
        else:
            cycles = self.cycle_type
            if cycles < 0:
                # F-cycle (:577-619): a recursive F-cycle first, pre-smoothing, then |m_cycle| V-cycles, post-smoothing
                cycles = -cycles
                op.restrict_residual(self.residual[depth + 1], correction, residual)
                ld_set(self.correction[depth + 1], 0.0)
                self.cycle(depth + 1, self.correction[depth + 1], self.residual[depth + 1])
                op.prolong_increment(correction, self.correction[depth + 1])
                op.relax(correction, residual, self.pre)
                for _ in range(cycles):
                    op.restrict_residual(self.residual[depth + 1], correction, residual)
                    ld_set(self.correction[depth + 1], 0.0)
                    self.cycle_type = 1            # "hack to get a V-cycle"
                    self.cycle(depth + 1, self.correction[depth + 1], self.residual[depth + 1])
                    self.cycle_type = -cycles
                    op.prolong_increment(correction, self.correction[depth + 1])
                op.relax(correction, residual, self.post)
                return
            op.relax(correction, residual, self.pre)
            op.restrict_residual(self.residual[depth + 1], correction, residual)
            ld_set(self.correction[depth + 1], 0.0)
            for _ in range(cycles):
                self.cycle(depth + 1, self.correction[depth + 1], self.residual[depth + 1])
            op.prolong_increment(correction, self.correction[depth + 1])
            op.relax(correction, residual, self.post)


class AMRMultiGrid:
    """MappedAMRMultiGrid<T> restricted to l_base == l_max == 0:
    solveNoInitResid, AMRElliptic/MappedAMRMultiGrid.H:979-1183; defaults :700-717."""

    def __init__(self, factory, bottomSolver, maxDepth=-1):
        self.eps, self.hang, self.normThresh = 1e-6, 1e-15, 1e-30
        self.imin, self.iterMax = 5, 20
        self.pre = self.post = self.bottom = 2
        self.numMG = 1
        self.convergenceMetric = 0.0
        self.bottomSolverEpsCushion = 1.0
        self.bottomSolver = bottomSolver
        self.mg = MultiGrid(factory, bottomSolver, maxDepth)
        self.op = self.mg.ops[0]
        self.exitStatus = 0
        self.history = []
        self.iters = 0

    def set_solver_parameters(self, pre, post, bottom, numMG, iterMax, eps, hang, normThresh):
        """setSolverParameters, MappedAMRMultiGrid.H:622-649."""
        self.pre, self.post, self.bottom, self.numMG = pre, post, bottom, numMG
        self.iterMax, self.eps, self.hang, self.normThresh = iterMax, eps, hang, normThresh
        self.mg.pre, self.mg.post, self.mg.bottom, self.mg.cycle_type = pre, post, bottom, numMG

    def compute_residual(self, resid, phi, rhs, homogeneous):
        self.op.residual(resid, phi, rhs, homogeneous)
        return self.op.local_max_norm(resid)

    def solve(self, phi, rhs, zeroPhi=True, forceHomogeneous=False):
        op = self.op
        self.mg.pre, self.mg.post, self.mg.bottom = self.pre, self.post, self.bottom
        self.mg.init(phi, rhs)
        uberCorrection = ld_create(phi)
        uberResidual = ld_create(rhs)
        bestPhi = ld_create(phi)
        ld_set(uberResidual, 0.0)
        ld_set(uberCorrection, 0.0)
        if zeroPhi:
            ld_set(phi, 0.0)
        ld_assign(bestPhi, phi)
        initial_rnorm = self.compute_residual(uberResidual, phi, rhs, forceHomogeneous)
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
            self.mg.one_cycle(uberCorrection, uberResidual)  # AMRVCycle, l_max == l_base (:1511-1514)
            # postVCycleOps (:1189-1215)
            ld_incr(phi, uberCorrection, 1.0)
            ld_set(uberCorrection, 0.0)
            rnorm = self.compute_residual(uberResidual, phi, rhs, forceHomogeneous)
            it += 1
            self.history.append(rnorm)
            if rnorm <= best_rnorm:
                best_rnorm = rnorm
                ld_assign(bestPhi, phi)
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
            ld_assign(phi, bestPhi)
        if rnorm > 10.0 * initial_rnorm and rnorm > 10.0 * self.eps:
            raise RuntimeError("kaboom")  # MayDay::Error (:1136)
        if (not somethingConverged) and rnorm >= initial_rnorm and rnorm >= self.eps:
            raise RuntimeError("MappedAMRMultiGrid solver blew up")  # (:1144)
        self.exitStatus = int(not goRedu) + int(not goIter) * 2 + int(not goHang) * 4 + int(not goNorm) * 8
        self.iters = it
        self.final_rnorm = rnorm
        self.initial_rnorm = initial_rnorm
        return rnorm


# ----------------------------------------------------------------------------
# Synthetic inputs shared by tests and bench
# ----------------------------------------------------------------------------
def split_domain(domainBox, boxSize):
    bs = _iv(boxSize)
    n = domainBox.size()
    out = []
    for k in range(domainBox.lo[2], domainBox.hi[2] + 1, bs[2]):
        for j in range(domainBox.lo[1], domainBox.hi[1] + 1, bs[1]):
            for i in range(domainBox.lo[0], domainBox.hi[0] + 1, bs[0]):
                out.append(Box((i, j, k), (min(i + bs[0], domainBox.lo[0] + n[0]) - 1,
                                           min(j + bs[1], domainBox.lo[1] + n[1]) - 1,
                                           min(k + bs[2], domainBox.lo[2] + n[2]) - 1)))
    return out


def stretch_factor(a, x, L):
    return 1.0 + 0.3 * np.sin(2.0 * np.pi * x / L + a)


def make_diagonal_metric(grids, dx, L, ndim=3, variant="stretched", domain=None):
    """C2 metric variants: 'cartesian' (all ones, CartesianMap.cpp:261-280) or the
    separable stretch s_a = 1 + 0.3 sin(2 pi x_a/L_a + a): Jg^{aa} = s_b s_c / s_a at
    a-faces, Jinv = 1/(s_0 s_1 s_2) at cell centres.  Off-diagonal comps are 0.
    With `domain`, indices are wrapped in periodic directions before the map is evaluated,
    so the two stored copies of a periodic face (index 0 and index n) are bitwise equal."""
    Jgup = FluxData(grids, 3 if ndim == 3 else 2, ndim)
    Jinv = LevelData(grids, 1, (0, 0, 0), 1.0)
    for i, g in enumerate(grids):
        if variant == "cartesian":
            for d in range(ndim):
                Jgup[i][d].a[..., d] = 1.0
            continue

        def coords(box, faceDir):
            xs = []
            for d in range(3):
                idx = np.arange(box.lo[d], box.hi[d] + 1, dtype=np.float64)
                if domain is not None and domain.periodic[d]:
                    nd = domain.box.size()[d]
                    idx = np.mod(idx - domain.box.lo[d], nd) + domain.box.lo[d]
                xs.append((idx if d == faceDir else idx + 0.5) * dx[d])
            return xs

        def s(a, x):
            return stretch_factor(a, x, L[a]) if a < ndim else np.ones_like(x)

        for d in range(ndim):
            fb = Jgup[i][d].box
            x = coords(fb, d)
            s0, s1, s2 = s(0, x[0])[:, None, None], s(1, x[1])[None, :, None], s(2, x[2])[None, None, :]
            sv = [s0, s1, s2]
            num = np.ones(fb.size())
            for e in range(3):
                if e != d:
                    num = num * sv[e]
            Jgup[i][d].a[..., d] = num / sv[d]
        x = coords(g, -1)
        Jinv[i].a[..., 0] = 1.0 / (s(0, x[0])[:, None, None] * s(1, x[1])[None, :, None] * s(2, x[2])[None, None, :])
    return Jgup, Jinv


def make_full_metric(grids, dx, L, domain, amp=(0.25, 0.2, 0.15), variant="sheared"):
    """A smooth NON-orthogonal map for the 19-point path (SURVEY config C5's role, synthetic), k_a = 2 pi / L_a:
        x = xi + a0 sin(k1 eta)/k1,  y = eta + a1 sin(k2 zeta)/k2,  z = zeta + a2 sin(k0 xi)/k0
    ('sheared': a_i = largest shear dx_i/dxi_j), or the constant skew x = xi + a0 eta, y = eta + a1 zeta, z = zeta
    ('skew', J = 1).
    J g^{ab} = J (F^-1 F^-T)^{ab} with F = d x / d xi, evaluated at face centres (all 3 comps per face direction,
    the layout of LevelGeometry::getFCJgup), Jinv = 1/det F at cell centres.  Indices wrap in periodic directions
    so that the two stored copies of a periodic face are bitwise equal."""
    Jgup = FluxData(grids, 3, 3)
    Jinv = LevelData(grids, 1, (0, 0, 0), 1.0)
    k = [2.0 * np.pi / L[d] for d in range(3)]

    def coords(box, faceDir):
        xs = []
        for d in range(3):
            idx = np.arange(box.lo[d], box.hi[d] + 1, dtype=np.float64)
            if domain.periodic[d]:
                nd = domain.box.size()[d]
                idx = np.mod(idx - domain.box.lo[d], nd) + domain.box.lo[d]
            xs.append((idx if d == faceDir else idx + 0.5) * dx[d])
        return np.meshgrid(*xs, indexing="ij")

    def jac(X):
        F = np.zeros(X[0].shape + (3, 3))
        for d in range(3):
            F[..., d, d] = 1.0
        if variant == "skew":
            F[..., 0, 1] = amp[0]
            F[..., 1, 2] = amp[1]
        else:
            F[..., 0, 1] = amp[0] * np.cos(k[1] * X[1])
            F[..., 1, 2] = amp[1] * np.cos(k[2] * X[2])
            F[..., 2, 0] = amp[2] * np.cos(k[0] * X[0])
        return F

    for i, g in enumerate(grids):
        for d in range(3):
            F = jac(coords(Jgup[i][d].box, d))
            Fi = np.linalg.inv(F)
            J = np.linalg.det(F)
            G = np.einsum("...ai,...bi->...ab", Fi, Fi) * J[..., None, None]
            for b in range(3):
                Jgup[i][d].a[..., b] = G[..., d, b]
        Jinv[i].a[..., 0] = 1.0 / np.linalg.det(jac(coords(g, -1)))
    return Jgup, Jinv


def make_terrain_metric(grids, dx, L, domain):
    """BASELINE config C5's terrain-following (bathymetric) map, SURVEY.md 8d: x = xi, y = eta,
    z = d + (1 - d/H) zeta (geometry/BathymetricBaseMapF.ChF:85-110) over the depth H s(x, y),
    s = 0.5 + 0.3 exp(-r^2/w^2), r measured from the domain centre, w = min(L0, L1)/4, H = L2:
    J = s, J g = [[s, 0, -z_xi], [0, s, -z_eta], [-z_xi, -z_eta, (1 + z_xi^2 + z_eta^2)/s]] with z_xi = s_x (zeta - H).
    Non-periodic directions only (a bump is not periodic).  The product-side generator somar_amd/synthetic.py::terrain_metric
    evaluates the same expressions; tests/test_synthetic.py compares them bit for bit."""
    assert not any(domain.periodic)
    Jgup = FluxData(grids, 3, 3)
    Jinv = LevelData(grids, 1, (0, 0, 0), 1.0)
    H = L[2]
    w = min(L[0], L[1]) / 4.0

    def coords(box, faceDir):
        xs = []
        for d in range(3):
            idx = np.arange(box.lo[d], box.hi[d] + 1, dtype=np.float64)
            xs.append((idx if d == faceDir else idx + 0.5) * dx[d])
        return np.meshgrid(*xs, indexing="ij")

    def fields(X):
        ex = np.exp(-((X[0] - 0.5 * L[0]) ** 2 + (X[1] - 0.5 * L[1]) ** 2) / (w * w))
        s = 0.5 + 0.3 * ex
        sx = -0.3 * ex * (2.0 * (X[0] - 0.5 * L[0]) / (w * w))
        sy = -0.3 * ex * (2.0 * (X[1] - 0.5 * L[1]) / (w * w))
        return s, sx * (X[2] - H), sy * (X[2] - H)

    for i, g in enumerate(grids):
        for d in range(3):
            s, zx, zy = fields(coords(Jgup[i][d].box, d))
            a = Jgup[i][d].a
            a[...] = 0.0
            if d == 0:
                a[..., 0], a[..., 2] = s, -zx
            elif d == 1:
                a[..., 1], a[..., 2] = s, -zy
            else:
                a[..., 0], a[..., 1], a[..., 2] = -zx, -zy, (1.0 + zx * zx + zy * zy) / s
        s, _, _ = fields(coords(g, -1))
        Jinv[i].a[..., 0] = 1.0 / s
    return Jgup, Jinv


def make_full_metric_2d(grids, dx, L, domain, amp=(0.25, 0.2)):
    """2-D counterpart of make_full_metric: x = xi + a0 sin(k1 eta)/k1, y = eta + a1 sin(k0 xi)/k0 on boxes one cell
    thick in z.  J g^{ab} (2 comps per face direction), Jinv = 1/det F."""
    Jgup = FluxData(grids, 2, 2)
    Jinv = LevelData(grids, 1, (0, 0, 0), 1.0)
    k = [2.0 * np.pi / L[d] for d in range(2)]

    def coords(box, faceDir):
        xs = []
        for d in range(2):
            idx = np.arange(box.lo[d], box.hi[d] + 1, dtype=np.float64)
            if domain.periodic[d]:
                nd = domain.box.size()[d]
                idx = np.mod(idx - domain.box.lo[d], nd) + domain.box.lo[d]
            xs.append((idx if d == faceDir else idx + 0.5) * dx[d])
        return np.meshgrid(*xs, indexing="ij")

    def jac(X):
        F = np.zeros(X[0].shape + (2, 2))
        F[..., 0, 0] = F[..., 1, 1] = 1.0
        F[..., 0, 1] = amp[0] * np.cos(k[1] * X[1])
        F[..., 1, 0] = amp[1] * np.cos(k[0] * X[0])
        return F

    for i, g in enumerate(grids):
        for d in range(2):
            F = jac(coords(Jgup[i][d].box, d))
            Fi = np.linalg.inv(F)
            J = np.linalg.det(F)
            G = np.einsum("...ai,...bi->...ab", Fi, Fi) * J[..., None, None]
            for b in range(2):
                Jgup[i][d].a[:, :, 0, b] = G[..., d, b]
        Jinv[i].a[:, :, 0, 0] = 1.0 / np.linalg.det(jac(coords(g, -1)))
    return Jgup, Jinv


def random_field(grids, seed, ghost=(0, 0, 0), domainBox=None):
    """uniform(-1,1) keyed by GLOBAL cell index so the field is independent of the box layout."""
    ld = LevelData(grids, 1, ghost)
    for g, f in zip(grids, ld.fabs):
        if domainBox is None:
            rng = np.random.default_rng(seed + hash(g) % 1000)
            f.view(g)[..., 0] = rng.uniform(-1.0, 1.0, g.size())
        else:
            n = domainBox.size()
            I, J, K = np.meshgrid(*[np.arange(g.lo[d], g.hi[d] + 1, dtype=np.uint64) for d in range(3)], indexing="ij")
            lin = (I - np.uint64(domainBox.lo[0])) + np.uint64(n[0]) * ((J - np.uint64(domainBox.lo[1])) + np.uint64(n[1]) * (K - np.uint64(domainBox.lo[2])))
            f.view(g)[..., 0] = hash_uniform(lin, seed)
    return ld


def hash_uniform(lin, seed):
    """splitmix64 of (cell index, seed) -> uniform(-1,1); cheap, layout independent, reproducible
    in C++/HIP (same integer recipe is used by bench.py for device-side fills)."""
    with np.errstate(over="ignore"):
        z = lin.astype(np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 / 9007199254740992.0) - 1.0


def remove_weighted_mean(rhs, Jinv):
    """Make a Neumann/periodic rhs compatible: subtract its J-weighted mean."""
    num = den = 0.0
    for g, f, j in zip(rhs.grids, rhs.fabs, Jinv.fabs):
        w = 1.0 / j.view(g)
        num += float(np.sum(f.view(g) * w))
        den += float(np.sum(w))
    m = num / den
    for g, f in zip(rhs.grids, rhs.fabs):
        f.view(g)[...] -= m
    return m


# ----------------------------------------------------------------------------
# MAC level projection (single level, velocity given in flux form J*u):
#   BaseProjector<FluxBox>::project           projection/BaseProjectorI.H:176-299
#   LevelMACProjector::computeDiv/computeGrad/applyCorrection   projection/LevelMACProjector.cpp:156-241
#   Divergence::levelDivergenceMAC            calculus/DivCurlGrad/Divergence.cpp:44-127   (no flux BC object:
#                                             boundary faces are taken as given)
#   Gradient::levelGradientMAC + singleBoxMacGrad   calculus/DivCurlGrad/Gradient.cpp:85-206, 946-1101
#   gradient BC = order-2 extrapolated ghosts  BCutil/PhysBCUtil.cpp:1432-1443
# ----------------------------------------------------------------------------
def level_divergence_mac(div, vel, Jinv, grids, dx, ndim=3):
    for i, g in enumerate(grids):
        lo, hi = _b(g)
        if ndim == 3:
            lib().orc_mappedfluxdivergence3d(*div[i].fra(), *vel[i][0].fran(), *vel[i][1].fran(), *vel[i][2].fran(),
                                             *Jinv[i].fra1(0), lo, hi, _rv(dx))
        else:
            lib().orc_mappedfluxdivergence2d(*div[i].fra(), *vel[i][0].fran(), *vel[i][1].fran(),
                                             *Jinv[i].fra1(0), lo, hi, _rv(dx))


def set_extrap_ghosts(phiF, valid, domain, order, ndim=3):
    """EllipticExtrapBCGhostClass::operator() -> setSideExtrapBC (cell-centred branch),
    BCInterface/EllipticBCUtils.cpp:1011-1050, 224-312."""
    for d in range(ndim):
        if domain.periodic[d]:
            continue
        for side in (0, 1):
            vend = valid.lo[d] if side == 0 else valid.hi[d]
            dend = domain.box.lo[d] if side == 0 else domain.box.hi[d]
            if vend != dend:
                continue
            dest = valid.adjCell(d, side, 1) & phiF.box
            if dest.isEmpty():
                continue
            lo, hi = _b(dest)
            rc = lib().orc_ellipticextrapbcghost(*phiF.fra(), lo, hi, d, 1 if side else -1, order)
            assert rc == 0


def level_gradient_mac(grad, phi, grids, domain, Jgup, dx, ndim=3, op=None):
    """one-component (normal) MAC gradient.  Diagonal metric: MAPPEDMACGRADORTHO.  Non-diagonal (pass the level
    operator): singleBoxMacGrad's sequence (Gradient.cpp:946-1101) -- extrap := fillExtrap-type extrapolation of phi
    (order 2, from the FAB clipped to the valid domain) BEFORE the gradient BC fills phi's own ghosts, then MAPPEDMACGRAD
    (DivCurlGradF.ChF:87-155), whose normal branch is MAPPEDGETFLUX with beta = 1 term for term."""
    exchange(phi, domain, phi.ghost)
    if op is not None and not op.isDiagonal:
        for i, g in enumerate(grids):
            extrap = Fab(phi[i].box, phi.ncomp, np.nan)
            op.fill_extrap(extrap, phi[i], 2)
            set_extrap_ghosts(phi[i], g, domain, 2, ndim)
            for d in range(ndim):
                op.get_flux_complete(grad[i][d], phi[i], extrap, g.faces(d), i, d)
        return
    for i, g in enumerate(grids):
        set_extrap_ghosts(phi[i], g, domain, 2, ndim)
        for d in range(ndim):
            eb = g.faces(d)
            lo, hi = _b(eb)
            lib().orc_mappedmacgradortho(*grad[i][d].fra1(0), *phi[i].fra1(0), *phi[i].fra1(0), *Jgup[i][d].fran(),
                                         lo, hi, C.c_double(dx[d]), d, d)


def mac_level_project(amr, vel, phi, dt, zeroPhi=True, ndim=3):
    """vel: FluxData (1 comp) holding J*u on faces; projected in place.  Returns the solve's final rnorm."""
    op = amr.op
    rhs = LevelData(op.grids, 1, (0, 0, 0))
    level_divergence_mac(rhs, vel, op.Jinv, op.grids, op.dx, ndim)
    if dt != 0.0:
        for f in rhs.fabs:
            f.a /= dt
    amr.solve(phi, rhs, zeroPhi=zeroPhi)
    corr = FluxData(op.grids, 1, ndim)
    level_gradient_mac(corr, phi, op.grids, op.domain, op.Jgup, op.dx, ndim, op=op)
    dtScale = -1.0 if dt == 0.0 else -dt
    for i in range(len(op.grids)):
        for d in range(ndim):
            vel[i][d].a += dtScale * corr[i][d].a   # FArrayBox::plus(src, scale)
    return rhs


# ----------------------------------------------------------------------------
# Cell-centred level projection (single level; velocity in flux form J*u, SpaceDim comps, >= 1 ghost layer filled by
# the caller -- the reference does not exchange the velocity either):
#   BaseProjector<FArrayBox>::project            projection/BaseProjectorI.H:176-299
#   LevelCCProjector::computeDiv/computeGrad/applyCorrection   projection/LevelCCProjector.cpp:163-255
#   Divergence::levelDivergenceCC, the branch built by default (USE_SIMPLE_STENCIL is commented out, Divergence.cpp:37):
#       CellToEdge, then levelDivergenceMAC with the velocity BC        Divergence.cpp:361-396, 44-127
#   velocity BC on the averaged faces: uStarFuncBC -> basicVelFuncBC -> BasicVelocityBCGhostClass with no inflow /
#       outflow side (BCutil/PhysBCUtil.cpp:793-801, 1261-1276): every non-periodic side is a solid wall, and both
#       its viscous and inviscid branches call setSideDiriBC(0) for the wall-normal component
#       (EllipticBCUtils.cpp:1284-1327), which on a face-centred FAB sets the boundary faces directly (:96-100)
#       => the normal flux on physical boundary faces is 0.
#   Gradient::levelGradientCC, default branch: levelGradientMAC, then EdgeToCell    Gradient.cpp:469-495
#   CellToEdge / EdgeToCell are Chombo 3.1 (EXTERNAL, not under /root/reference); restated from the published source:
#       edge_d(i) = half*(cell_d(i) + cell_d(i - e_d)) on the d-faces both of whose cells lie in the cell FAB,
#       cell_d(i) = half*(edge_d(i) + edge_d(i + e_d)) on the valid box.   No reference fixture => parity unpinned.
# ----------------------------------------------------------------------------
def cell_to_edge(cc, edge, ndim=3):
    """edge (1 comp) <- cc (comp d feeds the d-faces)"""
    for i in range(len(cc.grids)):
        cf = cc[i]
        for d in range(ndim):
            eb = cf.box.faces(d).growDir(d, -1) & edge[i][d].box
            if eb.isEmpty():
                continue
            sh = [0, 0, 0]
            sh[d] = -1
            hi_cells = eb                       # face i <-> cell i
            lo_cells = eb.shift(sh)             # ... and cell i - e_d
            edge[i][d].view(eb, 0)[...] = 0.5 * (cf.view(hi_cells, d) + cf.view(lo_cells, d))


def set_wall_normal_flux(edge, grids, domain, ndim=3):
    """solid walls: BasicVelocityBCGhostClass -> setSideDiriBC(0) on the FC normal faces"""
    for i, g in enumerate(grids):
        for d in range(ndim):
            if domain.periodic[d]:
                continue
            fb = g.faces(d)
            if g.lo[d] == domain.box.lo[d]:
                edge[i][d].view(fb.edgeCells(d, 0))[...] = 0.0
            if g.hi[d] == domain.box.hi[d]:
                edge[i][d].view(fb.edgeCells(d, 1))[...] = 0.0


def set_normal_flux_bc(edge, grids, domain, kind, value, ndim=3):
    """BasicVelocityBCGhostClass with inflow / outflow sides on the FC normal faces (EllipticBCUtils.cpp:1244-1327):
    kind[2*d + side] = 0 solid wall -> setSideDiriBC(0); 1 -> setSideDiriBC(value[2*d + side]) (both set the boundary faces of
    a face-centred FAB directly, :96-100); 2 outflow -> setSideExtrapBC order 0 = ELLIPTICEXTRAPBCGHOST on the boundary
    faces: state(i) = state(i + ii), the next face inside (EllipticBCUtilsF.ChF:148-154)"""
    for i, g in enumerate(grids):
        for d in range(ndim):
            if domain.periodic[d]:
                continue
            fb = g.faces(d)
            for side in (0, 1):
                if (g.lo[d] != domain.box.lo[d]) if side == 0 else (g.hi[d] != domain.box.hi[d]):
                    continue
                k, v = kind[2 * d + side], value[2 * d + side]
                dest = fb.edgeCells(d, side)
                if k == 2:
                    sh = [0, 0, 0]
                    sh[d] = 1 if side == 0 else -1
                    edge[i][d].view(dest)[...] = edge[i][d].view(dest.shift(sh))
                else:
                    edge[i][d].view(dest)[...] = v if k == 1 else 0.0


def edge_to_cell(edge, cc, ndim=3):
    for i, g in enumerate(cc.grids):
        for d in range(ndim):
            sh = [0, 0, 0]
            sh[d] = 1
            cc[i].view(g, d)[...] = 0.5 * (edge[i][d].view(g, 0) + edge[i][d].view(g.shift(sh), 0))


def level_divergence_cc(div, vel, Jinv, grids, domain, dx, ndim=3, wall=True, velbc=None):
    edge = FluxData(grids, 1, ndim)
    cell_to_edge(vel, edge, ndim)
    if wall and velbc is not None:
        set_normal_flux_bc(edge, grids, domain, velbc[0], velbc[1], ndim)
    elif wall:
        set_wall_normal_flux(edge, grids, domain, ndim)
    level_divergence_mac(div, edge, Jinv, grids, dx, ndim)
    return edge


def level_gradient_cc(grad, phi, grids, domain, Jgup, dx, ndim=3, op=None):
    edge = FluxData(grids, 1, ndim)
    level_gradient_mac(edge, phi, grids, domain, Jgup, dx, ndim, op=op)
    edge_to_cell(edge, grad, ndim)


def cc_level_project(amr, vel, phi, dt, zeroPhi=True, ndim=3, wall=True):
    """vel: LevelData (ndim comps, >= 1 ghost) holding J*u at cell centres; projected in place (valid cells).
    Returns the right-hand side of the solve."""
    op = amr.op
    rhs = LevelData(op.grids, 1, (0, 0, 0))
    level_divergence_cc(rhs, vel, op.Jinv, op.grids, op.domain, op.dx, ndim, wall)
    if dt != 0.0:
        for f in rhs.fabs:
            f.a /= dt
    amr.solve(phi, rhs, zeroPhi=zeroPhi)
    corr = LevelData(op.grids, ndim, (0, 0, 0))
    level_gradient_cc(corr, phi, op.grids, op.domain, op.Jgup, op.dx, ndim, op=op)
    dtScale = -1.0 if dt == 0.0 else -dt
    for i, g in enumerate(op.grids):
        vel[i].view(g)[...] += dtScale * corr[i].a   # FArrayBox::plus(src, scale) over the common box
    return rhs


# ----------------------------------------------------------------------------
# Viscous / diffusive Helmholtz solves through the same operator, single level:
#   MappedAMRPoissonOp::setAlphaAndBeta                        AMRElliptic/MappedAMRPoissonOp.cpp:582-619
#       m_alpha = a*aCoef, m_beta = b*bCoef with aCoef / bCoef the factory's alpha / beta (Factory.cpp:585-586); lapDiag
#       is refilled with the same values; the prolongation strategy chosen by the factory's null-space probe is kept
#   MappedBaseLevelHeatSolver::applyHelm / solveHelm / resetSolverAlphaAndBeta    AMRParabolic/MappedBaseLevelHeatSolver.cpp:154-270
#   MappedLevelBackwardEuler::updateSoln                       AMRParabolic/MappedLevelBackwardEuler.cpp:52-158
#   MappedLevelCrankNicolson::updateSoln                       AMRParabolic/MappedLevelCrankNicolson.cpp:52-152
#   MappedLevelTGA::updateSolnWithTimeIndependentOp            AMRParabolic/MappedLevelTGA.cpp:231-387
#   (diagonalScale / kappaScale are no-ops for this operator, MappedAMRPoissonOp.H:814-833; the flux-register
#   increments after the solve only matter with a second level and are not restated)
# ----------------------------------------------------------------------------
def reset_solver_alpha_and_beta(amr, a, b):
    for op in amr.mg.ops:
        if not hasattr(op, "aCoef"):
            op.aCoef, op.bCoef = op.alpha, op.beta      # what the factory handed over
        op.alpha = a * op.aCoef
        op.beta = b * op.bCoef


def level_backward_euler(amr, phiNew, phiOld, src, dt, zeroPhi=True):
    """(aCoef I - dt bCoef L) phiNew = phiOld.  The source does not enter (the reference comments its term out)."""
    phit = ld_create(phiNew)
    rhst = ld_create(src)
    ld_set(phit, 0.0)
    ld_set(rhst, 0.0)
    if zeroPhi:
        ld_set(phiNew, 0.0)
    ld_incr(phit, phiOld, 1.0)
    ld_incr(rhst, phit, 1.0)
    reset_solver_alpha_and_beta(amr, 1.0, -dt * 1.0)
    return amr.solve(phiNew, rhst, zeroPhi=zeroPhi)


def level_crank_nicolson(amr, phiNew, phiOld, src, dt, zeroPhi=True):
    """(I - dt/2 L) phiNew = dt src + (I + dt/2 L) phiOld, the explicit half with the inhomogeneous BCs"""
    phit = ld_create(phiNew)
    rhst = ld_create(src)
    ld_set(phit, 0.0)
    ld_set(rhst, 0.0)
    if zeroPhi:
        ld_set(phiNew, 0.0)
    reset_solver_alpha_and_beta(amr, 1.0, 0.5 * dt)
    amr.op.apply_op(phit, phiOld, False)
    ld_incr(rhst, src, dt)
    ld_incr(rhst, phit, 1.0)
    reset_solver_alpha_and_beta(amr, 1.0, -dt * 0.5)
    return amr.solve(phiNew, rhst, zeroPhi=zeroPhi)


def tga_coefficients():
    """MappedLevelTGA's constructor, AMRParabolic/MappedLevelTGA.cpp:30-56 -> (mu1, mu2, mu3, mu4, r1)"""
    tgaEpsilon = 1.e-12
    a = 2.0 - np.sqrt(2.0) - tgaEpsilon
    discr = np.sqrt(a * a - 4.0 * a + 2.0)
    return (float((a - discr) / 2.0), float((a + discr) / 2.0), float(1.0 - a), float(0.5 - a),
            float((2.0 * a - 1.0) / (a + discr)))


def level_tga(amr, phiNew, phiOld, src, dt, zeroPhi=True):
    """MappedLevelTGA::updateSolnWithTimeIndependentOp on one level (MappedLevelTGA.cpp:231-387):
    (I - mu1 dt L)(I - mu2 dt L) phiNew = (I + mu3 dt L) phiOld + (I + mu4 dt L) dt src, the source half with homogeneous
    BCs.  Returns the iteration counts of the two solves."""
    mu1, mu2, mu3, mu4, _ = tga_coefficients()
    op = amr.op
    rhst = ld_create(src)
    srct = ld_create(phiNew)
    phis = ld_create(phiNew)
    ld_set(srct, 0.0)
    ld_set(rhst, 0.0)
    ld_incr(srct, src, dt)
    if not zeroPhi:
        ld_set(phis, 0.0)
        ld_incr(phis, phiNew, 1.0)
    reset_solver_alpha_and_beta(amr, 1.0, mu4 * dt)                 # applyHelm(rhst, srct, mu4, homogeneous)
    op.apply_op(rhst, srct, True)
    reset_solver_alpha_and_beta(amr, 1.0, mu3 * dt)                 # applyHelm(phiNew, phiOld, mu3, inhomogeneous)
    op.apply_op(phiNew, phiOld, False)
    ld_incr(rhst, phiNew, 1.0)
    if not zeroPhi:
        ld_set(phiNew, 0.0)
        ld_incr(phiNew, phis, 1.0)
    reset_solver_alpha_and_beta(amr, 1.0, -dt * mu2)                # solveHelm(mu2)
    amr.solve(phiNew, rhst, zeroPhi=zeroPhi)
    it1 = amr.iters
    for d, f in zip(rhst.fabs, phiNew.fabs):                        # assign(rhst, phiNew)
        d.copy_from(f)
    if not zeroPhi:
        ld_set(phiNew, 0.0)
        ld_incr(phiNew, phis, 1.0)
    reset_solver_alpha_and_beta(amr, 1.0, -dt * mu1)                # solveHelm(mu1)
    amr.solve(phiNew, rhst, zeroPhi=zeroPhi)
    return it1, amr.iters


# ----------------------------------------------------------------------------
# AlteredMetric::fill_Jgup (projection/AlteredMetric.cpp:82-198): the whole-FAB statements after the map has been
# evaluated, one numpy statement per FArrayBox operation (same roundings).
# ----------------------------------------------------------------------------
def altered_jgup(nsq_fc, dximu_dz, dxinu_dz, gup, J, dt_theta, coriolis_f, hjac=None):
    theta = float(dt_theta)
    d = np.array(nsq_fc, dtype=np.float64)
    d = d * (theta * theta)
    tmp = d + 1.0
    d = d / tmp
    d = d * -1.0
    ftilde = float(coriolis_f) * theta
    ftildesq = ftilde * ftilde
    invfCoeff = 1.0 / (1.0 + ftildesq)
    d = d + ftildesq * invfCoeff
    d = d * np.asarray(dximu_dz)
    d = d * np.asarray(dxinu_dz)
    if hjac is not None:
        ix, jy, iy, jx = [np.asarray(a) for a in hjac]
        d = d + ftilde * invfCoeff * (ix * jy - iy * jx)
    d = d + np.asarray(gup) * invfCoeff
    return d * np.asarray(J)


# ----------------------------------------------------------------------------
# GeoSourceInterface's generic metric algebra: geometry/GeoSourceInterface.cpp
#   fill_dXidx (3-D, cofactors / det J) :200-291, fill_gup :373-415, fill_Jgup :417-450; ADDPROD2 / SUBPROD2
#   geometry/GeoSourceInterfaceF.ChF:251-300.  One numpy statement per whole-FAB statement of the reference.
# ----------------------------------------------------------------------------
def geo_fill_dXidx(dxdxi, detJ, mu, nu):
    """dxdxi[:, rho, sigma] = dx^rho / dXi^sigma"""
    mu1, mu2 = (nu + 1) % 3, (nu + 2) % 3
    nu1, nu2 = (mu + 1) % 3, (mu + 2) % 3
    d = np.zeros_like(detJ)
    d = d + dxdxi[:, mu1, nu1] * dxdxi[:, mu2, nu2]
    d = d - dxdxi[:, mu1, nu2] * dxdxi[:, mu2, nu1]
    return d / detJ


def geo_fill_jgup(dxdxi, detJ, mu, scale=1.0):
    """-> (n, 3): scale * J g^{mu nu}, nu = 0..2"""
    out = np.zeros((detJ.size, 3))
    for nu in range(3):
        g = np.zeros_like(detJ)
        for rho in range(3):
            g = g + geo_fill_dXidx(dxdxi, detJ, mu, rho) * geo_fill_dXidx(dxdxi, detJ, nu, rho)
        g = g * detJ
        if scale != 1.0:
            g = g * scale
        out[:, nu] = g
    return out
