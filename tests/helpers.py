"""Shared set-up for the parity tests: the SAME synthetic inputs go to the oracle (CPU restatement of
the reference) and, through the C ABI, to the HIP library."""
import numpy as np


def make_problem(so, n, boxsz, variant="stretched", periodic=(False, False, False), L=(1.0, 1.0, 1.0)):
    n = so._iv(n)
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), periodic)
    grids = so.split_domain(dom.box, boxsz)
    dx = tuple(L[d] / n[d] for d in range(3))
    Jgup, Jinv = so.make_diagonal_metric(grids, dx, L, 3, variant, domain=dom)
    return dom, grids, dx, Jgup, Jinv


def make_oracle_solver(so, dom, grids, dx, Jgup, Jinv, alpha=0.0, beta=1.0, pre=2, post=2, bottom=2, **kw):
    fac = so.Factory(dom, grids, dx, so.BCHolder(), Jgup, Jinv, alpha=alpha, beta=beta, **kw)
    amr = so.AMRMultiGrid(fac, so.BiCGStab())
    amr.pre, amr.post, amr.bottom = pre, post, bottom
    amr.mg.pre, amr.mg.post, amr.mg.bottom = pre, post, bottom
    return amr


def make_gpu_solver(dom, grids, dx, Jgup, Jinv, alpha=0.0, beta=1.0, pre=2, post=2, bottom=2, maxDepth=-1,
                    relaxMode=1, owner=None, comm=None, ndim=3, eps=None, bc_type=None, bc_values=None):
    from somar_amd import AMRPressureSolver
    s = AMRPressureSolver()
    p = s._p
    s.setSpaceDim(ndim)
    if eps is not None:
        p.eps = eps
    s.setAMRMGParameters(p.imin, p.imax, p.eps, maxDepth, p.num_smooth_precond, pre, post, bottom, p.precond_mode,
                         relaxMode, p.num_mg, p.hang, p.norm_thresh, 0)
    s.define(dom.box.lo, dom.box.hi, dom.periodic, dx, [(g.lo, g.hi) for g in grids], owner=owner, alpha=alpha,
             beta=beta, comm=comm, bc_type=bc_type)
    if bc_values is not None:
        s.setBCValues(bc_values)
    for p_ in range(s.num_local_patches):
        _, _, gi = s.patch_box(p_)
        jg = [np.asfortranarray(Jgup[gi][d].a[..., d]) for d in range(ndim)] + [None] * (3 - ndim)
        s.setMetricOrtho(p_, jg[0], jg[1], jg[2], np.asfortranarray(Jinv[gi].a[..., 0]))
    s.finalize()
    return s


def upload(s, field, ld, depth=0):
    for p in range(s.num_local_patches):
        _, _, gi = s.patch_box(p, depth)
        s.upload(field, p, np.asfortranarray(ld[gi].a[..., 0]), ld.ghost)


def download_valid(s, field, grids, depth=0):
    """-> list (per global box) of valid-region arrays"""
    out = [None] * len(grids)
    for p in range(s.num_local_patches):
        _, _, gi = s.patch_box(p, depth)
        out[gi] = s.download(field, p, (0, 0, 0), depth)
    return out


def valid_of(ld):
    return [f.view(g)[..., 0] for g, f in zip(ld.grids, ld.fabs)]


def max_rel_diff(a_list, b_list):
    num = max(float(np.max(np.abs(a - b))) for a, b in zip(a_list, b_list))
    den = max(float(np.max(np.abs(b))) for b in b_list)
    return num / den if den > 0 else num


def make_amr_levels(so, am, n, L, periodic, ratios, fine_boxes, variant="stretched", cbox=8, ndim=3):
    """A nested hierarchy: level 0 covers the domain (n cells, boxes of cbox), level l>0 is the list
    fine_boxes[l-1] given in level-l index space.  The metric is evaluated analytically at every level's own
    resolution, as the reference's LevelGeometry does."""
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), periodic)
    dx = tuple(L[d] / n[d] for d in range(3))
    grids = [so.split_domain(dom.box, cbox)] + [list(b) for b in fine_boxes]
    levels = []
    for l, g in enumerate(grids):
        if l > 0:
            dom = dom.refine(ratios[l - 1])
            dx = tuple(a / b for a, b in zip(dx, ratios[l - 1]))
        Jgup, Jinv = so.make_diagonal_metric(g, dx, L, ndim, variant=variant, domain=dom)
        levels.append(am.AMRLevel(dom, g, dx, Jgup, Jinv))
    return levels


def make_full_amr_levels(so, am, n, L, periodic, ratios, fine_boxes, cbox=8, ndim=3):
    """make_amr_levels with the sheared (non-diagonal) map evaluated at every level's own resolution"""
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), periodic)
    dx = tuple(L[d] / n[d] for d in range(3))
    grids = [so.split_domain(dom.box, cbox)] + [list(b) for b in fine_boxes]
    levels = []
    for l, g in enumerate(grids):
        if l > 0:
            dom = dom.refine(ratios[l - 1])
            dx = tuple(a / b for a, b in zip(dx, ratios[l - 1]))
        if ndim == 3:
            Jgup, Jinv = so.make_full_metric(g, dx, L, dom)
        else:
            Jgup, Jinv = so.make_full_metric_2d(g, dx, L[:2], dom)
        levels.append(am.AMRLevel(dom, g, dx, Jgup, Jinv))
    return levels


def make_gpu_amr(levels, ratios, alpha=0.0, beta=1.0, pre=2, post=2, bottom=2, maxDepth=-1, relaxMode=1, ndim=3,
                 full=False, imin=None, imax=None):
    """The same hierarchy (oracle AMRLevel list) on the GPU through the C ABI."""
    from somar_amd import AMRPressureSolver
    s = AMRPressureSolver()
    s.setSpaceDim(ndim)
    p = s._p
    s.setAMRMGParameters(p.imin if imin is None else imin, p.imax if imax is None else imax, p.eps, maxDepth,
                         p.num_smooth_precond, pre, post, bottom, p.precond_mode, relaxMode, p.num_mg, p.hang, p.norm_thresh, 0)
    L0 = levels[0]
    s.defineAMR(L0.domain.box.lo, L0.domain.box.hi, L0.domain.periodic, L0.dx, ratios,
                [[(g.lo, g.hi) for g in L.grids] for L in levels], alpha=alpha, beta=beta)
    for L, v in zip(levels, s.levels):
        for p_ in range(v.num_local_patches):
            _, _, gi = v.patch_box(p_)
            if full:
                jg = [np.asfortranarray(L.Jgup[gi][d].a) for d in range(ndim)] + [None] * (3 - ndim)
                v.setMetricFull(p_, jg[0], jg[1], jg[2], np.asfortranarray(L.Jinv[gi].a[..., 0]))
                continue
            jg = [np.asfortranarray(L.Jgup[gi][d].a[..., d]) for d in range(ndim)] + [None] * (3 - ndim)
            v.setMetricOrtho(p_, jg[0], jg[1], jg[2], np.asfortranarray(L.Jinv[gi].a[..., 0]))
    s.finalize()
    return s


def smooth_cc_velocity(so, dom, grids, ghost, ncomp=3):
    """a smooth cell field (ncomp comps) on valid + ghost cells, the same formula continued into every ghost cell"""
    vel = so.LevelData(grids, ncomp, ghost)
    n = dom.box.size()
    for f in vel.fabs:
        I, J, K = np.meshgrid(*[np.arange(f.box.lo[a], f.box.hi[a] + 1) for a in range(3)], indexing="ij")
        for d in range(ncomp):
            f.a[..., d] = (np.sin(2 * np.pi * (I + 0.5) / n[0] + 0.1 * d) * np.cos(2 * np.pi * (J + 0.5) / n[1] + 0.3)
                           * np.cos(2 * np.pi * (K + 0.5) / n[2] + d)) + 0.25 * d
    return vel
