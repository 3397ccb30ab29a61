"""The oracle's non-diagonal (19-point) path -- GSRBITER3D, GSRBBOUNDARYITER3D, MAPPEDGETFLUX, fillExtrap,
cross-term Neumann ghosts -- has no reference fixture either ("parity unpinned"); pinned here by
  * consistency: fed a DIAGONAL metric it reproduces the 7-point kernels (residual exactly, GSRB to rounding),
  * order of accuracy: on a constant-skew map the truncation error of L[cos cos cos] falls 4x per refinement,
  * constants are in the null space on a sheared map with Neumann walls."""
import numpy as np


def _prob(so, n, per, bs, L=(1.0, 1.0, 1.0)):
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), per)
    grids = so.split_domain(dom.box, bs)
    dx = tuple(L[d] / n[d] for d in range(3))
    return dom, grids, dx


def test_full_path_reduces_to_ortho_kernels_on_a_diagonal_metric(oracle):
    so = oracle
    dom, grids, dx = _prob(so, (16, 16, 16), (False, True, False), 8)
    Jg, Ji = so.make_diagonal_metric(grids, dx, (1, 1, 1), 3, "stretched", domain=dom)
    opD = so.Factory(dom, grids, dx, so.BCHolder(), Jg, Ji, isDiagonal=True, maxDepth=0).mg_new_op(0, None)
    opF = so.Factory(dom, grids, dx, so.BCHolder(), Jg, Ji, isDiagonal=False, maxDepth=0).mg_new_op(0, None)
    phi = so.random_field(grids, 3, (1, 1, 1), dom.box)
    rhs = so.random_field(grids, 4, (0, 0, 0), dom.box)
    p1, p2 = so.ld_create(phi), so.ld_create(phi)
    so.ld_assign(p1, phi)
    so.ld_assign(p2, phi)
    a, b = so.LevelData(grids, 1), so.LevelData(grids, 1)
    opD.residual(a, p1, rhs, True)
    opF.residual(b, p2, rhs, True)
    for g, x, y in zip(grids, a.fabs, b.fabs):
        np.testing.assert_array_equal(x.view(g), y.view(g))
    opD.relax(p1, rhs, 2)
    opF.relax(p2, rhs, 2)
    for g, x, y in zip(grids, p1.fabs, p2.fabs):
        np.testing.assert_allclose(x.view(g), y.view(g), rtol=0, atol=1e-14)


def test_full_operator_is_second_order_on_a_skewed_map(oracle):
    so = oracle
    A = np.array([[1, 0.3, 0], [0, 1, 0.2], [0, 0, 1.0]])
    Ai = np.linalg.inv(A)
    G = Ai @ Ai.T
    kk = 2 * np.pi
    errs = []
    for n in (16, 32):
        dom, grids, dx = _prob(so, (n, n, n), (True, True, True), n // 2)
        Jg, Ji = so.make_full_metric(grids, dx, (1, 1, 1), dom, amp=(0.3, 0.2, 0.0), variant="skew")
        op = so.Factory(dom, grids, dx, so.BCHolder(), Jg, Ji, isDiagonal=False, maxDepth=0).mg_new_op(0, None)
        phi, out = so.LevelData(grids, 1, (1, 1, 1)), so.LevelData(grids, 1)
        for f in phi.fabs:
            X = np.meshgrid(*[(np.arange(f.box.lo[d], f.box.hi[d] + 1) + 0.5) * dx[d] for d in range(3)], indexing="ij")
            f.a[..., 0] = np.cos(kk * X[0]) * np.cos(kk * X[1]) * np.cos(kk * X[2])
        op.apply_op(out, phi, True)
        worst = 0.0
        for g, o in zip(grids, out.fabs):
            X = np.meshgrid(*[(np.arange(g.lo[d], g.hi[d] + 1) + 0.5) * dx[d] for d in range(3)], indexing="ij")
            c = [np.cos(kk * X[d]) for d in range(3)]
            s = [np.sin(kk * X[d]) for d in range(3)]
            exact = 0.0
            for a_ in range(3):
                for b_ in range(3):
                    if a_ == b_:
                        exact = exact - G[a_, b_] * kk * kk * c[0] * c[1] * c[2]
                    else:
                        t = [c[0], c[1], c[2]]
                        t[a_], t[b_] = s[a_], s[b_]
                        exact = exact + G[a_, b_] * kk * kk * t[0] * t[1] * t[2]
            worst = max(worst, float(np.abs(o.view(g)[..., 0] - exact).max()))
        errs.append(worst)
    assert 3.6 < errs[0] / errs[1] < 4.4


def test_constants_are_in_the_null_space_with_cross_term_neumann_ghosts(oracle):
    so = oracle
    dom, grids, dx = _prob(so, (16, 16, 8), (False, True, False), 8, (2.0, 1.0, 0.5))
    Jg, Ji = so.make_full_metric(grids, dx, (2.0, 1.0, 0.5), dom)
    op = so.Factory(dom, grids, dx, so.BCHolder(), Jg, Ji, isDiagonal=False, maxDepth=0).mg_new_op(0, None)
    one, out = so.LevelData(grids, 1, (1, 1, 1)), so.LevelData(grids, 1)
    so.ld_set(one, 1.0)
    op.apply_op(out, one, True)
    assert so.ld_norm(out, 0) == 0.0
    assert op.zeroAvg


# ---- the same three pins for the 2-D (9-point) kernels: GSRBITER2D, GSRBBOUNDARYITER2D, MAPPEDGETFLUX (SpaceDim 2) ----
def _prob2(so, n, per, bs, L=(1.0, 1.0)):
    dom = so.Domain(so.Box((0, 0, 0), (n[0] - 1, n[1] - 1, 0)), (per[0], per[1], False))
    grids = so.split_domain(dom.box, (bs, bs, 1))
    dx = (L[0] / n[0], L[1] / n[1], 1.0)
    return dom, grids, dx


def test_2d_full_path_reduces_to_ortho_kernels_on_a_diagonal_metric(oracle):
    so = oracle
    dom, grids, dx = _prob2(so, (16, 16), (False, True), 8)
    Jg, Ji = so.make_diagonal_metric(grids, dx, (1, 1, 1), 2, "stretched", domain=dom)
    opD = so.Factory(dom, grids, dx, so.BCHolder(), Jg, Ji, isDiagonal=True, ndim=2, maxDepth=0).mg_new_op(0, None)
    opF = so.Factory(dom, grids, dx, so.BCHolder(), Jg, Ji, isDiagonal=False, ndim=2, maxDepth=0).mg_new_op(0, None)
    phi = so.random_field(grids, 3, (1, 1, 0), dom.box)
    rhs = so.random_field(grids, 4, (0, 0, 0), dom.box)
    p1, p2 = so.ld_create(phi), so.ld_create(phi)
    so.ld_assign(p1, phi)
    so.ld_assign(p2, phi)
    a, b = so.LevelData(grids, 1), so.LevelData(grids, 1)
    opD.residual(a, p1, rhs, True)
    opF.residual(b, p2, rhs, True)
    for g, x, y in zip(grids, a.fabs, b.fabs):
        np.testing.assert_array_equal(x.view(g), y.view(g))
    opD.relax(p1, rhs, 2)
    opF.relax(p2, rhs, 2)
    for g, x, y in zip(grids, p1.fabs, p2.fabs):
        np.testing.assert_allclose(x.view(g), y.view(g), rtol=0, atol=1e-14)


def test_2d_constants_in_null_space_and_multigrid_converges_on_a_sheared_map(oracle):
    so = oracle
    L = (2.0, 1.0)
    dom, grids, dx = _prob2(so, (32, 32), (False, False), 16, L)
    Jg, Ji = so.make_full_metric_2d(grids, dx, L, dom)
    fac = so.Factory(dom, grids, dx, so.BCHolder(), Jg, Ji, isDiagonal=False, ndim=2)
    amr = so.AMRMultiGrid(fac, so.BiCGStab())
    op = amr.op
    one, out = so.LevelData(grids, 1, (1, 1, 0)), so.LevelData(grids, 1)
    so.ld_set(one, 1.0)
    op.apply_op(out, one, True)
    assert so.ld_norm(out, 0) < 1e-12
    assert all(o.zeroAvg for o in amr.mg.ops)
    # compatible right-hand side: L[random]
    phi = so.random_field(grids, 5, (1, 1, 0), dom.box)
    rhs = so.LevelData(grids, 1)
    op.apply_op(rhs, phi, True)
    x = so.LevelData(grids, 1, (1, 1, 0))
    amr.solve(x, rhs)
    h = amr.history
    assert amr.exitStatus == 1 and h[-1] <= 1e-6 * h[0]
    assert amr.mg.depth >= 3


def test_nondiagonal_operator_depends_on_the_layout_where_a_box_edge_meets_a_wall(oracle):
    """Reference behaviour as restated (not fixed): setSideNeumBC extrapolates `extrap` with ExtrapolateFaceAndCopy from
    the box's OWN valid cells (EllipticBCUtils.cpp:128-214), edges included, so the corner ghost beyond a box edge that
    sits on a physical wall is extrapolated along the wall instead of holding the neighbour box's data.  The two copies
    of a face shared by two boxes then differ next to the wall: the non-diagonal operator is layout dependent in the
    two cells either side of every such junction (and only there), and exactly conservative on one box only."""
    so = oracle
    L = (2.0, 1.0)
    res = {}
    for bs in ((32, 16, 1), (16, 16, 1)):
        dom = so.Domain(so.Box((0, 0, 0), (31, 15, 0)), (False, False, False))
        grids = so.split_domain(dom.box, bs)
        dx = (L[0] / 32, L[1] / 16, 1.0)
        Jg, Ji = so.make_full_metric_2d(grids, dx, L, dom)
        op = so.Factory(dom, grids, dx, so.BCHolder(), Jg, Ji, isDiagonal=False, ndim=2, maxDepth=0).mg_new_op(0, None)
        phi = so.random_field(grids, 5, (1, 1, 0), dom.box)
        out = so.LevelData(grids, 1)
        op.apply_op(out, phi, True)
        full = np.zeros((32, 16))
        tot = 0.0
        for i, g in enumerate(grids):
            full[g.lo[0]:g.hi[0] + 1, g.lo[1]:g.hi[1] + 1] = out[i].view(g)[:, :, 0, 0]
            tot += float((out[i].view(g)[..., 0] / Ji[i].view(g)[..., 0]).sum())
        res[bs] = (full, tot)
    one, two = res[(32, 16, 1)], res[(16, 16, 1)]
    assert abs(one[1]) < 1e-9 * np.abs(one[0]).sum()           # one box: conservative
    differ = {tuple(q) for q in np.argwhere(np.abs(one[0] - two[0]) > 1e-9).tolist()}
    assert differ == {(15, 0), (15, 1), (15, 14), (15, 15), (16, 0), (16, 1), (16, 14), (16, 15)}
