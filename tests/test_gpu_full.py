"""Non-diagonal metric (19-point kernels) on the GPU vs the oracle's restatement of GSRBITER3D,
GSRBBOUNDARYITER3D, MAPPEDGETFLUX, fillExtrap / ExtrapolateFaceAndCopy and the cross-term Neumann ghost
(SURVEY.md rows a5/a6 full stencil, a9, a18, a19).  Synthetic sheared map (oracle.make_full_metric)."""
import numpy as np
import pytest

from helpers import download_valid, max_rel_diff, upload, valid_of

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["direct", "march", "fused"])
def kernel_path(request, monkeypatch):
    """direct: k_op_full / k_gsrb_full (small levels); march: the k-marching LDS kernels of large levels
    (full19_march.hip, psi kept in the boxes' frames only), forced onto these small cases.  Same bits either way."""
    monkeypatch.setenv("SOMAR_MARCH_MIN_CELLS", "0" if request.param != "direct" else "1000000000000")
    # fused: red + black in one marching launch three layers inside every box, then a shell pass (full19_fused.hip), forced onto
    # every level whose boxes are at least 8 cells wide (the default takes it on boxes of 192 and more)
    monkeypatch.setenv("SOMAR_FUSED19_MIN_BOX", "0" if request.param == "fused" else "-1")
    return request.param

CASES = [
    ((16, 16, 16), 8, (True, True, True), (1.0, 1.0, 1.0)),
    ((16, 16, 8), 8, (False, True, False), (2.0, 1.0, 0.5)),
    ((24, 16, 8), (12, 8, 8), (False, False, False), (1.5, 1.0, 0.5)),
    # tile columns of every lane class of the marching kernels (Level::build_march_tiles): 64 = 60 (two region rows per
    # wavefront) + 4 (sixteen); 128 = 124 (one row per wavefront) + 4
    ((64, 8, 8), (64, 8, 8), (False, True, False), (2.0, 1.0, 0.5)),
    ((128, 6, 4), (128, 6, 4), (True, False, False), (4.0, 1.0, 0.5)),
    # boxes with cells more than three layers inside them (the fused red+black kernel's own black updates)
    ((32, 16, 16), (16, 16, 16), (False, True, False), (2.0, 1.0, 1.0)),
]


def _setup(so, case, **kw):
    from somar_amd import AMRPressureSolver
    n, bs, per, L = case
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), per)
    grids = so.split_domain(dom.box, bs)
    dx = tuple(L[d] / n[d] for d in range(3))
    Jgup, Jinv = so.make_full_metric(grids, dx, L, dom)
    fac = so.Factory(dom, grids, dx, so.BCHolder(), Jgup, Jinv, isDiagonal=False, **kw)
    s = AMRPressureSolver()
    p = s._p
    s.setAMRMGParameters(p.imin, p.imax, p.eps, kw.get("maxDepth", -1), p.num_smooth_precond, 2, 2, 2, p.precond_mode, 1,
                         p.num_mg, p.hang, p.norm_thresh, 0)
    s.define(dom.box.lo, dom.box.hi, dom.periodic, dx, [(g.lo, g.hi) for g in grids])
    for q in range(s.num_local_patches):
        _, _, gi = s.patch_box(q)
        s.setMetricFull(q, *[np.asfortranarray(Jgup[gi][d].a) for d in range(3)], np.asfortranarray(Jinv[gi].a[..., 0]))
    s.finalize()
    return dom, grids, fac, s


@pytest.mark.parametrize("case", CASES)
def test_full_operator_and_gsrb_bit_exact(oracle, case, kernel_path):
    from somar_amd import api as F
    so = oracle
    dom, grids, fac, gpu = _setup(so, case)
    try:
        mg = so.MultiGrid(fac, so.BiCGStab())
        assert gpu.depth() == mg.depth
        for d in range(min(mg.depth, 2)):
            op = mg.ops[d]
            g = op.grids
            phi = so.random_field(g, 7 + d, (1, 1, 1), op.domain.box)
            rhs = so.random_field(g, 8 + d, (0, 0, 0), op.domain.box)
            fc, fr, fs = ((F.F_PHI, F.F_RHS, F.F_RES) if d == 0 else
                          (F.FIELD(d, F.F_CORR), F.FIELD(d, F.F_RES), F.FIELD(d, F.F_SCRATCH)))
            upload(gpu, fc, phi, depth=d)
            upload(gpu, fr, rhs, depth=d)
            res = so.LevelData(g, 1)
            op.residual(res, phi, rhs, True)
            gpu.residual(d, fs, fc, fr)
            for a, b in zip(download_valid(gpu, fs, g, d), valid_of(res)):
                np.testing.assert_array_equal(a, b)
            op.relax(phi, rhs, 2)
            n_fused = gpu.fused19Sweeps()
            gpu.relax(d, fc, fr, 2)
            for a, b in zip(download_valid(gpu, fc, g, d), valid_of(phi)):
                np.testing.assert_array_equal(a, b)
            if kernel_path == "fused" and all(min(b_.size()) >= 8 and b_.size()[0] % 2 == 0 for b_ in g):
                assert gpu.fused19Sweeps() == n_fused + 2     # the path under test is the one that ran
            else:
                assert gpu.fused19Sweeps() == n_fused
    finally:
        gpu.undefine()


def test_full_metric_solve_history_matches(oracle):
    so = oracle
    dom, grids, fac, gpu = _setup(so, CASES[1])
    try:
        amr = so.AMRMultiGrid(fac, so.BiCGStab())
        # compatible right-hand side: L[random phi]
        phi0 = so.random_field(grids, 3, (1, 1, 1), dom.box)
        b = so.LevelData(grids, 1)
        amr.op.apply_op(b, phi0, True)
        x = so.LevelData(grids, 1, (1, 1, 1))
        amr.solve(x, b)
        gx = [np.zeros(f.a.shape[:3], order="F") for f in x.fabs]
        gb = [np.asfortranarray(f.a[..., 0]) for f in b.fabs]
        st = gpu.solve(gx, gb, 0, 0, True, False)
        assert st["iters"] == amr.iters and st["exitStatus"] == amr.exitStatus
        np.testing.assert_allclose(st["history"], amr.history, rtol=1e-9, atol=1e-12 * amr.history[0])
        assert st["history"][-1] <= 1e-6 * st["history"][0]
    finally:
        gpu.undefine()


# ---- SpaceDim 2: the 9-point kernels (GSRBITER2D, GSRBBOUNDARYITER2D, MAPPEDGETFLUX / Neumann ghost with one cross term) ----
CASES_2D = [
    ((32, 32), 16, (False, False), (2.0, 1.0)),
    ((32, 16), 8, (False, True), (1.0, 1.0)),
    ((24, 16), (12, 8), (True, False), (1.5, 1.0)),
]


def _setup2(so, case, **kw):
    from somar_amd import AMRPressureSolver
    n, bs, per, L = case
    dom = so.Domain(so.Box((0, 0, 0), (n[0] - 1, n[1] - 1, 0)), (per[0], per[1], False))
    bsz = (bs, bs, 1) if isinstance(bs, int) else (bs[0], bs[1], 1)
    grids = so.split_domain(dom.box, bsz)
    dx = (L[0] / n[0], L[1] / n[1], 1.0)
    Jgup, Jinv = so.make_full_metric_2d(grids, dx, L, dom)
    fac = so.Factory(dom, grids, dx, so.BCHolder(), Jgup, Jinv, isDiagonal=False, ndim=2, **kw)
    s = AMRPressureSolver()
    s.setSpaceDim(2)
    p = s._p
    s.setAMRMGParameters(p.imin, p.imax, p.eps, kw.get("maxDepth", -1), p.num_smooth_precond, 2, 2, 2, p.precond_mode, 1,
                         p.num_mg, p.hang, p.norm_thresh, 0)
    s.define(dom.box.lo, dom.box.hi, dom.periodic, dx, [(g.lo, g.hi) for g in grids])
    for q in range(s.num_local_patches):
        _, _, gi = s.patch_box(q)
        s.setMetricFull(q, np.asfortranarray(Jgup[gi][0].a), np.asfortranarray(Jgup[gi][1].a), None,
                        np.asfortranarray(Jinv[gi].a[..., 0]))
    s.finalize()
    return dom, grids, fac, s


@pytest.mark.parametrize("case", CASES_2D)
def test_2d_full_operator_and_gsrb_bit_exact(oracle, case):
    from somar_amd import api as F
    so = oracle
    dom, grids, fac, gpu = _setup2(so, case)
    try:
        mg = so.MultiGrid(fac, so.BiCGStab())
        assert gpu.depth() == mg.depth
        for d in range(min(mg.depth, 2)):
            op = mg.ops[d]
            g = op.grids
            phi = so.random_field(g, 7 + d, (1, 1, 0), op.domain.box)
            rhs = so.random_field(g, 8 + d, (0, 0, 0), op.domain.box)
            fc, fr, fs = ((F.F_PHI, F.F_RHS, F.F_RES) if d == 0 else
                          (F.FIELD(d, F.F_CORR), F.FIELD(d, F.F_RES), F.FIELD(d, F.F_SCRATCH)))
            upload(gpu, fc, phi, depth=d)
            upload(gpu, fr, rhs, depth=d)
            res = so.LevelData(g, 1)
            op.residual(res, phi, rhs, True)
            gpu.residual(d, fs, fc, fr)
            for a, b in zip(download_valid(gpu, fs, g, d), valid_of(res)):
                np.testing.assert_array_equal(a, b)
            op.relax(phi, rhs, 2)
            n_fused = gpu.fused19Sweeps()
            gpu.relax(d, fc, fr, 2)
            for a, b in zip(download_valid(gpu, fc, g, d), valid_of(phi)):
                np.testing.assert_array_equal(a, b)
            if kernel_path == "fused" and all(min(b_.size()) >= 8 and b_.size()[0] % 2 == 0 for b_ in g):
                assert gpu.fused19Sweeps() == n_fused + 2     # the path under test is the one that ran
            else:
                assert gpu.fused19Sweeps() == n_fused
    finally:
        gpu.undefine()


def test_2d_full_metric_solve_history_matches(oracle):
    so = oracle
    dom, grids, fac, gpu = _setup2(so, CASES_2D[0])
    try:
        amr = so.AMRMultiGrid(fac, so.BiCGStab())
        phi0 = so.random_field(grids, 3, (1, 1, 0), dom.box)
        b = so.LevelData(grids, 1)
        amr.op.apply_op(b, phi0, True)
        x = so.LevelData(grids, 1, (1, 1, 0))
        amr.solve(x, b)
        gx = [np.zeros(f.a.shape[:3], order="F") for f in x.fabs]
        gb = [np.asfortranarray(f.a[..., 0]) for f in b.fabs]
        st = gpu.solve(gx, gb, 0, 0, True, False, phi_ghost=(1, 1, 0))
        assert st["iters"] == amr.iters and st["exitStatus"] == amr.exitStatus
        np.testing.assert_allclose(st["history"], amr.history, rtol=1e-9, atol=1e-12 * amr.history[0])
        assert st["history"][-1] <= 1e-6 * st["history"][0]
    finally:
        gpu.undefine()
