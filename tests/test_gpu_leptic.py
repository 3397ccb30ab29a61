"""Leptic level solver on the GPU (csrc/leptic.cpp, leptic_kernels.hip) against the oracle's restatement of
LevelLepticSolver (oracle/somar_leptic.py): same synthetic inputs, through the C ABI."""
import numpy as np
import pytest

from oracle import somar_leptic as sl
from oracle import somar_oracle as so
from tests.helpers import download_valid, make_oracle_solver, upload, valid_of

pytestmark = pytest.mark.gpu


def _problem(n, box, H, variant, seed=3):
    L = (1.0, 1.0, H)
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (False, False, False))
    grids = so.split_domain(dom.box, box)
    dx = tuple(L[d] / n[d] for d in range(3))
    Jgup, Jinv = so.make_diagonal_metric(grids, dx, L, 3, variant, domain=dom)
    rhs = so.random_field(grids, seed, domainBox=dom.box)
    so.remove_weighted_mean(rhs, Jinv)
    return dom, grids, dx, Jgup, Jinv, rhs


def _gpu_leptic(dom, grids, dx, Jgup, Jinv, maxOrder, H, full_relax=None):
    from somar_amd import LevelLepticSolver
    s = LevelLepticSolver()
    s.params.max_order = maxOrder
    s.params.domain_height = H
    if full_relax is not None:
        s.params.full.relax_mode, s.params.full.precond_mode = full_relax
    s.define(dom.box.lo, dom.box.hi, dom.periodic, dx, [(g.lo, g.hi) for g in grids])
    lv = s.level
    for p_ in range(lv.num_local_patches):
        _, _, gi = lv.patch_box(p_)
        jg = [np.asfortranarray(Jgup[gi][d].a[..., d]) for d in range(3)]
        lv.setMetricOrtho(p_, jg[0], jg[1], jg[2], np.asfortranarray(Jinv[gi].a[..., 0]))
    s.finalize()
    return s


CASES = [
    # n, box, H, variant, maxOrder
    ((32, 32, 8), (16, 16, 8), 0.005, "stretched", 4),
    ((32, 32, 8), (16, 16, 8), 0.005, "cartesian", 5),
    ((32, 16, 12), (8, 16, 12), 0.02, "stretched", 2),
    ((32, 32, 8), (32, 32, 8), 0.001, "stretched", 3),   # one box
    ((16, 16, 2), (8, 8, 2), 0.001, "stretched", 2),     # two-cell columns: the reference's a(0) sentinel row
]


@pytest.mark.parametrize("n,box,H,variant,maxOrder", CASES)
def test_leptic_solve_matches_oracle(n, box, H, variant, maxOrder):
    from somar_amd.api import F_PHI, F_RHS
    dom, grids, dx, Jgup, Jinv, rhs = _problem(n, box, H, variant)
    amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv)
    lep = sl.LevelLepticSolver(amr.op, maxOrder=maxOrder, domainHeight=H)
    phi = so.random_field(grids, 11, ghost=(1, 1, 1), domainBox=dom.box)   # non-trivial initial guess
    for f in phi.fabs:
        f.a[...] *= 1e-3
    phi0 = [v.copy() for v in valid_of(phi)]
    gpu = _gpu_leptic(dom, grids, dx, Jgup, Jinv, maxOrder, H)
    upload(gpu.level, F_PHI, phi)
    upload(gpu.level, F_RHS, rhs)
    status = lep.solve(phi, rhs)
    st = gpu.solve()
    assert st["exitStatus"] == status
    assert st["horizSolves"] == lep.horizSolves and st["usedFullSolver"] == lep.usedFullSolver
    assert len(st["resNorms"]) == len(lep.resNorms)
    got = download_valid(gpu.level, F_PHI, grids)
    want = valid_of(phi)
    if not lep.usedFullSolver:
        # the column kernels follow the Fortran term by term and every reduction that steers the flat multigrid
        # runs in the reference's order at these sizes (k_reduce_ordered) => bit-identical
        assert st["resNorms"] == lep.resNorms
        for g_, w_ in zip(got, want):
            np.testing.assert_array_equal(g_, w_)
        for name, view, fld in (("vertPhi", gpu.vert, F_PHI), ("horizPhi", gpu.horiz, F_PHI), ("horizRhs", gpu.horiz, F_RHS)):
            ld = lep.last[name]
            for g_, w_ in zip(download_valid(view, fld, ld.grids), valid_of(ld)):
                np.testing.assert_array_equal(g_, w_, err_msg=name)
    else:
        # the full 3-D multigrid (8192 cells at depth 0: tree sums in the zero-average prolongation) agrees to
        # round-off, not bit for bit
        np.testing.assert_allclose(st["resNorms"], lep.resNorms, rtol=1e-9)
        scale = max(float(np.max(np.abs(w - p0))) for w, p0 in zip(want, phi0))
        for g_, w_ in zip(got, want):
            np.testing.assert_allclose(g_, w_, rtol=0, atol=1e-12 * scale)


def _gpu_leptic_full(dom, grids, dx, Jgup, Jinv, maxOrder, H):
    from somar_amd import LevelLepticSolver
    s = LevelLepticSolver()
    s.params.max_order = maxOrder
    s.params.domain_height = H
    s.define(dom.box.lo, dom.box.hi, dom.periodic, dx, [(g.lo, g.hi) for g in grids])
    lv = s.level
    for p_ in range(lv.num_local_patches):
        _, _, gi = lv.patch_box(p_)
        lv.setMetricFull(p_, *[np.asfortranarray(Jgup[gi][d].a) for d in range(3)], np.asfortranarray(Jinv[gi].a[..., 0]))
    s.finalize()
    return s


FULL_CASES = [
    # n, box, L, metric, maxOrder
    ((32, 32, 8), (16, 16, 8), (64.0, 64.0, 1.0), "terrain", 4),
    ((32, 32, 8), (32, 32, 8), (64.0, 64.0, 1.0), "terrain", 2),     # one box
    ((32, 16, 12), (8, 16, 12), (64.0, 32.0, 1.0), "sheared", 3),    # every J g^{ab} non-zero
]


@pytest.mark.parametrize("n,box,L,metric,maxOrder", FULL_CASES)
def test_leptic_nondiagonal_matches_oracle(n, box, L, metric, maxOrder):
    """Non-diagonal metric: vertical boundary data from LEPTICVERTHORIZGRAD, MAPPEDMACGRAD with cross terms in the
    horizontal right-hand side, a horizontal solve at every order (oracle: somar_leptic.py, isDiagonal False)."""
    from somar_amd.api import F_PHI, F_RHS
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (False, False, False))
    grids = so.split_domain(dom.box, box)
    dx = tuple(L[d] / n[d] for d in range(3))
    if metric == "terrain":
        Jgup, Jinv = so.make_terrain_metric(grids, dx, L, dom)
    else:
        Jgup, Jinv = so.make_full_metric(grids, dx, L, dom, amp=(0.05, 0.04, 0.03))
    rhs = so.random_field(grids, 3, domainBox=dom.box)
    so.remove_weighted_mean(rhs, Jinv)
    amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv, isDiagonal=False)
    lep = sl.LevelLepticSolver(amr.op, maxOrder=maxOrder, domainHeight=L[2])
    phi = so.random_field(grids, 11, ghost=(1, 1, 1), domainBox=dom.box)
    for f in phi.fabs:
        f.a[...] *= 1e-3
    gpu = _gpu_leptic_full(dom, grids, dx, Jgup, Jinv, maxOrder, L[2])
    upload(gpu.level, F_PHI, phi)
    upload(gpu.level, F_RHS, rhs)
    status = lep.solve(phi, rhs)
    st = gpu.solve()
    assert not lep.usedFullSolver and not st["usedFullSolver"]
    assert st["exitStatus"] == status
    assert st["horizSolves"] == lep.horizSolves and lep.horizSolves == maxOrder + 1
    assert st["resNorms"] == lep.resNorms
    for g_, w_ in zip(download_valid(gpu.level, F_PHI, grids), valid_of(phi)):
        np.testing.assert_array_equal(g_, w_)
    for name, view, fld in (("vertPhi", gpu.vert, F_PHI), ("horizPhi", gpu.horiz, F_PHI), ("horizRhs", gpu.horiz, F_RHS)):
        ld = lep.last[name]
        for g_, w_ in zip(download_valid(view, fld, ld.grids), valid_of(ld)):
            np.testing.assert_array_equal(g_, w_, err_msg=name)


def test_leptic_full_multigrid_fallback_matches_oracle():
    """maxOrder 0: the O(1) residual exceeds the initial one by construction, the full 3-D multigrid (LINE_GSRB
    4/4/4, DiagLineRelax) takes over; both sides must take the same branch and agree."""
    from somar_amd.api import F_PHI, F_RHS
    n, box, H = (32, 32, 8), (16, 16, 8), 0.02
    dom, grids, dx, Jgup, Jinv, rhs = _problem(n, box, H, "stretched")
    amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv)
    lep = sl.LevelLepticSolver(amr.op, maxOrder=0, domainHeight=H)
    phi = so.LevelData(grids, 1, (1, 1, 1))
    gpu = _gpu_leptic(dom, grids, dx, Jgup, Jinv, 0, H)
    upload(gpu.level, F_PHI, phi)
    upload(gpu.level, F_RHS, rhs)
    status = lep.solve(phi, rhs)
    st = gpu.solve()
    assert lep.usedFullSolver and st["usedFullSolver"]
    assert st["exitStatus"] == status
    assert st["full"]["iters"] == lep.mgSolver.iters
    np.testing.assert_allclose(st["full"]["history"], lep.mgSolver.history, rtol=1e-8)
    np.testing.assert_allclose(st["resNorms"], lep.resNorms, rtol=1e-8)
    got = download_valid(gpu.level, F_PHI, grids)
    want = valid_of(phi)
    scale = max(float(np.max(np.abs(w))) for w in want)
    for g_, w_ in zip(got, want):
        np.testing.assert_allclose(g_, w_, rtol=0, atol=1e-8 * scale)


def test_leptic_rejects_what_the_reference_cannot_do():
    from somar_amd import LevelLepticSolver, SomarError
    s = LevelLepticSolver()
    with pytest.raises(SomarError, match="Vertical grids are ill-formed"):
        s.define((0, 0, 0), (15, 15, 7), (False, False, False), (0.1, 0.1, 0.01),
                 [((0, 0, 0), (15, 15, 3)), ((0, 0, 4), (15, 15, 7))])
    s = LevelLepticSolver()
    with pytest.raises(SomarError, match="periodic"):
        s.define((0, 0, 0), (15, 15, 7), (True, False, False), (0.1, 0.1, 0.01), [((0, 0, 0), (15, 15, 7))])


# ---- two ranks sharing the test box's GPU (shared-memory transport, as in test_gpu_multirank.py) -------------------
def _leptic_worker(rank, nranks, name, q):
    import os
    import sys
    import traceback
    try:
        here = os.path.dirname(os.path.abspath(__file__))
        sys.path.insert(0, os.path.dirname(here))
        from somar_amd import LevelLepticSolver
        from somar_amd import api as F
        comm = F.comm_create_shm(name, rank, nranks)
        n, box, H, maxOrder = (32, 32, 8), (16, 16, 8), 0.005, 3
        dom, grids, dx, Jgup, Jinv, rhs = _problem(n, box, H, "stretched")
        owner = [i % nranks for i in range(len(grids))]
        amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv)
        lep = sl.LevelLepticSolver(amr.op, maxOrder=maxOrder, domainHeight=H)
        phi = so.LevelData(grids, 1, (1, 1, 1))
        s = LevelLepticSolver()
        s.params.max_order = maxOrder
        s.params.domain_height = H
        s.define(dom.box.lo, dom.box.hi, dom.periodic, dx, [(g.lo, g.hi) for g in grids], owner=owner, comm=comm)
        lv = s.level
        assert lv.num_local_patches == len(grids) // nranks
        for p_ in range(lv.num_local_patches):
            _, _, gi = lv.patch_box(p_)
            assert owner[gi] == rank
            jg = [np.asfortranarray(Jgup[gi][d].a[..., d]) for d in range(3)]
            lv.setMetricOrtho(p_, jg[0], jg[1], jg[2], np.asfortranarray(Jinv[gi].a[..., 0]))
        s.finalize()
        upload(lv, F.F_PHI, phi)
        upload(lv, F.F_RHS, rhs)
        status = lep.solve(phi, rhs)
        st = s.solve()
        assert st["exitStatus"] == status and st["horizSolves"] == lep.horizSolves
        assert st["usedFullSolver"] == lep.usedFullSolver
        # rank-wise association of the scalar sums differs from the serial box order (BiCGStab amplifies it)
        np.testing.assert_allclose(st["resNorms"], lep.resNorms, rtol=1e-6)
        want = valid_of(phi)
        scale = max(float(np.max(np.abs(w))) for w in want)
        nmine = 0
        for g_, w_ in zip(download_valid(lv, F.F_PHI, grids), want):
            if g_ is None:
                continue
            np.testing.assert_allclose(g_, w_, rtol=0, atol=1e-8 * scale)
            nmine += 1
        assert nmine == len(grids) // nranks
        s.undefine()
        F.comm_destroy(comm)
        q.put((rank, "ok"))
    except Exception:
        q.put((rank, traceback.format_exc()))


def test_leptic_two_ranks_sharing_one_gpu():
    import multiprocessing as mp
    import uuid
    nranks = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "/somar_lep_%s" % uuid.uuid4().hex[:10]
    procs = [ctx.Process(target=_leptic_worker, args=(r, nranks, name, q)) for r in range(nranks)]
    for p in procs:
        p.start()
    out = {}
    try:
        for _ in procs:
            rank, msg = q.get(timeout=240)
            out[rank] = msg
    finally:
        for p in procs:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()
    assert out == {r: "ok" for r in range(nranks)}, "\n".join("rank %d: %s" % kv for kv in sorted(out.items()))


# ---- columns that END at a Dirichlet wall or a coarse-fine interface: LepticLapackVerticalSolver + dptsv -------------------
D_, N_ = 1, 0


@pytest.mark.parametrize("variant,H,maxOrder", [("cartesian", 0.02, 1), ("stretched", 0.005, 3), ("stretched", 0.02, 2)])
def test_dirichlet_topped_columns_match_the_oracle(variant, H, maxOrder):
    """Neumann below, Dirichlet above with a non-zero value: gatherVerticalBCTypes switches the horizontal problem off and
    every order is one dptsv per column (k_lep_vsolve_lapack).  Bit for bit unless the full multigrid takes over."""
    from somar_amd import LevelLepticSolver
    from somar_amd.api import F_PHI, F_RHS
    n, box = (16, 16, 8), (8, 8, 8)
    L = (1.0, 1.0, H)
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (False, False, False))
    grids = so.split_domain(dom.box, box)
    dx = tuple(L[d] / n[d] for d in range(3))
    Jgup, Jinv = so.make_diagonal_metric(grids, dx, L, 3, variant, domain=dom)
    types = [[N_, N_], [N_, N_], [N_, D_]]
    values = [[0.0, 0.0], [0.0, 0.0], [0.0, 0.3]]
    bc = so.BCHolder([list(t) for t in types], [list(v) for v in values])
    fac = so.Factory(dom, grids, dx, bc, Jgup, Jinv)
    op = so.AMRMultiGrid(fac, so.BiCGStab()).op
    lep = sl.LevelLepticSolver(op, maxOrder=maxOrder, domainHeight=H)
    assert not lep.doHorizSolve
    rhs = so.random_field(grids, 9, domainBox=dom.box)
    phi = so.random_field(grids, 11, ghost=(1, 1, 1), domainBox=dom.box)
    for f in phi.fabs:
        f.a[...] *= 1e-3
    s = LevelLepticSolver()
    s.params.max_order, s.params.domain_height = maxOrder, H
    s.define(dom.box.lo, dom.box.hi, dom.periodic, dx, [(g.lo, g.hi) for g in grids], bc_type=[t for q in types for t in q])
    try:
        assert s.horiz is None
        lv = s.level
        lv.setBCValues([x for q in values for x in q])
        for p_ in range(lv.num_local_patches):
            _, _, gi = lv.patch_box(p_)
            jg = [np.asfortranarray(Jgup[gi][d].a[..., d]) for d in range(3)]
            lv.setMetricOrtho(p_, jg[0], jg[1], jg[2], np.asfortranarray(Jinv[gi].a[..., 0]))
        s.finalize()
        upload(s.level, F_PHI, phi)
        upload(s.level, F_RHS, rhs)
        status = lep.solve(phi, rhs, False)
        st = s.solve(False)
        assert st["exitStatus"] == status and st["horizSolves"] == 0 and st["usedFullSolver"] == lep.usedFullSolver
        got, want = download_valid(s.level, F_PHI, grids), valid_of(phi)
        if not lep.usedFullSolver:
            assert st["resNorms"] == lep.resNorms
            for g_, w_ in zip(got, want):
                np.testing.assert_array_equal(g_, w_)
        else:
            np.testing.assert_allclose(st["resNorms"], lep.resNorms, rtol=1e-9)
            for g_, w_ in zip(got, want):
                np.testing.assert_allclose(g_, w_, rtol=0, atol=1e-10 * float(np.max(np.abs(w_))))
    finally:
        s.undefine()


def test_dirichlet_topped_columns_with_a_nondiagonal_metric():
    """the same column mode on a terrain-following (19-point) operator: the tridiagonal systems take J g^{zeta zeta} alone
    (LepticLapackVerticalSolver knows no cross terms), the residual test runs on the full operator"""
    from somar_amd import LevelLepticSolver
    from somar_amd.api import F_PHI, F_RHS
    n, box, L, maxOrder = (32, 32, 8), (16, 16, 8), (64.0, 64.0, 1.0), 3
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), (False, False, False))
    grids = so.split_domain(dom.box, box)
    dx = tuple(L[d] / n[d] for d in range(3))
    Jgup, Jinv = so.make_terrain_metric(grids, dx, L, dom)
    types = [[N_, N_], [N_, N_], [N_, D_]]
    bc = so.BCHolder([list(t) for t in types], [[0.0, 0.0], [0.0, 0.0], [0.0, 0.0]])
    fac = so.Factory(dom, grids, dx, bc, Jgup, Jinv, isDiagonal=False)
    op = so.AMRMultiGrid(fac, so.BiCGStab()).op
    lep = sl.LevelLepticSolver(op, maxOrder=maxOrder, domainHeight=L[2])
    assert not lep.doHorizSolve
    rhs = so.random_field(grids, 9, domainBox=dom.box)
    phi = so.LevelData(grids, 1, (1, 1, 1))
    s = LevelLepticSolver()
    s.params.max_order, s.params.domain_height = maxOrder, L[2]
    s.define(dom.box.lo, dom.box.hi, dom.periodic, dx, [(g.lo, g.hi) for g in grids], bc_type=[t for q in types for t in q])
    try:
        lv = s.level
        for p_ in range(lv.num_local_patches):
            _, _, gi = lv.patch_box(p_)
            lv.setMetricFull(p_, *[np.asfortranarray(Jgup[gi][d].a) for d in range(3)], np.asfortranarray(Jinv[gi].a[..., 0]))
        s.finalize()
        lv.setVal(F_PHI, 0.0)
        upload(lv, F_RHS, rhs)
        status = lep.solve(phi, rhs, True)
        st = s.solve(True)
        assert st["exitStatus"] == status and st["horizSolves"] == 0 and st["usedFullSolver"] == lep.usedFullSolver
        got, want = download_valid(lv, F_PHI, grids), valid_of(phi)
        if not lep.usedFullSolver:
            assert st["resNorms"] == lep.resNorms
            for g_, w_ in zip(got, want):
                np.testing.assert_array_equal(g_, w_)
        else:
            np.testing.assert_allclose(st["resNorms"], lep.resNorms, rtol=1e-8)
            scale = max(float(np.max(np.abs(w_))) for w_ in want)
            for g_, w_ in zip(got, want):
                np.testing.assert_allclose(g_, w_, rtol=0, atol=1e-9 * scale)
    finally:
        s.undefine()
