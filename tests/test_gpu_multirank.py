"""The sharded data path with TWO ranks (two processes) on the one GPU of the test box, over the shared-memory
transport (csrc/comm_shm.cpp; RCCL refuses two ranks on one device).  Same plans, packing kernels and
reductions as the RCCL path -- only the wire differs.  Every rank checks ITS patches against the oracle, which
computes the whole problem serially:
  * LevelGSRB (fused and two-pass), residual, fused residual+restriction, prolongation: bit-exact;
  * full solve: same iteration count, history to 1e-10 -- the north star's tolerance.  Small levels (where BiCGStab runs
    and amplifies last-bit differences) add their scalar sums in the serial box order on every rank: per-cell terms go
    into one vector in serial sequence, a sum-allreduce completes it, one wavefront walks it (PressureSolver::ordered_sums);
    only the mean of large levels is a tree sum whose association depends on the sharding (as it does on one rank);
  * two AMR levels, both sharded: CF interpolation + refluxed composite residual bit-exact, AMR V-cycle 1e-10."""
import multiprocessing as mp
import os
import traceback
import uuid

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, nranks, name, mode, q):
    try:
        os.environ["SOMAR_FUSED_MIN_CELLS"] = "0" if mode == "fused" else "1000000000000"
        # "fused": coarse depths agglomerated (replicated on both ranks, the default for small levels);
        # "twopass": every depth stays sharded (halo exchange and allreduce down to the bottom solver)
        if mode == "twopass":
            os.environ["SOMAR_AGGLOM_CELLS"] = "0"
        import sys
        here = os.path.dirname(os.path.abspath(__file__))
        sys.path.insert(0, here)
        sys.path.insert(0, os.path.dirname(here))
        from oracle import somar_oracle as so
        from oracle import somar_amr as am
        from somar_amd import api as F
        from helpers import (download_valid, make_amr_levels, make_gpu_solver, make_oracle_solver, make_problem, upload,
                             valid_of)
        comm = F.comm_create_shm(name, rank, nranks)

        def mine(got, want, what):
            n = 0
            for g, w in zip(got, want):
                if g is None:
                    continue
                np.testing.assert_array_equal(g, w, err_msg=what)
                n += 1
            assert n > 0, "rank owns no patch"

        # ---- single level, 8 boxes dealt round-robin to the ranks, periodic in y ----------------------------
        dom, grids, dx, Jgup, Jinv = make_problem(so, (32, 32, 32), 16, "stretched", (False, True, False), (2.0, 1.0, 1.0))
        owner = [i % nranks for i in range(len(grids))]
        amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv)
        op = amr.mg.ops[0]

        class Mine:  # the helpers index metric data by global box number: hand them only what this rank owns
            def __init__(self, x):
                self.x = x

            def __getitem__(self, gi):
                assert owner[gi] == rank
                return self.x[gi]

        gpu = make_gpu_solver(dom, grids, dx, Mine(Jgup), Mine(Jinv), owner=owner, comm=comm)
        assert gpu.num_local_patches == len(grids) // nranks
        assert gpu.depth() == amr.mg.depth and gpu.mgRefRatios() == [tuple(r) for r in amr.mg.mgRefRatios]
        phi = so.random_field(grids, 3, (1, 1, 1), dom.box)
        rhs = so.random_field(grids, 4, (0, 0, 0), dom.box)
        upload(gpu, F.F_PHI, phi)
        upload(gpu, F.F_RHS, rhs)
        op.relax(phi, rhs, 2)
        gpu.relax(0, F.F_PHI, F.F_RHS, 2)
        mine(download_valid(gpu, F.F_PHI, grids), valid_of(phi), "relax")
        res = so.LevelData(grids, 1)
        op.residual(res, phi, rhs, True)
        gpu.residual(0, F.F_RES, F.F_PHI, F.F_RHS)
        mine(download_valid(gpu, F.F_RES, grids), valid_of(res), "residual")
        cres = op.create_coarser(res)
        op.restrict_residual(cres, phi, rhs)
        gpu.restrictResidual(0, F.FIELD(1, F.F_RES), F.F_PHI, F.F_RHS)
        mine(download_valid(gpu, F.FIELD(1, F.F_RES), cres.grids, 1), valid_of(cres), "restrict")
        # norms: max is exact, sums agree to rounding
        assert gpu.norm(F.F_RES, 0) == so.ld_norm(res, 0)
        assert abs(gpu.norm(F.F_RES, 2) - so.ld_norm(res, 2)) <= 1e-13 * so.ld_norm(res, 2)
        # full solve
        b = so.random_field(grids, 12345, (0, 0, 0), dom.box)
        so.remove_weighted_mean(b, Jinv)
        x = so.LevelData(grids, 1, (1, 1, 1))
        amr.solve(x, b)
        upload(gpu, F.F_RHS, b)
        try:
            st = gpu.solveResident(True, False)
        except Exception:
            raise AssertionError("solve failed: %r vs oracle %r" % (gpu.stats, amr.history))
        assert st["iters"] == amr.iters and st["exitStatus"] == amr.exitStatus
        np.testing.assert_allclose(st["history"], amr.history, rtol=1e-10, atol=1e-10 * amr.history[0])
        gpu.undefine()

        if mode == "fused":
            # ---- exchange / compute overlap: boxes large enough to have tiles that read no remote ghost cell; the sweeps'
            #      exchanges travel on the second stream under those tiles (fused_overlap, solver.cpp).  Same bits.
            # (64 rows per box: with two region rows per wavefront a class-1 tile is 28 rows tall, and one of them must lie between the
            # two y faces)
            dom, grids, dx, Jgup, Jinv = make_problem(so, (64, 64 * nranks, 32), (64, 64, 32), "stretched", (False, True, False),
                                                      (2.0, 1.0, 1.0))
            assert len(grids) == nranks
            owner = list(range(nranks))
            op = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv).mg.ops[0]
            phi = so.random_field(grids, 13, (1, 1, 1), dom.box)
            rhs = so.random_field(grids, 14, (0, 0, 0), dom.box)
            op.relax(phi, rhs, 3)
            cres = op.create_coarser(rhs)
            op.restrict_residual(cres, phi, rhs)
            os.environ["SOMAR_MARCH_MIN_CELLS"] = "1"     # the k-marching residual + restriction on this level
            # (narrow: the tile tables hold the narrow lane classes -- 64 = 60 + 4 columns -- so the split into tiles that read no
            # remote ghost cell and the rest sees tiles of every class)
            for overlap, narrow in ((True, False), (False, False), (True, True)):
                if overlap:
                    os.environ.pop("SOMAR_NO_OVERLAP", None)
                else:
                    os.environ["SOMAR_NO_OVERLAP"] = "1"
                if narrow:
                    os.environ["SOMAR_NARROW_7PT"] = "1"
                else:
                    os.environ.pop("SOMAR_NARROW_7PT", None)
                g2 = make_gpu_solver(dom, grids, dx, Mine(Jgup), Mine(Jinv), owner=owner, comm=comm)
                p0 = so.random_field(grids, 13, (1, 1, 1), dom.box)
                upload(g2, F.F_PHI, p0)
                upload(g2, F.F_RHS, rhs)
                g2.relax(0, F.F_PHI, F.F_RHS, 3)
                assert (g2.counters()["overlapped_sweeps"] == 3) == overlap, g2.counters()
                mine(download_valid(g2, F.F_PHI, grids), valid_of(phi), "overlapped relax" if overlap else "serial relax")
                g2.restrictResidual(0, F.FIELD(1, F.F_RES), F.F_PHI, F.F_RHS)
                assert (g2.counters()["overlapped_sweeps"] == 4) == overlap, g2.counters()
                mine(download_valid(g2, F.FIELD(1, F.F_RES), cres.grids, 1), valid_of(cres), "residual + restriction")
                g2.undefine()
            os.environ.pop("SOMAR_NO_OVERLAP", None)
            os.environ.pop("SOMAR_NARROW_7PT", None)
            os.environ.pop("SOMAR_MARCH_MIN_CELLS", None)

        if nranks != 2:
            F.comm_destroy(comm)
            q.put((rank, "ok"))
            return
        # ---- two AMR levels, each sharded over both ranks --------------------------------------------------
        periodic, ratios = (True, False, False), [(2, 2, 1)]
        fb = [[so.Box((0, 8, 0), (15, 23, 7)), so.Box((24, 8, 0), (31, 23, 7))]]
        levels = make_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.5), periodic, ratios, fb)
        comp = am.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab())
        owners = [[i % nranks for i in range(len(L.grids))] for L in levels]
        owners[1] = [(i + 1) % nranks for i in range(len(levels[1].grids))]  # fine boxes NOT on their coarse box's rank
        from somar_amd import AMRPressureSolver
        s = AMRPressureSolver()
        L0 = levels[0]
        s.defineAMR(L0.domain.box.lo, L0.domain.box.hi, L0.domain.periodic, L0.dx, ratios,
                    [[(g.lo, g.hi) for g in L.grids] for L in levels], owners_per_level=owners, comm=comm)
        for L, v in zip(levels, s.levels):
            for p_ in range(v.num_local_patches):
                _, _, gi = v.patch_box(p_)
                jg = [np.asfortranarray(L.Jgup[gi][d].a[..., d]) for d in range(3)]
                v.setMetricOrtho(p_, jg[0], jg[1], jg[2], np.asfortranarray(L.Jinv[gi].a[..., 0]))
        s.finalize()
        phis = [so.random_field(L.grids, 5 + l, (1, 1, 1), L.domain.box) for l, L in enumerate(levels)]
        rhss = [so.random_field(L.grids, 50 + l, (0, 0, 0), L.domain.box) for l, L in enumerate(levels)]
        ress = [so.LevelData(L.grids, 1) for L in levels]
        comp.init(phis, rhss, 1, 0)
        for l, v in enumerate(s.levels):
            upload(v, F.F_PHI, phis[l])
            upload(v, F.F_RHS, rhss[l])
        comp.compute_amr_residual(ress, phis, rhss, 1, 0, True)
        for ilev in (0, 1):
            s.residualLevel(1, 0, ilev)
            if ilev == 0:
                s.zeroCovered(0, F.F_RES)
            got = download_valid(s.levels[ilev], F.F_RES, levels[ilev].grids)
            n = 0
            for g, w in zip(got, valid_of(ress[ilev])):
                if g is not None:
                    np.testing.assert_array_equal(g, w, err_msg="composite residual level %d" % ilev)
                    n += 1
            assert n > 0 or len(levels[ilev].grids) < nranks
        # AMR V-cycle
        res2 = [so.random_field(L.grids, 70 + l, (0, 0, 0), L.domain.box) for l, L in enumerate(levels)]
        comp.zero_covered(0, res2[0])
        zero = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
        comp.init(zero, res2, 1, 0)
        comp.set_bottom_solver(1, 0)
        corr = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
        for l, v in enumerate(s.levels):
            upload(v, F.F_RES, res2[l])
            v.setVal(F.F_CORR, 0.0)
        comp.amr_vcycle(corr, res2, 1, 1, 0)
        s.vcycleAMR(1, 0)
        for l in (0, 1):
            got = download_valid(s.levels[l], F.F_CORR, levels[l].grids)
            for g, w in zip(got, valid_of(corr[l])):
                if g is not None:
                    np.testing.assert_allclose(g, w, rtol=0, atol=1e-10 * float(np.max(np.abs(w))))
        s.undefine()
        F.comm_destroy(comm)
        q.put((rank, "ok"))
    except Exception:
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("mode,nranks", [("fused", 2), ("twopass", 2), ("fused", 4)])
def test_ranks_sharing_one_gpu(mode, nranks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "/somar_%s" % uuid.uuid4().hex[:12]
    procs = [ctx.Process(target=_worker, args=(r, nranks, name, mode, q)) for r in range(nranks)]
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
