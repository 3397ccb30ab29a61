"""N > 1 path rehearsed on CPUs (gloo, world_size 2).

The sharded data path of the library is: plan (host C++, `somar_plan_exchange`) -> pack -> one message per
neighbouring rank -> local copies -> unpack.  Here the SAME plan, produced by the library through the C ABI
(a pure host call, no GPU), is executed with numpy buffers and torch.distributed/gloo, and checked

  1. cell by cell against the definition of Chombo's exchange (every ghost cell covered by another box's
     valid region or a periodic image receives that value; nothing else is touched), and
  2. end to end: LevelGSRB sweeps + residual of the oracle run on 2 ranks with this exchange are bitwise
     equal to the single-process oracle.

The oracle is used as the checker (and, in 2., as the per-box kernel executor standing in for the HIP
kernels, which cannot run on this box); the product's planning code is what is under test.
"""
import os
import socket

import numpy as np
import pytest

GHOST = 2
DOMAIN = ((0, 0, 0), (31, 15, 15))
PERIODIC = (False, True, False)
BOXSZ = (16, 8, 8)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _layout(so):
    dom = so.Domain(so.Box(*DOMAIN), PERIODIC)
    grids = so.split_domain(dom.box, BOXSZ)
    owner = [(i * 7 + 3) % 2 if i % 3 else i % 2 for i in range(len(grids))]  # irregular but balanced-ish
    return dom, grids, owner


def _f(I, J, K):
    return 1000.0 * I + 10.0 * J + 0.1 * K + 0.5


class PlanExchanger:
    """Executes the library's exchange plan on numpy arrays held per GLOBAL box index."""

    def __init__(self, api, dist, dom, grids, owner, rank):
        self.dist, self.rank, self.grids = dist, rank, grids
        self.local, self.send, self.recv = api.plan_exchange(dom.box.lo, dom.box.hi, dom.periodic,
                                                             [(g.lo, g.hi) for g in grids], owner, rank, GHOST)

    @staticmethod
    def _sl(lo, n, ghost):
        return tuple(slice(l + ghost, l + ghost + m) for l, m in zip(lo, n))

    def exchange(self, arrays, ghost=GHOST):
        """arrays: {global box index: ndarray over box.grow(ghost)} for the boxes this rank owns."""
        import torch
        # remote: one message per peer, items in plan order
        peers = sorted(set(i["peer"] for i in self.send) | set(i["peer"] for i in self.recv))
        for q in peers:
            sbuf = np.concatenate([arrays[i["src"]][self._sl(i["src_lo"], i["n"], ghost)].ravel(order="F")
                                   for i in self.send if i["peer"] == q] or [np.zeros(0)])
            nrecv = sum(int(np.prod(i["n"])) for i in self.recv if i["peer"] == q)
            rbuf = torch.zeros(nrecv, dtype=torch.float64)
            st = torch.from_numpy(np.ascontiguousarray(sbuf))
            if self.rank < q:
                if st.numel():
                    self.dist.send(st, q)
                if nrecv:
                    self.dist.recv(rbuf, q)
            else:
                if nrecv:
                    self.dist.recv(rbuf, q)
                if st.numel():
                    self.dist.send(st, q)
            off = 0
            r = rbuf.numpy()
            for i in self.recv:
                if i["peer"] != q:
                    continue
                m = int(np.prod(i["n"]))
                arrays[i["dst"]][self._sl(i["dst_lo"], i["n"], ghost)] = r[off:off + m].reshape(i["n"], order="F")
                off += m
        for i in self.local:
            arrays[i["dst"]][self._sl(i["dst_lo"], i["n"], ghost)] = arrays[i["src"]][self._sl(i["src_lo"], i["n"], ghost)]


def _worker(rank, world, port, outdir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from oracle import somar_oracle as so
    from somar_amd import api
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        dom, grids, owner = _layout(so)
        mine = [i for i, o in enumerate(owner) if o == rank]
        ex = PlanExchanger(api, dist, dom, grids, owner, rank)

        # ---- 1. definition of exchange, cell by cell -------------------------------------------------
        arrays = {}
        for i in mine:
            g = grids[i].grow(GHOST)
            a = np.full(g.size(), np.nan)
            v = grids[i]
            I, J, K = np.meshgrid(*[np.arange(v.lo[d], v.hi[d] + 1) for d in range(3)], indexing="ij")
            a[tuple(slice(GHOST, GHOST + n) for n in v.size())] = _f(I, J, K)
            arrays[i] = a
        ex.exchange(arrays)
        n = dom.box.size()
        for i in mine:
            g = grids[i].grow(GHOST)
            I, J, K = np.meshgrid(*[np.arange(g.lo[d], g.hi[d] + 1) for d in range(3)], indexing="ij")
            idx = [I, J, K]
            inside = np.ones(I.shape, bool)
            W = []
            for d in range(3):
                if PERIODIC[d]:
                    W.append(np.mod(idx[d] - dom.box.lo[d], n[d]) + dom.box.lo[d])
                else:
                    W.append(idx[d])
                    inside &= (idx[d] >= dom.box.lo[d]) & (idx[d] <= dom.box.hi[d])
            want = np.where(inside, _f(*W), np.nan)
            np.testing.assert_array_equal(arrays[i], want)

        # ---- 2. distributed LevelGSRB + residual == serial oracle, bitwise -------------------------------
        dx = (1.0 / 32, 1.0 / 16, 1.0 / 16)
        L = (1.0, 1.0, 1.0)
        lgrids = [grids[i] for i in mine]
        Jgup, Jinv = so.make_diagonal_metric(lgrids, dx, L, 3, "stretched", domain=dom)
        fac = so.Factory(dom, lgrids, dx, so.BCHolder(), Jgup, Jinv)

        def dist_exchange(ld, domain, ghost=None):
            gh = ld.ghost if ghost is None else so._iv(ghost)
            assert max(gh) <= GHOST
            arr = {}
            for li, gi in enumerate(mine):
                big = np.zeros(grids[gi].grow(GHOST).size())
                pad = tuple(slice(GHOST - q, big.shape[d] - (GHOST - q)) for d, q in enumerate(ld.ghost))
                big[pad] = ld[li].a[..., 0]
                arr[gi] = big
            ex.exchange(arr)
            for li, gi in enumerate(mine):
                big = arr[gi]
                sub = tuple(slice(GHOST - q, big.shape[d] - (GHOST - q)) for d, q in enumerate(gh))
                dst = tuple(slice(ld.ghost[d] - gh[d], ld[li].a.shape[d] - (ld.ghost[d] - gh[d])) for d in range(3))
                valid = tuple(slice(ld.ghost[d], ld[li].a.shape[d] - ld.ghost[d]) for d in range(3))
                keep = ld[li].a[valid + (0,)].copy()
                ld[li].a[dst + (0,)] = big[sub]
                ld[li].a[valid + (0,)] = keep

        so.exchange = dist_exchange  # the oracle's serial exchange is replaced by the library's plan
        op = fac.mg_new_op(0, None)
        phi = so.random_field(lgrids, 41, (1, 1, 1), dom.box)
        rhs = so.random_field(lgrids, 42, (0, 0, 0), dom.box)
        op.relax(phi, rhs, 2)
        res = so.LevelData(lgrids, 1)
        op.residual(res, phi, rhs, True)
        np.savez(os.path.join(outdir, "rank%d.npz" % rank),
                 **{"phi%d" % gi: phi[li].view(grids[gi])[..., 0] for li, gi in enumerate(mine)},
                 **{"res%d" % gi: res[li].view(grids[gi])[..., 0] for li, gi in enumerate(mine)})
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_sharded_exchange_and_gsrb_match_serial_oracle(oracle, tmp_path):
    import torch.multiprocessing as mp
    from somar_amd import build
    build.build()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    so = oracle
    dom, grids, owner = _layout(so)
    dx = (1.0 / 32, 1.0 / 16, 1.0 / 16)
    Jgup, Jinv = so.make_diagonal_metric(grids, dx, (1.0, 1.0, 1.0), 3, "stretched", domain=dom)
    fac = so.Factory(dom, grids, dx, so.BCHolder(), Jgup, Jinv)
    op = fac.mg_new_op(0, None)
    phi = so.random_field(grids, 41, (1, 1, 1), dom.box)
    rhs = so.random_field(grids, 42, (0, 0, 0), dom.box)
    op.relax(phi, rhs, 2)
    res = so.LevelData(grids, 1)
    op.residual(res, phi, rhs, True)
    got = {}
    for r in range(2):
        got.update(np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)))
    assert len(got) == 2 * len(grids)
    for gi, g in enumerate(grids):
        np.testing.assert_array_equal(got["phi%d" % gi], phi[gi].view(g)[..., 0])
        np.testing.assert_array_equal(got["res%d" % gi], res[gi].view(g)[..., 0])


def test_plan_is_symmetric_between_ranks(oracle):
    """Every send item of rank a to rank b is the i-th receive item of b from a, with equal extents."""
    from somar_amd import api, build
    build.build()
    so = oracle
    dom, grids, owner = _layout(so)
    plans = [api.plan_exchange(dom.box.lo, dom.box.hi, dom.periodic, [(g.lo, g.hi) for g in grids], owner, r, GHOST)
             for r in range(2)]
    for a in range(2):
        b = 1 - a
        s = [i for i in plans[a][1] if i["peer"] == b]
        r = [i for i in plans[b][2] if i["peer"] == a]
        assert len(s) == len(r) and len(s) > 0
        for x, y in zip(s, r):
            assert (x["src"], x["dst"], x["n"], x["src_lo"], x["dst_lo"]) == (y["src"], y["dst"], y["n"], y["src_lo"], y["dst_lo"])
    # single rank: everything local, nothing remote
    loc, snd, rcv = api.plan_exchange(dom.box.lo, dom.box.hi, dom.periodic, [(g.lo, g.hi) for g in grids],
                                      [0] * len(grids), 0, GHOST)
    assert not snd and not rcv and len(loc) == len(plans[0][0]) + len(plans[1][0]) + len(plans[0][1]) + len(plans[1][1])
