"""The C/OpenMP-orchestrated V-cycle that bench.py times as the CPU baseline (oracle/cpu_vcycle.c) against the
numpy-orchestrated oracle (oracle/somar_oracle.py) on the same one-box stretched-metric Neumann problem: same
hierarchy, same null-space decisions, the same bits after one V-cycle with one thread; round-off with several
(only the zero-average sums associate differently)."""
import numpy as np
import pytest

from helpers import make_oracle_solver, make_problem


@pytest.mark.parametrize("n", [16, 32])
def test_c_orchestrated_vcycle_matches_the_python_orchestration(oracle, n):
    so = oracle
    from oracle import cpu_vcycle as cv
    dom, grids, dx, Jgup, Jinv = make_problem(so, n, n, "stretched")
    amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv)
    res = so.random_field(grids, 12345, (0, 0, 0), dom.box)
    so.remove_weighted_mean(res, Jinv)
    corr = so.LevelData(grids, 1, (1, 1, 1))
    amr.mg.init(corr, res)
    amr.mg.one_cycle(corr, res)
    want = corr[0].a[..., 0]

    jg = [np.asfortranarray(Jgup[0][d].a[..., d]) for d in range(3)]
    jinv = np.asfortranarray(Jinv[0].a[..., 0])
    h = cv.CpuVCycle((n,) * 3, dx, jg, jinv, nthreads=1, fast=False)
    assert h.depth() == amr.mg.depth
    assert [h.zero_avg(d) for d in range(h.depth())] == [op.zeroAvg for op in amr.mg.ops]
    r = np.asfortranarray(res[0].a[..., 0])
    got = np.zeros((n + 2,) * 3, order="F")
    h.vcycle(got, r)
    np.testing.assert_array_equal(got[1:-1, 1:-1, 1:-1], want[1:-1, 1:-1, 1:-1])
    h.set_threads(4)
    got4 = np.zeros((n + 2,) * 3, order="F")
    h.vcycle(got4, r)
    np.testing.assert_allclose(got4[1:-1, 1:-1, 1:-1], want[1:-1, 1:-1, 1:-1], rtol=0, atol=1e-12 * np.abs(want).max())
    h.close()


def test_triad_reports_a_bandwidth():
    from oracle import cpu_vcycle as cv
    assert cv.triad_gbs(2, n=1 << 22, reps=2) > 0.1
