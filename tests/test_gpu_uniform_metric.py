"""Uniform-metric fast path (PressureSolver::detect_uniform_metric): on a Cartesian map the k-marching sweep / residual
kernels take J g^{aa}, J^{-1} from their parameter block instead of streaming the arrays.  Same arithmetic => the same
bits as the streaming kernels, and as the oracle."""
import os

import numpy as np
import pytest

from oracle import somar_oracle as so
from tests.helpers import download_valid, make_gpu_solver, make_oracle_solver, upload, valid_of

pytestmark = pytest.mark.gpu


def _problem(n, box, variant, periodic=(False, False, False), L=(1.0, 1.0, 1.0)):
    dom = so.Domain(so.Box((0, 0, 0), tuple(a - 1 for a in n)), periodic)
    grids = so.split_domain(dom.box, box)
    dx = tuple(L[d] / n[d] for d in range(3))
    Jgup, Jinv = so.make_diagonal_metric(grids, dx, L, 3, variant, domain=dom)
    rhs = so.random_field(grids, 5, domainBox=dom.box)
    so.remove_weighted_mean(rhs, Jinv)
    return dom, grids, dx, Jgup, Jinv, rhs


@pytest.fixture
def march_everything(monkeypatch):
    # small levels normally take the direct kernels; force the k-marching ones (the kernels with the fast path)
    monkeypatch.setenv("SOMAR_MARCH_MIN_CELLS", "1")
    yield


def _solve(dom, grids, dx, Jgup, Jinv, rhs, uniform):
    from somar_amd.api import F_PHI, F_RHS
    os.environ["SOMAR_NO_UNIFORM"] = "0" if uniform else "1"
    try:
        gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv, pre=2, post=2, bottom=2)
    finally:
        os.environ.pop("SOMAR_NO_UNIFORM", None)
    upload(gpu, F_RHS, rhs)
    st = gpu.solveResident(True, False)
    return gpu, st, download_valid(gpu, F_PHI, grids)


@pytest.mark.parametrize("n,box,periodic", [((64, 64, 32), (32, 64, 32), (False, False, False)),
                                            ((64, 32, 32), (64, 32, 32), (False, True, False))])
def test_uniform_detected_and_bit_identical_to_streaming(march_everything, n, box, periodic):
    dom, grids, dx, Jgup, Jinv, rhs = _problem(n, box, "cartesian", periodic)
    gU, stU, phiU = _solve(dom, grids, dx, Jgup, Jinv, rhs, True)
    gS, stS, phiS = _solve(dom, grids, dx, Jgup, Jinv, rhs, False)
    assert gU.metricUniform(0) is not None and gS.metricUniform(0) is None
    c = gU.metricUniform(0)
    assert c[3] == float(Jinv[0].a.flat[0]) and c[0] == float(Jgup[0][0].a[..., 0].flat[0])
    for d in range(gU.depth()):
        assert gU.metricUniform(d) is not None      # averages of equal numbers stay equal on every coarser depth
    assert stU["history"] == stS["history"] and stU["iters"] == stS["iters"]
    for a, b in zip(phiU, phiS):
        np.testing.assert_array_equal(a, b)
    # and against the oracle
    amr = make_oracle_solver(so, dom, grids, dx, Jgup, Jinv)
    phi = so.LevelData(grids, 1, (1, 1, 1))
    amr.solve(phi, rhs)
    assert stU["iters"] == amr.iters
    np.testing.assert_allclose(stU["history"], amr.history, rtol=0, atol=1e-10 * amr.history[0])


def test_stretched_metric_is_not_uniform():
    dom, grids, dx, Jgup, Jinv, rhs = _problem((32, 32, 32), (32, 32, 32), "stretched")
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv)
    assert gpu.metricUniform(0) is None


def test_one_odd_face_value_turns_the_fast_path_off():
    dom, grids, dx, Jgup, Jinv, rhs = _problem((32, 32, 32), (16, 32, 32), "cartesian")
    Jgup[1][2].a[3, 4, 5, 2] *= 1.0 + 2.0 ** -52    # one ulp on one z-face of the second box
    gpu = make_gpu_solver(dom, grids, dx, Jgup, Jinv)
    assert gpu.metricUniform(0) is None


@pytest.mark.parametrize("layout_id", [0, 1, 3])
def test_uniform_metric_kernels_on_amr_levels_bit_exact(oracle, layout_id, monkeypatch):
    """The combination the C3 / C4 benches run: the uniform-metric instantiations (k_gsrb_fused<.., UNI>, k_resid_march<.., UNI>,
    lean interior tiles included) on REFINED levels -- coarse-fine ghosts between the colours, the prolongation folded into the
    first post-sweep, the lean AMRVCycle, residual + restriction in one pass -- forced onto small hierarchies
    (SOMAR_FUSED_MIN_CELLS=0, SOMAR_MARCH_MIN_CELLS=1).  One AMR V-cycle must equal the oracle's and the SOMAR_NO_UNIFORM=1
    twin's bit for bit."""
    from oracle import somar_amr as am
    from somar_amd import api as F
    from helpers import download_valid, make_amr_levels, make_gpu_amr, upload, valid_of
    from test_gpu_amr import LAYOUTS
    so = oracle
    periodic, ratios, boxes = LAYOUTS[layout_id]
    fb = [[so.Box(lo, hi) for lo, hi in lev] for lev in boxes]
    levels = make_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.5), periodic, ratios, fb, variant="cartesian")
    comp = am.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab())
    lmax = len(levels) - 1
    phi = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
    res = [so.random_field(L.grids, 70 + l, (0, 0, 0), L.domain.box) for l, L in enumerate(levels)]
    for l in range(lmax):
        comp.zero_covered(l, res[l])
    comp.init(phi, res, lmax, 0)
    comp.set_bottom_solver(lmax, 0)
    corr = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
    comp.amr_vcycle(corr, res, lmax, lmax, 0)
    monkeypatch.setenv("SOMAR_FUSED_MIN_CELLS", "0")
    monkeypatch.setenv("SOMAR_MARCH_MIN_CELLS", "1")
    got = {}
    for uni in (True, False):
        if uni:
            monkeypatch.delenv("SOMAR_NO_UNIFORM", raising=False)
        else:
            monkeypatch.setenv("SOMAR_NO_UNIFORM", "1")
        gpu = make_gpu_amr(levels, ratios)
        try:
            for v in gpu.levels:
                assert (v.metricUniform(0) is not None) == uni
            for l, v in enumerate(gpu.levels):
                upload(v, F.F_RES, res[l])
                v.setVal(F.F_CORR, 0.0)
            gpu.vcycleAMR(lmax, 0)
            got[uni] = [download_valid(gpu.levels[l], F.F_CORR, levels[l].grids) for l in range(lmax + 1)]
        finally:
            gpu.undefine()
    for l in range(lmax + 1):
        for a, b, w in zip(got[True][l], got[False][l], valid_of(corr[l])):
            np.testing.assert_array_equal(a, b)
            np.testing.assert_array_equal(a, w)
