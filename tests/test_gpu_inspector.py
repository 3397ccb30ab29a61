"""MappedAMRMultiGridInspector (MappedAMRMultiGrid.H:260-298, 1064-1065, 1083-1084) on the GPU solver: the callback sees the
composite residual before every V-cycle and the correction after it -- the same fields, bit for bit, as the oracle's solve
holds at those points; tools/inspect_solve.py turns them into the files OutputMappedAMRMultiGridInspector would write."""
import os
import sys

import numpy as np
import pytest

from helpers import make_amr_levels, make_gpu_amr, upload, valid_of

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_inspector_sees_residuals_and_corrections(oracle, tmp_path):
    from oracle import somar_amr as am
    from somar_amd import api as F
    import inspect_solve
    so = oracle
    periodic, ratios = (True, False, False), [(2, 2, 1)]
    fb = [[so.Box((0, 8, 0), (15, 23, 7)), so.Box((24, 8, 0), (31, 23, 7))]]
    levels = make_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.5), periodic, ratios, fb)
    comp = am.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab())
    gpu = make_gpu_amr(levels, ratios)
    try:
        phi = [so.random_field(L.grids, 5 + l, (1, 1, 1), L.domain.box) for l, L in enumerate(levels)]
        zero = [so.LevelData(L.grids, 1) for L in levels]
        rhs = [so.LevelData(L.grids, 1) for L in levels]
        comp.init(phi, zero, 1, 0)
        comp.compute_amr_residual(rhs, phi, zero, 1, 0, True)
        for r in rhs:
            so.ld_scale(r, -1.0)
        # the oracle's first residual and first correction
        sol = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
        res0 = [so.LevelData(L.grids, 1) for L in levels]
        comp.init(sol, rhs, 1, 0)
        comp.compute_amr_residual(res0, sol, rhs, 1, 0, False)
        for l, v in enumerate(gpu.levels):
            upload(v, F.F_RHS, rhs[l])
        files = inspect_solve.attach(gpu, str(tmp_path / "run"))
        calls = []
        st = gpu.solveAMR(1, 0)
        assert len(files) == 2 * st["iters"]
        first = np.load(files[0])
        assert list(first["meta"]) == [0, 1, 0] and files[0].endswith(".residual.iter.0.npz")
        for l, L in enumerate(levels):
            for gi, g in enumerate(L.grids):
                np.testing.assert_array_equal(first["l%d_b%d" % (l, gi)], res0[l][gi].view(g)[..., 0])
                assert first["l%d_b%d_box" % (l, gi)].tolist() == [list(g.lo), list(g.hi)]
        corr = np.load(files[1])
        assert files[1].endswith(".correction.iter.0.npz")
        assert max(float(np.abs(corr[k]).max()) for k in corr.files if not k.endswith("_box") and k != "meta") > 0.0
        gpu.setInspector(None)
        st2 = gpu.solveAMR(1, 0)
        assert st2["history"] == st["history"] and len(files) == 2 * st["iters"]   # removed: no more files, same solve
        del calls
    finally:
        gpu.undefine()


def test_inspector_writes_chombo_hdf5_level_dumps(oracle, tmp_path):
    """the same fields as Chombo plot files (tools/chombo_hdf5.py, the layout WriteAnisotropicAMRHierarchyHDF5 produces):
    <name>.residual.iter.N.hdf5 read back box by box equals the oracle's composite residual"""
    from oracle import somar_amr as am
    from somar_amd import api as F
    import chombo_hdf5 as ch
    import inspect_solve
    if ch.lib() is None:
        pytest.skip("no HDF5 C library on this box")
    so = oracle
    periodic, ratios = (True, False, False), [(2, 2, 1)]
    fb = [[so.Box((0, 8, 0), (15, 23, 7)), so.Box((24, 8, 0), (31, 23, 7))]]
    levels = make_amr_levels(so, am, (16, 16, 8), (2.0, 1.0, 0.5), periodic, ratios, fb)
    comp = am.AMRComposite(levels, ratios, so.BCHolder(), so.BiCGStab())
    gpu = make_gpu_amr(levels, ratios)
    try:
        phi = [so.random_field(L.grids, 5 + l, (1, 1, 1), L.domain.box) for l, L in enumerate(levels)]
        zero = [so.LevelData(L.grids, 1) for L in levels]
        rhs = [so.LevelData(L.grids, 1) for L in levels]
        comp.init(phi, zero, 1, 0)
        comp.compute_amr_residual(rhs, phi, zero, 1, 0, True)
        for r in rhs:
            so.ld_scale(r, -1.0)
        sol = [so.LevelData(L.grids, 1, (1, 1, 1)) for L in levels]
        res0 = [so.LevelData(L.grids, 1) for L in levels]
        comp.init(sol, rhs, 1, 0)
        comp.compute_amr_residual(res0, sol, rhs, 1, 0, False)
        for l, v in enumerate(gpu.levels):
            upload(v, F.F_RHS, rhs[l])
        files = inspect_solve.attach(gpu, str(tmp_path / "run"), "hdf5", levels[0].dx, ratios)
        st = gpu.solveAMR(1, 0)
        assert len(files) == 2 * st["iters"] and files[0].endswith(".residual.iter.0.hdf5")
        for l, L in enumerate(levels):
            got = ch.read_level(files[0], l)
            assert got["boxes"].tolist() == [list(g.lo) + list(g.hi) for g in L.grids]
            assert got["vec_dx"] == tuple(L.dx)
            for gi, g in enumerate(L.grids):
                a = got["data"][got["offsets"][gi]:got["offsets"][gi + 1]]
                np.testing.assert_array_equal(a, res0[l][gi].view(g)[..., 0].ravel(order="F"))
    finally:
        gpu.undefine()
