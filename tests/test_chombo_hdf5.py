"""tools/chombo_hdf5.py: the inspector's level dumps in Chombo's plot-file layout, written through ctypes on libhdf5 (no h5py in
the image).  CPU test: a two-level hierarchy is written and read back through the same library; the file starts with the HDF5
signature.  Skipped where no libhdf5 can be loaded.  (Layout unpinned: the reference ships no HDF5 file.)"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_write_and_read_back_a_two_level_hierarchy(tmp_path):
    import chombo_hdf5 as ch
    if ch.lib() is None:
        pytest.skip("no HDF5 C library in this environment")
    rng = np.random.default_rng(3)
    l0 = [((0, 0, 0), (7, 7, 3), rng.uniform(-1, 1, (8, 8, 4))), ((8, 0, 0), (15, 7, 3), rng.uniform(-1, 1, (8, 8, 4)))]
    l1 = [((8, 4, 0), (23, 11, 3), rng.uniform(-1, 1, (16, 8, 4)))]
    path = str(tmp_path / "run.residual.iter.3.hdf5")
    ch.write_hierarchy(path, [l0, l1], ((0, 0, 0), (15, 7, 3)), (0.5, 0.25, 0.125), [(2, 2, 1)], time=1.5)
    assert open(path, "rb").read(8) == b"\x89HDF\r\n\x1a\n"
    a = ch.read_level(path, 0)
    np.testing.assert_array_equal(a["boxes"], [[0, 0, 0, 7, 7, 3], [8, 0, 0, 15, 7, 3]])
    np.testing.assert_array_equal(a["offsets"], [0, 256, 512])
    np.testing.assert_array_equal(a["data"][:256], l0[0][2].ravel(order="F"))
    np.testing.assert_array_equal(a["data"][256:], l0[1][2].ravel(order="F"))
    assert a["vec_dx"] == (0.5, 0.25, 0.125) and a["prob_domain"] == (0, 0, 0, 15, 7, 3)
    b = ch.read_level(path, 1)
    np.testing.assert_array_equal(b["boxes"], [[8, 4, 0, 23, 11, 3]])
    np.testing.assert_array_equal(b["data"], l1[0][2].ravel(order="F"))
    assert b["vec_dx"] == (0.25, 0.125, 0.125) and b["prob_domain"] == (0, 0, 0, 31, 15, 3)   # refined by (2, 2, 1)
