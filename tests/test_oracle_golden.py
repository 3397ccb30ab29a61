"""The oracle against its own committed outputs (tests/golden/oracle_small.npz, made by tests/golden/make_golden.py).
Not reference vectors -- the reference has none for this path -- but a bit-level regression pin of the checker."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_oracle_reproduces_its_committed_vectors(oracle):
    import make_golden
    want = np.load(os.path.join(HERE, "golden", "oracle_small.npz"))
    got = make_golden.compute()
    assert sorted(got) == sorted(want.files)
    for k in want.files:
        if k in ("solve_history", "tga_history", "amr_tga_history"):
            np.testing.assert_allclose(got[k], want[k], rtol=1e-12)   # numpy's pairwise sums may differ across builds
        elif k in ("tga_phi_box0", "amr_tga_fine_phi_box0"):
            np.testing.assert_allclose(got[k], want[k], rtol=0, atol=1e-12)    # behind two iterative solves
        else:
            np.testing.assert_array_equal(got[k], want[k], err_msg=k)
