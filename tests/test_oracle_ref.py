"""oracle/_ref = the reference's own utils/ThomasAlgorithm.f90 compiled with flang
(oracle/Makefile).  It pins the oracle's tridiagonal restatement against real reference code."""
import ctypes as C
import os

import numpy as np
import pytest

REF = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libthomas_ref.so")


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built (needs /root/reference + flang)")
def test_tridiag_matches_reference_fortran(oracle):
    ref = C.CDLL(REF)
    rng = np.random.default_rng(9)
    P = C.POINTER(C.c_double)
    for n in (3, 8, 64, 257):
        a = -rng.uniform(0.5, 1.0, n - 1)
        c = -rng.uniform(0.5, 1.0, n - 1)
        b = 2.5 + rng.uniform(0.0, 1.0, n)
        d = rng.uniform(-1, 1, n)
        x1, x2 = np.zeros(n), np.zeros(n)
        oracle.lib().orc_solve_tridiag(a.ctypes.data_as(P), b.ctypes.data_as(P), c.ctypes.data_as(P),
                                       d.ctypes.data_as(P), x1.ctypes.data_as(P), n)
        nn = C.c_int(n)
        ref.solve_tridiag_(a.ctypes.data_as(P), b.ctypes.data_as(P), c.ctypes.data_as(P),
                           d.ctypes.data_as(P), x2.ctypes.data_as(P), C.byref(nn))
        # flang may contract a*b+c into FMA; allow a few ulp
        np.testing.assert_allclose(x1, x2, rtol=1e-14, atol=1e-16)
