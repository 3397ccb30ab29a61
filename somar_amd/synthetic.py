"""Synthetic inputs of the BASELINE configs, for bench.py and tools/: plain numpy, no oracle.

The C2 "mapped-Cartesian" metric is the separable stretch s_a(x) = 1 + 0.3 sin(2 pi x / L_a + a):
Jg^{aa} = s_b s_c / s_a at a-face centres, Jinv = 1 / (s_0 s_1 s_2) at cell centres -- a diagonal, non-uniform
metric with the structure of the reference's CylindricalMap, handed to the solver as full coefficient arrays
exactly as LevelGeometry::getFCJgupPtr / getCCJinvPtr would (projection/AMRPressureSolver.cpp:163-463).
tests/test_synthetic.py checks these arrays against the oracle's generator bit for bit, so the timed problem and
the parity-tested problem are the same one.
"""
import numpy as np


def stretch_factor(a, x, L):
    return 1.0 + 0.3 * np.sin(2.0 * np.pi * x / L + a)


def stretched_diagonal_metric(lo, hi, dx, L, ndim=3):
    """Metric of one box [lo, hi] (cell indices, inclusive).
    -> ([Jg^{00} on x-faces, Jg^{11} on y-faces, Jg^{22} on z-faces] (None beyond ndim), Jinv), Fortran order;
    face array d has one more entry in direction d, face index i = low face of cell i."""
    def coords(face_dir):
        xs = []
        for d in range(3):
            n = hi[d] - lo[d] + 1 + (1 if d == face_dir else 0)
            idx = np.arange(lo[d], lo[d] + n, dtype=np.float64)
            xs.append((idx if d == face_dir else idx + 0.5) * dx[d])
        return xs

    def s(a, x):
        return stretch_factor(a, x, L[a]) if a < ndim else np.ones_like(x)

    def planes(x):
        return [s(0, x[0])[:, None, None], s(1, x[1])[None, :, None], s(2, x[2])[None, None, :]]

    jg = [None, None, None]
    for d in range(ndim):
        sv = planes(coords(d))
        num = np.ones(tuple(hi[q] - lo[q] + 1 + (1 if q == d else 0) for q in range(3)))
        for e in range(3):
            if e != d:
                num = num * sv[e]
        jg[d] = np.asfortranarray(num / sv[d])
    sv = planes(coords(-1))
    jinv = np.asfortranarray(1.0 / (sv[0] * sv[1] * sv[2]))
    return jg, jinv


def slab_partition(n, nparts):
    """One box per GPU for the 512^3 problem: split z, then y; x is never split (x rows stay long: coalescing, and
    the marching kernels' tile columns divide a 512-wide box evenly).  -> [(lo, hi)] in rank order."""
    split = {1: (1, 1, 1), 2: (1, 1, 2), 4: (1, 2, 2), 8: (1, 2, 4)}.get(nparts)
    if split is None:
        raise ValueError("number of parts must be 1, 2, 4 or 8")
    sz = [n // s for s in split]
    boxes = []
    for k in range(split[2]):
        for j in range(split[1]):
            for i in range(split[0]):
                lo = (i * sz[0], j * sz[1], k * sz[2])
                boxes.append((lo, tuple(a + b - 1 for a, b in zip(lo, sz))))
    return boxes
