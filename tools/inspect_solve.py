#!/usr/bin/env python3
"""Solver inspector: dumps what the reference's OutputMappedAMRMultiGridInspector writes -- the composite residual before
every V-cycle and the correction after it, per level and box -- from a solve on the GPU (calculus/AMRElliptic/
MappedAMRMultiGrid.H:305-362: "<name>.residual.iter.N.hdf5", "<name>.correction.iter.N.hdf5" through outputAMR).

    python tools/inspect_solve.py --config c3 --scale 8 --name /tmp/run1

h5py / Chombo's HDF5 writer are not in this image, so the files are NumPy archives with the same content and naming:
<name>.residual.iter.N.npz / <name>.correction.iter.N.npz holding, per level l and box b, `l{l}_b{b}` (the valid-region
array, Fortran order) and `l{l}_b{b}_box` (lo, hi); `meta` = (l_min, l_max, iter).  A Chombo-side reader can diff them
against the reference's HDF5 level data box by box.  Used by tests/test_gpu_inspector.py."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402


def attach(gpu, name):
    """register a dumping inspector on a hierarchy (AMRPressureSolver after defineAMR + finalize); -> list of files written"""
    from somar_amd import api as F
    written = []

    def record(kind, it, lmin, lmax):
        field = F.F_RES if kind == 0 else F.F_CORR
        out = {"meta": np.array([lmin, lmax, it])}
        for l in range(lmin, lmax + 1):
            v = gpu.levels[l]
            for p in range(v.num_local_patches):
                lo, hi, gi = v.patch_box(p)
                out["l%d_b%d" % (l, gi)] = v.download(field, p, (0, 0, 0))
                out["l%d_b%d_box" % (l, gi)] = np.array([lo, hi])
        path = "%s.%s.iter.%d.npz" % (name, "residual" if kind == 0 else "correction", it)
        np.savez(path, **out)
        written.append(path)

    gpu.setInspector(record)
    return written


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--scale", type=int, default=8)
    ap.add_argument("--name", default="/tmp/somar_inspect")
    args = ap.parse_args()
    from bench_amr import build_hierarchy
    from somar_amd import api as F
    gpu, levels, cells, _, dx0, ratios = build_hierarchy(args.config, args.scale)
    nlev = len(levels)
    for l, v in enumerate(gpu.levels):
        v.fillHash(F.F_PHI, 12345 + l)
        v.setVal(F.F_RHS, 0.0)
    for ilev in range(nlev):
        gpu.residualLevel(nlev - 1, 0, ilev)
    for l in range(nlev - 1):
        gpu.zeroCovered(l, F.F_RES)
    for v in gpu.levels:
        for q in range(v.num_local_patches):
            v.upload(F.F_RHS, q, v.download(F.F_RES, q, (0, 0, 0)), (0, 0, 0))
    files = attach(gpu, args.name)
    st = gpu.solveAMR(nlev - 1, 0)
    print("iters %d, exit %d, %d files: %s ..." % (st["iters"], st["exitStatus"], len(files), files[:2]))
    gpu.undefine()


if __name__ == "__main__":
    main()
