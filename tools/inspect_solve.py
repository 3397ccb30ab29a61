#!/usr/bin/env python3
"""Solver inspector: dumps what the reference's OutputMappedAMRMultiGridInspector writes -- the composite residual before
every V-cycle and the correction after it, per level and box -- from a solve on the GPU (calculus/AMRElliptic/
MappedAMRMultiGrid.H:305-362: "<name>.residual.iter.N.hdf5", "<name>.correction.iter.N.hdf5" through outputAMR).

    python tools/inspect_solve.py --config c3 --scale 8 --name /tmp/run1

Two formats, same content and naming:
  * <name>.residual.iter.N.hdf5 / <name>.correction.iter.N.hdf5 in Chombo's plot-file layout (what
    WriteAnisotropicAMRHierarchyHDF5 produces, utils/Printing.cpp:736-827), written by tools/chombo_hdf5.py through ctypes on
    libhdf5 -- when that library can be loaded (it is in /opt/conda/lib of this image; h5py is not) and every level's boxes
    are on this rank;
  * <name>....npz NumPy archives otherwise / additionally (--npz): per level l and box b `l{l}_b{b}` (the valid-region array,
    Fortran order) and `l{l}_b{b}_box` (lo, hi); `meta` = (l_min, l_max, iter).
Used by tests/test_gpu_inspector.py."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np  # noqa: E402


def attach(gpu, name, fmt="auto", dx0=None, ratios=None):
    """register a dumping inspector on a hierarchy (AMRPressureSolver after defineAMR + finalize); -> list of files written.
    fmt: "npz", "hdf5" (needs dx0 = level-0 spacing and ratios = refinement ratio per level) or "auto" (hdf5 when possible)"""
    from somar_amd import api as F
    import chombo_hdf5 as ch
    written = []
    use_h5 = fmt == "hdf5" or (fmt == "auto" and ch.lib() is not None and dx0 is not None and ratios is not None)
    if fmt == "hdf5" and ch.lib() is None:
        raise RuntimeError("no HDF5 C library (tools/chombo_hdf5.py); use fmt='npz'")

    def record(kind, it, lmin, lmax):
        field = F.F_RES if kind == 0 else F.F_CORR
        what = "residual" if kind == 0 else "correction"
        if use_h5:
            levels = []
            for l in range(lmin, lmax + 1):
                v = gpu.levels[l]
                levels.append([(lo, hi, v.download(field, p, (0, 0, 0))) for p in range(v.num_local_patches)
                               for lo, hi, gi in [v.patch_box(p)]])
            info = gpu.levels[lmin].levelInfo(0)
            dxl = [float(x) for x in info["dx"]]
            path = "%s.%s.iter.%d.hdf5" % (name, what, it)
            ch.write_hierarchy(path, levels, info["domain"], dxl, [tuple(r) for r in ratios[lmin:lmax]])
            written.append(path)
            return
        out = {"meta": np.array([lmin, lmax, it])}
        for l in range(lmin, lmax + 1):
            v = gpu.levels[l]
            for p in range(v.num_local_patches):
                lo, hi, gi = v.patch_box(p)
                out["l%d_b%d" % (l, gi)] = v.download(field, p, (0, 0, 0))
                out["l%d_b%d_box" % (l, gi)] = np.array([lo, hi])
        path = "%s.%s.iter.%d.npz" % (name, what, it)
        np.savez(path, **out)
        written.append(path)

    gpu.setInspector(record)
    return written


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--scale", type=int, default=8)
    ap.add_argument("--name", default="/tmp/somar_inspect")
    ap.add_argument("--format", default="auto", choices=["auto", "hdf5", "npz"])
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
    files = attach(gpu, args.name, args.format, dx0, ratios)
    st = gpu.solveAMR(nlev - 1, 0)
    print("iters %d, exit %d, %d files: %s ..." % (st["iters"], st["exitStatus"], len(files), files[:2]))
    gpu.undefine()


if __name__ == "__main__":
    main()
