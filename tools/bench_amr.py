#!/usr/bin/env python3
"""AMR V-cycle timing on BASELINE configs C3 / C4 (Cartesian LockExchange-shaped hierarchies), 1 GPU.

    python tools/bench_amr.py --config c3 [--scale 1] [--steps 10]

C3: 512x512x64 base, L = (15,3,2), y periodic, level 1 = (2,2,1) refinement of the central half in x.
C4: 1024x1024x128 base + two (2,2,1) levels (central half, central quarter in x).
c5: BASELINE C5's shape -- terrain-following NON-diagonal metric (19-point kernels), 512x512x64 base + three (2,2,1) levels
    nested around the topographic bump, boxes 64x64x64 (somar_amd/synthetic.py::c5_hierarchy).
le3d: the reference's own exec/inputs.LockExchange_Cartesian3D.machine shape -- base nx = 64 x 96 x 64 times
      --mult (default 4: 256x384x256), one level refined by (4,1,1) (amr.refratio_lev0) over the central half in x:
      forced (2,1,1) MG depth + mini V-cycles on the fine level.
le2d: exec/inputs.LockExchange_Cartesian2D.machine -- 2-D (space_dim 2), base nx = 128 x 64 times --mult, L = (15, 2),
      one level refined by (4,1) over the central half in x.
--scale s divides every extent by s (parity-sized runs).  A step = one AMRVCycle (MappedAMRMultiGrid.H:1498) from a
zero correction on a hash-random residual with covered cells zeroed; pre/post/bottom = 4/4/2.
Prints one JSON line.  Not the driver's bench (that is bench.py, config C2)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def build_hierarchy(config="c3", scale=1, box=128, pre=4, post=4, bottom=2, mult=4, comm=None, nranks=1, alpha=0.0, beta=1.0):
    """-> (AMRPressureSolver (finalized), levels' boxes, LOCAL cells per level, define seconds, dx0, ratios)"""
    from somar_amd import api as F
    from somar_amd import synthetic
    H = synthetic.c5_hierarchy(scale, min(box, 64), nranks) if config == "c5" else synthetic.lockexchange_hierarchy(config, scale, box, mult, nranks)
    levels, ratios, dx0, flat, n0 = H["levels"], H["ratios"], H["dx0"], H["flat"], H["n0"]
    gpu = F.AMRPressureSolver()
    if flat:
        gpu.setSpaceDim(2)
    p = gpu._p
    gpu.setAMRMGParameters(p.imin, p.imax, p.eps, -1, p.num_smooth_precond, pre, post, bottom, p.precond_mode, 1, p.num_mg,
                           p.hang, p.norm_thresh, 0)
    t0 = time.perf_counter()
    tm = {"host_metric_arrays": 0.0, "metric_upload": 0.0}
    gpu.defineAMR((0, 0, 0), tuple(a - 1 for a in n0), H["periodic"], dx0, ratios, levels,
                  owners_per_level=H["owners"], comm=comm, alpha=alpha, beta=beta)
    tm["define_tables"] = time.perf_counter() - t0
    cells = []
    dxl = list(dx0)
    for l, v in enumerate(gpu.levels):
        tot = 0
        if l > 0:
            dxl = [a / b for a, b in zip(dxl, ratios[l - 1])]
        for q in range(v.num_local_patches):
            lo, hi, _ = v.patch_box(q)
            shp = [h - a + 1 for a, h in zip(lo, hi)]
            tot += shp[0] * shp[1] * shp[2]
            if H.get("metric") == "terrain" and not os.environ.get("SOMAR_BENCH_HOST_METRIC"):
                continue   # produced on the device below (setMetricMap) from the level's nodal depth
            if H.get("metric") == "terrain":
                ta = time.perf_counter()
                jg, jinv = synthetic.terrain_metric(lo, hi, dxl, H["L"])   # non-diagonal: the 19-point kernels
                tb = time.perf_counter()
                v.setMetricFull(q, jg[0], jg[1], jg[2], jinv)
                tm["host_metric_arrays"] += tb - ta
                tm["metric_upload"] += time.perf_counter() - tb
                continue
            if not os.environ.get("SOMAR_BENCH_HOST_METRIC"):
                continue   # Cartesian: J = 1, Jg^aa = 1 -- written on the device below (setMetricUniform), no host arrays
            ta = time.perf_counter()
            ones = [np.ones((shp[0] + (d == 0), shp[1] + (d == 1), shp[2] + (d == 2)), order="F") for d in range(3)]
            one_c = np.ones(shp, order="F")
            tb = time.perf_counter()
            v.setMetricOrtho(q, ones[0], ones[1], None if flat else ones[2], one_c)   # Cartesian: J = 1, Jg^aa = 1
            tm["host_metric_arrays"] += tb - ta
            tm["metric_upload"] += time.perf_counter() - tb
        cells.append(tot)
        if H.get("metric") == "terrain" and not os.environ.get("SOMAR_BENCH_HOST_METRIC") and v.num_local_patches:
            # BathymetricBaseMap on the device: only the NODAL depth d = H (1 - s) of the region this rank's boxes cover
            # (nodes lo-1 .. hi+2) is evaluated on the host, a 2-D array
            ta = time.perf_counter()
            bx = [v.patch_box(q) for q in range(v.num_local_patches)]
            nlo = [min(b[0][d] for b in bx) - 1 for d in range(2)]
            nhi = [max(b[1][d] for b in bx) + 2 for d in range(2)]
            depth = synthetic.terrain_nodal_depth(nlo, nhi, dxl, H["L"])
            tb = time.perf_counter()
            v.setMetricMap(F.MAP_BATHYMETRIC, H["L"], depth, nlo)
            tm["host_metric_arrays"] += tb - ta
            tm["metric_upload"] += time.perf_counter() - tb
        if H.get("metric") != "terrain" and not os.environ.get("SOMAR_BENCH_HOST_METRIC"):
            ta = time.perf_counter()
            v.setMetricUniform(1.0, 1.0, 1.0, 1.0)   # CartesianMap::fill_Jgup / fill_Jinv on the device
            tm["metric_upload"] += time.perf_counter() - ta
    ta = time.perf_counter()
    gpu.finalize()
    gpu.levels[0].sync()
    tm["finalize"] = time.perf_counter() - ta
    build_hierarchy.last_breakdown = {k: round(v, 4) for k, v in tm.items()}
    return gpu, levels, cells, time.perf_counter() - t0, dx0, ratios


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3")
    ap.add_argument("--scale", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--box", type=int, default=128, help="box edge in x,y (z is never split)")
    ap.add_argument("--mult", type=int, default=4, help="le3d: multiple of the reference input's 64x96x64 base grid")
    args = ap.parse_args()
    from somar_amd import api as F
    s = args.scale
    gpu, levels, cells, t_def, dx0, ratios = build_hierarchy(args.config, s, args.box, mult=args.mult)
    nlev = len(levels)
    # a compatible composite residual: RES := 0 - L_composite[hash-random phi], covered coarse cells zeroed
    for l, v in enumerate(gpu.levels):
        v.fillHash(F.F_PHI, 12345 + l)
        v.setVal(F.F_RHS, 0.0)
    for ilev in range(nlev):
        gpu.residualLevel(nlev - 1, 0, ilev)
    for l in range(nlev - 1):
        gpu.zeroCovered(l, F.F_RES)

    def step():
        for v in gpu.levels:
            v.setVal(F.F_CORR, 0.0)
        gpu.vcycleAMR(nlev - 1, 0)

    for _ in range(args.warmup):
        step()
    gpu.levels[0].sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    gpu.levels[0].sync()
    dt = (time.perf_counter() - t0) / args.steps
    r0 = max(v.norm(F.F_RES, 0) for v in gpu.levels)
    for ilev in range(nlev):
        gpu.residualLevel(nlev - 1, 0, ilev, res_field=F.F_SCRATCH, phi_field=F.F_CORR, rhs_field=F.F_RES)
    for l in range(nlev - 1):
        gpu.zeroCovered(l, F.F_SCRATCH)
    r1 = max(v.norm(F.F_SCRATCH, 0) for v in gpu.levels)
    print(json.dumps({"config": args.config, "amr_vcycle_contraction": r1 / r0, "scale": s, "levels": nlev, "cells_per_level": cells,
                      "boxes_per_level": [len(b) for b in levels], "define_seconds": t_def, "define_breakdown_s": getattr(build_hierarchy, "last_breakdown", None), "ms_per_amr_vcycle": dt * 1e3,
                      "amr_vcycles_per_s": 1.0 / dt, "mg_depth_per_level": [v.depth() for v in gpu.levels],
                      "cell_updates_per_s": sum(cells) * 8 / dt}))
    gpu.undefine()


if __name__ == "__main__":
    main()
