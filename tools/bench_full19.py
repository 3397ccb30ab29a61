#!/usr/bin/env python3
"""The 19-point (non-diagonal metric) kernels at size: one level, one box n^3, synthetic sheared map evaluated in numpy.

    python tools/bench_full19.py --n 256 [--reps 10] [--path march|direct]

Reports per-launch HIP-event times of the GSRB colour pass (prof slot 0) and of the residual (slot 1), the
algorithmic rates (GSRB sweep 120 B/cell, residual 112 B/cell, unit = 232 B/cell) and the fraction of
the 8 TB/s HBM peak.  --path direct forces the direct-load kernels of full19.hip (the round-1 state) for A/B runs."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def sheared_metric(n, dx, L, amp=(0.25, 0.2, 0.15)):
    """x = xi + a0 sin(2 pi eta / L1), y = eta + a1 sin(2 pi zeta / L2), z = zeta + a2 sin(2 pi xi / L0) scaled:
    a smooth non-orthogonal map; J g^{ab} on a-faces (3 comps, comp slowest) and 1/J at cell centres."""
    def at(face_dir):
        xs = []
        for d in range(3):
            m = n + (1 if d == face_dir else 0)
            idx = np.arange(m, dtype=np.float64)
            xs.append((idx if d == face_dir else idx + 0.5) * dx[d])
        X = np.meshgrid(*xs, indexing="ij")
        # Jacobian dx_phys/dxi of the sheared map
        k = [2 * np.pi / L[d] for d in range(3)]
        A = np.zeros(X[0].shape + (3, 3))
        A[..., 0, 0] = 1.0
        A[..., 0, 1] = amp[0] * L[0] / (2 * np.pi) * k[1] * np.cos(k[1] * X[1]) * 0.5
        A[..., 1, 1] = 1.0
        A[..., 1, 2] = amp[1] * L[1] / (2 * np.pi) * k[2] * np.cos(k[2] * X[2]) * 0.5
        A[..., 2, 2] = 1.0
        A[..., 2, 0] = amp[2] * L[2] / (2 * np.pi) * k[0] * np.cos(k[0] * X[0]) * 0.5
        Jdet = np.linalg.det(A)
        Ainv = np.linalg.inv(A)           # dxi/dx
        gup = Ainv @ np.swapaxes(Ainv, -1, -2)
        return Jdet, gup
    jg = []
    for d in range(3):
        Jdet, gup = at(d)
        arr = np.empty(Jdet.shape + (3,), order="F")
        for b in range(3):
            arr[..., b] = Jdet * gup[..., d, b]
        jg.append(arr)
    Jdet, _ = at(-1)
    return jg, np.asfortranarray(1.0 / Jdet)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--path", default="march", choices=["march", "direct"])
    ap.add_argument("--metric", default="sheared", choices=["sheared", "bathy"],
                    help="bathy: BathymetricBaseMap of a bump, produced on the device (no host arrays)")
    ap.add_argument("--box", type=int, default=0, help="cut the n^3 domain into boxes of this many cells per side (default: one box)")
    args = ap.parse_args()
    if args.path == "direct":
        os.environ["SOMAR_MARCH_MIN_CELLS"] = "1000000000000"
    from somar_amd import api as F
    n = args.n
    L = (1.0, 1.0, 1.0)
    dx = tuple(L[d] / n for d in range(3))
    t0 = time.perf_counter()
    if args.metric == "sheared":
        jg, jinv = sheared_metric(n, dx, L)
    t_metric = time.perf_counter() - t0
    s = F.AMRPressureSolver()
    p = s._p
    s.setAMRMGParameters(p.imin, p.imax, p.eps, -1, p.num_smooth_precond, 2, 2, 2, p.precond_mode, 1, p.num_mg, p.hang,
                         p.norm_thresh, 0)
    bs = args.box or n
    boxes = [((i, j, k), (i + bs - 1, j + bs - 1, k + bs - 1)) for k in range(0, n, bs) for j in range(0, n, bs) for i in range(0, n, bs)]
    s.define((0, 0, 0), (n - 1,) * 3, (False, False, False), dx, boxes)
    if args.metric == "sheared":
        assert len(boxes) == 1, "--box needs --metric bathy (the map is evaluated per box on the device)"
        s.setMetricFull(0, jg[0], jg[1], jg[2], jinv)
        del jg, jinv
    else:
        t0 = time.perf_counter()
        x = (np.arange(-1, n + 3) * dx[0])[:, None]
        y = (np.arange(-1, n + 3) * dx[1])[None, :]
        depth = L[2] * (0.5 - 0.3 * np.exp(-((x - 0.5) ** 2 + (y - 0.5) ** 2) / 0.0625))
        s.setMetricMap(F.MAP_BATHYMETRIC, L, depth, (-1, -1))
        t_metric = time.perf_counter() - t0
    s.finalize()
    s.fillHash(F.F_RHS, 12345)
    s.removeMean(F.F_RHS)
    s.setVal(F.F_PHI, 0.0)
    s.relax(0, F.F_PHI, F.F_RHS, 1)
    s.residual(0, F.F_RES, F.F_PHI, F.F_RHS)
    s.sync()
    s.profileEnable(True)
    s.relax(0, F.F_PHI, F.F_RHS, args.reps)
    for _ in range(args.reps):
        s.residual(0, F.F_RES, F.F_PHI, F.F_RHS)
    s.sync()
    n_g, ms_g = s.profileGet(0)
    n_r, ms_r = s.profileGet(1)
    s.profileEnable(False)
    # whole sweeps / residuals incl. exchange + ghost programs, host-timed
    s.sync()
    t0 = time.perf_counter()
    s.relax(0, F.F_PHI, F.F_RHS, args.reps)
    s.sync()
    t_sweep = (time.perf_counter() - t0) / args.reps
    t0 = time.perf_counter()
    for _ in range(args.reps):
        s.residual(0, F.F_RES, F.F_PHI, F.F_RHS)
    s.sync()
    t_res = (time.perf_counter() - t0) / args.reps
    r0 = s.norm(F.F_RHS, 0)
    r1 = s.norm(F.F_RES, 0)
    cells = n ** 3
    pass_ms = ms_g / max(n_g, 1)
    res_ms = ms_r / max(n_r, 1)
    out = {"n": n, "box": bs, "fused19_sweeps": s.fused19Sweeps(), "path": args.path, "metric": args.metric, "rows": os.environ.get("SOMAR_FULL_ROWS", "8"), "cells": cells, "metric_seconds": t_metric, "mg_depth": s.depth(),
           "gsrb_colour_pass_ms": pass_ms, "gsrb_sweep_kernel_ms": 2 * pass_ms, "residual_kernel_ms": res_ms,
           "gsrb_sweep_wall_ms": t_sweep * 1e3, "residual_wall_ms": t_res * 1e3,
           "gsrb_sweep_alg_GBs": 120.0 * cells / (2 * pass_ms * 1e-3) / 1e9,
           "residual_alg_GBs": 112.0 * cells / (res_ms * 1e-3) / 1e9,
           "unit_kernel_frac_of_8TBs": 232.0 * cells / ((2 * pass_ms + res_ms) * 1e-3) / 8e12,
           "unit_wall_frac_of_8TBs": 232.0 * cells / (t_sweep + t_res) / 8e12,
           "residual_norm_ratio_after_sweeps": r1 / r0}
    print(json.dumps(out))
    s.undefine()


if __name__ == "__main__":
    main()
