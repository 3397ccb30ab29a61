#!/usr/bin/env python3
"""Viscous-type Helmholtz solve on one level: one Crank-Nicolson step (somar_heat_step, scheme 1) of
d(phi)/dt = nu L[phi] + src on n^3 cells, stretched diagonal metric, homogeneous Dirichlet walls on all sides -- the
shape of the reference's single-component viscous solves (MappedLevelCrankNicolson through AMRNavierStokes::
defineViscousMGSolver).  Prints one JSON line: solve time, V-cycles, ms per V-cycle.  Not the driver's bench (bench.py, C2).

    python tools/bench_heat.py --n 512 --nu 1e-3 --dt 0.1 [--scheme 1]
    SOMAR_FUSED_MIN_CELLS=100000000000 python tools/bench_heat.py ...     # A/B: two-pass smoother, direct-load residual
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--nu", type=float, default=1e-3)
    ap.add_argument("--dt", type=float, default=0.1)
    ap.add_argument("--scheme", type=int, default=1)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    from somar_amd import api as F
    from somar_amd import synthetic
    n = args.n
    L = (1.0, 1.0, 1.0)
    dx = tuple(L[d] / n for d in range(3))
    s = F.AMRPressureSolver()
    s.define((0, 0, 0), (n - 1,) * 3, (False, False, False), dx, [((0, 0, 0), (n - 1,) * 3)], bc_type=[F.BC_DIRI] * 6,
             alpha=1.0, beta=args.nu)
    s.setBCValues([0.0] * 6)
    jg, jinv = synthetic.stretched_diagonal_metric((0, 0, 0), (n - 1,) * 3, dx, L)
    s.setMetricOrtho(0, jg[0], jg[1], jg[2], jinv)
    del jg, jinv
    s.finalize()
    s.fillHash(F.F_HEAT_OLD, 12345)
    s.fillHash(F.F_HEAT_SRC, 54321)
    st = s.heatStep(args.scheme, args.dt)          # warm-up (graph capture, first-touch)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        st = s.heatStep(args.scheme, args.dt)
    s.sync()
    dt = (time.perf_counter() - t0) / args.reps
    cycles = st["iters"] * (2 if args.scheme == 2 else 1)
    print(json.dumps({"workload": "Crank-Nicolson" if args.scheme == 1 else ("backward Euler" if args.scheme == 0 else "TGA"),
                      "n": n, "nu": args.nu, "dt": args.dt, "ms_per_step": 1e3 * dt, "vcycles_last_solve": st["iters"],
                      "ms_per_vcycle_incl_residuals": 1e3 * dt / max(cycles, 1), "exit_status": st["exitStatus"],
                      "reduction": st["history"][-1] / st["history"][0],
                      "fused_min_cells": os.environ.get("SOMAR_FUSED_MIN_CELLS", "default")}))
    s.undefine()


if __name__ == "__main__":
    main()
