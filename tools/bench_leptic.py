#!/usr/bin/env python3
"""Leptic level solve timing on a thin (leptic) domain, 1 GPU.

    python tools/bench_leptic.py [--n 512 512 64] [--height 0.01] [--box 128] [--order 2] [--steps 5]

Domain n cells over L = (15, 15, height), separable stretched diagonal metric, Neumann boundaries, vertically
complete boxes of box x box x nz.  A step = one LevelLepticSolver::solve (LevelLepticSolver.cpp:646-956) from
phi = 0 on a hash-random, mean-free right-hand side.  Also times the level's own multigrid (LevelGSRB V-cycles) on
the same problem for context.  Prints one JSON line.  Not the driver's bench (that is bench.py, config C2)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def stretch(a, x, L):
    return 1.0 + 0.3 * np.sin(2.0 * np.pi * x / L + a)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, nargs=3, default=[512, 512, 64])
    ap.add_argument("--height", type=float, default=0.01)
    ap.add_argument("--box", type=int, default=128)
    ap.add_argument("--order", type=int, default=2)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    a = ap.parse_args()
    from somar_amd import LevelLepticSolver
    from somar_amd.api import F_PHI, F_RHS
    n = tuple(a.n)
    L = (15.0, 15.0, a.height)
    dx = tuple(L[d] / n[d] for d in range(3))
    boxes = [((i, j, 0), (min(i + a.box, n[0]) - 1, min(j + a.box, n[1]) - 1, n[2] - 1))
             for j in range(0, n[1], a.box) for i in range(0, n[0], a.box)]
    s = LevelLepticSolver()
    s.params.max_order = a.order
    s.params.domain_height = a.height
    t0 = time.perf_counter()
    s.define((0, 0, 0), tuple(x - 1 for x in n), (False, False, False), dx, boxes)
    lv = s.level
    for q in range(lv.num_local_patches):
        lo, hi, _ = lv.patch_box(q)
        sv = []
        for d in range(3):
            cc = (np.arange(lo[d], hi[d] + 1) + 0.5) * dx[d]
            fc = np.arange(lo[d], hi[d] + 2) * dx[d]
            sv.append((stretch(d, cc, L[d]), stretch(d, fc, L[d])))

        def field(face):
            v = [sv[d][1] if d == face else sv[d][0] for d in range(3)]
            return v[0][:, None, None], v[1][None, :, None], v[2][None, None, :]
        jg = []
        for d in range(3):
            s0, s1, s2 = field(d)
            t = [s0, s1, s2]
            num = 1.0
            for e in range(3):
                if e != d:
                    num = num * t[e]
            jg.append(np.asfortranarray(num / t[d]))
        s0, s1, s2 = field(-1)
        lv.setMetricOrtho(q, jg[0], jg[1], jg[2], np.asfortranarray(1.0 / (s0 * s1 * s2)))
    s.finalize()
    t_define = time.perf_counter() - t0
    lv.fillHash(F_RHS, 7)
    lv.removeMean(F_RHS)

    times, st = [], None
    for it in range(a.warmup + a.steps):
        lv.setVal(F_PHI, 0.0)
        lv.sync()
        t0 = time.perf_counter()
        st = s.solve()
        lv.sync()
        if it >= a.warmup:
            times.append(time.perf_counter() - t0)
    # the level's own multigrid on the same problem
    lv.setVal(F_PHI, 0.0)
    lv.sync()
    t0 = time.perf_counter()
    mg = lv.solveResident(zeroPhi=True)
    lv.sync()
    t_mg = time.perf_counter() - t0
    cells = n[0] * n[1] * n[2]
    out = {"metric": "leptic level solves/sec", "value": 1.0 / (sum(times) / len(times)), "unit": "solves/s",
           "ms_per_solve": 1e3 * sum(times) / len(times), "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": "leptic solve, %dx%dx%d over L=(15,15,%g), stretched diagonal metric, %d boxes %dx%dx%d, "
                                  "max_order %d" % (n + (a.height, len(boxes), a.box, a.box, n[2], a.order)),
                      "cells": cells, "define_seconds": t_define},
           "leptic": {"exitStatus": st["exitStatus"], "orders": st["orders"], "horizSolves": st["horizSolves"],
                      "usedFullSolver": st["usedFullSolver"], "relResNorms": [x / st["resNorms"][0] for x in st["resNorms"]],
                      "horiz_iters": st["horiz"]["iters"]},
           "level_multigrid": {"ms": 1e3 * t_mg, "iters": mg["iters"], "exitStatus": mg["exitStatus"],
                               "rel_final": mg["final_rnorm"] / mg["initial_rnorm"]}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
