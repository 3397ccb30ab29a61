#!/usr/bin/env python3
"""Diagnostic: the composite solve of tests/test_gpu_amr_fullsize.py on config c3 / c4 at several scales, printing the
residual history, exit status and the first AMR V-cycle's contraction on a COMPATIBLE residual."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402


def run(config, scale, box):
    from bench_amr import build_hierarchy
    from somar_amd import api as F
    gpu, levels, cells, _, dx0, ratios = build_hierarchy(config, scale, box)
    nlev = len(levels)
    try:
        for l, v in enumerate(gpu.levels):
            v.fillHash(F.F_PHI, 5 + l)
            v.setVal(F.F_RHS, 0.0)
        for ilev in range(nlev):
            gpu.residualLevel(nlev - 1, 0, ilev)
        for l in range(nlev - 1):
            gpu.zeroCovered(l, F.F_RES)
        total, mag = 0.0, 0.0
        dx = list(dx0)
        for l, v in enumerate(gpu.levels):
            if l > 0:
                dx = [a / b for a, b in zip(dx, ratios[l - 1])]
            v.setVal(F.F_SCRATCH, 1.0)
            total += v.dotProduct(F.F_RES, F.F_SCRATCH) * float(np.prod(dx))
            mag = max(mag, v.norm(F.F_RES, 0))
        for v in gpu.levels:
            for q in range(v.num_local_patches):
                v.upload(F.F_RHS, q, v.download(F.F_RES, q, (0, 0, 0)), (0, 0, 0))
        # one AMR V-cycle on the compatible residual RES (= RHS), from zero
        for v in gpu.levels:
            v.setVal(F.F_CORR, 0.0)
        gpu.vcycleAMR(nlev - 1, 0)
        r0 = max(v.norm(F.F_RES, 0) for v in gpu.levels)
        for ilev in range(nlev):
            gpu.residualLevel(nlev - 1, 0, ilev, res_field=F.F_SCRATCH, phi_field=F.F_CORR, rhs_field=F.F_RES)
        for l in range(nlev - 1):
            gpu.zeroCovered(l, F.F_SCRATCH)
        r1 = max(v.norm(F.F_SCRATCH, 0) for v in gpu.levels)
        per_level = [v.norm(F.F_SCRATCH, 0) for v in gpu.levels]
        try:
            st = gpu.solveAMR(nlev - 1, 0)
        except Exception as e:   # noqa: BLE001
            st = dict(gpu.stats or {}, error=str(e))
        print(json.dumps({"config": config, "scale": scale, "box": box, "cells": cells, "conservation": total / (mag * 90.0),
                          "vcycle_contraction": r1 / r0, "vcycle_res_per_level": per_level, "r0": r0,
                          "iters": st.get("iters"), "exit": st.get("exitStatus"), "history": st.get("history"),
                          "mg_depth": [v.depth() for v in gpu.levels],
                          "ratios0": [v.mgRefRatios() for v in gpu.levels]}), flush=True)
    finally:
        gpu.undefine()


if __name__ == "__main__":
    for spec in sys.argv[1:]:
        c, s, b = spec.split(":")
        run(c, int(s), int(b))
