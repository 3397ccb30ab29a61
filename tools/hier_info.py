#!/usr/bin/env python3
"""Prints the multigrid hierarchy (cells and boxes per depth) of every AMR level of a bench config: where the bottom solver runs.
    python tools/hier_info.py c3 c4 c5"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_amr import build_hierarchy  # noqa: E402

for cfg in sys.argv[1:]:
    gpu, levels, cells, t_def, dx0, ratios = build_hierarchy(cfg)
    out = {"config": cfg, "levels": []}
    for v in gpu.levels:
        out["levels"].append([v.levelInfo(d) for d in range(v.depth())])
    print(json.dumps(out))
    gpu.undefine()
