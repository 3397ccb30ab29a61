#!/usr/bin/env python3
"""Kernels longer than a threshold inside the last `window_ms` of a rocprofv3 --kernel-trace CSV, in launch order, and the
totals by kernel over that window: where a cycle's time goes besides its sweeps.
    python tools/big_kernels.py <dir> <out.txt> [window_ms=160] [min_us=50]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d, out = sys.argv[1], sys.argv[2]
    window = float(sys.argv[3]) if len(sys.argv) > 3 else 160.0
    min_us = float(sys.argv[4]) if len(sys.argv) > 4 else 50.0
    f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    end = int(rows[-1]["End_Timestamp"])
    rows = [r for r in rows if int(r["Start_Timestamp"]) >= end - window * 1e6]

    def name(r):
        return r["Kernel_Name"].split("(")[0].replace("void ", "").replace("somar::", "")[:60]

    tot = defaultdict(lambda: [0, 0.0])
    lines = []
    for r in rows:
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        t = tot[name(r)]
        t[0] += 1
        t[1] += us
        if us >= min_us:
            lines.append("%-62s %10d %9.1f" % (name(r), int(r["Grid_Size_X"]), us))
    busy = sum(v[1] for v in tot.values())
    head = ["last %.0f ms: %d dispatches, busy %.2f ms" % (window, len(rows), busy / 1e3), "", "totals by kernel (ms):"]
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:25]:
        head.append("  %-60s x%-6d %8.2f" % (k, v[0], v[1] / 1e3))
    open(out, "w").write("\n".join(head + ["", "kernels >= %.0f us in order:" % min_us] + lines) + "\n")


if __name__ == "__main__":
    main()
