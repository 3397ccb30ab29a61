#!/usr/bin/env python3
"""The launch sequence of ONE V-cycle from a rocprofv3 --kernel-trace CSV: kernel, grid, duration, gap to the previous
kernel -- what the launch-bound coarse tail consists of.

    python tools/tail_sequence.py <dir-with-*_kernel_trace.csv> <out.txt> [anchor-kernel-substring]

A cycle = the dispatches between two consecutive launches of the anchor (default: the INMODE 1 fused sweep at its largest
grid, the first kernel of a V-cycle from a zero correction)."""
import csv
import glob
import os
import sys
from collections import Counter


def main():
    d, out = sys.argv[1], sys.argv[2]
    anchor = sys.argv[3] if len(sys.argv) > 3 else "k_gsrb_fused<16, 1"
    f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))

    def grid(r):
        return int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)

    def name(r):
        return r["Kernel_Name"].split("(")[0].replace("void ", "").replace("somar::", "")

    big = max(grid(r) for r in rows if anchor in r["Kernel_Name"])
    idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"] and grid(r) == big]
    m = len(idx) // 2          # a cycle of the timed loop (the last ones belong to bench.py's own profiled pass)
    a, b = idx[m - 1], idx[m]
    cyc = rows[a:b]
    t0 = int(cyc[0]["Start_Timestamp"])
    lines = ["one V-cycle: %d dispatches, %.3f ms from first start to next cycle's start" %
             (len(cyc), (int(rows[b]["Start_Timestamp"]) - t0) / 1e6)]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in cyc)
    lines.append("busy %.3f ms" % (busy / 1e6))
    small = [r for r in cyc if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) < 30000]
    lines.append("dispatches shorter than 30 us: %d, their busy time %.3f ms" %
                 (len(small), sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in small) / 1e6))
    c = Counter(name(r) for r in small)
    lines.append("by kernel (short ones): " + ", ".join("%s x%d" % kv for kv in c.most_common()))
    lines.append("")
    lines.append("%-58s %10s %9s %8s" % ("kernel", "grid", "dur us", "gap us"))
    prev_end = None
    for r in cyc:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        lines.append("%-58s %10d %9.1f %8.1f" % (name(r)[:58], grid(r), (e - s) / 1e3, gap))
        prev_end = e
    open(out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
