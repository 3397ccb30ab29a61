#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into the small summaries kept under profiles/.

  python tools/prof_summary.py stats  <dir-with-*_kernel_stats.csv/_kernel_trace.csv>  <out.md>
  python tools/prof_summary.py pmc    <fetch-dir> <write-dir> <out.json> [cells [kernel-substring,kernel-substring,...]]

PMC recipe (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE come from separate passes;
both are in KiB; on gfx950 FETCH_SIZE counts 128-B fills as 64 B, so read bytes are calibrated on a kernel
of the SAME run whose traffic is known exactly -- k_incr (y += a*x over a whole depth-0 field: reads 16 B,
writes 8 B per element) -- and the resulting factor is applied to the stencil kernels.
"""
import csv
import glob
import json
import os
import sys


def _one(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    if not f:
        raise SystemExit("no %s under %s" % (pat, d))
    return f[0]


def stats(d, out):
    rows = list(csv.DictReader(open(_one(d, "*_kernel_stats.csv"))))
    tr = list(csv.DictReader(open(_one(d, "*_kernel_trace.csv"))))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    lines = ["| kernel | calls | total ms | avg us | % | max us |", "|---|---|---|---|---|---|"]
    for r in rows[:16]:
        name = r["Name"].split("(")[0].replace("void ", "").replace("somar::", "")
        lines.append("| %s | %s | %.2f | %.1f | %.1f | %.1f |" % (name, r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                                float(r["AverageNs"]) / 1e3, float(r["Percentage"]),
                                                                float(r["MaxNs"]) / 1e3))
    lines.append("")
    lines.append("total kernel time %.2f ms over %d dispatches" % (tot / 1e6, len(tr)))
    for key in ("k_gsrb_ortho", "k_gsrb_fused", "k_op_ortho<0>"):
        big = {}
        for r in tr:
            if key in r["Kernel_Name"]:
                g = int(r["Grid_Size_X"])
                big.setdefault(g, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        if big:
            g = max(big)
            v = big[g]
            lines.append("%s, largest grid (%d threads = depth 0): %d launches, avg %.1f us, min %.1f, max %.1f"
                         % (key, g, len(v), sum(v) / len(v), min(v), max(v)))
    # by (kernel, grid): one row per multigrid depth, and how much of the wall span the GPU sat idle between kernels
    by = {}
    for r in tr:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("somar::", "")
        key = (name, int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1))
        by.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    lines += ["", "| kernel | grid threads | calls | total ms | avg us |", "|---|---|---|---|---|"]
    for (name, g), v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:28]:
        lines.append("| %s | %d | %d | %.2f | %.1f |" % (name, g, len(v), sum(v) / 1e3, sum(v) / len(v)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in tr)
    # the steady-state tail: the last half of the dispatches (the timed loop), gaps under 1 ms only (host pauses excluded)
    tail = ev[len(ev) // 2:]
    busy = sum(e - s_ for s_, e in tail)
    gaps = [tail[i + 1][0] - max(x[1] for x in tail[max(0, i - 3):i + 1]) for i in range(len(tail) - 1)]
    small = [g for g in gaps if 0 < g < 1e6]
    lines += ["", "last half of the dispatches: busy %.2f ms, idle between kernels %.2f ms in %d gaps (median %.1f us)"
              % (busy / 1e6, sum(small) / 1e6, len(small), (sorted(small)[len(small) // 2] / 1e3) if small else 0.0)]
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


def _pmc_rows(d, counter):
    f = _one(d, "*_counter_collection.csv")
    out = []
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            out.append((r["Kernel_Name"], int(r["Grid_Size"]) if "Grid_Size" in r else 0, float(r["Counter_Value"])))
    return out


def pmc(fd, wd, out, cells, keys=None):
    fetch = _pmc_rows(fd, "FETCH_SIZE")
    write = _pmc_rows(wd, "WRITE_SIZE")

    def biggest(rows, key):
        sel = [r for r in rows if key in r[0]]
        if not sel:
            return None
        g = max(r[1] for r in sel)
        v = [r[2] for r in sel if r[1] == g]
        return sum(v) / len(v) * 1024.0, len(v)

    res = {"cells": cells, "note": "bytes per launch at depth 0 (largest grid of each kernel); the gfx950 FETCH_SIZE correction is "
                                   "calibrated in the same run on kernels of exactly known traffic: k_sub_mean (8 B read + 8 B written "
                                   "per element) when present, and bench.py's k_stream_probe<1> (reads 8 B per cell of a 512^3 array, "
                                   "writes nothing) / k_stream_probe<2> (48 B read + 8 B written per cell)"}
    inc_f, inc_w = biggest(fetch, "k_sub_mean"), biggest(write, "k_sub_mean")
    res["k_sub_mean_raw_fetch"], res["k_sub_mean_raw_write"] = inc_f, inc_w
    p1, p2f, p2w = biggest(fetch, "k_stream_probe<1>"), biggest(fetch, "k_stream_probe<2>"), biggest(write, "k_stream_probe<2>")
    if p1:
        res["k_stream_probe_read_raw_fetch"] = p1[0]
        res["fetch_correction_from_read_probe"] = 8.0 * 512 ** 3 / p1[0]
    if p2f and p2w:
        res["k_stream_probe_mix_raw_fetch"], res["k_stream_probe_mix_write"] = p2f[0], p2w[0]
        res["fetch_correction_from_mix_probe"] = 48.0 * 512 ** 3 / p2f[0]
        res["write_check_from_mix_probe"] = p2w[0] / (8.0 * 512 ** 3)
    res["raw"] = {}
    for key in (keys or ("k_gsrb_ortho", "k_gsrb_fused", "k_op_ortho<0>", "k_resid_march<0", "k_resid_march<2", "k_restrict", "k_prolong")):
        f, w = biggest(fetch, key), biggest(write, key)
        if f and w:
            res["raw"][key] = {"fetch_bytes_raw": f[0], "write_bytes": w[0], "launches": f[1]}
    corr = res.get("fetch_correction_from_read_probe") or 2.0
    res["corrected"] = {k: {"bytes_per_launch": v["fetch_bytes_raw"] * corr + v["write_bytes"], "per_cell": (v["fetch_bytes_raw"] * corr + v["write_bytes"]) / cells,
                            "fetch_correction_used": corr} for k, v in res["raw"].items()}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]) if len(sys.argv) > 5 else 512 ** 3,
            sys.argv[6].split(",") if len(sys.argv) > 6 else None)
