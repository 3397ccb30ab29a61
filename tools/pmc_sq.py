#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSV per kernel name: python tools/pmc_sq.py <dir> [substr ...]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
keys = sys.argv[2:] or ["k_gsrb_fused", "k_resid_march"]
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    for k in keys:
        if k in name:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in acc:
    print(k, "dispatches", len(n[k]))
    for c, v in sorted(acc[k].items()):
        print("   %-24s %.4g per dispatch" % (c, v / len(n[k])))
