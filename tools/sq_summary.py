#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes of SQ counters (one counter per pass): per kernel of the Chamfer step, the mean
counter value per launch.  usage: sq_summary.py <dir with ch_<COUNTER>/ subdirs written by tools/regen_profiles.sh>"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
table = collections.defaultdict(dict)      # kernel -> counter -> (mean, n)
for d in sorted(glob.glob(os.path.join(root, "ch_*"))):
    val = collections.defaultdict(lambda: collections.defaultdict(float))
    cname = os.path.basename(d)[3:]
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            val[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, disp in val.items():
        short = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
        if not short.startswith(("grid_", "nmdist_")):
            continue
        v = list(disp.values())
        table[short][cname] = (sum(v) / len(v), len(v))
for k in sorted(table):
    print(k)
    for c in sorted(table[k]):
        m, n = table[k][c]
        print("    %-28s mean %.4g  (n=%d)" % (c, m, n))
