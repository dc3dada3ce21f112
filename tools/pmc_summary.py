#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes: per kernel name, mean counter value (KB) and mean duration.
usage: pmc_summary.py <dir with <tag>_<COUNTER>/ subdirs written by tools/refresh_profiles.sh>"""
import csv, glob, os, sys, collections
root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "*_*"))):
    if not os.path.isdir(d):
        continue
    cc = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
    kt = glob.glob(os.path.join(d, "*", "*kernel_trace.csv"))
    if not cc:
        continue
    dur = collections.defaultdict(list)
    for f in kt:
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    val = collections.defaultdict(lambda: collections.defaultdict(float))   # name -> dispatch -> sum
    cname = None
    for f in cc:
        for r in csv.DictReader(open(f)):
            cname = r["Counter_Name"]
            val[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    print(os.path.basename(d))
    for k, disp in sorted(val.items(), key=lambda kv: -sum(kv[1].values())):
        v = list(disp.values())
        mean = sum(v) / len(v)
        du = dur.get(k, [0])
        short = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
        print("   %-60s %s launches=%d mean=%.0f KB = %.1f MB   mean duration %.1f us"
              % (short, cname, len(v), mean, mean / 1024, sum(du) / len(du) / 1e3))
