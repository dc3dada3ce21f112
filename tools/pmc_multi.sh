#!/bin/bash
# tools/pmc_multi.sh <tag> "<counters of pass 1>" "<counters of pass 2>" ... -- <python script and args>
# one rocprofv3 --pmc pass per counter group (kernel-trace only), summary per kernel printed
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
tag=$1; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
i=0
for g in "${groups[@]}"; do
  (timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d gpurun_out/pmcm_$tag/p$i -- python3 "$@" > gpurun_out/pmcm_$tag.p$i.log 2>&1)
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
val = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcm_$tag/p*/*/*counter_collection.csv"):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Kernel_Name"], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, d, c), v in per.items():
        val[k][c].append(v)
for k, cs in val.items():
    short = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:50]
    if short.startswith("at::"): continue
    print(short)
    for c, v in sorted(cs.items()):
        print("    %-28s mean %.4g  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
