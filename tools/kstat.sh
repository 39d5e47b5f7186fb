#!/bin/bash
# Usage (GPU box): bash tools/kstat.sh <label> <python script and args...>   -> average/min of the accumulate + update kernels
LABEL=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $REPO/gpurun_out/kstat_$LABEL
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/kstat_$LABEL -- python3 $REPO/"$@" > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$REPO/gpurun_out/kstat_$LABEL/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "icp_" in r["Name"]: print("$LABEL", r["Name"].split("(")[0][-40:], r["Calls"], "avg %.1f us min %.1f us" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
