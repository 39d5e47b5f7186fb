#!/bin/bash
# Usage (GPU box): bash tools/kstat_all.sh <label> <python script and args...>
# rocprofv3 --kernel-trace --stats of the command; grouped per-kernel table -> gpurun_out/kstat_<label>.txt
LABEL=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $REPO/gpurun_out/kstat_$LABEL
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/kstat_$LABEL -- python3 $REPO/"$@" > $REPO/gpurun_out/kstat_$LABEL.stdout 2>&1
python3 - "$LABEL" "$*" <<PY
import csv, glob, re, sys, collections
label, cmd = sys.argv[1], sys.argv[2]
f = glob.glob("$REPO/gpurun_out/kstat_%s/*/*kernel_stats.csv" % label)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
agg = collections.OrderedDict()
for r in rows:
    n = r["Name"]
    if "rocprim" in n: k = "rocprim::*"
    else:
        m = re.search(r"(ope::(?:\(anonymous namespace\)::)?\w+(?:<[^>]*>)?)", n)
        k = (m.group(1) if m else n[:48]).replace("(anonymous namespace)::", "")
    a = agg.setdefault(k, [0, 0, 10**18, 0]); a[0] += int(r["Calls"]); a[1] += int(r["TotalDurationNs"])
    a[2] = min(a[2], int(r["MinNs"])); a[3] = max(a[3], int(r["MaxNs"]))
out = ["rocprofv3 --kernel-trace --stats of \`python3 %s\`, one MI355X.  rocPRIM kernels are grouped." % cmd,
       "kernel | calls | total us | average us | min us | max us | % of GPU time"]
for k, (c, t, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    out.append("%s | %d | %.1f | %.2f | %.2f | %.2f | %.2f" % (k, c, t / 1e3, t / 1e3 / c, mn / 1e3, mx / 1e3, 100.0 * t / tot))
out.append("total GPU kernel time %.2f ms" % (tot / 1e6))
open("$REPO/gpurun_out/kstat_%s.txt" % label, "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
PY
