"""Condense a tools/profile.sh output directory into a few lines per kernel (means per dispatch)."""
import csv, glob, os, sys, collections
out = sys.argv[1]
def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("trace/**/*kernel_stats.csv"):
    for i, row in enumerate(csv.DictReader(open(f))):
        if i < 8:
            print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
print("== PMC (mean per dispatch, per kernel whose name contains 'accumulate') ==")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in find("pmc_*/**/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "")
        if "accumulate" not in name:
            continue
        short = name.split("(")[0].replace("void ", "").replace("ope::", "")
        agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for short in sorted(agg):
    print(f"-- {short}")
    for k in sorted(agg[short]):
        v = agg[short][k]
        print(f"{k:36s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
