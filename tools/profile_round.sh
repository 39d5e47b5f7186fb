#!/bin/bash
# Round profile session (GPU box, product build): bash tools/profile_round.sh <round label, e.g. r2>
# Keeps the summaries small enough to travel back (gpurun_out is limited to 64 MiB): traces are deleted once condensed.
R=${1:-r2}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/profile.sh ${R}_c3 --steps 30 --warmup 5 --no-cpu-baseline > $REPO/gpurun_out/profile_${R}_c3.txt 2>&1
cp $REPO/gpurun_out/prof_${R}_c3/summary.txt $REPO/gpurun_out/${R}_c3_summary.txt
cp $(find $REPO/gpurun_out/prof_${R}_c3/trace -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/${R}_c3_kernel_stats.csv
cp $REPO/gpurun_out/prof_${R}_c3/bench_trace.json $REPO/gpurun_out/${R}_c3_bench_under_trace.json
rm -rf $REPO/gpurun_out/prof_${R}_c3
for cfg in "0.10 tree g1tree" "0.10 grid g1grid" "0.0 tree g0tree" "0.0 grid g0grid"; do
  set -- $cfg
  bash tools/pmc_script.sh ${R}_$3 tools/grid_probe3.py $1 $2 > $REPO/gpurun_out/pmc_${R}_$3.txt 2>&1
  rm -rf $REPO/gpurun_out/pmcs_${R}_$3
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kstat_c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstat_c5 -- python3 $REPO/tools/c5_device.py 8 500000 > $REPO/gpurun_out/${R}_c5_under_trace.txt 2>&1
cp $(find /tmp/kstat_c5 -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/${R}_c5_8x500k_kernel_stats.csv
cd $REPO && python3 tools/c5_device.py 8 500000 > gpurun_out/${R}_c5_device.txt 2>&1
