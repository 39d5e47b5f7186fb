#!/bin/bash
# Round profile session (GPU box, product build): bash tools/profile_round.sh <round label, e.g. r3> [part]
#   part 1: rocprofv3 kernel trace + separate PMC passes of `bench.py --steps 30 --warmup 5 --no-cpu-baseline` (the bench's own launches:
#           the timed kernel, the normal-shooting leg, the front end)
#   part 2: kernel trace of the device-resident BuildModel loop (8 frames x 500 k) and its wall-clock without the profiler
# Keeps the summaries small enough to travel back (gpurun_out is limited to 64 MiB): traces are deleted once condensed.
R=${1:-r3}; PART=${2:-all}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
if [ "$PART" = "1" ] || [ "$PART" = "all" ]; then
  bash tools/profile.sh ${R}_c3 --steps 30 --warmup 5 --no-cpu-baseline > $REPO/gpurun_out/profile_${R}_c3.txt 2>&1
  cp $REPO/gpurun_out/prof_${R}_c3/summary.txt $REPO/gpurun_out/${R}_c3_summary.txt
  cp $(find $REPO/gpurun_out/prof_${R}_c3/trace -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/${R}_c3_kernel_stats.csv
  cp $REPO/gpurun_out/prof_${R}_c3/bench_trace.json $REPO/gpurun_out/${R}_c3_bench_under_trace.json
  rm -rf $REPO/gpurun_out/prof_${R}_c3
fi
if [ "$PART" = "2" ] || [ "$PART" = "all" ]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/kstat_c5
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstat_c5 -- python3 $REPO/tools/c5_device.py 8 500000 > $REPO/gpurun_out/${R}_c5_under_trace.txt 2>&1
  cp $(find /tmp/kstat_c5 -name "*kernel_stats.csv" | head -1) $REPO/gpurun_out/${R}_c5_8x500k_kernel_stats.csv
  cd $REPO && python3 tools/c5_device.py 8 500000 > gpurun_out/${R}_c5_device.txt 2>&1
fi
