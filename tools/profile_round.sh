#!/bin/bash
# Round profile session (GPU box, product build): bash tools/profile_round.sh <round label, e.g. r2>
R=${1:-r2}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/profile.sh ${R}_c3 --steps 30 --warmup 5 --no-cpu-baseline
for cfg in "0.10 tree g1tree" "0.10 grid g1grid" "0.0 tree g0tree" "0.0 grid g0grid"; do
  set -- $cfg
  bash tools/pmc_script.sh ${R}_$3 tools/grid_probe3.py $1 $2 > $REPO/gpurun_out/pmc_${R}_$3.txt 2>&1
done
bash tools/kstat.sh ${R}_c5_8x500k tools/c5_device.py 8 500000 > $REPO/gpurun_out/kstat_${R}_c5.txt 2>&1
cd $REPO && python3 tools/c5_device.py 8 500000 > gpurun_out/${R}_c5_device.txt 2>&1
