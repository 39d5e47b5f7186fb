#!/bin/bash
# Developer probe: the facade binary N times on the developer library with OPE_DUMP_HASH=1; prints, per run, the checksums of every
# ICP run's inputs and the fine poses.  usage: flake_hash.sh <log name> <runs> [variant]
out=gpurun_out/$1.log; runs=$2; v=${3:-dev}
mkdir -p gpurun_out /tmp/flk/lib_$v; : > $out
cp object-pose-estimation_amd/libope_hip_$v.so /tmp/flk/lib_$v/libope_hip.so
[ -f /tmp/flk/model.pcd ] || bash tools/flake_facade.sh _prep 0 > /dev/null
for i in $(seq 1 $runs); do
  echo "== run $i" >> $out
  mkdir -p gpurun_out/$1.dumps/run$i
  env LD_LIBRARY_PATH=/tmp/flk/lib_$v:$LD_LIBRARY_PATH OPE_DUMP_HASH=1 OPE_DUMP_DIR=gpurun_out/$1.dumps/run$i object-pose-estimation_amd/build/detect_and_localize /tmp/flk/model.pcd /tmp/flk/s1.pcd /tmp/flk/s2.pcd --seed 3 2>&1 \
    | grep -E "^frame |\[hash\]" | awk '/^frame/ {printf "frame it %s fine ", $10; for (i = 46; i <= 49; ++i) printf "%s,", $i; print ""; next} {print}' >> $out
done
for d in gpurun_out/$1.dumps/run*; do ls $d | sort | tail -n +17 | sed "s#^#$d/#" | xargs -r rm -f; done
# which lines vary between runs
awk '/^==/ {k = 0; next} {k++; key = sprintf("%03d: %s", k, $0); c[key]++} END {for (key in c) print key " x" c[key]}' $out | sort | cut -c1-200 > gpurun_out/$1.summary
# first position at which runs disagree
awk '{pos = substr($0, 1, 3); n[pos]++} END {for (p in n) if (n[p] > 1) print p}' gpurun_out/$1.summary | sort | head -3
