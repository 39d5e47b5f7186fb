#!/bin/bash
# A/B of library variants (make VARIANT=...): the chain floor (clutter only, shard-size mixes) and the C3 frame.
# usage: ab_variants.sh <log name> <variant> [<variant> ...]
out=gpurun_out/$1.log; shift
mkdir -p gpurun_out; : > $out
for v in "$@"; do
  echo "== $v faronly" >> $out; PROBE_LIB=$v python3 tools/faronly_probe.py >> $out 2>&1 || echo "FAILED rc=$?" >> $out
  echo "== $v c3" >> $out; PROBE_LIB=$v python3 tools/c3_probe.py steady,window 2 >> $out 2>&1 || echo "FAILED rc=$?" >> $out
done
grep -v amdgpu.ids $out | cut -c1-300
