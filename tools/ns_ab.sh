#!/bin/bash
# A/B of library variants on normal shooting: exactness hashes, then C3 timing.  usage: ns_ab.sh <log name> <variant|-> ...
out=gpurun_out/$1.log; shift
mkdir -p gpurun_out; : > $out
for v in "$@"; do
  lib=$v; [ "$v" = "-" ] && lib=
  echo "== ${v} hashes" >> $out; PROBE_LIB=$lib python3 tools/ns_determinism_probe.py 2>&1 | grep -E "^K=|converging|Error|error" >> $out || echo "FAILED" >> $out
  echo "== ${v} timing" >> $out; PROBE_LIB=$lib python3 tools/ns_bench.py >> $out 2>&1 || echo "FAILED rc=$?" >> $out
done
grep -v amdgpu.ids $out | cut -c1-200
