#!/bin/bash
out=gpurun_out/${1:-window}.log
mkdir -p gpurun_out; : > $out
run() { echo "== $*" >> $out; env PROBE_LIB=dev "$@" python3 tools/window_ab.py 3 >> $out 2>&1 || echo "FAILED rc=$?" >> $out; }
run OPE_PLAN_SYNC=1
run OPE_X=default
run OPE_PLAN_SYNC=1
run OPE_X=default
grep -v amdgpu.ids $out | cut -c1-400
