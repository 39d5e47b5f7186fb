#!/bin/bash
out=gpurun_out/${1:-faronly}.log
mkdir -p gpurun_out; : > $out
run() { echo "== $*" >> $out; env "$@" python3 tools/faronly_probe.py >> $out 2>&1 || echo "FAILED rc=$?" >> $out; }
run OPE_X=base
run OPE_HEAVY_FACTOR=0
run OPE_HEAVY_FACTOR=1.5
run OPE_SPLIT=1
grep -v amdgpu.ids $out | cut -c1-300
