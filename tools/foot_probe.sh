#!/bin/bash
# Developer A/B: the C3 frame on the tree kernel, baseline vs far list (DEVELOPER=1 build).  Output: gpurun_out/$1.log
out=gpurun_out/${1:-far}.log
mkdir -p gpurun_out
: > $out
run() { echo "== $*" >> $out; env "$@" python3 tools/foot_probe.py ${MODES:-steady,window,chunks} ${REPS:-2} >> $out 2>&1 || echo "FAILED rc=$?" >> $out; }
run OPE_X=base
run OPE_FAR=1
run OPE_FAR=1 OPE_HEAVY_LOAD=0
run OPE_FAR=1 OPE_FAR_THR=0.0015
run OPE_FAR=1 OPE_FAR_THR=0.004
run OPE_FAR=1 OPE_NO_ALONE=1
grep -v amdgpu.ids $out | cut -c1-420
