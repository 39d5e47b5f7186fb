#!/bin/bash
# Developer A/B on the C3 frame, tree kernel (DEVELOPER=1 build).  Output: gpurun_out/$1.log
out=gpurun_out/${1:-ab}.log
mkdir -p gpurun_out
: > $out
run() { echo "== $*" >> $out; env "$@" python3 tools/foot_probe.py ${MODES:-steady,window} ${REPS:-2} >> $out 2>&1 || echo "FAILED rc=$?" >> $out; }
run OPE_X=base
run OPE_SPLIT=1
run OPE_SPLIT=1 OPE_HEAVY_LOAD=1.2
run OPE_SPLIT=1 OPE_HEAVY_LOAD=1.0
run OPE_SPLIT=1 OPE_HEAVY_LOAD=0.8
run OPE_X=base OPE_HEAVY_LOAD=1.2
MODES=chunks run OPE_SPLIT=1
grep -v amdgpu.ids $out | cut -c1-420
