#!/bin/bash
set -e
make -C object-pose-estimation_amd clean > /dev/null
make -C object-pose-estimation_amd DEVELOPER=1 -j16 libope_hip.so > /dev/null 2>&1
export OPE_TRACE_GRID=1
python tools/grid_probe2.py
