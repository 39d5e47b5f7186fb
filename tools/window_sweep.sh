#!/bin/bash
set -e
make -C object-pose-estimation_amd clean > /dev/null
make -C object-pose-estimation_amd DEVELOPER=1 -j16 libope_hip.so > /dev/null 2>&1
python tools/window_probe.py
OPE_NO_ALONE=1 OPE_HEAVY_LOAD=0 OPE_ACC_BLOCKS=1024 python tools/window_probe.py
OPE_NO_ALONE=1 OPE_HEAVY_LOAD=0 python tools/window_probe.py
OPE_HEAVY_LOAD=0 python tools/window_probe.py
