#!/bin/bash
set -e
make -C object-pose-estimation_amd clean > /dev/null
make -C object-pose-estimation_amd DEVELOPER=1 -j16 libope_hip.so > /dev/null 2>&1
for h in 1.45 1.6 1.8 2.0 2.4; do OPE_HEAVY_LOAD=$h python tools/ab_probe.py tree 3; done
OPE_HEAVY_LOAD=1.6 OPE_ACC_BLOCKS=1024 python tools/ab_probe.py tree 3
OPE_HEAVY_LOAD=1.6 OPE_ACC_BLOCKS=896 python tools/ab_probe.py tree 3
OPE_HEAVY_LOAD=1.6 python tools/config_probe.py tree,grid
python tools/fill_probe.py
