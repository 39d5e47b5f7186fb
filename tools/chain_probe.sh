#!/bin/bash
set -e
make -C object-pose-estimation_amd clean > /dev/null
make -C object-pose-estimation_amd DEVELOPER=1 -j16 libope_hip.so > /dev/null 2>&1
python tools/chain_probe.py
OPE_HEAVY_LOAD=1.2 python tools/chain_probe.py
OPE_HEAVY_LOAD=0.9 python tools/chain_probe.py
