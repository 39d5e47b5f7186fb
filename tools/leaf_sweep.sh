#!/bin/bash
set -e
make -C object-pose-estimation_amd clean > /dev/null
make -C object-pose-estimation_amd DEVELOPER=1 -j16 libope_hip.so > /dev/null 2>&1
for l in 4 8 12 16 24 32; do PROBE_LEAF=$l python tools/config_probe.py tree,grid; done
