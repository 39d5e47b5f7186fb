#!/bin/bash
set -e
make -C object-pose-estimation_amd clean > /dev/null
make -C object-pose-estimation_amd DEVELOPER=1 -j16 libope_hip.so > /dev/null 2>&1
for h in 0 1.0 1.3 1.6 2.0; do echo "== OPE_HEAVY_LOAD=$h"; OPE_HEAVY_LOAD=$h python tools/config_probe.py tree; done
python tools/cost_probe.py
python -m pytest tests/test_gpu_icp.py -x -q --timeout 600 2>&1 | tail -5
