#!/bin/bash
# GPU box: developer build, then plan / heavy-threshold sweep of the TREE kernel on the C3 frame (clutter 10 %)
set -e
make -C object-pose-estimation_amd clean > /dev/null
make -C object-pose-estimation_amd DEVELOPER=1 -j16 libope_hip.so > /dev/null 2>&1
echo "== default"; python tools/grid_probe3.py 0.1 tree 140
echo "== OPE_NO_PLAN=1"; OPE_NO_PLAN=1 python tools/grid_probe3.py 0.1 tree 140
for f in 0 2 3 8 20; do echo "== OPE_HEAVY_FACTOR=$f"; OPE_HEAVY_FACTOR=$f python tools/grid_probe3.py 0.1 tree 140; done
echo "== OPE_NO_PACKET=1"; OPE_NO_PACKET=1 python tools/grid_probe3.py 0.1 tree 140
echo "== OPE_NO_PACKET=1 OPE_HEAVY_FACTOR=0"; OPE_NO_PACKET=1 OPE_HEAVY_FACTOR=0 python tools/grid_probe3.py 0.1 tree 140
