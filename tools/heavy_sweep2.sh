#!/bin/bash
# GPU box: developer build, then the tree-part heavy-chunk threshold sweep of the grid kernel (clutter 10 %)
set -e
make -C object-pose-estimation_amd clean > /dev/null
make -C object-pose-estimation_amd DEVELOPER=1 -j16 libope_hip.so > /dev/null 2>&1
for f in -1 0 0.7 1.0 1.5 2.5 4; do
  if [ "$f" = "-1" ]; then unset OPE_HEAVY_FACTOR; else export OPE_HEAVY_FACTOR=$f; fi
  echo "== OPE_HEAVY_FACTOR=$f"; python tools/grid_probe3.py 0.1 grid 100
done
unset OPE_HEAVY_FACTOR
python tools/grid_probe3.py 0.1 tree 100
