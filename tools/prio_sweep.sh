#!/bin/bash
set -e
make -C object-pose-estimation_amd clean > /dev/null
make -C object-pose-estimation_amd DEVELOPER=1 -j16 libope_hip.so > /dev/null 2>&1
for f in 0 1.5 2 3 5; do OPE_PRIO_FACTOR=$f python tools/prio_probe.py; done
