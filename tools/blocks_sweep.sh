#!/bin/bash
set -e
make -C object-pose-estimation_amd clean > /dev/null
make -C object-pose-estimation_amd DEVELOPER=1 -j16 libope_hip.so > /dev/null 2>&1
for b in 512 640 704 768 832 1024; do echo "== OPE_ACC_BLOCKS=$b"; OPE_ACC_BLOCKS=$b python tools/config_probe.py; done
