#!/bin/bash
set -e
make -C object-pose-estimation_amd clean > /dev/null
make -C object-pose-estimation_amd DEVELOPER=1 -j16 libope_hip.so > /dev/null 2>&1
python tools/chunk_profile.py C3
python tools/far_probe.py
OPE_NO_PACKET=1 python tools/far_probe.py
