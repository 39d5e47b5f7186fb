#!/bin/bash
# Developer hunt: which preceding activity makes the first overlapped run of a fresh bench process fall back?
q() { timeout -k 10 300 python bench.py --no-cpu-baseline --steady 0 --full-run 0 --no-ns 2>&1 >/dev/null | grep -c "gave up"; }
echo "== after nothing"; q; q
echo "== after smoke"; timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > /dev/null 2>&1; q; q
echo "== after the p2p fault test"; timeout -k 10 300 python -m pytest tests/test_gpu_sharded.py -q -m gpu -k "missing_peer" > /dev/null 2>&1; q; q
echo "== after the sharded file"; timeout -k 10 600 python -m pytest tests/test_gpu_sharded.py -q -m gpu -k "not c5" > /dev/null 2>&1; q; q
echo "== after the icp file"; timeout -k 10 600 python -m pytest tests/test_gpu_icp.py -q -m gpu > /dev/null 2>&1; q; q
