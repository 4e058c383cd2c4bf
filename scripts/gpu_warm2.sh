#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
echo "=== c3 properties"; timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py::test_c3_properties -x -q -m gpu 2>&1 | grep -E "^E|assert|passed|failed" | head -20
echo "=== diag grid walkers (default)"; timeout -s KILL 200 python scripts/diag_passes.py 2>&1 | tail -15
echo "=== diag SGA_GRID=0"; SGA_GRID=0 timeout -s KILL 200 python scripts/diag_passes.py 2>&1 | tail -14
echo "=== bench A/B"; SKIP_TESTS=1 bash scripts/gpu_ab.sh "SGA_GRID=1" "SGA_WARM_SPLIT=0" "SGA_GRID=1" "SGA_WARM_SPLIT=0" 2>&1 | tail -8
