#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
echo "=== warm tests"; timeout -s KILL 600 python -m pytest tests/test_warm_pass.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -6
echo "=== diag split=1"; SGA_WARM_SPLIT=1 timeout -s KILL 200 python scripts/diag_passes.py 2>&1 | tail -15
echo "=== diag split=0"; SGA_WARM_SPLIT=0 timeout -s KILL 200 python scripts/diag_passes.py 2>&1 | tail -15
echo "=== kstats split=1"; SGA_WARM_SPLIT=1 KSTATS_TOP=14 bash scripts/kstats.sh ws1 python /root/repo/scripts/diag_passes.py 2>&1 | grep -E "linearize|queue|reduce|walkers"
echo "=== bench A/B"; SKIP_TESTS=1 bash scripts/gpu_ab.sh "SGA_WARM_SPLIT=1" "SGA_WARM_SPLIT=0" "SGA_WARM_SPLIT=1" "SGA_WARM_SPLIT=0" 2>&1 | tail -8
