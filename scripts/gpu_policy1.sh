#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
echo "=== policy test"; timeout -s KILL 500 python -m pytest tests/test_integration_policy.py -x -q -s 2>&1 | tail -15
echo "=== policy_bench C3 1M"; timeout -s KILL 400 python scripts/policy_bench.py GICP 1000000 5 2>&1 | tail -3
echo "=== policy_bench C2 100k"; timeout -s KILL 200 python scripts/policy_bench.py PLANE_ICP 100000 20 2>&1 | tail -3
echo "=== policy_bench C3 1M, 2 shards on one device"; timeout -s KILL 400 python scripts/policy_bench.py GICP 1000000 3 2 2>&1 | tail -3
echo "=== ring1 with 4 waves per workgroup"; SGA_GRID=1 KSTATS_TOP=12 bash scripts/kstats.sh g4w python /root/repo/scripts/diag_passes.py 2>&1 | grep -E "grid_"
