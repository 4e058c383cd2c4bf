#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
echo "=== policy + multi tests"; timeout -s KILL 500 python -m pytest tests/test_integration_policy.py tests/test_distributed_gpu.py -x -q -m gpu 2>&1 | tail -8
echo "=== policy_bench C3 1M"; timeout -s KILL 400 python scripts/policy_bench.py GICP 1000000 5 2>&1 | tail -3
echo "=== policy_bench C2 100k"; timeout -s KILL 200 python scripts/policy_bench.py PLANE_ICP 100000 20 2>&1 | tail -3
