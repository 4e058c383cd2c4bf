#!/bin/bash
cd /root/repo
timeout -s KILL 600 python -m pytest tests/test_integration_policy.py tests/test_distributed_gpu.py -q -m gpu -x 2>&1 | tail -3
echo "=== policy bench C3"; timeout -s KILL 300 python scripts/policy_bench.py GICP 1000000 5 | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print({k:j[k] for k in ('first_align_s','first_bind_s','whole_align_iterations_per_s','policy_calls_iterations_per_s','lean','per_align_ms')})"
echo "=== policy bench C2"; timeout -s KILL 300 python scripts/policy_bench.py PLANE_ICP 100000 20 | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print({k:j[k] for k in ('first_align_s','first_bind_s','whole_align_iterations_per_s','policy_calls_iterations_per_s','lean')})"
