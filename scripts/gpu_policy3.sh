#!/bin/bash
# GPU box: the reference-side policy's rates (oracle/_ref/policy_bench through scripts/policy_bench.py): C3, C2 and VGICP at C4
cd /root/repo
for cfg in "GICP 1000000 5" "PLANE_ICP 100000 20" "VGICP 1000000 5"; do
  echo "=== policy bench $cfg"
  timeout -s KILL 400 python scripts/policy_bench.py $cfg | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print({k:j.get(k) for k in ('first_align_s','first_bind_s','whole_align_iterations_per_s','policy_calls_iterations_per_s','lean','per_align_ms')})"
done
