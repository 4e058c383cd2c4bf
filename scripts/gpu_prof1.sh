#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
echo "=== shards"; timeout -s KILL 400 python scripts/diag_shards.py 2>&1 | tail -5
echo "=== profile"; COMMIT=$(cat .commit 2>/dev/null || echo r04) bash scripts/profile_gpu.sh r04 60 2>&1 | tail -45
