#!/bin/bash
# GPU box: the per-pass table of one C3 registration (scripts/diag_passes.py) under each environment setting given as an argument
#   bash scripts/gpu_cert_pad.sh "" "SGA_CERT_PAD=0.3" "SGA_CERT_PAD=0.5 SGA_SLACK_MAX=0.03"
mkdir -p gpurun_out
cd /root/repo
i=0
for cfg in "$@"; do
  i=$((i+1))
  echo "=== [$cfg]"
  env $cfg SGA_ADJ_STATS=1 timeout -s KILL 120 python scripts/diag_passes.py 2>&1 | grep -E "^pass|^total" | sed -e 's/grid_open=.*unsettled=/unsettled=/' | tee gpurun_out/cert_pad_$i.txt
done
