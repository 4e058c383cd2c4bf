#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
for c in 0.16 0.25 0.35; do echo "=== kstats grid=1 cell=$c"; SGA_GRID=1 SGA_GRID_CELL=$c KSTATS_TOP=8 bash scripts/kstats.sh g$c python /root/repo/scripts/diag_passes.py 2>&1 | tail -9; done
