#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
echo "=== suite with adjacency forced everywhere"; SGA_ADJ_MIN_POINTS=16 SGA_ADJ_PASS=2 timeout -s KILL 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6
for m in 0 1 2; do echo "=== diag ADJ_PASS=$m"; SGA_ADJ_PASS=$m timeout -s KILL 200 python scripts/diag_passes.py 2>&1 | tail -14 | head -13; done
