#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
echo "=== test_cell_grid"; timeout -s KILL 400 python -m pytest tests/test_cell_grid.py -x -q 2>&1 | tail -8
for c in 0.16 0.25; do echo "=== kstats grid=1 cell=$c"; SGA_GRID=1 SGA_GRID_CELL=$c KSTATS_TOP=9 bash scripts/kstats.sh g$c python /root/repo/scripts/diag_passes.py 2>&1 | grep -E "grid_|search_lin|linearize_kernel|reduce|queue"; done
echo "=== diag_passes grid=1"; SGA_GRID=1 timeout -s KILL 200 python scripts/diag_passes.py 2>&1 | tail -15
echo "=== full suite, grid forced (mode 2, every target)"; SGA_GRID=2 SGA_GRID_MIN_POINTS=16 timeout -s KILL 600 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6
