#!/bin/bash
# round 4, session 1: the cell grid — correctness (new tests, then the whole suite with the grid forced on), then per-pass timing A/B
mkdir -p gpurun_out
cd /root/repo
export SGA_GRID_VERBOSE=1
echo "=== test_cell_grid"; timeout -s KILL 400 python -m pytest tests/test_cell_grid.py -x -q 2>&1 | tail -25
echo "=== diag_passes grid=1"; SGA_GRID=1 timeout -s KILL 200 python scripts/diag_passes.py 2>&1 | tail -16
echo "=== diag_passes grid=0"; SGA_GRID=0 timeout -s KILL 200 python scripts/diag_passes.py 2>&1 | tail -16
echo "=== diag_passes grid=2"; SGA_GRID=2 timeout -s KILL 200 python scripts/diag_passes.py 2>&1 | tail -16
for c in 0.125 0.15 0.25; do echo "=== diag_passes grid=1 cell=$c"; SGA_GRID=1 SGA_GRID_CELL=$c timeout -s KILL 200 python scripts/diag_passes.py 2>&1 | tail -14; done
echo "=== full suite, default"; timeout -s KILL 600 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8
echo "=== full suite, grid forced (mode 2, every target)"; SGA_GRID=2 SGA_GRID_MIN_POINTS=16 timeout -s KILL 600 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -12
