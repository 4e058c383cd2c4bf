#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
echo "=== warm + parity tests"; timeout -s KILL 600 python -m pytest tests/test_warm_pass.py tests/test_cell_grid.py "tests/test_gpu_parity.py::test_c3_matches_reference" "tests/test_gpu_parity.py::test_c3_properties" -x -q -m gpu 2>&1 | tail -3
for m in 1 3 5 7; do echo "=== diag GRID_WALK=$m"; SGA_GRID_WALK=$m timeout -s KILL 200 python scripts/diag_passes.py 2>&1 | tail -14 | head -12 | cut -c1-90; done
echo "=== bench A/B"; SKIP_TESTS=1 bash scripts/gpu_ab.sh "SGA_GRID_WALK=1" "SGA_GRID_WALK=7" "SGA_GRID_WALK=3" "SGA_GRID_WALK=5" "SGA_GRID_WALK=1" "SGA_GRID_WALK=7" 2>&1 | tail -7
