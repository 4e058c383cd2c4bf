#!/bin/bash
# two-phase walk of certify_linearize_kernel (SGA_GRID_WALK bit 32) with the large-motion warm passes routed to it (SGA_SPLIT_DELTA=0.1)
cd /root/repo
export SGA_SPLIT_DELTA=0.1
echo "=== tests under SGA_SPLIT_DELTA=0.1"; timeout -s KILL 900 python -m pytest tests/test_warm_pass.py tests/test_cell_grid.py "tests/test_gpu_parity.py::test_c3_matches_reference" "tests/test_gpu_parity.py::test_c3_properties" -x -q -m gpu 2>&1 | tail -3
timeout -s KILL 300 python scripts/soak_exactness.py 2>&1 | tail -2
for m in 15 47; do echo "=== diag SGA_SPLIT_DELTA=0.1 GRID_WALK=$m"; SGA_GRID_WALK=$m timeout -s KILL 200 python scripts/diag_passes.py 2>&1 | tail -14 | head -7 | cut -c1-75; done
unset SGA_SPLIT_DELTA
SKIP_TESTS=1 bash scripts/gpu_ab.sh "SGA_SPLIT_DELTA=0.002 SGA_GRID_WALK=15" "SGA_SPLIT_DELTA=0.1 SGA_GRID_WALK=15" "SGA_SPLIT_DELTA=0.1 SGA_GRID_WALK=47" "SGA_SPLIT_DELTA=0.002 SGA_GRID_WALK=15" "SGA_SPLIT_DELTA=0.1 SGA_GRID_WALK=15" "SGA_SPLIT_DELTA=0.1 SGA_GRID_WALK=47" 2>&1 | grep "it/s"
