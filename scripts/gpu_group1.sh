#!/bin/bash
# round 4: ring 1 by a group of lanes per walker (SGA_GRID_WALK bit 8; bit 16: wave-wide settling ring for a lone unsettled walker) against one lane per walker (7)
mkdir -p gpurun_out
cd /root/repo
echo "=== warm + parity tests"; timeout -s KILL 600 python -m pytest tests/test_warm_pass.py tests/test_cell_grid.py "tests/test_gpu_parity.py::test_c3_matches_reference" "tests/test_gpu_parity.py::test_c3_properties" -x -q -m gpu 2>&1 | tail -3
echo "=== soak"; timeout -s KILL 300 python scripts/soak_exactness.py 2>&1 | tail -3
for m in 7 15 31; do echo "=== diag GRID_WALK=$m"; SGA_GRID_WALK=$m timeout -s KILL 200 python scripts/diag_passes.py 2>&1 | tail -14 | head -13 | cut -c1-75; done
echo "=== bench A/B"; SKIP_TESTS=1 bash scripts/gpu_ab.sh "SGA_GRID_WALK=7" "SGA_GRID_WALK=15" "SGA_GRID_WALK=31" "SGA_GRID_WALK=7" "SGA_GRID_WALK=15" "SGA_GRID_WALK=31" 2>&1 | tail -7
