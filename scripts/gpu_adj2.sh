#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
for m in 1 2; do echo "=== diag ADJ_PASS=$m"; SGA_ADJ_STATS=1 SGA_ADJ_PASS=$m timeout -s KILL 200 python scripts/diag_passes.py 2>&1 | tail -14 | head -13; done
echo "=== parity subset, adjacency forced"; SGA_ADJ_MIN_POINTS=16 SGA_ADJ_PASS=2 timeout -s KILL 600 python -m pytest tests/test_warm_pass.py tests/test_cell_grid.py "tests/test_gpu_parity.py::test_c3_matches_reference" "tests/test_gpu_parity.py::test_c3_properties" -x -q -m gpu 2>&1 | tail -4
