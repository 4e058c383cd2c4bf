#!/bin/bash
cd /root/repo
export SGA_SPLIT_DELTA=0.1
echo "=== tests under SGA_SPLIT_DELTA=0.1"; timeout -s KILL 900 python -m pytest tests/test_warm_pass.py tests/test_cell_grid.py "tests/test_gpu_parity.py::test_c3_matches_reference" "tests/test_gpu_parity.py::test_c3_properties" -x -q -m gpu 2>&1 | tail -3
timeout -s KILL 300 python scripts/soak_exactness.py 2>&1 | tail -2
for d in 0.1; do echo "=== diag SGA_SPLIT_DELTA=$d"; SGA_SPLIT_DELTA=$d timeout -s KILL 200 python scripts/diag_passes.py 2>&1 | tail -14 | head -13 | cut -c1-75; done
unset SGA_SPLIT_DELTA
SKIP_TESTS=1 STEPS=400 bash scripts/gpu_ab.sh "SGA_SPLIT_DELTA=0.002" "SGA_SPLIT_DELTA=0.1" "SGA_SPLIT_DELTA=0.002" "SGA_SPLIT_DELTA=0.1" 2>&1 | grep "it/s"
