#!/bin/bash
# GPU box: compile-time variants of the library (make variant V=..) against the product build: parity tests of the walk with the LAST
# variant, then the small-problem pass times (scripts/diag_small.py) and the C3 headline for every library, alternating, twice.
#   bash scripts/gpu_ab_variants.sh libsmall_gicp_amd.so libsmall_gicp_amd_pf1.so ...
mkdir -p gpurun_out
cd /root/repo
last="${@: -1}"
SGA_LIB_PATH=/root/repo/small_gicp_amd/lib/$last timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py tests/test_warm_pass.py tests/test_cell_grid.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    SGA_LIB_PATH=/root/repo/small_gicp_amd/lib/$lib timeout -s KILL 300 python scripts/diag_small.py 2>&1 | tail -4
    if [ "${SKIP_C3:-0}" != "1" ]; then SGA_LIB_PATH=/root/repo/small_gicp_amd/lib/$lib SKIP_TESTS=1 bash scripts/gpu_ab.sh "SGA_VARIANT=$lib"; fi
  done
done
