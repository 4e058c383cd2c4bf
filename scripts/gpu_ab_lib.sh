#!/bin/bash
# GPU box: same-box A/B of compile-time variants of the library (make variant V=..) on the C3 headline: arguments = library file names
# under small_gicp_amd/lib/ ("libsmall_gicp_amd.so" = the product build), each run twice, alternating; per-pass table for each first.
mkdir -p gpurun_out
cd /root/repo
for lib in "$@"; do
  echo "== $lib"
  SGA_LIB_PATH=/root/repo/small_gicp_amd/lib/$lib timeout -s KILL 200 python scripts/diag_passes.py 2>&1 | grep "^pass\|total"
done
for rep in 1 2; do
  for lib in "$@"; do
    SGA_LIB_PATH=/root/repo/small_gicp_amd/lib/$lib SKIP_TESTS=1 bash scripts/gpu_ab.sh "SGA_VARIANT=$lib"
  done
done
