#!/bin/bash
# same-box A/B of library builds (SGA_LIB_PATH)
cd /root/repo
L=/root/repo/small_gicp_amd/lib
timeout -s KILL 600 python -m pytest tests/test_warm_pass.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
SKIP_TESTS=1 bash scripts/gpu_ab.sh "SGA_LIB_PATH=$L/libsmall_gicp_amd_base.so" "A=1" "SGA_REDUCE_GROUPS=64" "SGA_REDUCE_GROUPS=128" "SGA_LIB_PATH=$L/libsmall_gicp_amd_base.so" "A=1" "SGA_REDUCE_GROUPS=64" "SGA_REDUCE_GROUPS=128" 2>&1 | grep "it/s" | sed 's|/root/repo/small_gicp_amd/lib/||'
