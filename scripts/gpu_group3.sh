#!/bin/bash
# same-box A/B of library builds (SGA_LIB_PATH) and walker modes
cd /root/repo
L=/root/repo/small_gicp_amd/lib
SKIP_TESTS=1 bash scripts/gpu_ab.sh "SGA_LIB_PATH=$L/libsmall_gicp_amd_base.so SGA_GRID_WALK=7" "SGA_GRID_WALK=7" "SGA_GRID_WALK=15" "SGA_LIB_PATH=$L/libsmall_gicp_amd_gonly.so SGA_GRID_WALK=15" "SGA_LIB_PATH=$L/libsmall_gicp_amd_base.so SGA_GRID_WALK=7" "SGA_GRID_WALK=7" "SGA_GRID_WALK=15" "SGA_LIB_PATH=$L/libsmall_gicp_amd_gonly.so SGA_GRID_WALK=15" 2>&1 | grep "it/s" | sed 's|/root/repo/small_gicp_amd/lib/||'
