#!/bin/bash
# same-box A/B of two library builds (SGA_LIB_PATH) after the exactness tests with the default build
cd /root/repo
L=/root/repo/small_gicp_amd/lib
timeout -s KILL 900 python -m pytest tests/test_warm_pass.py tests/test_cell_grid.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
bash scripts/gpu_ab2.sh "SGA_LIB_PATH=$L/libsmall_gicp_amd_${1}.so" "A=new" "SGA_LIB_PATH=$L/libsmall_gicp_amd_${1}.so" "A=new" "SGA_LIB_PATH=$L/libsmall_gicp_amd_${1}.so" "A=new" 2>&1 | grep "it/s"
