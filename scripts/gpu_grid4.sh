#!/bin/bash
# counters of the grid kernels and of the kd kernels on the same registration (SGA_GRID=2: pass 1 kd, passes 2-4 grid)
mkdir -p gpurun_out
cd /root/repo
SGA_GRID=2 PMC_EXTRA=1 bash scripts/pmc_k1.sh grid > gpurun_out/pmc_grid.txt 2>&1
SGA_GRID=0 PMC_EXTRA=1 bash scripts/pmc_k1.sh kd > gpurun_out/pmc_kd.txt 2>&1
tail -3 gpurun_out/pmc_grid.txt
