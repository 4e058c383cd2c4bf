#!/bin/bash
# quick GPU validation: warm-pass tests, parity tests, per-pass diagnostics (each command under its own hard timeout)
mkdir -p gpurun_out
cd /root/repo
timeout -s KILL 150 python -m pytest tests/test_warm_pass.py -x -q 2>&1 | tail -4
timeout -s KILL 400 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -6
for lim in ${LIMITS:-0.1}; do echo "== warm limit $lim"; timeout -s KILL 100 python scripts/diag_passes.py 1000000 $lim 2>&1 | tail -11; done
