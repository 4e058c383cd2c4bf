#!/bin/bash
# GPU box: selected tests first (arguments), then the whole -m gpu suite.   bash scripts/gpu_suite.sh tests/test_coordinate_range.py
mkdir -p gpurun_out
cd /root/repo
if [ $# -gt 0 ]; then timeout -s KILL 900 python -m pytest "$@" -q -m gpu -s 2>&1 | tail -${TAIL:-40}; fi
if [ "${SKIP_SUITE:-0}" != "1" ]; then timeout -s KILL 1500 python -m pytest tests/ -q -m gpu -rs 2>&1 | tail -15; fi
