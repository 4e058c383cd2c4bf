#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
echo "=== new tests (-s)"; timeout -s KILL 900 python -m pytest tests/test_contexts.py tests/test_distributed_gpu.py "tests/test_gpu_parity.py::test_normals_covariances_match_oracle" "tests/test_gpu_parity.py::test_covariances_at_scale_match_reference" "tests/test_gpu_parity.py::test_c3_with_cpu_estimated_covariances" -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -25
echo "=== full suite"; timeout -s KILL 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6
