#!/bin/bash
# first GPU validation of the warm pass: tests, then a bench line, then kernel stats
set -x
mkdir -p gpurun_out
cd /root/repo
timeout 900 python -m pytest tests/test_warm_pass.py -x -q 2>&1 | tail -15 > gpurun_out/warm_tests.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -15 > gpurun_out/parity_tests.log
timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --odom-frames 0 --no-vgicp > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
cat gpurun_out/warm_tests.log gpurun_out/parity_tests.log gpurun_out/bench_a.json
