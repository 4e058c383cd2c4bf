#!/bin/bash
cd /root/repo
timeout -s KILL 300 python -m pytest tests/test_reference_python_suite.py tests/test_integration_policy.py -x -q -m gpu 2>&1 | tail -30
timeout -s KILL 300 python -m pytest tests/test_gpu_parity.py -x -q -k "c3_matches or c4_matches or module_surface" 2>&1 | tail -12
