#!/bin/bash
cd /root/repo
timeout -s KILL 300 python -m pytest tests/test_gpu_parity.py -x -q -k "restrict or rejector or gauss" 2>&1 | tail -12
