#!/bin/bash
# full GPU suite, per-pass diagnostics, quick bench
mkdir -p gpurun_out
cd /root/repo
timeout -s KILL 900 python -m pytest tests/ -x -q -m gpu -rs 2>&1 | tail -15
timeout -s KILL 200 python scripts/diag_passes.py 2>&1 | tail -14
timeout -s KILL 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --sustain-s 1 > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; tail -c 300 gpurun_out/bench_quick.err
python - <<PY
import json
j=json.load(open('gpurun_out/bench_quick.json'))
print('value', j['value'], 'ms/step', j['ms_per_step'], 'frac', j['roofline']['frac'], 'K1', j['roofline']['avg_launch_us'])
print('odom', {k:v for k,v in j.get('kitti_odom',{}).items() if 'ms' in k})
print('vgicp', j.get('vgicp_c4',{}).get('value'), 'fp64', j.get('fp64',{}).get('iterations_per_s'), 'parity', j.get('parity_vs_reference'))
PY
