#!/bin/bash
mkdir -p gpurun_out
cd /root/repo
timeout -s KILL 300 python -m pytest tests/test_gpu_parity.py -x -q -k "c3_matches or c4_matches or odometry_matches" 2>&1 | tail -12
timeout -s KILL 300 python -m pytest tests/test_distributed_gpu.py -x -q -rs 2>&1 | tail -8
timeout -s KILL 300 python bench.py --steps 200 --warmup 20 > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; tail -c 600 gpurun_out/bench_n1.err; python - <<'PY'
import json
j=json.load(open('gpurun_out/bench_n1.json'))
for k in ('value','ms_per_step','sustained','fp64','parity_vs_reference','gpu_over_cpu'):
    print(k, j.get(k))
print('roofline', {k:j['roofline'][k] for k in ('frac','avg_launch_us','cold_pass_avg_us','warm_pass_avg_us','warm_pass_search_avg_us','nn_search_kernel_avg_us','linearize_kernel_avg_us','error_kernel_avg_us','pass_stats')})
print('cpu', j['cpu_baseline']['value'], j['cpu_baseline']['cores'], j['cpu_baseline']['kind'])
print('odom', {k:v for k,v in j.get('kitti_odom',{}).items() if 'ms' in k})
print('vgicp', j.get('vgicp_c4',{}).get('value'))
PY
timeout -s KILL 300 python bench.py --gpus 2 --oversubscribe --steps 40 --warmup 10 --odom-frames 4 --sustain-s 0 > gpurun_out/bench_n2_over.json 2> gpurun_out/bench_n2_over.err; echo "rc=$?"; tail -c 1500 gpurun_out/bench_n2_over.err; cut -c1-900 gpurun_out/bench_n2_over.json
